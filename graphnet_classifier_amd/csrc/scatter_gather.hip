// K1 scatter-sum aggregation over a destination-sorted CSR, K2 row gather, K6 edge features.
// All three are HBM-bandwidth-bound row movers; the design rules are: 16 B per lane, whole
// rows per lane group, several independent loads in flight per lane, no atomics.
//
// K1 lane mapping (D = feature width):  LPR = D/4 lanes own one destination row (one float4
// column slice each), so a 64-lane wave works on 64/LPR destinations at once (D=64: 16 lanes
// per row, 4 rows per wave instruction = 1 KiB per load instruction).  A wave owns a chunk of
// 64 consecutive destinations: its lanes fetch the 65 row pointers of the chunk with one
// coalesced load and hand them to the lane groups by cross-lane reads, then each group walks
// its segment in ascending sorted position k (= original edge order, the sort is stable) and
// keeps the running sum in registers.  Every destination row is written exactly once.
#include <stdlib.h>

#include "gnc_common.h"

namespace {

using gnc::kWave;

struct alignas(16) f4 { float x, y, z, w; };

__device__ __forceinline__ f4 ld4(const float* p) { return *reinterpret_cast<const f4*>(p); }
// Streamed-once rows (the messages of K1): nontemporal 16-B load, so the 2.5 GB stream does not push the
// row pointers / output lines out of L2 and MALL.  Measured at c3 size: 0.592 -> 0.529 ms per launch.
__device__ __forceinline__ f4 ld4_stream(const float* p) {
  typedef float v4 __attribute__((ext_vector_type(4)));
  const v4 v = __builtin_nontemporal_load(reinterpret_cast<const v4*>(p));
  return f4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ void st4(float* p, f4 v) { *reinterpret_cast<f4*>(p) = v; }
__device__ __forceinline__ void acc4(f4& a, const f4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }

constexpr int kChunk = 64;  // destinations per wave iteration

// ---------------------------------------------------------------------------------------
// K1, vectorised: feat_dim % 4 == 0, feat_dim <= 4*LPR, 16-B aligned rows.
// ---------------------------------------------------------------------------------------
template <int LPR, bool HAS_PERM>
__global__ __launch_bounds__(gnc::kBlock) void scatter_sum_csr_vec4(
    const float* __restrict__ src, int64_t ld_src, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ perm, int32_t num_nodes, int32_t feat_dim, float* __restrict__ out,
    int64_t ld_out) {
  constexpr int GPW = kWave / LPR;      // lane groups (destinations) per wave
  constexpr int ROUNDS = kChunk / GPW;  // rounds to cover the 64-destination chunk
  const int lane = threadIdx.x & (kWave - 1);
  const int grp = lane / LPR;
  const int col = (lane % LPR) * 4;
  const bool col_ok = col < feat_dim;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave;
  const int64_t num_waves = ((int64_t)gridDim.x * blockDim.x) / kWave;
  const int64_t num_chunks = ((int64_t)num_nodes + kChunk - 1) / kChunk;

  for (int64_t chunk = wave; chunk < num_chunks; chunk += num_waves) {
    const int64_t base = chunk * kChunk;
    // lane l holds rowptr[base + l] (clamped), lane 63's upper bound is fetched separately
    const int64_t vl = base + lane;
    const int32_t rp = rowptr[vl < num_nodes ? vl : num_nodes];
    const int64_t vlast = base + kChunk;
    const int32_t rp_last = rowptr[vlast < num_nodes ? vlast : num_nodes];

#pragma unroll 1
    for (int r = 0; r < ROUNDS; ++r) {
      const int slot = r * GPW + grp;  // destination inside the chunk
      const int64_t v = base + slot;
      const int32_t start = __shfl(rp, slot, kWave);
      const int32_t hi = __shfl(rp, (slot + 1) & (kWave - 1), kWave);
      const int32_t end = (slot + 1 < kWave) ? hi : rp_last;
      if (base + r * GPW >= num_nodes) break;  // wave-uniform: nothing left in this chunk
      const bool live = (v < num_nodes) & col_ok;

      f4 a = {0.f, 0.f, 0.f, 0.f};
      if (live) {
        int32_t k = start;
        for (; k + 4 <= end; k += 4) {
          int32_t i0 = k, i1 = k + 1, i2 = k + 2, i3 = k + 3;
          if (HAS_PERM) { i0 = perm[k]; i1 = perm[k + 1]; i2 = perm[k + 2]; i3 = perm[k + 3]; }
          const f4 r0 = ld4_stream(src + (int64_t)i0 * ld_src + col);
          const f4 r1 = ld4_stream(src + (int64_t)i1 * ld_src + col);
          const f4 r2 = ld4_stream(src + (int64_t)i2 * ld_src + col);
          const f4 r3 = ld4_stream(src + (int64_t)i3 * ld_src + col);
          acc4(a, r0); acc4(a, r1); acc4(a, r2); acc4(a, r3);  // ascending k: reference edge order
        }
        for (; k < end; ++k) {
          const int32_t i0 = HAS_PERM ? perm[k] : k;
          acc4(a, ld4_stream(src + (int64_t)i0 * ld_src + col));
        }
        st4(out + v * ld_out + col, a);
      }
    }
  }
}

// K1 for a small graph (the reference's one-graph-per-call regime): the kernel above gives a wave 64 destinations and, at
// 128 features, walks them two at a time - 32 dependent rounds of (ids, rows) per wave while most of the chip is idle
// (1024 nodes: 16 busy waves, 25..37 us).  Here every lane group takes ONE destination, so the launch is one chain of
// row pointers -> ids -> rows per group.  Same sums in the same order (bit-identical).
template <int LPR, bool HAS_PERM>
__global__ __launch_bounds__(gnc::kBlock) void scatter_sum_csr_small(
    const float* __restrict__ src, int64_t ld_src, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ perm, int32_t num_nodes, int32_t feat_dim, float* __restrict__ out,
    int64_t ld_out) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t v = tid / LPR;
  const int col = (int)(tid % LPR) * 4;
  if (v >= num_nodes || col >= feat_dim) return;
  const int32_t start = rowptr[v], end = rowptr[v + 1];
  f4 a = {0.f, 0.f, 0.f, 0.f};
  int32_t k = start;
  for (; k + 4 <= end; k += 4) {
    int32_t i0 = k, i1 = k + 1, i2 = k + 2, i3 = k + 3;
    if (HAS_PERM) { i0 = perm[k]; i1 = perm[k + 1]; i2 = perm[k + 2]; i3 = perm[k + 3]; }
    const f4 r0 = ld4(src + (int64_t)i0 * ld_src + col);
    const f4 r1 = ld4(src + (int64_t)i1 * ld_src + col);
    const f4 r2 = ld4(src + (int64_t)i2 * ld_src + col);
    const f4 r3 = ld4(src + (int64_t)i3 * ld_src + col);
    acc4(a, r0); acc4(a, r1); acc4(a, r2); acc4(a, r3);  // ascending k: reference edge order
  }
  for (; k < end; ++k) {
    const int32_t i0 = HAS_PERM ? perm[k] : k;
    acc4(a, ld4(src + (int64_t)i0 * ld_src + col));
  }
  st4(out + v * ld_out + col, a);
}

// Fix-up of the fused aggregation epilogue (mlp_resident.hip) in ONE launch: blocks [0, fix_blocks) sum the few
// destinations cut by a wave's row-range boundary from the stored rows, one 16-lane group (4 floats per lane and 64
// features) each; the blocks behind them give the destinations without any row their zeros (one thread per destination;
// far cheaper than a memset of the whole [N, D] aggregate in front of the edge kernel).
// VEC: whole 16-B pieces (feat_dim, ld_src % 4 == 0, aligned base); eight rows are requested before the first is added -
// a destination's rows are still ADDED in ascending order starting from 0.0, as K1 does, only their loads overlap
// (one dependent load per row made this 16 us for a dozen rows per destination).
template <bool VEC>
__global__ __launch_bounds__(gnc::kBlock) void agg_fixup_kernel(const float* __restrict__ src, int64_t ld_src,
                                                                const int32_t* __restrict__ rowptr,
                                                                const int32_t* __restrict__ fix, int32_t n_fix, int32_t fix_blocks,
                                                                int32_t num_nodes, int32_t feat_dim,
                                                                float* __restrict__ out, int64_t ld_out) {
  if ((int)blockIdx.x >= fix_blocks) {
    const int64_t v = (int64_t)(blockIdx.x - fix_blocks) * blockDim.x + threadIdx.x;
    if (v >= num_nodes || rowptr[v] != rowptr[v + 1]) return;
    for (int c = 0; c < feat_dim; ++c) out[v * ld_out + c] = 0.f;
    return;
  }
  const int j = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 4);
  const int col = (threadIdx.x & 15) * 4;
  if (j >= n_fix) return;
  const int32_t v = fix[j];
  if (v < 0 || v >= num_nodes) return;
  const int32_t k0 = rowptr[v], k1 = rowptr[v + 1];
  if (k0 == k1) return;  // no rows: the zero blocks write it
  for (int c = col; c < feat_dim; c += 64) {
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (VEC) {
      int32_t k = k0;
      for (; k + 8 <= k1; k += 8) {
        float4 r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) r[u] = *reinterpret_cast<const float4*>(src + (int64_t)(k + u) * ld_src + c);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          a[0] += r[u].x; a[1] += r[u].y; a[2] += r[u].z; a[3] += r[u].w;
        }
      }
      float4 r[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int32_t kk = k + u < k1 ? k + u : k1 - 1;  // clamped: every load is legal, the surplus is not added
        r[u] = *reinterpret_cast<const float4*>(src + (int64_t)kk * ld_src + c);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (k + u < k1) {
          a[0] += r[u].x; a[1] += r[u].y; a[2] += r[u].z; a[3] += r[u].w;
        }
    } else {
      for (int32_t k = k0; k < k1; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (c + q < feat_dim) a[q] += src[(int64_t)k * ld_src + c + q];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (c + q < feat_dim) out[(int64_t)v * ld_out + c + q] = a[q];
  }
}

// K1, generic: any feat_dim / alignment; one wave per destination, scalar columns.
template <bool HAS_PERM>
__global__ __launch_bounds__(gnc::kBlock) void scatter_sum_csr_scalar(
    const float* __restrict__ src, int64_t ld_src, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ perm, int32_t num_nodes, int32_t feat_dim, float* __restrict__ out,
    int64_t ld_out) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave;
  const int64_t num_waves = ((int64_t)gridDim.x * blockDim.x) / kWave;
  for (int64_t v = wave; v < num_nodes; v += num_waves) {
    const int32_t start = rowptr[v], end = rowptr[v + 1];
    for (int c = lane; c < feat_dim; c += kWave) {
      float a = 0.f;
      for (int32_t k = start; k < end; ++k) {
        const int32_t i = HAS_PERM ? perm[k] : k;
        a += src[(int64_t)i * ld_src + c];
      }
      out[v * ld_out + c] = a;
    }
  }
}

// ---------------------------------------------------------------------------------------
// K2 row gather: out[r, :] = table[index[r], :]
// ---------------------------------------------------------------------------------------
template <int LPR, bool ADD = false>
__global__ __launch_bounds__(gnc::kBlock) void gather_rows_vec4(const float* __restrict__ table, int64_t ld_table,
                                                                const int32_t* __restrict__ index,
                                                                int64_t num_rows, int32_t feat_dim,
                                                                float* __restrict__ out, int64_t ld_out,
                                                                const float* __restrict__ addend = nullptr, int64_t ld_add = 0) {
  constexpr int RPB = gnc::kBlock / LPR;  // rows per block pass
  const int sub = threadIdx.x % LPR;
  const int col = sub * 4;
  if (col >= feat_dim) return;
  int64_t r = (int64_t)blockIdx.x * RPB + threadIdx.x / LPR;
  const int64_t stride = (int64_t)gridDim.x * RPB;
  // two rows in flight per lane
  for (; r + stride < num_rows; r += 2 * stride) {
    const int32_t i0 = index[r], i1 = index[r + stride];
    f4 a = ld4(table + (int64_t)i0 * ld_table + col);
    f4 b = ld4(table + (int64_t)i1 * ld_table + col);
    if (ADD) {
      acc4(a, ld4_stream(addend + r * ld_add + col));
      acc4(b, ld4_stream(addend + (r + stride) * ld_add + col));
    }
    st4(out + r * ld_out + col, a);
    st4(out + (r + stride) * ld_out + col, b);
  }
  if (r < num_rows) {
    f4 a = ld4(table + (int64_t)index[r] * ld_table + col);
    if (ADD) acc4(a, ld4_stream(addend + r * ld_add + col));
    st4(out + r * ld_out + col, a);
  }
}

template <bool ADD = false>
__global__ __launch_bounds__(gnc::kBlock) void gather_rows_scalar(const float* __restrict__ table, int64_t ld_table,
                                                                  const int32_t* __restrict__ index,
                                                                  int64_t num_rows, int32_t feat_dim,
                                                                  float* __restrict__ out, int64_t ld_out,
                                                                  const float* __restrict__ addend = nullptr, int64_t ld_add = 0) {
  const int64_t total = num_rows * feat_dim;
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; t < total; t += stride) {
    const int64_t r = t / feat_dim;
    const int c = (int)(t - r * feat_dim);
    float v = table[(int64_t)index[r] * ld_table + c];
    if (ADD) v += addend[r * ld_add + c];
    out[r * ld_out + c] = v;
  }
}

// ---------------------------------------------------------------------------------------
// K6 edge features: out[e] = [pos[dst]-pos[src], sum|pos[dst]-pos[src]|]
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(gnc::kBlock) void edge_features_kernel(const float* __restrict__ pos, int32_t space_dim,
                                                                    const int32_t* __restrict__ src,
                                                                    const int32_t* __restrict__ dst,
                                                                    int64_t num_edges, float* __restrict__ out,
                                                                    int64_t od) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; e < num_edges; e += stride) {
    const float* ps = pos + (int64_t)src[e] * space_dim;
    const float* pd = pos + (int64_t)dst[e] * space_dim;
    float dist = 0.f;
    for (int d = 0; d < space_dim; ++d) {
      const float rel = pd[d] - ps[d];
      out[e * od + d] = rel;
      dist += fabsf(rel);  // summed in dimension order like torch.sum(dim=1) on a short row
    }
    out[e * od + space_dim] = dist;
    for (int64_t d = space_dim + 1; d < od; ++d) out[e * od + d] = 0.f;
  }
}

int pow2_lanes_for(int feat_dim) {  // smallest power-of-two lane count whose float4 slices cover a row
  int lanes = 1;
  while (lanes * 4 < feat_dim) lanes *= 2;
  return lanes;
}

bool vec4_ok(const void* p, int64_t ld, int feat_dim) {
  return feat_dim % 4 == 0 && feat_dim <= 256 && ld % 4 == 0 && gnc::aligned16(p);
}

template <bool HAS_PERM>
int launch_scatter(const float* src, int64_t ld_src, const int32_t* rowptr, const int32_t* perm, int32_t n,
                   int32_t d, float* out, int64_t ld_out, hipStream_t stream) {
  const int64_t chunks = gnc::ceil_div(n, kChunk);
  const int waves_per_block = gnc::kBlock / kWave;
  if (vec4_ok(src, ld_src, d) && vec4_ok(out, ld_out, d)) {
    static const bool no_small = getenv("GNC_NO_SMALL_K1") != nullptr;  // A/B switch
    if (!no_small && n <= 64 * gnc::num_cu()) {  // a small graph: one lane group per destination
      const int lpr = pow2_lanes_for(d);
      const int64_t threads = (int64_t)n * lpr;
      dim3 grid((unsigned)gnc::ceil_div(threads, (int64_t)gnc::kBlock)), block(gnc::kBlock);
      switch (lpr) {
#define GNC_CASE(L)                                                                                                \
  case L:                                                                                                          \
    scatter_sum_csr_small<L, HAS_PERM><<<grid, block, 0, stream>>>(src, ld_src, rowptr, perm, n, d, out, ld_out); \
    break;
        GNC_CASE(1) GNC_CASE(2) GNC_CASE(4) GNC_CASE(8) GNC_CASE(16) GNC_CASE(32) GNC_CASE(64)
#undef GNC_CASE
        default:
          gnc::set_error("scatter_sum: internal lane mapping error for D=%d", d);
          return GNC_ERR_UNSUPPORTED;
      }
      return gnc::check_launch("scatter_sum_csr_small");
    }
    int64_t blocks = gnc::ceil_div(chunks, waves_per_block);
    const int64_t cap = gnc::num_cu() * 16;
    if (blocks > cap) blocks = cap;
    dim3 grid((unsigned)blocks), block(gnc::kBlock);
    switch (pow2_lanes_for(d)) {
#define GNC_CASE(L)                                                                                               \
  case L:                                                                                                         \
    scatter_sum_csr_vec4<L, HAS_PERM><<<grid, block, 0, stream>>>(src, ld_src, rowptr, perm, n, d, out, ld_out); \
    break;
      GNC_CASE(1) GNC_CASE(2) GNC_CASE(4) GNC_CASE(8) GNC_CASE(16) GNC_CASE(32) GNC_CASE(64)
#undef GNC_CASE
      default:
        gnc::set_error("scatter_sum: internal lane mapping error for D=%d", d);
        return GNC_ERR_UNSUPPORTED;
    }
    return gnc::check_launch("scatter_sum_csr_vec4");
  }
  int64_t blocks = gnc::ceil_div(n, waves_per_block);
  const int64_t cap = gnc::num_cu() * 16;
  if (blocks > cap) blocks = cap;
  scatter_sum_csr_scalar<HAS_PERM><<<dim3((unsigned)blocks), dim3(gnc::kBlock), 0, stream>>>(src, ld_src, rowptr, perm,
                                                                                              n, d, out, ld_out);
  return gnc::check_launch("scatter_sum_csr_scalar");
}

}  // namespace

extern "C" int gnc_scatter_sum_csr_f32(const float* src, int64_t ld_src, const int32_t* rowptr, const int32_t* perm,
                                       int64_t num_nodes, int64_t num_edges, int32_t feat_dim, float* out,
                                       int64_t ld_out, void* stream_) {
  GNC_REQUIRE(num_nodes >= 0 && num_edges >= 0 && feat_dim >= 0, "gnc_scatter_sum_csr_f32: negative size");
  GNC_REQUIRE(num_nodes < INT32_MAX && num_edges < INT32_MAX, "gnc_scatter_sum_csr_f32: sizes exceed int32");
  if (num_nodes == 0 || feat_dim == 0) return GNC_OK;
  GNC_REQUIRE(rowptr && out, "gnc_scatter_sum_csr_f32: null rowptr/out");
  GNC_REQUIRE(num_edges == 0 || src, "gnc_scatter_sum_csr_f32: null src");
  GNC_REQUIRE(ld_src >= feat_dim && ld_out >= feat_dim, "gnc_scatter_sum_csr_f32: leading dimension < feat_dim");
  hipStream_t stream = (hipStream_t)stream_;
  if (perm)
    return launch_scatter<true>(src, ld_src, rowptr, perm, (int32_t)num_nodes, feat_dim, out, ld_out, stream);
  return launch_scatter<false>(src, ld_src, rowptr, nullptr, (int32_t)num_nodes, feat_dim, out, ld_out, stream);
}

extern "C" int gnc_agg_fixup_f32(const float* src, int64_t ld_src, const int32_t* rowptr, const int32_t* fix, int32_t n_fix,
                                 int64_t num_nodes, int32_t feat_dim, float* out, int64_t ld_out, void* stream_) {
  GNC_REQUIRE(n_fix >= 0 && num_nodes >= 0 && num_nodes < INT32_MAX && feat_dim >= 0, "gnc_agg_fixup_f32: bad sizes");
  if (feat_dim == 0 || num_nodes == 0) return GNC_OK;
  GNC_REQUIRE(rowptr && out && (n_fix == 0 || (src && fix)), "gnc_agg_fixup_f32: null pointer");
  GNC_REQUIRE(ld_src >= feat_dim && ld_out >= feat_dim, "gnc_agg_fixup_f32: leading dimension < feat_dim");
  const int64_t fix_blocks = n_fix > 0 ? gnc::ceil_div((int64_t)n_fix * 16, gnc::kBlock) : 0;
  const int64_t blocks = fix_blocks + gnc::ceil_div(num_nodes, gnc::kBlock);
  const bool vec = feat_dim % 4 == 0 && ld_src % 4 == 0 && gnc::aligned16(src);
  if (vec)
    agg_fixup_kernel<true><<<dim3((unsigned)blocks), dim3(gnc::kBlock), 0, (hipStream_t)stream_>>>(
        src, ld_src, rowptr, fix, n_fix, (int32_t)fix_blocks, (int32_t)num_nodes, feat_dim, out, ld_out);
  else
    agg_fixup_kernel<false><<<dim3((unsigned)blocks), dim3(gnc::kBlock), 0, (hipStream_t)stream_>>>(
        src, ld_src, rowptr, fix, n_fix, (int32_t)fix_blocks, (int32_t)num_nodes, feat_dim, out, ld_out);
  return gnc::check_launch("agg_fixup_kernel");
}

namespace {
template <bool ADD>
int launch_gather(const float* table, int64_t ld_table, const int32_t* index, const float* addend, int64_t ld_add,
                  int64_t num_rows, int32_t feat_dim, float* out, int64_t ld_out, hipStream_t stream) {
  const int64_t cap = gnc::num_cu() * 16;
  if (vec4_ok(table, ld_table, feat_dim) && vec4_ok(out, ld_out, feat_dim) && (!ADD || vec4_ok(addend, ld_add, feat_dim))) {
    const int lpr = pow2_lanes_for(feat_dim);
    int64_t blocks = gnc::ceil_div(num_rows, 2 * (gnc::kBlock / lpr));
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    dim3 grid((unsigned)blocks), block(gnc::kBlock);
    switch (lpr) {
#define GNC_CASE(L)                                                                                                            \
  case L:                                                                                                                      \
    gather_rows_vec4<L, ADD><<<grid, block, 0, stream>>>(table, ld_table, index, num_rows, feat_dim, out, ld_out, addend, ld_add); \
    break;
      GNC_CASE(1) GNC_CASE(2) GNC_CASE(4) GNC_CASE(8) GNC_CASE(16) GNC_CASE(32) GNC_CASE(64)
#undef GNC_CASE
      default:
        gnc::set_error("gather_rows: internal lane mapping error for D=%d", feat_dim);
        return GNC_ERR_UNSUPPORTED;
    }
    return gnc::check_launch("gather_rows_vec4");
  }
  int64_t blocks = gnc::ceil_div(num_rows * feat_dim, gnc::kBlock);
  if (blocks > cap) blocks = cap;
  gather_rows_scalar<ADD><<<dim3((unsigned)blocks), dim3(gnc::kBlock), 0, stream>>>(table, ld_table, index, num_rows, feat_dim,
                                                                                     out, ld_out, addend, ld_add);
  return gnc::check_launch("gather_rows_scalar");
}
}  // namespace

extern "C" int gnc_gather_rows_f32(const float* table, int64_t ld_table, const int32_t* index, int64_t num_rows,
                                   int32_t feat_dim, float* out, int64_t ld_out, void* stream_) {
  GNC_REQUIRE(num_rows >= 0 && feat_dim >= 0, "gnc_gather_rows_f32: negative size");
  if (num_rows == 0 || feat_dim == 0) return GNC_OK;
  GNC_REQUIRE(table && index && out, "gnc_gather_rows_f32: null pointer");
  GNC_REQUIRE(ld_table >= feat_dim && ld_out >= feat_dim, "gnc_gather_rows_f32: leading dimension < feat_dim");
  return launch_gather<false>(table, ld_table, index, nullptr, 0, num_rows, feat_dim, out, ld_out, (hipStream_t)stream_);
}

extern "C" int gnc_gather_rows_add_f32(const float* table, int64_t ld_table, const int32_t* index, const float* addend,
                                       int64_t ld_addend, int64_t num_rows, int32_t feat_dim, float* out, int64_t ld_out,
                                       void* stream_) {
  GNC_REQUIRE(num_rows >= 0 && feat_dim >= 0, "gnc_gather_rows_add_f32: negative size");
  if (num_rows == 0 || feat_dim == 0) return GNC_OK;
  GNC_REQUIRE(table && index && addend && out, "gnc_gather_rows_add_f32: null pointer");
  GNC_REQUIRE(ld_table >= feat_dim && ld_out >= feat_dim && ld_addend >= feat_dim,
              "gnc_gather_rows_add_f32: leading dimension < feat_dim");
  return launch_gather<true>(table, ld_table, index, addend, ld_addend, num_rows, feat_dim, out, ld_out, (hipStream_t)stream_);
}

// deferred validation (topology.py): a result computed from an edge_index with out-of-range ids must not look plausible.
// One launch that returns at once unless a flag is set (torch.where(flags.any(), nan, out) is three launches per forward).
__global__ __launch_bounds__(gnc::kBlock) void poison_if_flagged_kernel(float* __restrict__ out, int64_t count,
                                                                        const int32_t* __restrict__ flags, int nflags) {
  int any = 0;
  for (int k = 0; k < nflags; ++k) any |= flags[k];
  if (!any) return;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < count; t += (int64_t)gridDim.x * blockDim.x)
    out[t] = __builtin_nanf("");
}

extern "C" int gnc_poison_if_flagged_f32(float* out, int64_t count, const int32_t* flags, int32_t nflags, void* stream_) {
  GNC_REQUIRE(count >= 0 && nflags >= 0 && nflags <= 16, "gnc_poison_if_flagged_f32: bad sizes");
  if (count == 0 || nflags == 0) return GNC_OK;
  GNC_REQUIRE(out && flags, "gnc_poison_if_flagged_f32: null pointer");
  int64_t blocks = gnc::ceil_div(count, gnc::kBlock);
  const int64_t cap = gnc::num_cu() * 4;
  if (blocks > cap) blocks = cap;
  poison_if_flagged_kernel<<<dim3((unsigned)blocks), dim3(gnc::kBlock), 0, (hipStream_t)stream_>>>(out, count, flags, nflags);
  return gnc::check_launch("poison_if_flagged_kernel");
}

extern "C" int gnc_edge_features_f32(const float* pos, int32_t space_dim, const int32_t* src, const int32_t* dst,
                                     int64_t num_edges, float* out, int64_t ld_out, void* stream_) {
  GNC_REQUIRE(num_edges >= 0 && space_dim >= 1 && space_dim <= 16 && ld_out >= space_dim + 1,
              "gnc_edge_features_f32: bad sizes");
  if (num_edges == 0) return GNC_OK;
  GNC_REQUIRE(pos && src && dst && out, "gnc_edge_features_f32: null pointer");
  int64_t blocks = gnc::ceil_div(num_edges, gnc::kBlock);
  const int64_t cap = gnc::num_cu() * 16;
  if (blocks > cap) blocks = cap;
  edge_features_kernel<<<dim3((unsigned)blocks), dim3(gnc::kBlock), 0, (hipStream_t)stream_>>>(pos, space_dim, src, dst,
                                                                                               num_edges, out, ld_out);
  return gnc::check_launch("edge_features_kernel");
}
