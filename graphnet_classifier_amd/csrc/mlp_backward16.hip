// K8 data kernel for widths 129..256: the 16-row formulation of mlp_stream16.hip run backwards.
//
// Same contract as mlp_backward_stream_kernel (mlp_backward.hip: recompute the forward of the tile, LayerNorm backward,
// da_l = W_{l+1}^T dz_{l+1}, dz_l = da_l * (a_l > 0), dx = W_0^T dz_0; emits a_l, dz_l, y_hat, dx as whole rows), but
// at 256 features the 32-row form would need four 128-register accumulator sets.  With v_mfma_f32_16x16x4_f32 a wave
// owns 16 data rows, one set is 64 registers, the two live ones (hid / next going forward, dz / da going backward) fit
// next to the weight-chunk prefetch, and 8 waves run per CU.  Post-activation tensors leave the kernel as soon as they
// are formed; what stays of them is one bit per value (the ReLU mask).
//
// Transposed products.  Accumulator register s of tile t on lane (i = lane & 15, g = lane >> 4) holds feature
// n = 16t + 4g + s of data row i, which is the B operand (k slot g) of an MFMA whose A operand is
// W[n][64c + 16cb + i]: read down a column of the staged [n][64 k] chunk (ds_read_b32, the four lane groups hit
// four different banks groups: conflict-free with the 68-float row pitch).  The chunk sequence of a tile - forward
// chunks, the last Linear for the LayerNorm statistics, then the same chunks in reverse layer order, then the first
// Linear's again for dx - streams through the single 70 KB LDS buffer exactly as in the forward kernel (register
// prefetch under the MFMAs of the previous chunk, two barriers per chunk).
#include <stdlib.h>

#include <type_traits>

#include "mlp_device16.h"

using namespace gnc_mlp;

// Phase probe (developer build: make probe_b16, tools/profile_bwd16.py): cycles per phase of the tile loop, per wave.
#ifdef GNC_PHASE_PROBE
__device__ unsigned long long gnc_phase_probe_b16[4096 * 12];
extern "C" int gnc_phase_probe_b16_read(void* dst, size_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(gnc_phase_probe_b16), bytes); }
#define B16P_BEGIN() unsigned long long pr_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pr_tp, pr_tn; \
  const unsigned long long pr_c0 = __builtin_readcyclecounter(), pr_w0 = wall_clock64()
#define B16P_TILE() pr_tp = __builtin_readcyclecounter()
#define B16P(k) do { pr_tn = __builtin_readcyclecounter(); pr_acc[k] += pr_tn - pr_tp; pr_tp = pr_tn; } while (0)
#define B16P_END() do { if ((threadIdx.x & 63) == 0) {                                                          \
    unsigned long long* o = gnc_phase_probe_b16 + (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 12;  \
    for (int k = 0; k < 8; ++k) o[k] = pr_acc[k];                                                               \
    o[8] = __builtin_readcyclecounter() - pr_c0; o[9] = wall_clock64() - pr_w0; } } while (0)
#else
#define B16P_BEGIN() do {} while (0)
#define B16P_TILE() do {} while (0)
#define B16P(k) do {} while (0)
#define B16P_END() do {} while (0)
#endif

namespace {

constexpr int B16_MAX_STEPS = 16;
constexpr int B16_MAX_WCHUNKS = 72;

struct BPlan16 {
  int num_steps;    // first-Linear staging steps per tile
  int num_wchunks;  // weight chunks per tile (forward + backward + dx)
  int saved;        // 1: b.act[] are the forward's saved post-activations (inputs): no forward chunks but the last Linear's
  struct { short seg, c0, add, pad; } step[B16_MAX_STEPS];
  struct { short layer, kbase, klimit, pad; } wc[B16_MAX_WCHUNKS];
};

// ReLU in place; bit (4 * (t % 8) + r) of m[t / 8] = value was positive
template <int NTL>
__device__ __forceinline__ void relu_mask16(f32x4 (&acc)[NTL], unsigned (&m)[(NTL + 7) / 8]) {
#pragma unroll
  for (int w = 0; w < (NTL + 7) / 8; ++w) {
    unsigned bits = 0;
#pragma unroll
    for (int tt = 0; tt < 8; ++tt) {
      const int t = 8 * w + tt;
      if (t < NTL) {
        float* v = reinterpret_cast<float*>(&acc[t < NTL ? t : 0]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool pos = v[r] > 0.f;
          bits |= pos ? (1u << (4 * tt + r)) : 0u;
          v[r] = pos ? v[r] : 0.f;
        }
      }
    }
    m[w] = bits;
  }
}

template <int NTL>
__device__ __forceinline__ void apply_mask16(f32x4 (&acc)[NTL], const unsigned (&m)[(NTL + 7) / 8]) {
#pragma unroll
  for (int t = 0; t < NTL; ++t) {
    float* v = reinterpret_cast<float*>(&acc[t]);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = (m[t / 8] >> (4 * (t % 8) + r)) & 1u ? v[r] : 0.f;
  }
}

// dst[DBASE + cb] += (columns [16 cb, 16 cb + 16) of the staged chunk)^T * src for cb = 0..3: contraction over the
// chunk's rows n (the NTI tiles of src)
template <int NTI, int NTO, int DBASE, int WINDOW = 4>
__device__ __forceinline__ void mma16_transposed_chunk(f32x4 (&dst)[NTO], const f32x4 (&src)[NTI], const float* wbuf, int i,
                                                       int g) {
  const float* col = wbuf + (4 * g) * LDSW + i;
#pragma unroll
  for (int t = 0; t < NTI; ++t) {
    const float bs[4] = {src[t].x, src[t].y, src[t].z, src[t].w};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        if constexpr (DBASE + 3 < NTO) {
          dst[DBASE + cb] = mfma16(col[(16 * t + s) * LDSW + 16 * cb], bs[s], dst[DBASE + cb]);
        }
      }
    }
    if ((t & (WINDOW - 1)) == WINDOW - 1) __builtin_amdgcn_sched_barrier(0);  // a bounded window of column reads in flight (16 per t)
  }
}

// LayerNorm backward on the 16-row accumulator layout (see layer_norm_backward_tiles in mlp_device.h):
//   in : y = pre-LayerNorm output (features >= out_dim exact zeros), gr = grad wrt the LayerNorm output
//   out: y = y_hat (normalised, pre-affine), gr = grad wrt the pre-LayerNorm output
template <int NTL>
__device__ __forceinline__ void layer_norm_backward16(f32x4 (&y)[NTL], f32x4 (&gr)[NTL], const float* pg, int out_dim, float eps,
                                                      int g) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const float inv_n = 1.f / (float)out_dim;
  const bool padded = out_dim != NTL * 16;
  f32x2 sa = {0.f, 0.f}, sb = {0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NTL; ++t) {
    sa += f32x2{y[t].x, y[t].y};
    sb += f32x2{y[t].z, y[t].w};
  }
  const float mean = add_quarters((sa.x + sa.y) + (sb.x + sb.y)) * inv_n;
  const f32x4 m4 = {mean, mean, mean, mean};
#pragma unroll
  for (int t = 0; t < NTL; ++t) y[t] -= m4;
  if (padded) {
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
      const int f = 16 * t + 4 * g;
      y[t].x = f + 0 < out_dim ? y[t].x : 0.f; y[t].y = f + 1 < out_dim ? y[t].y : 0.f;
      y[t].z = f + 2 < out_dim ? y[t].z : 0.f; y[t].w = f + 3 < out_dim ? y[t].w : 0.f;
      gr[t].x = f + 0 < out_dim ? gr[t].x : 0.f; gr[t].y = f + 1 < out_dim ? gr[t].y : 0.f;
      gr[t].z = f + 2 < out_dim ? gr[t].z : 0.f; gr[t].w = f + 3 < out_dim ? gr[t].w : 0.f;
    }
  }
  f32x2 va = {0.f, 0.f}, vb = {0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NTL; ++t) {
    const f32x2 da = {y[t].x, y[t].y}, db = {y[t].z, y[t].w};
    va = __builtin_elementwise_fma(da, da, va);
    vb = __builtin_elementwise_fma(db, db, vb);
  }
  const float rstd = __frsqrt_rn(add_quarters((va.x + va.y) + (vb.x + vb.y)) * inv_n + eps);
  const f32x4 r4 = {rstd, rstd, rstd, rstd};
  f32x4 m1v = {0.f, 0.f, 0.f, 0.f}, m2v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NTL; ++t) {
    const f32x4 gm = *reinterpret_cast<const f32x4*>(pg + 16 * t + 4 * g);  // gamma, zero padded
    y[t] = y[t] * r4;        // y_hat
    gr[t] = gr[t] * gm;      // g * gamma
    m1v += gr[t];
    m2v = __builtin_elementwise_fma(gr[t], y[t], m2v);
  }
  const float m1 = add_quarters((m1v.x + m1v.y) + (m1v.z + m1v.w)) * inv_n;
  const float m2 = add_quarters((m2v.x + m2v.y) + (m2v.z + m2v.w)) * inv_n;
  const f32x4 nm1 = {-m1, -m1, -m1, -m1}, nm2 = {-m2, -m2, -m2, -m2};
#pragma unroll
  for (int t = 0; t < NTL; ++t) gr[t] = __builtin_elementwise_fma(y[t], nm2, gr[t] + nm1) * r4;  // rstd * (g - m1 - y_hat * m2)
  if (padded) {
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
      const int f = 16 * t + 4 * g;
      gr[t].x = f + 0 < out_dim ? gr[t].x : 0.f; gr[t].y = f + 1 < out_dim ? gr[t].y : 0.f;
      gr[t].z = f + 2 < out_dim ? gr[t].z : 0.f; gr[t].w = f + 3 < out_dim ? gr[t].w : 0.f;
    }
  }
}

// NTL: 16-feature tiles of the widest layer (hidden / output): 16 for 129..256 features
// SAVED (= pl.saved, as a template flag: the run-time branch alone cost the recomputing instance 460 B of scratch)
template <int NTL, bool SAVED = false>
__global__ __launch_bounds__(NT16) void mlp_backward_stream16_kernel(const gnc_mlp_desc_t d, const BwdArgs b, const BPlan16 pl,
                                                                      const int num_tiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int WROWS = NTL * 16;
  constexpr int CH = WROWS * LDSW;
  constexpr int PSTRIDE = WROWS;
  constexpr int RPP = NT16 / 16;
  constexpr int NW = WROWS / RPP;
  constexpr int NCH = (NTL + 3) / 4;
  constexpr int MW = (NTL + 7) / 8;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15;
  const int g = lane >> 4;
  const int c4 = lane & 15;
  const int rs = lane >> 4;
  const int wc4 = tid & 15;
  const int wr0 = tid >> 4;
  const int L = d.num_linear;
  const int out_dim = d.out_dim[L - 1];
  const int rows = (int)d.rows;
  float* wbuf = lds;
  float* pbuf = lds + CH;
  float* abuf = pbuf + (L + 2) * PSTRIDE + wave * R16 * LDSW;
  // LayerNorm parameter sums of this wave's rows (b.ln_partial): [colsum(grad_out) | colsum(grad_out * y_hat)], in LDS
  float* lnbuf = pbuf + (L + 2) * PSTRIDE + W16 * R16 * LDSW + wave * 2 * PSTRIDE;
  if (b.ln_partial) {
    for (int j = lane; j < 2 * PSTRIDE; j += 64) lnbuf[j] = 0.f;
  }

  stage_params<NT16>(pbuf, d, PSTRIDE, tid);

  auto wload = [&](f32x4 (&wr)[NW], int q) {
    const int layer = pl.wc[q].layer;
    const float* W = d.weight[layer];
    const int ldw = ldw_of(d, layer);
    const int nrows = d.out_dim[layer];
    const int klimit = pl.wc[q].klimit;
    const int kbase = pl.wc[q].kbase;
    if (kbase + KC <= klimit && (int64_t)nrows * ldw * 4 <= 0xffffffffll) {
      // full 64-column chunk (the plan guarantees aligned weights): window over the matrix, rows >= nrows read as 0
      const __amdgpu_buffer_rsrc_t w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, nrows * ldw * 4, 0x00020000);
      const uint32_t off0 = (uint32_t)(wr0 * ldw + kbase + wc4 * 4) * 4u;
#pragma unroll
      for (int p = 0; p < NW; ++p) wr[p] = window_load(w, off0 + (uint32_t)(p * RPP * 4) * (uint32_t)ldw);
      return;
    }
    const int col = kbase + wc4 * 4;
    const int colc = col < klimit ? col : kbase;
#pragma unroll
    for (int p = 0; p < NW; ++p) {
      const int n = p * RPP + wr0;
      const int nc = n < nrows ? n : nrows - 1;
      f32x4 v = *reinterpret_cast<const f32x4*>(W + (int64_t)nc * ldw + colc);
      const bool rowok = n < nrows;
      v.x = (rowok && col + 0 < klimit) ? v.x : 0.f; v.y = (rowok && col + 1 < klimit) ? v.y : 0.f;
      v.z = (rowok && col + 2 < klimit) ? v.z : 0.f; v.w = (rowok && col + 3 < klimit) ? v.w : 0.f;
      wr[p] = v;
    }
  };
  auto wstore = [&](const f32x4 (&wr)[NW]) {
#pragma unroll
    for (int p = 0; p < NW; ++p) *reinterpret_cast<f32x4*>(wbuf + (p * RPP + wr0) * LDSW + wc4 * 4) = wr[p];
  };
  const int last_tile = num_tiles - 1;
  auto load_idx = [&](int tile, int s) -> int {
    const int tc = tile < last_tile ? tile : last_tile;
    int r = (tc * W16 + wave) * R16 + (lane & 15);
    r = r < rows ? r : rows - 1;
    const int32_t* ip = d.seg[s].index;
    return ip ? ip[r] : r;
  };
  // rows of one 64-column slab of a table: row-ordered tables through a window at the wave's first row (rows past the
  // end read 0), gathered tables through a window over the stated table (< 4 GiB: guaranteed by the support query)
  auto load_rows = [&](f32x4 (&pre)[NP16], const float* base, int ld, const int32_t* index, int64_t table_rows, int c0, int idxv,
                       int tile_of) {
    const int col = c0 + c4 * 4 < ld ? c0 + c4 * 4 : 0;
    if (index == nullptr) {
      load_tile_rows<false, NP16>(pre, base, ld, ((int64_t)tile_of * W16 + wave) * R16, rows, (uint32_t)(rs * ld + col) * 4u);
    } else {
      const __amdgpu_buffer_rsrc_t w =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)(uint32_t)(table_rows * (int64_t)ld * 4), 0x00020000);
      const int row_bytes = idxv * (ld * 4);
#pragma unroll
      for (int p = 0; p < NP16; ++p) pre[p] = window_load(w, (uint32_t)__shfl(row_bytes, p * 4 + rs, 64) + (uint32_t)(col * 4));
    }
  };
  auto stage = [&](const f32x4 (&pre)[NP16], int c0, int width) {
    compiler_lds_barrier();
    const int c = c0 + c4 * 4;
    if (c0 + KC <= width) {
#pragma unroll
      for (int p = 0; p < NP16; ++p) *reinterpret_cast<f32x4*>(abuf + (p * 4 + rs) * LDSW + c4 * 4) = pre[p];
    } else {
#pragma unroll
      for (int p = 0; p < NP16; ++p) {
        f32x4 v = pre[p];
        v.x = c + 0 < width ? v.x : 0.f; v.y = c + 1 < width ? v.y : 0.f;
        v.z = c + 2 < width ? v.z : 0.f; v.w = c + 3 < width ? v.w : 0.f;
        *reinterpret_cast<f32x4*>(abuf + (p * 4 + rs) * LDSW + c4 * 4) = v;
      }
    }
    compiler_lds_barrier();
  };
  // accumulator tiles -> whole rows of dst ([rows, ld], columns < width), 64 columns at a time through the wave's tile
  auto emit = [&](const f32x4 (&acc)[NTL], float* dst, int ld, int width, int row0) {
#pragma unroll
    for (int cc = 0; cc < NCH; ++cc) {
      if (cc * KC < width) {
        compiler_lds_barrier();
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
          if (4 * cc + cb < NTL) *reinterpret_cast<f32x4*>(abuf + i * LDSW + 16 * cb + 4 * g) = acc[4 * cc + cb < NTL ? 4 * cc + cb : 0];
        compiler_lds_barrier();
        store_staged_rows<NP16, false>(abuf, dst + cc * KC, ld, width - cc * KC < KC ? width - cc * KC : KC, row0, rows, c4, rs);
      }
    }
    compiler_lds_barrier();
  };

  f32x4 wreg[NW];
  wload(wreg, 0);
  wstore(wreg);
  __syncthreads();
  auto prefetch_next_chunk = [&](int q) { wload(wreg, q + 1 < pl.num_wchunks ? q + 1 : 0); };
  auto publish_next_chunk = [&]() {
    __syncthreads();  // everyone is done reading the buffer
    wstore(wreg);
    __syncthreads();
  };

  // the row stream of a tile: the first Linear's staging steps, then the grad_out slabs; item k + 1 is requested while
  // item k is being consumed, the next tile's first item while the last slab of this one is
  int ids[GNC_MAX_SEGMENTS], ids_next[GNC_MAX_SEGMENTS];
  int tile = blockIdx.x;
#pragma unroll
  for (int s = 0; s < GNC_MAX_SEGMENTS; ++s) {
    ids[s] = s < d.num_segments ? load_idx(tile, s) : 0;
    ids_next[s] = s < d.num_segments ? load_idx(tile + (int)gridDim.x, s) : 0;
  }
  auto id_of = [&](int s) {
    int v = 0;
#pragma unroll
    for (int k = 0; k < GNC_MAX_SEGMENTS; ++k) v = k == s ? ids[k] : v;
    return v;
  };
  auto load_step = [&](f32x4 (&pre)[NP16], int st, int tile_of) {
    const gnc_mlp_segment_t& sg = d.seg[pl.step[st].seg];
    load_rows(pre, sg.ptr, sg.ld, sg.index, sg.table_rows, pl.step[st].c0, id_of(pl.step[st].seg), tile_of);
  };
  // SAVED (pl.saved): the row stream of a tile is the slabs of the saved post-activations a_0 .. a_{L-2} (ReLU masks; the
  // last tensor stays in registers for the LayerNorm recompute), then the grad_out slabs; no gathered input is read
  auto load_first = [&](f32x4 (&pre)[NP16], int tile_of) {
    if constexpr (SAVED) load_rows(pre, b.act[0], d.out_dim[0], nullptr, 0, 0, 0, tile_of);
    else load_step(pre, 0, tile_of);
  };
  f32x4 cur[NP16];
  load_first(cur, tile);
  const int n_gslabs = (out_dim + KC - 1) / KC;

  B16P_BEGIN();
  while (tile < num_tiles) {
    B16P_TILE();
    const int row0 = (tile * W16 + wave) * R16;
    const int ntile = tile + gridDim.x;
    int q = 0;

    f32x4 hid[NTL];
    unsigned mask[GNC_MAX_LINEAR - 1][MW];
    if constexpr (SAVED) {
#pragma unroll
      for (int l = 0; l < GNC_MAX_LINEAR - 1; ++l) {
        if (l < L - 1) {
          const int width = d.out_dim[l];
#pragma unroll
          for (int cc = 0; cc < NCH; ++cc) {
            if (cc * KC < width) {
              stage(cur, cc * KC, width);
              // the stream's next item: this tensor's next slab, the next tensor's first, or grad_out's first
              if ((cc + 1) * KC < width) load_rows(cur, b.act[l], width, nullptr, 0, (cc + 1) * KC, 0, tile);
              else if (l + 1 < L - 1) load_rows(cur, b.act[l + 1 < GNC_MAX_LINEAR ? l + 1 : 0], d.out_dim[l + 1 < GNC_MAX_LINEAR ? l + 1 : 0], nullptr, 0, 0, 0, tile);
              else load_rows(cur, b.grad_out, b.ld_grad_out, nullptr, 0, 0, 0, tile);
#pragma unroll
              for (int cb = 0; cb < 4; ++cb)
                if (4 * cc + cb < NTL) hid[4 * cc + cb < NTL ? 4 * cc + cb : 0] = *reinterpret_cast<const f32x4*>(abuf + i * LDSW + 16 * cb + 4 * g);
              compiler_lds_barrier();
            } else {
#pragma unroll
              for (int cb = 0; cb < 4; ++cb)
                if (4 * cc + cb < NTL) hid[4 * cc + cb < NTL ? 4 * cc + cb : 0] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
          }
          relu_mask16<NTL>(hid, mask[l]);  // values are post-ReLU already: this only forms the bit masks
        }
      }
    } else {
    // ------------------------------------------------------------------ forward recompute: first Linear
    init_bias16<NTL>(hid, pbuf, g);
    auto stage_and_advance = [&](int st) {
      stage(cur, pl.step[st].c0, d.seg[pl.step[st].seg].width);
      if (st + 1 < pl.num_steps) load_step(cur, st + 1, tile);
      else load_rows(cur, b.grad_out, b.ld_grad_out, nullptr, 0, 0, 0, tile);  // first grad_out slab
    };
    int st = 0;
    for (; st < pl.num_steps && !pl.step[st].add; ++st) {
      const int c0 = pl.step[st].c0, width = d.seg[pl.step[st].seg].width;
      stage_and_advance(st);
      prefetch_next_chunk(q);
      const int kc = width - c0 < KC ? width - c0 : KC;
      mma16_chunk_from_lds<NTL>(hid, abuf, wbuf, (kc + 15) >> 4, i, g);
      publish_next_chunk();
      ++q;
    }
    // additive steps, slab-major in the plan: static accumulator tile index (see mlp_stream16.hip)
#define GNC_ADD_SLAB(CC_)                                                        \
  if constexpr (CC_ < NCH) {                                                    \
    for (; st < pl.num_steps && pl.step[st].c0 == CC_ * KC; ++st) {               \
      stage_and_advance(st);                                                     \
      add_slab16<NTL, 4 * CC_>(hid, abuf, i, g);                                  \
    }                                                                            \
  }
    GNC_ADD_SLAB(0) GNC_ADD_SLAB(1) GNC_ADD_SLAB(2) GNC_ADD_SLAB(3)
#undef GNC_ADD_SLAB
    relu_mask16<NTL>(hid, mask[0]);
    emit(hid, b.act[0], d.out_dim[0], d.out_dim[0], row0);

    // ------------------------------------------------------------------ hidden layers
#pragma unroll
    for (int l = 1; l < GNC_MAX_LINEAR - 1; ++l) {
      if (l < L - 1) {
        f32x4 nxt[NTL];
        init_bias16<NTL>(nxt, pbuf + l * PSTRIDE, g);
#define GNC_B16_FWD(C_, DST_, LAYER_)                                                \
  if constexpr (C_ < NCH) {                                                          \
    if (C_ * KC < d.in_dim[LAYER_]) {                                                \
      prefetch_next_chunk(q);                                                        \
      mma16_chunk_from_regs<NTL, NTL, C_>(DST_, hid, wbuf, d.in_dim[LAYER_], i, g);  \
      publish_next_chunk();                                                          \
      ++q;                                                                           \
    }                                                                                \
  }
        GNC_B16_FWD(0, nxt, l) GNC_B16_FWD(1, nxt, l) GNC_B16_FWD(2, nxt, l) GNC_B16_FWD(3, nxt, l)
        relu_mask16<NTL>(nxt, mask[l]);
#pragma unroll
        for (int t = 0; t < NTL; ++t) hid[t] = nxt[t];
        emit(hid, b.act[l], d.out_dim[l], d.out_dim[l], row0);
      }
    }
    }  // forward recompute
    B16P(0);  // saved tiles -> masks (or the forward recompute)

    // ------------------------------------------------------------------ grad_out tile (slab by slab), LayerNorm backward
    f32x4 gr[NTL];
#pragma unroll
    for (int cc = 0; cc < NCH; ++cc) {
      if (cc < n_gslabs) {
        stage(cur, cc * KC, out_dim);
        if (cc + 1 < n_gslabs) {
          load_rows(cur, b.grad_out, b.ld_grad_out, nullptr, 0, (cc + 1) * KC, 0, tile);
        } else {  // the next tile's first staging step
#pragma unroll
          for (int k = 0; k < GNC_MAX_SEGMENTS; ++k) {
            ids[k] = ids_next[k];
            if (k < d.num_segments) ids_next[k] = load_idx(ntile + (int)gridDim.x, k);
          }
          load_first(cur, ntile);
        }
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
          if (4 * cc + cb < NTL) gr[4 * cc + cb < NTL ? 4 * cc + cb : 0] = *reinterpret_cast<const f32x4*>(abuf + i * LDSW + 16 * cb + 4 * g);
        compiler_lds_barrier();
      } else {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
          if (4 * cc + cb < NTL) gr[4 * cc + cb < NTL ? 4 * cc + cb : 0] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    B16P(1);  // grad_out slabs
    if (d.ln_gamma) {
      f32x4 y[NTL];
      init_bias16<NTL>(y, pbuf + (L - 1) * PSTRIDE, g);
      GNC_B16_FWD(0, y, L - 1) GNC_B16_FWD(1, y, L - 1) GNC_B16_FWD(2, y, L - 1) GNC_B16_FWD(3, y, L - 1)
      B16P(2);  // last Linear (LayerNorm statistics)
      layer_norm_backward16<NTL>(y, gr, pbuf + L * PSTRIDE, out_dim, d.ln_eps, g);
      if (b.ln_partial) {
        // d beta / d gamma in flight (as in mlp_backward_stream_kernel): per 64-column slab y_hat goes through the wave's
        // LDS tile into whole-row pieces, grad_out's pieces come back through the window (L2 hits; rows past the end read
        // 0), the four row groups are folded and added to the wave's LDS totals.  No y_hat tensor, no colsum pass.
#pragma unroll
        for (int cc = 0; cc < NCH; ++cc) {
          if (cc * KC < out_dim) {
            const int col = cc * KC + c4 * 4;
            f32x4 gp[NP16];
            load_tile_rows<false, NP16>(gp, b.grad_out, b.ld_grad_out, row0, rows,
                                        (uint32_t)(rs * b.ld_grad_out + (col < out_dim ? col : 0)) * 4u);
            compiler_lds_barrier();
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
              if (4 * cc + cb < NTL) *reinterpret_cast<f32x4*>(abuf + i * LDSW + 16 * cb + 4 * g) = y[4 * cc + cb < NTL ? 4 * cc + cb : 0];
            compiler_lds_barrier();
            f32x4 sg = {0.f, 0.f, 0.f, 0.f}, sgy = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int p = 0; p < NP16; ++p) {
              const f32x4 yh = *reinterpret_cast<const f32x4*>(abuf + (p * 4 + rs) * LDSW + c4 * 4);
              sg += gp[p];
              sgy = __builtin_elementwise_fma(gp[p], yh, sgy);
            }
            sg.x = add_quarters(sg.x); sg.y = add_quarters(sg.y); sg.z = add_quarters(sg.z); sg.w = add_quarters(sg.w);
            sgy.x = add_quarters(sgy.x); sgy.y = add_quarters(sgy.y); sgy.z = add_quarters(sgy.z); sgy.w = add_quarters(sgy.w);
            if (rs == 0) {
              f32x4* pg = reinterpret_cast<f32x4*>(lnbuf + col);
              f32x4* py = reinterpret_cast<f32x4*>(lnbuf + PSTRIDE + col);
              *pg = *pg + sg;
              *py = *py + sgy;
            }
            compiler_lds_barrier();
          }
        }
      } else {
        emit(y, b.yhat, out_dim, out_dim, row0);  // normalised pre-affine output: d gamma = colsum(grad_out * y_hat)
      }
    } else {
#pragma unroll
      for (int t = 0; t < NTL; ++t) {
        const int f = 16 * t + 4 * g;
        gr[t].x = f + 0 < out_dim ? gr[t].x : 0.f; gr[t].y = f + 1 < out_dim ? gr[t].y : 0.f;
        gr[t].z = f + 2 < out_dim ? gr[t].z : 0.f; gr[t].w = f + 3 < out_dim ? gr[t].w : 0.f;
      }
    }
#undef GNC_B16_FWD
    B16P(3);  // LayerNorm backward + parameter sums
    emit(gr, b.dz[L - 1], out_dim, out_dim, row0);
    B16P(4);  // emit dz_{L-1}

    // ------------------------------------------------------------------ back through the Linear layers
#define GNC_B16_BWD(C_, DST_, NTO_, LAYER_)                                          \
  if constexpr (C_ < NCH) {                                                          \
    if (C_ * KC < d.in_dim[LAYER_]) {                                                \
      prefetch_next_chunk(q);                                                        \
      mma16_transposed_chunk<NTL, NTO_, 4 * C_>(DST_, gr, wbuf, i, g);               \
      publish_next_chunk();                                                          \
      ++q;                                                                           \
    }                                                                                \
  }
#pragma unroll
    for (int l = GNC_MAX_LINEAR - 2; l >= 0; --l) {
      if (l < L - 1) {
        f32x4 da[NTL];
#pragma unroll
        for (int t = 0; t < NTL; ++t) da[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        GNC_B16_BWD(0, da, NTL, l + 1) GNC_B16_BWD(1, da, NTL, l + 1) GNC_B16_BWD(2, da, NTL, l + 1) GNC_B16_BWD(3, da, NTL, l + 1)
        apply_mask16<NTL>(da, mask[l]);
#pragma unroll
        for (int t = 0; t < NTL; ++t) gr[t] = da[t];
        emit(gr, b.dz[l], d.out_dim[l], d.out_dim[l], row0);
      }
    }
    B16P(5);  // transposed products + masks + emits of dz_l
#undef GNC_B16_BWD
    if (b.dx) {  // one 64-column slab of dx per MATMUL step, in step order (= the order of the plan's last chunks)
      for (int st = 0; st < pl.num_steps; ++st) {
        if (pl.step[st].add) continue;
        const int s = pl.step[st].seg, c0 = pl.step[st].c0;
        f32x4 dxs[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) dxs[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        prefetch_next_chunk(q);
        mma16_transposed_chunk<NTL, 4, 0>(dxs, gr, wbuf, i, g);
        publish_next_chunk();
        ++q;
        compiler_lds_barrier();
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) *reinterpret_cast<f32x4*>(abuf + i * LDSW + 16 * cb + 4 * g) = dxs[cb];
        compiler_lds_barrier();
        // residual path (the segment is also the residual): + the same columns of grad_out, read back as whole rows
        const bool fold = b.dx_add_grad_out == s + 1;
        store_staged_rows<NP16, false>(abuf, b.dx + d.seg[s].wcol + c0, b.ld_dx, d.seg[s].width - c0 < KC ? d.seg[s].width - c0 : KC,
                                       row0, rows, c4, rs, fold ? b.grad_out + c0 : nullptr, b.ld_grad_out);
        compiler_lds_barrier();
      }
    }
    B16P(6);  // dx
    tile = ntile;
  }
  B16P_END();
  if (b.ln_partial) {  // one row of partials per wave: [d beta (out_dim) | d gamma (out_dim)]
    compiler_lds_barrier();
    float* dst = b.ln_partial + (int64_t)((int)blockIdx.x * W16 + wave) * 2 * out_dim;
    for (int j = lane; j < PSTRIDE; j += 64)
      if (j < out_dim) {
        dst[j] = lnbuf[j];
        dst[out_dim + j] = lnbuf[PSTRIDE + j];
      }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// SAVED instance, round 3: the same arithmetic with the ROW STREAM (saved post-activation slabs, grad_out slabs) consumed
// INSIDE the matrix phases.  The phase probe of the first SAVED instance (tools/profile_bwd16.py, c5 edge processor:
// 574 k cycles per 16-row tile and wave against 262 k of MFMA for the wave pair) showed 151 k cycles in two phases that
// do nothing but wait for memory - eight saved slabs and four grad_out slabs, each requested one item ahead with only a
// few vector instructions between them, all eight waves of the workgroup in step (they share the weight-chunk barriers),
// so nothing covers the 11-13 k cycles a row request takes while the whole chip issues them - and 42 k around the
// LayerNorm backward, most of it four synchronous re-reads of grad_out.  Here:
//   * the LayerNorm-statistics product y = W_{L-1} a_{L-2} consumes a_{L-2} ONE 64-column slab per weight chunk (chunk c
//     contracts over exactly those columns), so slab c is staged right in front of chunk c and slab c + 1 is in flight
//     under chunk c's 256 MFMAs; only the slab's 4 accumulator tiles are live, not all 16;
//   * grad_out slab c rides along with chunk c through a second request buffer;
//   * the masks of the layers below come from slabs streamed under the transposed product of the layer above;
//   * the next tile's first two items are requested under this tile's last chunks;
//   * the re-reads of grad_out for the LayerNorm parameter sums are requested two slabs ahead, the first two before the
//     LayerNorm backward arithmetic.
// ---------------------------------------------------------------------------------------------------------------------
// bit (4 * tt + r) = value r of slab tile tt was positive (the slab holds post-ReLU values)
__device__ __forceinline__ unsigned slab_mask16(const f32x4 (&sl)[4]) {
  unsigned bits = 0;
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) {
    bits |= sl[tt].x > 0.f ? (1u << (4 * tt + 0)) : 0u;
    bits |= sl[tt].y > 0.f ? (1u << (4 * tt + 1)) : 0u;
    bits |= sl[tt].z > 0.f ? (1u << (4 * tt + 2)) : 0u;
    bits |= sl[tt].w > 0.f ? (1u << (4 * tt + 3)) : 0u;
  }
  return bits;
}

// dst[t] += W_chunk * slab for the 64-column chunk whose k values are the slab's 4 tiles (cb = 0..3; columns >= kvalid unused)
template <int NTO>
__device__ __forceinline__ void mma16_chunk_from_slab(f32x4 (&dst)[NTO], const f32x4 (&sl)[4], const float* wbuf, int kvalid, int i,
                                                      int g) {
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    if (16 * cb < kvalid) {
      const float bs[4] = {sl[cb].x, sl[cb].y, sl[cb].z, sl[cb].w};
#pragma unroll
      for (int th = 0; th < NTO; th += 4) {
        f32x4 a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = *reinterpret_cast<const f32x4*>(wbuf + (16 * (th + u) + i) * LDSW + 16 * cb + 4 * g);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float av = s == 0 ? a[u].x : s == 1 ? a[u].y : s == 2 ? a[u].z : a[u].w;
            dst[th + u] = mfma16(av, bs[s], dst[th + u]);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

// x summed over the 16 lanes of a DPP row (lanes with the same lane >> 4 = the 16 data rows of the wave's tile); the total
// lands in the row's last lane (lane & 15 == 15).  row_shr with bound_ctrl: lanes without a source read 0.
template <int SH>
__device__ __forceinline__ float dpp_row_shr(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x110 + SH, 0xf, 0xf, true));
}
__device__ __forceinline__ float row_total16(float v) {
  v += dpp_row_shr<1>(v);
  v += dpp_row_shr<2>(v);
  v += dpp_row_shr<4>(v);
  v += dpp_row_shr<8>(v);
  return v;
}

// layer_norm_backward16 that also forms the LayerNorm parameter sums of the tile's 16 rows IN the accumulator layout:
// d beta[f] += sum_rows grad_out, d gamma[f] += sum_rows grad_out * y_hat, reduced over the rows (= over the lanes of a
// DPP row) on the vector unit and added to the wave's LDS totals by the row's last lane.  Replaces the round-2 form
// (y_hat transposed through the LDS tile, grad_out re-read from L2 as whole-row pieces): no loads, nothing to wait for,
// 32 registers of request buffers less at the kernel's register peak.
template <int NTL>
__device__ __forceinline__ void layer_norm_backward16_sums(f32x4 (&y)[NTL], f32x4 (&gr)[NTL], const float* pg, int out_dim, float eps,
                                                           int g, int i, float* lnbuf, int pstride, bool want_sums) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const float inv_n = 1.f / (float)out_dim;
  const bool padded = out_dim != NTL * 16;
  f32x2 sa = {0.f, 0.f}, sb = {0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NTL; ++t) {
    sa += f32x2{y[t].x, y[t].y};
    sb += f32x2{y[t].z, y[t].w};
  }
  const float mean = add_quarters((sa.x + sa.y) + (sb.x + sb.y)) * inv_n;
  const f32x4 m4 = {mean, mean, mean, mean};
#pragma unroll
  for (int t = 0; t < NTL; ++t) y[t] -= m4;
  if (padded) {
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
      const int f = 16 * t + 4 * g;
      y[t].x = f + 0 < out_dim ? y[t].x : 0.f; y[t].y = f + 1 < out_dim ? y[t].y : 0.f;
      y[t].z = f + 2 < out_dim ? y[t].z : 0.f; y[t].w = f + 3 < out_dim ? y[t].w : 0.f;
      gr[t].x = f + 0 < out_dim ? gr[t].x : 0.f; gr[t].y = f + 1 < out_dim ? gr[t].y : 0.f;
      gr[t].z = f + 2 < out_dim ? gr[t].z : 0.f; gr[t].w = f + 3 < out_dim ? gr[t].w : 0.f;
    }
  }
  f32x2 va = {0.f, 0.f}, vb = {0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NTL; ++t) {
    const f32x2 da = {y[t].x, y[t].y}, db = {y[t].z, y[t].w};
    va = __builtin_elementwise_fma(da, da, va);
    vb = __builtin_elementwise_fma(db, db, vb);
  }
  const float rstd = __frsqrt_rn(add_quarters((va.x + va.y) + (vb.x + vb.y)) * inv_n + eps);
  const f32x4 r4 = {rstd, rstd, rstd, rstd};
  f32x4 m1v = {0.f, 0.f, 0.f, 0.f}, m2v = {0.f, 0.f, 0.f, 0.f};
  const bool writer = want_sums && i == 15;
#pragma unroll
  for (int t = 0; t < NTL; ++t) {
    const f32x4 gm = *reinterpret_cast<const f32x4*>(pg + 16 * t + 4 * g);  // gamma, zero padded
    y[t] = y[t] * r4;        // y_hat
    if (want_sums) {         // wave-uniform
      const f32x4 gy = gr[t] * y[t];
      const f32x4 sg = {row_total16(gr[t].x), row_total16(gr[t].y), row_total16(gr[t].z), row_total16(gr[t].w)};
      const f32x4 sgy = {row_total16(gy.x), row_total16(gy.y), row_total16(gy.z), row_total16(gy.w)};
      if (writer) {
        f32x4* pb_ = reinterpret_cast<f32x4*>(lnbuf + 16 * t + 4 * g);
        f32x4* pc_ = reinterpret_cast<f32x4*>(lnbuf + pstride + 16 * t + 4 * g);
        *pb_ = *pb_ + sg;
        *pc_ = *pc_ + sgy;
      }
    }
    gr[t] = gr[t] * gm;      // g * gamma
    m1v += gr[t];
    m2v = __builtin_elementwise_fma(gr[t], y[t], m2v);
  }
  const float m1 = add_quarters((m1v.x + m1v.y) + (m1v.z + m1v.w)) * inv_n;
  const float m2 = add_quarters((m2v.x + m2v.y) + (m2v.z + m2v.w)) * inv_n;
  const f32x4 nm1 = {-m1, -m1, -m1, -m1}, nm2 = {-m2, -m2, -m2, -m2};
#pragma unroll
  for (int t = 0; t < NTL; ++t) gr[t] = __builtin_elementwise_fma(y[t], nm2, gr[t] + nm1) * r4;  // rstd * (g - m1 - y_hat * m2)
  if (padded) {
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
      const int f = 16 * t + 4 * g;
      gr[t].x = f + 0 < out_dim ? gr[t].x : 0.f; gr[t].y = f + 1 < out_dim ? gr[t].y : 0.f;
      gr[t].z = f + 2 < out_dim ? gr[t].z : 0.f; gr[t].w = f + 3 < out_dim ? gr[t].w : 0.f;
    }
  }
}

template <int NTL>
__global__ __launch_bounds__(NT16) void mlp_backward_saved16_kernel(const gnc_mlp_desc_t d, const BwdArgs b, const BPlan16 pl,
                                                                     const int num_tiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int WROWS = NTL * 16;
  constexpr int CH = WROWS * LDSW;
  constexpr int PSTRIDE = WROWS;
  constexpr int RPP = NT16 / 16;
  constexpr int NW = WROWS / RPP;
  constexpr int NCH = (NTL + 3) / 4;
  constexpr int MW = (NTL + 7) / 8;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15;
  const int g = lane >> 4;
  const int c4 = lane & 15;
  const int rs = lane >> 4;
  const int wc4 = tid & 15;
  const int wr0 = tid >> 4;
  const int L = d.num_linear;
  const int out_dim = d.out_dim[L - 1];
  const int rows = (int)d.rows;
  float* wbuf = lds;
  float* pbuf = lds + CH;
  float* abuf = pbuf + (L + 2) * PSTRIDE + wave * R16 * LDSW;
  float* lnbuf = pbuf + (L + 2) * PSTRIDE + W16 * R16 * LDSW + wave * 2 * PSTRIDE;
  if (b.ln_partial) {
    for (int j = lane; j < 2 * PSTRIDE; j += 64) lnbuf[j] = 0.f;
  }
  stage_params<NT16>(pbuf, d, PSTRIDE, tid);

  auto wload = [&](f32x4 (&wr)[NW], int q) {  // SAVED plans hold full, aligned chunks or tails; same two forms as the first instance
    const int layer = pl.wc[q].layer;
    const float* W = d.weight[layer];
    const int ldw = ldw_of(d, layer);
    const int nrows = d.out_dim[layer];
    const int klimit = pl.wc[q].klimit;
    const int kbase = pl.wc[q].kbase;
    if (kbase + KC <= klimit && (int64_t)nrows * ldw * 4 <= 0xffffffffll) {
      const __amdgpu_buffer_rsrc_t w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, nrows * ldw * 4, 0x00020000);
      const uint32_t off0 = (uint32_t)(wr0 * ldw + kbase + wc4 * 4) * 4u;
#pragma unroll
      for (int p = 0; p < NW; ++p) wr[p] = window_load(w, off0 + (uint32_t)(p * RPP * 4) * (uint32_t)ldw);
      return;
    }
    const int col = kbase + wc4 * 4;
    const int colc = col < klimit ? col : kbase;
#pragma unroll
    for (int p = 0; p < NW; ++p) {
      const int n = p * RPP + wr0;
      const int nc = n < nrows ? n : nrows - 1;
      f32x4 v = *reinterpret_cast<const f32x4*>(W + (int64_t)nc * ldw + colc);
      const bool rowok = n < nrows;
      v.x = (rowok && col + 0 < klimit) ? v.x : 0.f; v.y = (rowok && col + 1 < klimit) ? v.y : 0.f;
      v.z = (rowok && col + 2 < klimit) ? v.z : 0.f; v.w = (rowok && col + 3 < klimit) ? v.w : 0.f;
      wr[p] = v;
    }
  };
  auto wstore = [&](const f32x4 (&wr)[NW]) {
#pragma unroll
    for (int p = 0; p < NW; ++p) *reinterpret_cast<f32x4*>(wbuf + (p * RPP + wr0) * LDSW + wc4 * 4) = wr[p];
  };
  // request one 64-column slab of a row-ordered [rows, ld] tensor for wave tile `tile_of` (rows past the end read 0)
  auto request = [&](f32x4 (&pre)[NP16], const float* base, int ld, int c0, int tile_of) {
    const int fl = fresh_lane();  // roles recomputed where they are used: nothing lane-derived is carried (and spilled) across the loop
    const int c4 = fl & 15, rs = fl >> 4;
    const int col = c0 + c4 * 4 < ld ? c0 + c4 * 4 : 0;
    load_tile_rows<false, NP16>(pre, base, ld, ((int64_t)tile_of * W16 + wave) * R16, rows, (uint32_t)(rs * ld + col) * 4u);
  };
  auto stage = [&](const f32x4 (&pre)[NP16], int c0, int width) {
    compiler_lds_barrier();
    const int fl = fresh_lane();
    const int c4 = fl & 15, rs = fl >> 4;
    const int c = c0 + c4 * 4;
    if (c0 + KC <= width) {
#pragma unroll
      for (int p = 0; p < NP16; ++p) *reinterpret_cast<f32x4*>(abuf + (p * 4 + rs) * LDSW + c4 * 4) = pre[p];
    } else {
#pragma unroll
      for (int p = 0; p < NP16; ++p) {
        f32x4 v = pre[p];
        v.x = c + 0 < width ? v.x : 0.f; v.y = c + 1 < width ? v.y : 0.f;
        v.z = c + 2 < width ? v.z : 0.f; v.w = c + 3 < width ? v.w : 0.f;
        *reinterpret_cast<f32x4*>(abuf + (p * 4 + rs) * LDSW + c4 * 4) = v;
      }
    }
    compiler_lds_barrier();
  };
  auto slab_from_lds = [&](f32x4 (&sl)[4]) {
    const int fl = fresh_lane();
    const int i = fl & 15, g = fl >> 4;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) sl[cb] = *reinterpret_cast<const f32x4*>(abuf + i * LDSW + 16 * cb + 4 * g);
    compiler_lds_barrier();
  };
  auto emit = [&](const f32x4 (&acc)[NTL], float* dst, int ld, int width, int row0) {
    const int fl = fresh_lane();
    const int i = fl & 15, g = fl >> 4, c4 = fl & 15, rs = fl >> 4;
#pragma unroll
    for (int cc = 0; cc < NCH; ++cc) {
      if (cc * KC < width) {
        compiler_lds_barrier();
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
          if (4 * cc + cb < NTL) *reinterpret_cast<f32x4*>(abuf + i * LDSW + 16 * cb + 4 * g) = acc[4 * cc + cb < NTL ? 4 * cc + cb : 0];
        compiler_lds_barrier();
        store_staged_rows<NP16, false>(abuf, dst + cc * KC, ld, width - cc * KC < KC ? width - cc * KC : KC, row0, rows, c4, rs);
      }
    }
    compiler_lds_barrier();
  };

  f32x4 wreg[NW];
  wload(wreg, 0);
  wstore(wreg);
  __syncthreads();
  auto prefetch_next_chunk = [&](int q) { wload(wreg, q + 1 < pl.num_wchunks ? q + 1 : 0); };
  auto publish_next_chunk = [&]() {
    __syncthreads();  // everyone is done reading the buffer
    wstore(wreg);
    __syncthreads();
  };

  // The A stream (saved post-activations, top layer first): ONE request buffer, the next slab in flight under the current
  // chunk's MFMAs.  grad_out: its (up to four) slabs are requested TOGETHER under the last chunk of the LayerNorm
  // product, when nothing but y is live, and staged behind it.  Register budget (256 per wave, and a single spilled
  // register costs more than it saves: hipcc drains vmcnt(0) around every scratch access, which also drains the
  // prefetched requests): LayerNorm product y 64 + slab 16 + A 16 + weight chunk 32 (+ 64 for the grad_out requests in
  // its last chunk); transposed products gr 64 + da 64 + weight chunk 32 + A 16 + slab 16 + 16 column reads in flight.
  f32x4 curA[NP16];
  const int la = L - 2;  // the layer whose saved output feeds the LayerNorm-statistics product
  int tile = blockIdx.x;
  request(curA, b.act[la], d.out_dim[la], 0, tile);
  const int n_gslabs = (out_dim + KC - 1) / KC;

  B16P_BEGIN();
  while (tile < num_tiles) {
    B16P_TILE();
    const int row0 = (tile * W16 + wave) * R16;
    const int ntile = tile + gridDim.x;
    int q = 0;
    unsigned m_now[MW], m_next[MW];  // ReLU masks: of the layer whose product is next / of the layer below it (being streamed)
#pragma unroll
    for (int w = 0; w < MW; ++w) m_now[w] = m_next[w] = 0u;

    // ---------------------------------------------------------------- LayerNorm statistics: y = W_{L-1} a_{L-2} + b, the
    // saved a_{L-2} consumed one 64-column slab per weight chunk
    f32x4 y[NTL];
    f32x4 gb[NCH][NP16];  // grad_out request buffers
    init_bias16<NTL>(y, pbuf + (L - 1) * PSTRIDE, g);
    const int wa = d.out_dim[la];
    const int last_c = (wa + KC - 1) / KC - 1;
    // one body for the chunks in front of the last one (a run-time loop: the chunk index only moves the mask bits and the
    // requested columns), one for the last chunk, which also requests the grad_out slabs
    auto ln_chunk = [&](int c, bool last) {
      f32x4 sl[4];
#ifdef GNC_B16_WLOAD_FIRST_LN
      if (d.ln_gamma) prefetch_next_chunk(q);
#endif
      stage(curA, c * KC, wa);
      if (!last) {
        request(curA, b.act[la], wa, (c + 1) * KC, tile);
      } else {
#pragma unroll
        for (int cc = 0; cc < NCH; ++cc)
          if (cc < n_gslabs) request(gb[cc], b.grad_out, b.ld_grad_out, cc * KC, tile);
      }
      slab_from_lds(sl);
      const unsigned bits = slab_mask16(sl) << (16 * (c & 1));
#pragma unroll
      for (int w = 0; w < MW; ++w) m_now[w] |= (c >> 1) == w ? bits : 0u;
      if (d.ln_gamma) {
#ifndef GNC_B16_WLOAD_FIRST_LN
        prefetch_next_chunk(q);  // (in front of the row request it would not queue behind it at the barrier - but it costs 240 B of scratch here)
#endif
        mma16_chunk_from_slab<NTL>(y, sl, wbuf, wa - c * KC, i, g);
        publish_next_chunk();
        ++q;
      }
    };
    for (int c = 0; c < last_c; ++c) ln_chunk(c, false);
    ln_chunk(last_c, true);
    B16P(2);  // LayerNorm-statistics product, the saved slabs riding along
    // ---------------------------------------------------------------- the grad_out tile (requested under the last chunk)
    f32x4 gr[NTL];
#pragma unroll
    for (int cc = 0; cc < NCH; ++cc) {
      if (cc < n_gslabs) {
        stage(gb[cc], cc * KC, out_dim);
        f32x4 gs[4];
        slab_from_lds(gs);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
          if (4 * cc + cb < NTL) gr[4 * cc + cb < NTL ? 4 * cc + cb : 0] = gs[cb];
      } else {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
          if (4 * cc + cb < NTL) gr[4 * cc + cb < NTL ? 4 * cc + cb : 0] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    // the A stream moves on: the layer below (its masks are due at the end of the NEXT product), or the next tile
    if (la > 0) request(curA, b.act[la - 1 >= 0 ? la - 1 : 0], d.out_dim[la - 1 >= 0 ? la - 1 : 0], 0, tile);
    else request(curA, b.act[la], wa, 0, ntile);
    B16P(1);  // grad_out slabs
    if (d.ln_gamma) {
      // rows past the end of the table: their grad_out slabs read zeros (window bounds check), so they add nothing
      compiler_lds_barrier();
      layer_norm_backward16_sums<NTL>(y, gr, pbuf + L * PSTRIDE, out_dim, d.ln_eps, g, i, lnbuf, PSTRIDE, b.ln_partial != nullptr);
      compiler_lds_barrier();
      if (!b.ln_partial) emit(y, b.yhat, out_dim, out_dim, row0);
    } else {
#pragma unroll
      for (int t = 0; t < NTL; ++t) {
        const int f = 16 * t + 4 * g;
        gr[t].x = f + 0 < out_dim ? gr[t].x : 0.f; gr[t].y = f + 1 < out_dim ? gr[t].y : 0.f;
        gr[t].z = f + 2 < out_dim ? gr[t].z : 0.f; gr[t].w = f + 3 < out_dim ? gr[t].w : 0.f;
      }
    }
    B16P(3);  // LayerNorm backward + parameter sums
    emit(gr, b.dz[L - 1], out_dim, out_dim, row0);
    B16P(4);  // emit dz_{L-1}

    // ---------------------------------------------------------------- back through the Linear layers; the saved slabs of
    // the layer BELOW (their masks are needed at the end of the next product) ride along, one per chunk
#define GNC_S16_BWD(C_, DST_, LAYER_, LBELOW_)                                                               \
  if constexpr (C_ < NCH) {                                                                                  \
    if (C_ * KC < d.in_dim[LAYER_]) {                                                                        \
      /* the weight chunk FIRST: vmcnt retires in order, and the chunk (an L2 hit, due at this chunk's barrier) must not queue behind the row request (HBM, due a whole chunk later) */ \
      prefetch_next_chunk(q);                                                                                \
      if (LBELOW_ >= 0 && C_ * KC < d.out_dim[LBELOW_ >= 0 ? LBELOW_ : 0]) {                                  \
        f32x4 sl[4];                                                                                         \
        const int lb_ = LBELOW_ >= 0 ? LBELOW_ : 0, wb_ = d.out_dim[lb_];                                     \
        stage(curA, C_ * KC, wb_);                                                                           \
        if ((C_ + 1) * KC < wb_) request(curA, b.act[lb_], wb_, (C_ + 1) * KC, tile);                          \
        else if (lb_ > 0) request(curA, b.act[lb_ - 1 >= 0 ? lb_ - 1 : 0], d.out_dim[lb_ - 1 >= 0 ? lb_ - 1 : 0], 0, tile); \
        else request(curA, b.act[la], wa, 0, ntile);                                                          \
        slab_from_lds(sl);                                                                                   \
        m_next[C_ >> 1] |= slab_mask16(sl) << (16 * (C_ & 1));                                                \
      }                                                                                                      \
      mma16_transposed_chunk<NTL, NTL, 4 * C_, 1>(DST_, gr, wbuf, i, g);                                     \
      publish_next_chunk();                                                                                  \
      ++q;                                                                                                   \
    }                                                                                                        \
  }
#pragma unroll
    for (int l = GNC_MAX_LINEAR - 2; l >= 0; --l) {
      if (l < L - 1) {
        f32x4 da[NTL];
#pragma unroll
        for (int t = 0; t < NTL; ++t) da[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int lb = l - 1;  // the layer whose masks the NEXT product needs
        GNC_S16_BWD(0, da, l + 1, lb) GNC_S16_BWD(1, da, l + 1, lb) GNC_S16_BWD(2, da, l + 1, lb) GNC_S16_BWD(3, da, l + 1, lb)
        apply_mask16<NTL>(da, m_now);
#pragma unroll
        for (int w = 0; w < MW; ++w) {
          m_now[w] = m_next[w];
          m_next[w] = 0u;
        }
#pragma unroll
        for (int t = 0; t < NTL; ++t) gr[t] = da[t];
        emit(gr, b.dz[l], d.out_dim[l], d.out_dim[l], row0);
      }
    }
#undef GNC_S16_BWD
    B16P(5);  // transposed products + masks + emits of dz_l
    if (b.dx) {
      for (int st = 0; st < pl.num_steps; ++st) {
        if (pl.step[st].add) continue;
        const int s = pl.step[st].seg, c0 = pl.step[st].c0;
        f32x4 dxs[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) dxs[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        prefetch_next_chunk(q);
        mma16_transposed_chunk<NTL, 4, 0, 1>(dxs, gr, wbuf, i, g);
        publish_next_chunk();
        ++q;
        compiler_lds_barrier();
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) *reinterpret_cast<f32x4*>(abuf + i * LDSW + 16 * cb + 4 * g) = dxs[cb];
        compiler_lds_barrier();
        const bool fold = b.dx_add_grad_out == s + 1;
        store_staged_rows<NP16, false>(abuf, b.dx + d.seg[s].wcol + c0, b.ld_dx, d.seg[s].width - c0 < KC ? d.seg[s].width - c0 : KC,
                                       row0, rows, c4, rs, fold ? b.grad_out + c0 : nullptr, b.ld_grad_out);
        compiler_lds_barrier();
      }
    }
    B16P(6);  // dx
    tile = ntile;
  }
  B16P_END();
  if (b.ln_partial) {
    compiler_lds_barrier();
    float* dst = b.ln_partial + (int64_t)((int)blockIdx.x * W16 + wave) * 2 * out_dim;
    for (int j = lane; j < PSTRIDE; j += 64)
      if (j < out_dim) {
        dst[j] = lnbuf[j];
        dst[out_dim + j] = lnbuf[PSTRIDE + j];
      }
  }
}

bool al16b(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

bool make_plan(const gnc_mlp_desc_t& d, bool want_dx, BPlan16* pl, bool saved = false) {
  const int L = d.num_linear;
  if (L < 2 || L > GNC_MAX_LINEAR - 1 || d.activation != GNC_ACT_RELU || d.rows < 1 || d.rows >= INT32_MAX - (1 << 22)) return false;
  int wmax = 0;
  for (int l = 0; l < L; ++l) {
    if (d.out_dim[l] > wmax) wmax = d.out_dim[l];
    if (l > 0 && d.in_dim[l] > wmax) wmax = d.in_dim[l];
    if (ldw_of(d, l) % 4 != 0 || !al16b(d.weight[l])) return false;
  }
  if (wmax > 256) return false;
  *pl = BPlan16{};
  for (int s = 0; s < d.num_segments; ++s) {  // MATMUL segments first, chunk by chunk
    const gnc_mlp_segment_t& sg = d.seg[s];
    if (sg.ld % 4 != 0 || !al16b(sg.ptr)) return false;
    if (sg.index && (sg.table_rows <= 0 || sg.table_rows * (int64_t)sg.ld * 4 > 0xffffffffll)) return false;
    if (sg.mode == GNC_SEG_ADD) continue;
    if (sg.wcol % 4 != 0) return false;
    for (int c0 = 0; c0 < sg.width; c0 += KC) {
      if (pl->num_steps >= B16_MAX_STEPS || pl->num_wchunks >= B16_MAX_WCHUNKS) return false;
      pl->step[pl->num_steps++] = {(short)s, (short)c0, (short)0, 0};
      pl->wc[pl->num_wchunks++] = {(short)0, (short)(sg.wcol + c0), (short)(sg.wcol + sg.width), 0};
    }
  }
  for (int c0 = 0; c0 < d.out_dim[0]; c0 += KC)  // then the additive ones, slab-major
    for (int s = 0; s < d.num_segments; ++s) {
      if (d.seg[s].mode != GNC_SEG_ADD) continue;
      if (pl->num_steps >= B16_MAX_STEPS) return false;
      pl->step[pl->num_steps++] = {(short)s, (short)c0, (short)1, 0};
    }
  if (pl->num_wchunks == 0) return false;
  const int mm_chunks = pl->num_wchunks;
  std::remove_reference_t<decltype(pl->wc[0])> mm_wc[B16_MAX_WCHUNKS];
  for (int k = 0; k < mm_chunks; ++k) mm_wc[k] = pl->wc[k];
  if (saved) {  // the forward's post-activations are inputs: no chunk of the first Linear or of the hidden layers going forward
    pl->saved = 1;
    pl->num_wchunks = 0;
  }
  auto push_layer = [&](int l) {
    for (int c = 0; c * KC < d.in_dim[l]; ++c) {
      if (pl->num_wchunks >= B16_MAX_WCHUNKS) return false;
      pl->wc[pl->num_wchunks++] = {(short)l, (short)(c * KC), (short)d.in_dim[l], 0};
    }
    return true;
  };
  if (!saved)
    for (int l = 1; l < L - 1; ++l)
      if (!push_layer(l)) return false;                   // forward, hidden layers
  if (d.ln_gamma && !push_layer(L - 1)) return false;     // the last Linear only feeds the LayerNorm statistics
  for (int l = L - 2; l >= 0; --l)
    if (!push_layer(l + 1)) return false;                 // backward: W_{l+1}^T
  if (want_dx)
    for (int k = 0; k < mm_chunks; ++k) {                 // dx: the first Linear's chunks again, in step order
      if (pl->num_wchunks >= B16_MAX_WCHUNKS) return false;
      pl->wc[pl->num_wchunks] = mm_wc[k];
      ++pl->num_wchunks;
    }
  return true;
}

}  // namespace

int gnc_mlp::bwd_stream16_ln_partial_rows(int64_t rows) {  // one row of LayerNorm partial sums per wave of the grid
  const int64_t num_tiles = gnc::ceil_div(rows, (int64_t)W16 * R16);
  return (int)(num_tiles < gnc::num_cu() ? num_tiles : gnc::num_cu()) * W16;
}

bool gnc_mlp::bwd_stream16_supported(const gnc_mlp_desc_t& d, bool want_dx) {
  BPlan16 pl;
  return make_plan(d, want_dx, &pl);
}

int gnc_mlp::launch_bwd_stream16(const gnc_mlp_desc_t& d, const BwdArgs& b, hipStream_t stream, bool saved) {
  BPlan16 pl;
  if (!make_plan(d, b.dx != nullptr, &pl, saved)) {
    gnc::set_error("mlp_backward_stream16: shape outside the kernel");
    return GNC_ERR_UNSUPPORTED;
  }
  constexpr int NTL = 16;
  const size_t smem = ((size_t)NTL * 16 * LDSW + (size_t)(d.num_linear + 2) * NTL * 16 + (size_t)W16 * R16 * LDSW +
                       (size_t)W16 * 2 * NTL * 16) * sizeof(float);
  static_assert(((size_t)16 * 16 * LDSW + (size_t)(GNC_MAX_LINEAR + 2) * 256 + (size_t)W16 * R16 * LDSW + (size_t)W16 * 2 * 256) *
                        sizeof(float) <= 160 * 1024,
                "backward16: LDS budget");
  const int64_t num_tiles = gnc::ceil_div(d.rows, (int64_t)W16 * R16);
  const int64_t grid = num_tiles < gnc::num_cu() ? num_tiles : gnc::num_cu();
  auto go = [&](auto kernel, bool* attr_set) -> int {
    if (!*attr_set) {
      int rc = gnc::check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                              "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
      if (rc) return rc;
      *attr_set = true;
    }
    kernel<<<dim3((unsigned)grid), dim3(NT16), smem, stream>>>(d, b, pl, (int)num_tiles);
    return GNC_OK;
  };
  static bool attr_rc = false, attr_sv = false, attr_s3 = false;
  static const bool first_saved = getenv("GNC_B16_FIRST_SAVED") != nullptr;  // A/B switch: the round-2 SAVED instance
  const int rc = saved ? (first_saved ? go(&mlp_backward_stream16_kernel<NTL, true>, &attr_sv) : go(&mlp_backward_saved16_kernel<NTL>, &attr_s3))
                       : go(&mlp_backward_stream16_kernel<NTL, false>, &attr_rc);
  if (rc) return rc;
  return gnc::check_launch("mlp_backward_stream16_kernel");
}
