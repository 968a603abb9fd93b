// Weight gradients of a small batch (the reference's one-graph-per-step loop, utils/train_model.py:35-45): dW = A^T B and
// db = column sums of A over a few thousand rows, for SEVERAL (A, B) pairs in one launch.
//
// gnc_xty_f32 is built for millions of rows: per-wave partial matrices + a reduction launch per weight block - at ~2000 rows
// that is 69 launches of 5..8 us per training step for 60 MFLOP of work each.  Here one workgroup owns one 16 x 16 tile of
// one job's dW: its four waves take every fourth group of four rows (A operand = A[r + g][m0 + i], B operand =
// B[r + g][k0 + i]: one dword per lane and MFMA, 64 contiguous bytes per row), the four partial tiles meet in LDS and are
// added in wave order (bitwise reproducible); the tiles of column block 0 also leave db.  grid.y = job.
// Job kind 1 sums the rows of a [P, W] matrix (the per-tile LayerNorm partial sums of mlp_bwd_col16.hip) the same way.
#include <stdlib.h>

#include "mlp_device16.h"

using namespace gnc_mlp;

namespace {


struct XJobs {
  gnc_xty_job_t j[GNC_XTY_MAX_JOBS];
};

// Row splits of ONE job: a function of its own row count only, so that a job's bits do not depend on which other jobs share
// its launch (a launch runs max-over-jobs waves per workgroup; the waves a job does not use contribute zeros).  The launch is
// latency-bound (two dword loads per MFMA, a few hundred workgroups on the chip): beyond a thousand rows more waves per tile -
// more loads in flight - are what it needs.  Captured training step of R x R pixel graphs with 4 / 8 / 16 splits everywhere
// (tools/latency_sizes.py, same box): R = 32 0.514 / 0.490 / 0.491 ms, R = 64 1.069 / 0.904 / 0.817, R = 96 1.678 / 1.516 /
// 1.474, R = 128 (the reference's default image) 2.403 / 2.145 / 2.155; N = 144: equal.
__host__ __device__ inline int xty_splits(int64_t rows) { return rows >= 4096 ? 16 : (rows >= 1024 ? 8 : 4); }

// XW: waves per workgroup (>= the splits of every job of the launch)
template <int XW>
__global__ __launch_bounds__(XW * 64) void xty_small_kernel(const XJobs js) {
  __shared__ float part[XW][64 * 4];
  __shared__ float csum[XW][64];
  const gnc_xty_job_t& jb = js.j[blockIdx.y];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int M = jb.m, K = jb.k;
  const int rows = (int)jb.rows;
  if (jb.kind == 1) {  // out[c] = sum over p of a[p][c]; M = columns, rows = P; 64 columns per workgroup
    const int c = (int)blockIdx.x * 64 + lane;
    if ((int)blockIdx.x * 64 >= M) return;
    const __amdgpu_buffer_rsrc_t wa = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(jb.a), 0, (int)(((int64_t)(rows - 1) * jb.lda + M) * 4), 0x00020000);
    float s = 0.f;
    constexpr int XS1 = 4;  // four splits whatever the launch's wave count
    for (int p0 = w; w < XS1 && p0 < rows; p0 += XS1 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)  // rows past the end are outside the window: zeros
        v[u] = c < M ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wa, (uint32_t)(((int64_t)(p0 + u * XS1) * jb.lda + c) * 4), 0, 0)) : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    csum[w][lane] = s;
    __syncthreads();
    if (w == 0 && c < M) jb.dw[c] = ((csum[0][lane] + csum[1][lane]) + csum[2][lane]) + csum[3][lane];
    return;
  }
  const int tiles_k = (K + 15) >> 4, tiles_m = (M + 15) >> 4;
  const int tile = blockIdx.x;
  if (tile >= tiles_m * tiles_k) return;
  const int tm = tile / tiles_k, tk = tile - tm * tiles_k;
  const int m0 = tm * 16, k0 = tk * 16;
  const __amdgpu_buffer_rsrc_t wa = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(jb.a), 0, (int)(((int64_t)(rows - 1) * jb.lda + M) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t wb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(jb.b), 0, (int)(((int64_t)(rows - 1) * jb.ldb + K) * 4), 0x00020000);
  const bool aok = m0 + i < M, bok = k0 + i < K;
  // wave w (< xs, the job's splits) takes the row groups w, w + xs, ...; eight groups (32 rows) per iteration, all 16 loads in flight
  const int xs = xty_splits(rows);
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  float asum = 0.f;
  const uint32_t lda4 = (uint32_t)jb.lda * 4u, ldb4 = (uint32_t)jb.ldb * 4u;
  for (int r0 = 4 * w; w < xs && r0 < rows; r0 += 4 * xs * 8) {
    float av[8], bv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const uint32_t r = (uint32_t)(r0 + u * 4 * xs + g);
      av[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wa, r * lda4 + (uint32_t)(m0 + i) * 4u, 0, 0));
      bv[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wb, r * ldb4 + (uint32_t)(k0 + i) * 4u, 0, 0));
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float a = aok ? av[u] : 0.f, b = bok ? bv[u] : 0.f;  // (a column past the width reads the row's next floats)
      asum += a;
      if (u & 1) acc1 = mfma16(a, b, acc1);
      else acc0 = mfma16(a, b, acc0);
    }
  }
  const f32x4 acc = acc0 + acc1;
  *reinterpret_cast<f32x4*>(&part[w][lane * 4]) = acc;
  csum[w][lane] = asum;
  __syncthreads();
  if (w != 0) return;
  f32x4 t = *reinterpret_cast<const f32x4*>(&part[0][lane * 4]);
#pragma unroll
  for (int ww = 1; ww < XW; ++ww) t += *reinterpret_cast<const f32x4*>(&part[ww][lane * 4]);
  // accumulator register r of lane (n = lane & 15, g) = dW[m0 + 4g + r][k0 + n]
  if (k0 + i < K) {
    float* o = jb.dw + (int64_t)(m0 + 4 * g) * jb.ld_dw + k0 + i;
    if (m0 + 4 * g + 0 < M) o[0] = t.x;
    if (m0 + 4 * g + 1 < M) o[jb.ld_dw] = t.y;
    if (m0 + 4 * g + 2 < M) o[2 * jb.ld_dw] = t.z;
    if (m0 + 4 * g + 3 < M) o[3 * jb.ld_dw] = t.w;
  }
  if (tk == 0 && jb.db && lane < 16 && m0 + lane < M) {  // db[m0 + i] = sum over the 4 row phases g and the waves, in that order
    float s = 0.f;
#pragma unroll
    for (int ww = 0; ww < XW; ++ww)
#pragma unroll
      for (int gg = 0; gg < 4; ++gg) s += csum[ww][gg * 16 + lane];
    jb.db[m0 + lane] = s;
  }
}

}  // namespace

extern "C" int gnc_xty_small_max_rows(void) {
  // 96 x CUs (24,576): measured against the partial-sum products per captured training step of R x R pixel graphs
  // (tools/latency_sizes.py): 1.67 -> 1.53 ms at 16 k edge rows, 1.96 -> 1.75 at 18 k, but 2.45 -> 2.61 at 32 k
  static const int v = getenv("GNC_XTY_SMALL_MAX_ROWS") ? atoi(getenv("GNC_XTY_SMALL_MAX_ROWS")) : 6 * R16 * gnc::num_cu();
  return v;
}

extern "C" int gnc_xty_small_f32(const gnc_xty_job_t* jobs, int32_t njobs, void* stream_) {
  GNC_REQUIRE(jobs && njobs >= 1 && njobs <= GNC_XTY_MAX_JOBS, "gnc_xty_small_f32: 1..%d jobs", GNC_XTY_MAX_JOBS);
  XJobs js = {};
  int gx = 1;
  for (int q = 0; q < njobs; ++q) {
    const gnc_xty_job_t& j = jobs[q];
    GNC_REQUIRE(j.a && j.dw && j.rows >= 1 && j.rows <= gnc_xty_small_max_rows() && j.m >= 1 && j.lda >= j.m,
                "gnc_xty_small_f32: job %d: bad operand", q);
    GNC_REQUIRE(j.rows * j.lda * 4 <= 0x7fffffffll, "gnc_xty_small_f32: job %d: operand A beyond 2 GiB", q);
    int tiles;
    if (j.kind == 1) {
      tiles = (j.m + 63) / 64;
    } else {
      GNC_REQUIRE(j.kind == 0 && j.b && j.k >= 1 && j.ldb >= j.k && j.ld_dw >= j.k && j.m <= 4096 && j.k <= 4096,
                  "gnc_xty_small_f32: job %d: bad operand B / output", q);
      GNC_REQUIRE(j.rows * j.ldb * 4 <= 0x7fffffffll, "gnc_xty_small_f32: job %d: operand B beyond 2 GiB", q);
      tiles = ((j.m + 15) / 16) * ((j.k + 15) / 16);
    }
    gx = tiles > gx ? tiles : gx;
    js.j[q] = j;
  }
  int xw = 4;  // waves per workgroup: the largest split count among the jobs
  for (int q = 0; q < njobs; ++q)
    if (jobs[q].kind == 0 && xty_splits(jobs[q].rows) > xw) xw = xty_splits(jobs[q].rows);
  const dim3 grid((unsigned)gx, (unsigned)njobs);
  if (xw >= 16) xty_small_kernel<16><<<grid, dim3(16 * 64), 0, (hipStream_t)stream_>>>(js);
  else if (xw >= 8) xty_small_kernel<8><<<grid, dim3(8 * 64), 0, (hipStream_t)stream_>>>(js);
  else xty_small_kernel<4><<<grid, dim3(4 * 64), 0, (hipStream_t)stream_>>>(js);
  return gnc::check_launch("xty_small_kernel");
}
