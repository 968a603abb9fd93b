// K4, weights-resident variant of the fused MLP (formulation: see mlp_fused.hip).
//
// When every Linear's weights fit in the 160 KB LDS of a CU (hidden width <= 64: the config-c3
// model needs 5 chunks x 64 x 68 floats = 87 KB for the edge processor) the kernel keeps them
// there for its whole life: one persistent 8-wave workgroup per CU stages all weight chunks
// once, and from then on the waves never meet at a barrier again.  Each wave walks its own
// 32-row tiles; the only per-tile memory traffic is the tile's input rows and its output rows.
//
// Latency hiding is explicit: the global loads of step k+1 (one step = one 64-column chunk of
// one concat segment of one tile) are issued into registers before the MFMAs of step k run, and
// the gather indices are fetched two steps ahead with one coalesced load per step, so the
// dependent chain index -> row address -> row never sits on the critical path.  Two waves per
// SIMD cover each other's LDS writes, LayerNorm and epilogue with MFMA work.
#include <stdlib.h>

#include "mlp_device.h"

using namespace gnc_mlp;

namespace {

constexpr int RWAVES = 8;
constexpr int RNT = RWAVES * 64;

struct Step {  // one staging/compute step of a wave: tile `wt`, concat segment `s`, column offset `c0`
  int64_t wt;
  int s;
  int c0;
};

template <int HT, int OT>
__global__ __launch_bounds__(RNT) void mlp_resident_kernel(const gnc_mlp_desc_t d, const int64_t num_wtiles,
                                                           const int total_chunks) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int WT = HT > OT ? HT : OT;
  constexpr int CH = WT * 32 * LDSW;  // floats per resident weight chunk
  constexpr int PSTRIDE = WT * 32;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31;
  const int h = lane >> 5;
  const int c4 = lane & 15;  // staging role: 16 lanes x 16 B cover one 64-float row chunk
  const int rs = lane >> 4;
  const int L = d.num_linear;
  const int nseg = d.num_segments;
  const int out_dim = d.out_dim[L - 1];
  float* wres = lds;
  float* pbuf = lds + total_chunks * CH;
  float* abuf = pbuf + (L + 2) * PSTRIDE + wave * RPW * LDSW;

  // ---- one-time: parameters and every weight chunk into LDS ---------------------------------
  stage_params<RNT>(pbuf, d, PSTRIDE, tid);
  int l0_chunks = 0;
  {
    int chunk = 0, koff = 0;
    const bool w0v = (d.in_dim[0] % 4 == 0) && ((reinterpret_cast<uintptr_t>(d.weight[0]) & 15u) == 0);
    for (int s = 0; s < nseg; ++s) {
      const int width = d.seg[s].width;
      for (int c0 = 0; c0 < width; c0 += KC, ++chunk)
        stage_weights<WT * 32, RNT>(wres + chunk * CH, d.weight[0], d.in_dim[0], d.out_dim[0], koff + c0, koff + width,
                                    16, w0v && (koff % 4 == 0), tid);
      koff += width;
    }
    l0_chunks = chunk;
    for (int l = 1; l < L; ++l) {
      const bool wv = (d.in_dim[l] % 4 == 0) && ((reinterpret_cast<uintptr_t>(d.weight[l]) & 15u) == 0);
      for (int c = 0; c * KC < d.in_dim[l]; ++c, ++chunk)
        stage_weights<WT * 32, RNT>(wres + chunk * CH, d.weight[l], d.in_dim[l], d.out_dim[l], c * KC, d.in_dim[l], 16,
                                    wv, tid);
    }
  }
  const int hid_chunks = (d.in_dim[1] + KC - 1) / KC;  // every Linear after the first has in_dim = H
  __syncthreads();

  // ---- per-wave pipeline over (tile, segment, column chunk) steps ----------------------------
  const int64_t total_waves = (int64_t)gridDim.x * RWAVES;

  auto advance = [&](Step c) {
    c.c0 += KC;
    if (c.c0 >= d.seg[c.s].width) {
      c.c0 = 0;
      if (++c.s >= nseg) { c.s = 0; c.wt += total_waves; }
    }
    return c;
  };
  // row ids of the step's 32 rows, one per lane (lanes 32..63 mirror 0..31): ONE coalesced load
  auto load_idx = [&](const Step& c) -> int {
    const int64_t r = c.wt * RPW + (lane & 31);
    int v = (int)r;
    const int32_t* ip = d.seg[c.s].index;
    if (ip && c.wt < num_wtiles && r < d.rows) v = ip[r];
    return v;
  };
  auto load_rows = [&](f32x4 (&pre)[RPW / 4], const Step& c, int idxv) {
    const gnc_mlp_segment_t seg = d.seg[c.s];
    const bool vec = (seg.ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(seg.ptr) & 15u) == 0);
    const int col = c.c0 + c4 * 4;
#pragma unroll
    for (int p = 0; p < RPW / 4; ++p) {
      const int j = p * 4 + rs;
      const int tr = __shfl(idxv, j, 64);
      const int64_t r = c.wt * RPW + j;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (c.wt < num_wtiles && r < d.rows) v = load4_guarded(seg.ptr + (int64_t)tr * seg.ld + col, col, seg.width, vec);
      pre[p] = v;
    }
  };

  Step c1 = {(int64_t)blockIdx.x * RWAVES + wave, 0, 0};
  f32x4 pre[RPW / 4];
  load_rows(pre, c1, load_idx(c1));
  Step c2 = advance(c1);
  int idx2 = load_idx(c2);

  while (c1.wt < num_wtiles) {
    const int64_t row0 = c1.wt * RPW;

    // ------------------------------------------------------------------ layer 0 over the staged steps
    f32x16 hid[HT];
    init_bias<HT>(hid, pbuf, h);
    for (int chunk = 0; chunk < l0_chunks; ++chunk) {
      const int kc = d.seg[c1.s].width - c1.c0 < KC ? d.seg[c1.s].width - c1.c0 : KC;
      const int kc8 = (kc + 7) >> 3;
      wave_lds_fence();
#pragma unroll
      for (int p = 0; p < RPW / 4; ++p) *reinterpret_cast<f32x4*>(abuf + (p * 4 + rs) * LDSW + c4 * 4) = pre[p];
      wave_lds_fence();
      load_rows(pre, c2, idx2);  // next step's rows fly while this step's MFMAs run
      const Step c3 = advance(c2);
      const int idx3 = load_idx(c3);
      mma_chunk_from_lds<HT>(hid, abuf, wres + chunk * CH, kc8, i, h);
      c1 = c2;
      c2 = c3;
      idx2 = idx3;
    }
    activate_tiles<HT>(hid, d.activation, d.act_param);

    // ------------------------------------------------------------------ hidden layers 1 .. L-2 (weights resident)
    for (int l = 1; l < L - 1; ++l) {
      f32x16 nxt[HT];
      init_bias<HT>(nxt, pbuf + l * PSTRIDE, h);
      const float* wl = wres + (l0_chunks + (l - 1) * hid_chunks) * CH;
#pragma unroll
      for (int c = 0; c < (HT + 1) / 2; ++c)
        if (c * KC < d.in_dim[l]) mma_chunk_from_regs<HT, HT>(nxt, hid, wl + c * CH, c, d.in_dim[l], i, h);
      activate_tiles<HT>(nxt, d.activation, d.act_param);
#pragma unroll
      for (int t = 0; t < HT; ++t) hid[t] = nxt[t];
    }

    // ------------------------------------------------------------------ last Linear, LayerNorm, store
    f32x16 o[OT];
    init_bias<OT>(o, pbuf + (L - 1) * PSTRIDE, h);
    {
      const float* wl = wres + (l0_chunks + (L - 2) * hid_chunks) * CH;
#pragma unroll
      for (int c = 0; c < (HT + 1) / 2; ++c)
        if (c * KC < d.in_dim[L - 1]) mma_chunk_from_regs<HT, OT>(o, hid, wl + c * CH, c, d.in_dim[L - 1], i, h);
    }
    if (d.ln_gamma) layer_norm_tiles<OT>(o, pbuf + L * PSTRIDE, pbuf + (L + 1) * PSTRIDE, out_dim, d.ln_eps, h);
    store_tiles<OT>(o, abuf, d, row0, out_dim, lane, i, h);
  }
}

template <int HT, int OT>
int launch(const gnc_mlp_desc_t& d, int total_chunks, size_t smem, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    int rc = gnc::check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_resident_kernel<HT, OT>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                            "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    if (rc) return rc;
    attr_set = true;
  }
  const int64_t num_wtiles = gnc::ceil_div(d.rows, RPW);
  int64_t grid = gnc::ceil_div(num_wtiles, RWAVES);
  if (grid > gnc::kNumCU) grid = gnc::kNumCU;  // one persistent workgroup per CU
  mlp_resident_kernel<HT, OT><<<dim3((unsigned)grid), dim3(RNT), smem, stream>>>(d, num_wtiles, total_chunks);
  return gnc::check_launch("mlp_resident_kernel");
}

}  // namespace

int gnc_mlp::launch_resident(const gnc_mlp_desc_t& d, int T, bool narrow_out, hipStream_t stream, bool* launched) {
  *launched = false;
  static const bool disabled = getenv("GNC_MLP_NO_RESIDENT") != nullptr;  // A/B switch for benchmarking
  if (disabled) return GNC_OK;
  if (T > 2 || d.rows >= INT32_MAX) return GNC_OK;  // wider layers never fit: streaming kernel
  const int L = d.num_linear;
  int total_chunks = 0;
  for (int s = 0; s < d.num_segments; ++s) total_chunks += (d.seg[s].width + KC - 1) / KC;
  total_chunks += (L - 1) * ((d.in_dim[1] + KC - 1) / KC);
  const size_t floats = (size_t)total_chunks * T * 32 * LDSW + (size_t)(L + 2) * T * 32 + (size_t)RWAVES * RPW * LDSW;
  const size_t smem = floats * sizeof(float);
  if (smem > 160 * 1024) return GNC_OK;
  *launched = true;
  if (T == 1) return launch<1, 1>(d, total_chunks, smem, stream);
  return narrow_out ? launch<2, 1>(d, total_chunks, smem, stream) : launch<2, 2>(d, total_chunks, smem, stream);
}
