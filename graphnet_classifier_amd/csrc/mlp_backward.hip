// Backward of the fused MLP (K8 of SURVEY.md section 2.2: the autograd of K4), split in two kernels.
//
// The reference's caller that needs it is utils/train_model.py:41 (`loss.backward()`): PyTorch's autograd
// through models/MLP.py:45-47 and the concat / residual around it (models/GNN.py:58-62, :100-102).
//
// mlp_backward_data_kernel  (one launch per MLP call, same tiling as the weights-resident forward):
//   recomputes the forward of its 32-row tile on the fp32 MFMA path (this kernel serves the shapes whose forward
//   saves nothing: two MATMUL segments, the node processors; the streamed and the fused kernels read the training
//   forward's saved post-activations instead, gnc_mlp_bwd_desc_t.act_given), then walks the chain backwards with the SAME resident weight chunks read
//   transposed (ds_read_b32 down a column):
//        g            -> LayerNorm backward -> dy            (and per-wave partial sums of d gamma, d beta)
//        da_l = W_{l+1}^T dz_{l+1},  dz_l = da_l * (a_l > 0)   for l = L-2 .. 0   (ReLU masks: 1 bit per value)
//        dx   = W_0^T dz_0                                     (grad of the MATMUL part of the input concat)
//   The accumulator layout keeps the feature index in the register and the data row on the lane in both
//   directions, so dz_{l+1} is the B operand of the next product as it stands.  It writes what the weight
//   gradients need: a_l (layer inputs) and dz_l, as whole 256-B rows.
// xty_kernel  (one launch per Linear): dW = dz^T a and db = colsum(dz): a row-streaming skinny GEMM,
//   rows on the MFMA k axis, per-wave partial results (summed afterwards in a fixed order, so weight
//   gradients are bitwise reproducible; no atomics).
#include <stdlib.h>

#include <type_traits>

#include "mlp_backward_fused.h"


// Phase probe of the streamed kernel (developer build: make probe_b32, tools/profile_bwd16.py <iters> 128): cycles per
// phase of the tile loop, per wave.
#if defined(GNC_PHASE_PROBE) && !defined(GNC_BWD_PROBE_SYMBOL)
__device__ unsigned long long gnc_phase_probe_b32[4096 * 12];
extern "C" int gnc_phase_probe_b32_read(void* dst, size_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(gnc_phase_probe_b32), bytes); }
#define B32P_BEGIN() unsigned long long pr_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pr_tp, pr_tn; \
  const unsigned long long pr_c0 = __builtin_readcyclecounter(), pr_w0 = wall_clock64()
#define B32P_TILE() pr_tp = __builtin_readcyclecounter()
#define B32P(k) do { pr_tn = __builtin_readcyclecounter(); pr_acc[k] += pr_tn - pr_tp; pr_tp = pr_tn; } while (0)
#define B32P_END() do { if ((threadIdx.x & 63) == 0) {                                                          \
    unsigned long long* o = gnc_phase_probe_b32 + (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 12;  \
    for (int k = 0; k < 8; ++k) o[k] = pr_acc[k];                                                               \
    o[8] = __builtin_readcyclecounter() - pr_c0; o[9] = wall_clock64() - pr_w0; } } while (0)
#else
#define B32P_BEGIN() do {} while (0)
#define B32P_TILE() do {} while (0)
#define B32P(k) do {} while (0)
#define B32P_END() do {} while (0)
#endif

namespace {

constexpr int BWAVES = 8;
constexpr int BNT = BWAVES * 64;

// HT = tiles of the hidden AND output width (both <= 64).  NMM / NADD as in mlp_resident.hip.
// SAVED: b.act[] are the training forward's saved post-activations (inputs): no forward recompute but the last Linear's
// (LayerNorm statistics), nothing emitted for them
template <int HT, int NMM, int NADD, bool SAVED = false>
__global__ __launch_bounds__(BNT) void mlp_backward_data_kernel(const gnc_mlp_desc_t d, const BwdArgs b,
                                                                const int num_wtiles, const int total_chunks) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int CH = HT * 32 * LDSW;
  constexpr int PSTRIDE = HT * 32;
  constexpr int NS = NMM + NADD;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31;
  const int h = lane >> 5;
  const int c4 = lane & 15;
  const int rs = lane >> 4;
  const int L = d.num_linear;
  const int out_dim = d.out_dim[L - 1];
  const int rows = (int)d.rows;
  float* wres = lds;
  float* pbuf = lds + total_chunks * CH;
  float* abuf = pbuf + (L + 2) * PSTRIDE + wave * RPW * LDSW;

  stage_params<BNT>(pbuf, d, PSTRIDE, tid);
  {
    int chunk = 0;
    const int ldw0 = ldw_of(d, 0);
    const bool w0v = (ldw0 % 4 == 0) && ((reinterpret_cast<uintptr_t>(d.weight[0]) & 15u) == 0);
#pragma unroll
    for (int s = 0; s < NMM; ++s, ++chunk)
      stage_weights<HT * 32, BNT>(wres + chunk * CH, d.weight[0], ldw0, d.out_dim[0], d.seg[s].wcol,
                                  d.seg[s].wcol + d.seg[s].width, 16, w0v && (d.seg[s].wcol % 4 == 0), tid);
    for (int l = 1; l < L; ++l) {
      const int ldw = ldw_of(d, l);
      const bool wv = (ldw % 4 == 0) && ((reinterpret_cast<uintptr_t>(d.weight[l]) & 15u) == 0);
      stage_weights<HT * 32, BNT>(wres + chunk * CH, d.weight[l], ldw, d.out_dim[l], 0, d.in_dim[l], 16, wv, tid);
      ++chunk;
    }
  }
  __syncthreads();

  SegView sv[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) sv[s] = {d.seg[s].ptr, d.seg[s].index, d.seg[s].ld, d.seg[s].width};

  const int total_waves = (int)gridDim.x * BWAVES;
  const int last_wt = num_wtiles - 1;

  auto load_idx = [&](int wt, const SegView& s) -> int {
    const int wtc = wt < last_wt ? wt : last_wt;
    int r = wtc * RPW + (lane & 31);
    r = r < rows ? r : rows - 1;
    return s.index ? s.index[r] : r;
  };
  auto load_rows = [&](f32x4 (&pre)[NP], const float* base, int ld, int idxv) {
    const int col = c4 * 4 < ld ? c4 * 4 : 0;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int tr = __shfl(idxv, p * 4 + rs, 64);
      pre[p] = *reinterpret_cast<const f32x4*>(base + (int64_t)tr * ld + col);
    }
  };
  // row-ordered table: one buffer window per tile, no address arithmetic on the vector unit, rows past the end read 0
  auto load_rows_seq = [&](f32x4 (&pre)[NP], const float* base, int ld, int row0) {
    const int col = c4 * 4 < ld ? c4 * 4 : 0;
    load_tile_rows(pre, base, ld, row0, rows, (uint32_t)(rs * ld + col) * 4u);
  };
  auto load_seg = [&](f32x4 (&pre)[NP], const SegView& sg, int wt) {
    if (sg.index) load_rows(pre, sg.ptr, sg.ld, load_idx(wt, sg));
    else load_rows_seq(pre, sg.ptr, sg.ld, wt * RPW);
  };
  auto stage = [&](const f32x4 (&pre)[NP], int width) {
    compiler_lds_barrier();
    const int c = c4 * 4;
    if (width >= KC) {
#pragma unroll
      for (int p = 0; p < NP; ++p) *reinterpret_cast<f32x4*>(abuf + (p * 4 + rs) * LDSW + c) = pre[p];
    } else {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        f32x4 v = pre[p];
        v.x = c + 0 < width ? v.x : 0.f; v.y = c + 1 < width ? v.y : 0.f;
        v.z = c + 2 < width ? v.z : 0.f; v.w = c + 3 < width ? v.w : 0.f;
        *reinterpret_cast<f32x4*>(abuf + (p * 4 + rs) * LDSW + c) = v;
      }
    }
    compiler_lds_barrier();
  };
  // accumulator tiles -> whole rows of a [rows, ld] tensor (columns < width), through the wave's LDS tile
  auto emit = [&](const f32x16 (&acc)[HT], float* dst, int ld, int width, int row0) {
    compiler_lds_barrier();
    tiles_to_lds<HT>(acc, abuf, i, h);
    compiler_lds_barrier();
    store_staged_rows(abuf, dst, ld, width, row0, rows, c4, rs);
    compiler_lds_barrier();
  };

  // LayerNorm parameter gradients formed in flight (b.ln_partial): lane (rs, c4) sums columns c4*4..+3 of the rows
  // p*4+rs of all its tiles - 8 registers instead of a [rows, out_dim] y_hat tensor written here and read back
  f32x4 sum_g = {0.f, 0.f, 0.f, 0.f}, sum_gy = {0.f, 0.f, 0.f, 0.f};
  for (int wt = (int)blockIdx.x * BWAVES + wave; wt < num_wtiles; wt += total_waves) {
    const int row0 = wt * RPW;

    f32x16 hid[HT];
    f32x4 gpre[NP];  // grad_out rows of this tile: requested early, consumed after the forward recompute
    unsigned mask[GNC_MAX_LINEAR - 1];
    if constexpr (SAVED) {
      // the saved tiles give the ReLU masks; the last one stays in registers for the LayerNorm recompute
      f32x4 pre[NP];
      load_rows_seq(pre, b.act[0], d.out_dim[0], row0);
#pragma unroll
      for (int l = 0; l < GNC_MAX_LINEAR - 1; ++l) {
        if (l < L - 1) {
          stage(pre, d.out_dim[l]);
          if (l + 1 < L - 1) load_rows_seq(pre, b.act[l + 1 < GNC_MAX_LINEAR ? l + 1 : 0], d.out_dim[l + 1 < GNC_MAX_LINEAR ? l + 1 : 0], row0);
          else load_rows_seq(gpre, b.grad_out, b.ld_grad_out, row0);
          tile_from_lds<HT>(hid, abuf, i, h);
          compiler_lds_barrier();
          mask[l] = relu_tiles_mask<HT>(hid);
        }
      }
    } else {
    // ------------------------------------------------------------------ forward recompute
    init_bias<HT>(hid, pbuf, h);
    {
      f32x4 pre[NP], pre2[NP], pre3[NP];
      load_seg(pre, sv[0], wt);
#pragma unroll
      for (int s = 0; s < NMM; ++s) {
        stage(pre, sv[s].width);
        // request the next step's rows (and, at the last MATMUL step, the additive gathers and grad_out) before the MFMAs
        if (s + 1 < NMM) {
          load_seg(pre, sv[s + 1 < NMM ? s + 1 : 0], wt);
        } else {
          if constexpr (NADD > 0) {
            load_seg(pre2, sv[NMM], wt);
            load_seg(pre3, sv[NMM + 1], wt);
          }
          load_rows_seq(gpre, b.grad_out, b.ld_grad_out, row0);
        }
        mma_chunk_from_lds<HT>(hid, abuf, wres + s * CH, (sv[s].width + 7) >> 3, i, h);
      }
      if constexpr (NADD > 0) {
#pragma unroll
        for (int p = 0; p < NP; ++p) pre2[p] += pre3[p];
        stage(pre2, sv[NMM].width);
        add_tile_from_lds<HT>(hid, abuf, i, h);
      }
    }
    mask[0] = relu_tiles_mask<HT>(hid);
    emit(hid, b.act[0], d.out_dim[0], d.out_dim[0], row0);
#pragma unroll
    for (int l = 1; l < GNC_MAX_LINEAR - 1; ++l) {
      if (l < L - 1) {
        f32x16 nxt[HT];
        init_bias<HT>(nxt, pbuf + l * PSTRIDE, h);
        mma_chunk_from_regs<HT, HT>(nxt, hid, wres + (NMM + l - 1) * CH, 0, d.in_dim[l], i, h);
        mask[l] = relu_tiles_mask<HT>(nxt);
#pragma unroll
        for (int t = 0; t < HT; ++t) hid[t] = nxt[t];
        emit(hid, b.act[l], d.out_dim[l], d.out_dim[l], row0);
      }
    }
    }  // forward recompute

    // ------------------------------------------------------------------ grad of the pre-LayerNorm output
    f32x16 g[HT];
    {
      stage(gpre, out_dim);
      tile_from_lds<HT>(g, abuf, i, h);
      compiler_lds_barrier();
    }
    if (d.ln_gamma) {
      f32x16 y[HT];
      init_bias<HT>(y, pbuf + (L - 1) * PSTRIDE, h);
      mma_chunk_from_regs<HT, HT>(y, hid, wres + (NMM + L - 2) * CH, 0, d.in_dim[L - 1], i, h);
      layer_norm_backward_tiles<HT>(y, g, pbuf + L * PSTRIDE, out_dim, d.ln_eps, h);
      if (b.ln_partial) {
        // y_hat goes through the wave's LDS tile into whole-row pieces; grad_out's pieces come back through the
        // window (this tile read them a moment ago: L2 hits; rows past the end read 0 and drop out)
        compiler_lds_barrier();
        tiles_to_lds<HT>(y, abuf, i, h);
        compiler_lds_barrier();
        f32x4 gp[NP];
        const int gcol = c4 * 4 < out_dim ? c4 * 4 : 0;
        load_tile_rows(gp, b.grad_out, b.ld_grad_out, row0, rows, (uint32_t)(rs * b.ld_grad_out + gcol) * 4u);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const f32x4 yh = *reinterpret_cast<const f32x4*>(abuf + (p * 4 + rs) * LDSW + c4 * 4);
          sum_g += gp[p];
          sum_gy = __builtin_elementwise_fma(gp[p], yh, sum_gy);
        }
        compiler_lds_barrier();
      } else {
        emit(y, b.yhat, out_dim, out_dim, row0);  // normalised pre-affine output: d gamma = colsum(grad_out * yhat)
      }
    } else {
#pragma unroll
      for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) g[t][r] = (feat_of(t, r, h) < out_dim) ? g[t][r] : 0.f;
    }
    emit(g, b.dz[L - 1], out_dim, out_dim, row0);  // dy

    // ------------------------------------------------------------------ back through the Linear layers
#pragma unroll
    for (int l = GNC_MAX_LINEAR - 2; l >= 0; --l) {
      if (l < L - 1) {
        f32x16 da[HT];
        zero_tiles<HT>(da);
        mma_transposed_from_regs<HT, HT>(da, g, wres + (NMM + l) * CH, i, h);  // W_{l+1}^T dz_{l+1}
        apply_mask<HT>(da, mask[l]);
#pragma unroll
        for (int t = 0; t < HT; ++t) g[t] = da[t];
        emit(g, b.dz[l], d.out_dim[l], d.out_dim[l], row0);
      }
    }
    if (b.dx) {
#pragma unroll
      for (int s = 0; s < NMM; ++s) {
        f32x16 dxs[HT];
        zero_tiles<HT>(dxs);
        mma_transposed_from_regs<HT, HT>(dxs, g, wres + s * CH, i, h);  // columns wcol_s .. of W_0^T dz_0
        if (b.dx_add_grad_out && s == NMM - 1) {  // residual path: + grad_out, read back as whole rows (L2 hits)
          compiler_lds_barrier();
          tiles_to_lds<HT>(dxs, abuf, i, h);
          compiler_lds_barrier();
          store_staged_rows(abuf, b.dx + d.seg[s].wcol, b.ld_dx, sv[s].width, row0, rows, c4, rs, b.grad_out, b.ld_grad_out);
          compiler_lds_barrier();
        } else {
          emit(dxs, b.dx + d.seg[s].wcol, b.ld_dx, sv[s].width, row0);
        }
      }
    }
  }
  if (b.ln_partial) {  // fold the four row groups (lanes with the same c4), then one row of partials per wave
    float* dst = b.ln_partial + (int64_t)((int)blockIdx.x * BWAVES + wave) * 2 * out_dim;
    const float sg[4] = {add_quarters(sum_g.x), add_quarters(sum_g.y), add_quarters(sum_g.z), add_quarters(sum_g.w)};
    const float sy[4] = {add_quarters(sum_gy.x), add_quarters(sum_gy.y), add_quarters(sum_gy.z), add_quarters(sum_gy.w)};
    if (rs == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (c4 * 4 + k < out_dim) {
          dst[c4 * 4 + k] = sg[k];
          dst[out_dim + c4 * 4 + k] = sy[k];
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Streamed variant of the data kernel for widths 65..128: the weights do not fit in LDS, so the chunk
// sequence of a tile - forward chunks, then the SAME [features][64 k] chunks again in reverse layer order
// for the transposed products - streams through the double buffer of mlp_stream.hip (registers ->
// alternate LDS buffer -> one barrier per chunk).  4 waves x 512 registers: the chain needs four
// accumulator sets of 64 registers (hid, next, g, da) at width 128.
// ---------------------------------------------------------------------------------------------------
constexpr int SB_MAX_STEPS = 16;
constexpr int SB_MAX_WCHUNKS = 64;

struct BwdPlan {
  int num_steps;     // first-Linear staging steps per tile
  int num_wchunks;   // weight chunks per tile (forward + backward)
  int saved;         // 1: b.act[] are the forward's saved post-activations (inputs): no forward chunks but the last Linear's
  struct { short seg, seg2, c0, add; } step[SB_MAX_STEPS];
  struct { short layer, kbase, klimit, pad; } wc[SB_MAX_WCHUNKS];
};

// dst[2c], dst[2c+1] += (columns [64c, 64c+64) of W)^T * src   (chunk buffer = [n][64 k] of those columns)
template <int TI, int TO, int C>
__device__ __forceinline__ void mma_transposed_chunk(f32x16 (&dst)[TO], const f32x16 (&src)[TI], const float* wbuf, int i,
                                                     int h) {
#pragma unroll
  for (int t = 0; t < TI; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
      if constexpr (2 * C < TO) dst[2 * C] = mfma(wbuf[n * LDSW + i], src[t][r], dst[2 * C]);
      if constexpr (2 * C + 1 < TO) dst[2 * C + 1] = mfma(wbuf[n * LDSW + 32 + i], src[t][r], dst[2 * C + 1]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// SAVED (= pl.saved, as a template flag: the run-time branch alone cost the recomputing instances 280 B of scratch)
template <int HT, int WAVES, bool SAVED = false>
__global__ __launch_bounds__(WAVES * 64) void mlp_backward_stream_kernel(const gnc_mlp_desc_t d, const BwdArgs b, const BwdPlan pl,
                                                                         const int num_tiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NT = WAVES * 64;
  constexpr int CH = HT * 32 * LDSW;
  constexpr int PSTRIDE = HT * 32;
  constexpr int RPP = NT / 16;
  constexpr int NW = (HT * 32) / RPP;
  constexpr int NCH = (HT + 1) / 2;  // 64-column chunks of a full-width tile
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31;
  const int h = lane >> 5;
  const int L = d.num_linear;
  const int out_dim = d.out_dim[L - 1];
  const int rows = (int)d.rows;
  float* wbuf = lds;
  float* pbuf = lds + 2 * CH;
  float* abuf = pbuf + (L + 2) * PSTRIDE + wave * RPW * LDSW;
  // LayerNorm parameter sums of the rows this wave processes (b.ln_partial): [colsum(grad_out) | colsum(grad_out * y_hat)],
  // kept in LDS so that nothing is carried in registers across the tile loop
  float* lnbuf = pbuf + (L + 2) * PSTRIDE + WAVES * RPW * LDSW + wave * 2 * PSTRIDE;
  if (b.ln_partial) {
    for (int j = lane; j < 2 * PSTRIDE; j += 64) lnbuf[j] = 0.f;
  }

  stage_params<NT>(pbuf, d, PSTRIDE, tid);

  auto wload = [&](f32x4 (&wr)[NW], int q) {
    const int ftid = wave * 64 + fresh_lane();  // roles recomputed per use (see fresh_lane): nothing lane-derived is carried
    const int wc4 = ftid & 15, wr0 = ftid >> 4;
    const int layer = pl.wc[q].layer;
    const float* W = d.weight[layer];
    const int ldw = ldw_of(d, layer);
    const int nrows = d.out_dim[layer];
    const int klimit = pl.wc[q].klimit;
    if (pl.wc[q].kbase + KC <= klimit && (int64_t)nrows * ldw * 4 <= 0xffffffffll) {
      // full 64-column chunk (the plan guarantees aligned weights): a window over the matrix, rows >= nrows read as 0
      // through the bounds check - no masks, one 32-bit add per load (as in mlp_stream.hip; the masked form below cost
      // ~10 vector instructions per piece, 40 per chunk and thread)
      const __amdgpu_buffer_rsrc_t w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, nrows * ldw * 4, 0x00020000);
      const uint32_t off0 = (uint32_t)(wr0 * ldw + pl.wc[q].kbase + wc4 * 4) * 4u;
#pragma unroll
      for (int p = 0; p < NW; ++p) wr[p] = window_load(w, off0 + (uint32_t)(p * RPP * 4) * (uint32_t)ldw);
      return;
    }
    const int col = pl.wc[q].kbase + wc4 * 4;
    const int colc = col < klimit ? col : pl.wc[q].kbase;
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
  auto wstore = [&](const f32x4 (&wr)[NW], float* buf) {
    const int ftid = wave * 64 + fresh_lane();
    const int wc4 = ftid & 15, wr0 = ftid >> 4;
#pragma unroll
    for (int p = 0; p < NW; ++p) *reinterpret_cast<f32x4*>(buf + (p * RPP + wr0) * LDSW + wc4 * 4) = wr[p];
  };
  auto load_idx = [&](int row0, int s) -> int {
    int r = row0 + (fresh_lane() & 31);
    r = r < rows ? r : rows - 1;
    const int32_t* ip = d.seg[s].index;
    return ip ? ip[r] : r;
  };
  auto load_rows = [&](f32x4 (&pre)[NP], const float* base, int ld, int c0, int idxv) {
    const int fl = fresh_lane();
    const int c4 = fl & 15, rs = fl >> 4;
    const int col = c0 + c4 * 4 < ld ? c0 + c4 * 4 : 0;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int tr = __shfl(idxv, p * 4 + rs, 64);
      pre[p] = *reinterpret_cast<const f32x4*>(base + (int64_t)tr * ld + col);
    }
  };
  auto stage = [&](const f32x4 (&pre)[NP], int c0, int width) {
    compiler_lds_barrier();
    const int fl = fresh_lane();
    const int c4 = fl & 15, rs = fl >> 4;
    const int c = c0 + c4 * 4;
    if (c0 + KC <= width) {  // full slab: nothing to mask
#pragma unroll
      for (int p = 0; p < NP; ++p) *reinterpret_cast<f32x4*>(abuf + (p * 4 + rs) * LDSW + c4 * 4) = pre[p];
    } else {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        f32x4 v = pre[p];
        v.x = c + 0 < width ? v.x : 0.f; v.y = c + 1 < width ? v.y : 0.f;
        v.z = c + 2 < width ? v.z : 0.f; v.w = c + 3 < width ? v.w : 0.f;
        *reinterpret_cast<f32x4*>(abuf + (p * 4 + rs) * LDSW + c4 * 4) = v;
      }
    }
    compiler_lds_barrier();
  };
  // 64 columns [64*cc, +64) of `acc` -> rows of dst (columns < width)
  auto emit_chunk = [&](float* dst, int ld, int width, int row0, int cc) {
    const int fl = fresh_lane();
    const int c4 = fl & 15, rs = fl >> 4;
    compiler_lds_barrier();
    store_staged_rows(abuf, dst + cc * KC, ld, width - cc * KC < KC ? width - cc * KC : KC, row0, rows, c4, rs);
    compiler_lds_barrier();
  };
  auto emit = [&](const f32x16 (&acc)[HT], float* dst, int ld, int width, int row0) {
#pragma unroll
    for (int cc = 0; cc < NCH; ++cc) {
      if (cc * KC < width) {
        compiler_lds_barrier();
        switch (cc) {
          case 0: chunk_to_lds<HT, 0>(acc, abuf, i, h); break;
          case 1: chunk_to_lds<HT, 2>(acc, abuf, i, h); break;
          case 2: chunk_to_lds<HT, 4>(acc, abuf, i, h); break;
          default: chunk_to_lds<HT, 6>(acc, abuf, i, h); break;
        }
        emit_chunk(dst, ld, width, row0, cc);
      }
    }
  };

  f32x4 wreg[NW];
  wload(wreg, 0);
  wstore(wreg, wbuf);
  __syncthreads();
  int gq = 0;
  int q = 0;  // chunk index inside the tile's sequence
  auto prefetch_next_chunk = [&]() { wload(wreg, q + 1 < pl.num_wchunks ? q + 1 : 0); };
  auto publish_next_chunk = [&]() {
    wstore(wreg, wbuf + ((gq + 1) & 1) * CH);
    __syncthreads();
    ++gq;
    q = q + 1 < pl.num_wchunks ? q + 1 : 0;
  };
  auto cur_chunk = [&]() -> const float* { return wbuf + (gq & 1) * CH; };

  B32P_BEGIN();
  for (int tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
    B32P_TILE();
    const int row0 = (tile * WAVES + wave) * RPW;

    unsigned mask[GNC_MAX_LINEAR - 1][NCH];  // ReLU masks: 32 bits per 64-column chunk and layer
    auto relu_mask = [&](f32x16 (&acc)[HT], int l) {
#pragma unroll
      for (int cc = 0; cc < NCH; ++cc) {
        unsigned m = 0;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (2 * cc + tt < HT) {
              const bool pos = acc[2 * cc + tt < HT ? 2 * cc + tt : 0][r] > 0.f;
              m |= pos ? (1u << (16 * tt + r)) : 0u;
              acc[2 * cc + tt < HT ? 2 * cc + tt : 0][r] = pos ? acc[2 * cc + tt < HT ? 2 * cc + tt : 0][r] : 0.f;
            }
          }
        mask[l][cc] = m;
      }
    };
    // a [32, width] tile of a row-ordered tensor in accumulator layout, 64 columns at a time through the wave's LDS tile
    auto load_acc_rows = [&](f32x16 (&acc)[HT], const float* base, int ld, int width) {
#pragma unroll
      for (int cc = 0; cc < NCH; ++cc) {
        if (cc * KC < width) {
          f32x4 pre[NP];
          {  // row-ordered: one window per tile, no per-piece address arithmetic; rows past the end read 0
            const int fl = fresh_lane();
            const int col = cc * KC + (fl & 15) * 4;
            load_tile_rows(pre, base, ld, row0, rows, (uint32_t)((fl >> 4) * ld + (col < ld ? col : 0)) * 4u);
          }
          stage(pre, cc * KC, width);
#pragma unroll
          for (int tt = 0; tt < 2; ++tt)
            if (2 * cc + tt < HT) {
#pragma unroll
              for (int qq = 0; qq < 4; ++qq) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(abuf + i * LDSW + 32 * tt + 8 * qq + 4 * h);
                const int t = 2 * cc + tt < HT ? 2 * cc + tt : 0;
                acc[t][4 * qq + 0] = v.x; acc[t][4 * qq + 1] = v.y; acc[t][4 * qq + 2] = v.z; acc[t][4 * qq + 3] = v.w;
              }
            }
          compiler_lds_barrier();
        } else {
#pragma unroll
          for (int tt = 0; tt < 2; ++tt)
            if (2 * cc + tt < HT)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[2 * cc + tt < HT ? 2 * cc + tt : 0][r] = 0.f;
        }
      }
    };
    auto apply_mask_l = [&](f32x16 (&acc)[HT], int l) {
#pragma unroll
      for (int cc = 0; cc < NCH; ++cc)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (2 * cc + tt < HT) {
              const int t = 2 * cc + tt < HT ? 2 * cc + tt : 0;
              acc[t][r] = (mask[l][cc] >> (16 * tt + r)) & 1u ? acc[t][r] : 0.f;
            }
    };
    f32x16 hid[HT];
    if constexpr (SAVED) {
      // the forward kept its post-activations (gnc_mlp_bwd_desc_t.act_given): their tiles give the ReLU masks, the last
      // one stays in registers for the LayerNorm recompute; no forward chunk but the last Linear's is in the plan, and the
      // weight-gradient products (gnc_xty_f32) read the saved tensors instead of copies emitted here
#pragma unroll
      for (int l = 0; l < GNC_MAX_LINEAR - 1; ++l)
        if (l < L - 1) {
          load_acc_rows(hid, b.act[l], d.out_dim[l], d.out_dim[l]);
          relu_mask(hid, l);
        }
    } else {
    // ------------------------------------------------------------------ forward recompute
    init_bias<HT>(hid, pbuf, h);
    for (int st = 0; st < pl.num_steps; ++st) {
      const int s = pl.step[st].seg, s2 = pl.step[st].seg2, c0 = pl.step[st].c0;
      f32x4 pre[NP];
      load_rows(pre, d.seg[s].ptr, d.seg[s].ld, c0, load_idx(row0, s));
      if (s2 >= 0) {
        f32x4 pre2[NP];
        load_rows(pre2, d.seg[s2].ptr, d.seg[s2].ld, c0, load_idx(row0, s2));
#pragma unroll
        for (int p = 0; p < NP; ++p) pre[p] += pre2[p];
      }
      stage(pre, c0, d.seg[s].width);
      if (pl.step[st].add) {
        switch (c0 >> 6) {
          case 0: add_chunk_from_lds<HT, 0>(hid, abuf, i, h); break;
          case 1: add_chunk_from_lds<HT, 2>(hid, abuf, i, h); break;
          case 2: add_chunk_from_lds<HT, 4>(hid, abuf, i, h); break;
          default: add_chunk_from_lds<HT, 6>(hid, abuf, i, h); break;
        }
      } else {
        prefetch_next_chunk();
        const int kc = d.seg[s].width - c0 < KC ? d.seg[s].width - c0 : KC;
        mma_chunk_from_lds<HT>(hid, abuf, cur_chunk(), (kc + 7) >> 3, i, h);
        publish_next_chunk();
      }
    }
    relu_mask(hid, 0);
    emit(hid, b.act[0], d.out_dim[0], d.out_dim[0], row0);
#pragma unroll
    for (int l = 1; l < GNC_MAX_LINEAR - 1; ++l) {
      if (l < L - 1) {
        f32x16 nxt[HT];
        init_bias<HT>(nxt, pbuf + l * PSTRIDE, h);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          if (c * KC < d.in_dim[l]) {
            prefetch_next_chunk();
            mma_chunk_from_regs<HT, HT>(nxt, hid, cur_chunk(), c, d.in_dim[l], i, h);
            publish_next_chunk();
          }
        }
        relu_mask(nxt, l);
#pragma unroll
        for (int t = 0; t < HT; ++t) hid[t] = nxt[t];
        emit(hid, b.act[l], d.out_dim[l], d.out_dim[l], row0);
      }
    }
    }  // forward recompute
    B32P(0);  // saved tiles -> masks (or the forward recompute)

    // ------------------------------------------------------------------ LayerNorm recompute FIRST (the last hidden
    // activations die with it), THEN the grad_out tile: never more than two accumulator sets live
    f32x16 y[HT];
    if (d.ln_gamma) {
      init_bias<HT>(y, pbuf + (L - 1) * PSTRIDE, h);
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (c * KC < d.in_dim[L - 1]) {
          prefetch_next_chunk();
          mma_chunk_from_regs<HT, HT>(y, hid, cur_chunk(), c, d.in_dim[L - 1], i, h);
          publish_next_chunk();
        }
      }
    }
    B32P(1);  // last Linear (LayerNorm statistics)
    f32x16 g[HT];
#pragma unroll
    for (int cc = 0; cc < NCH; ++cc) {
      if (cc * KC < out_dim) {
        f32x4 pre[NP];
        {  // row-ordered: one window per tile (rows past the end read 0: their gradients are never stored)
          const int fl = fresh_lane();
          const int col = cc * KC + (fl & 15) * 4;
          load_tile_rows(pre, b.grad_out, b.ld_grad_out, row0, rows, (uint32_t)((fl >> 4) * b.ld_grad_out + (col < b.ld_grad_out ? col : 0)) * 4u);
        }
        stage(pre, cc * KC, out_dim);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
          if (2 * cc + tt < HT) {
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
              const f32x4 v = *reinterpret_cast<const f32x4*>(abuf + i * LDSW + 32 * tt + 8 * qq + 4 * h);
              const int t = 2 * cc + tt < HT ? 2 * cc + tt : 0;
              g[t][4 * qq + 0] = v.x; g[t][4 * qq + 1] = v.y; g[t][4 * qq + 2] = v.z; g[t][4 * qq + 3] = v.w;
            }
          }
        compiler_lds_barrier();
      } else {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
          if (2 * cc + tt < HT)
#pragma unroll
            for (int r = 0; r < 16; ++r) g[2 * cc + tt < HT ? 2 * cc + tt : 0][r] = 0.f;
      }
    }
    B32P(2);  // grad_out slabs
    if (d.ln_gamma) {
      layer_norm_backward_tiles<HT>(y, g, pbuf + L * PSTRIDE, out_dim, d.ln_eps, h);
      if (b.ln_partial) {
        // d beta / d gamma in flight, one 64-column slab at a time: y_hat goes through the wave's LDS tile into whole-row
        // pieces, grad_out's pieces come back through the window (the tile read them a moment ago: L2 hits; rows past
        // the end read 0 and drop out); the four row groups are folded and the slab's 2 x 64 sums added to the wave's
        // LDS totals.  Replaces the y_hat tensor ([rows, out] written here, read back by colsum_pair with grad_out).
#pragma unroll
        for (int cc = 0; cc < NCH; ++cc) {
          if (cc * KC < out_dim) {
            const int fl = fresh_lane();
            const int c4 = fl & 15, rs = fl >> 4;
            const int col = cc * KC + c4 * 4;
            f32x4 gp[NP];
            load_tile_rows(gp, b.grad_out, b.ld_grad_out, row0, rows, (uint32_t)(rs * b.ld_grad_out + (col < out_dim ? col : 0)) * 4u);
            compiler_lds_barrier();
            switch (cc) {
              case 0: chunk_to_lds<HT, 0>(y, abuf, i, h); break;
              case 1: chunk_to_lds<HT, 2>(y, abuf, i, h); break;
              case 2: chunk_to_lds<HT, 4>(y, abuf, i, h); break;
              default: chunk_to_lds<HT, 6>(y, abuf, i, h); break;
            }
            compiler_lds_barrier();
            f32x4 sg = {0.f, 0.f, 0.f, 0.f}, sgy = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int p = 0; p < NP; ++p) {
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
        emit(y, b.yhat, out_dim, out_dim, row0);
      }
    } else {
#pragma unroll
      for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) g[t][r] = (feat_of(t, r, h) < out_dim) ? g[t][r] : 0.f;
    }
    B32P(3);  // LayerNorm backward + parameter sums
    emit(g, b.dz[L - 1], out_dim, out_dim, row0);
    B32P(4);  // emit dz_{L-1}

    // ------------------------------------------------------------------ back through the Linear layers
#pragma unroll
    for (int l = GNC_MAX_LINEAR - 2; l >= 0; --l) {
      if (l < L - 1) {
        f32x16 da[HT];
        zero_tiles<HT>(da);
        // chunks of W_{l+1} by 64 input columns: chunk c yields da tiles 2c, 2c+1
        if (0 * KC < d.in_dim[l + 1]) { prefetch_next_chunk(); mma_transposed_chunk<HT, HT, 0>(da, g, cur_chunk(), i, h); publish_next_chunk(); }
        if constexpr (NCH > 1) if (1 * KC < d.in_dim[l + 1]) { prefetch_next_chunk(); mma_transposed_chunk<HT, HT, 1>(da, g, cur_chunk(), i, h); publish_next_chunk(); }
        if constexpr (NCH > 2) if (2 * KC < d.in_dim[l + 1]) { prefetch_next_chunk(); mma_transposed_chunk<HT, HT, 2>(da, g, cur_chunk(), i, h); publish_next_chunk(); }
        if constexpr (NCH > 3) if (3 * KC < d.in_dim[l + 1]) { prefetch_next_chunk(); mma_transposed_chunk<HT, HT, 3>(da, g, cur_chunk(), i, h); publish_next_chunk(); }
        apply_mask_l(da, l);
#pragma unroll
        for (int t = 0; t < HT; ++t) g[t] = da[t];
        emit(g, b.dz[l], d.out_dim[l], d.out_dim[l], row0);
      }
    }
    B32P(5);  // transposed products + masks + emits
    if (b.dx) {  // one 64-column slab of dx per MATMUL step, in step order (= the order of the plan's last chunks)
      for (int st = 0; st < pl.num_steps; ++st) {
        if (pl.step[st].add) continue;
        const int s = pl.step[st].seg, c0 = pl.step[st].c0;
        f32x16 dxs[2];
        zero_tiles<2>(dxs);
        prefetch_next_chunk();
        mma_transposed_chunk<HT, 2, 0>(dxs, g, cur_chunk(), i, h);
        publish_next_chunk();
        compiler_lds_barrier();
        chunk_to_lds<2, 0>(dxs, abuf, i, h);
        // columns c0 .. c0+63 of segment s -> dx[:, wcol_s + c0 ...]
        {
          const int fl = fresh_lane();
          compiler_lds_barrier();
          // residual path (the segment is also the residual): + the same columns of grad_out, read back as whole rows
          const bool fold = b.dx_add_grad_out == s + 1;
          store_staged_rows(abuf, b.dx + d.seg[s].wcol + c0, b.ld_dx, d.seg[s].width - c0 < KC ? d.seg[s].width - c0 : KC, row0,
                            rows, fl & 15, fl >> 4, fold ? b.grad_out + c0 : nullptr, b.ld_grad_out);
          compiler_lds_barrier();
        }
      }
    }
    B32P(6);  // dx
  }
  B32P_END();
  if (b.ln_partial) {  // one row of partials per wave: [d beta (out_dim) | d gamma (out_dim)]
    compiler_lds_barrier();
    float* dst = b.ln_partial + (int64_t)((int)blockIdx.x * WAVES + wave) * 2 * out_dim;
    for (int j = fresh_lane(); j < PSTRIDE; j += 64)
      if (j < out_dim) {
        dst[j] = lnbuf[j];
        dst[out_dim + j] = lnbuf[PSTRIDE + j];
      }
  }
}

// columns [c0, c0 + 64) of rows [32 tile, 32 tile + 32) of an operand ([rows, ld], `width` columns in all) as 8 pieces of
// 16 B per lane (lane = (rs, c4): row 4 p + rs, columns 4 c4 ..), zero beyond `width` and past the last row
__device__ __forceinline__ void xty_load_tile(f32x4 (&pre)[NP], const float* base, int64_t ld, int width, int c0, bool vec,
                                              int tile, int rows, int c4, int rs) {
  const int col = c0 + c4 * 4;
  if (vec && c0 + KC <= width) {  // full slab: one buffer window per tile, rows past the end read as 0
    load_tile_rows<true>(pre, base, (int)ld, (int64_t)tile * RPW, rows, (uint32_t)(rs * (int)ld + col) * 4u);
    return;
  }
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int r = tile * RPW + p * 4 + rs;
    const int rc = r < rows ? r : rows - 1;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (vec) {
      const int cc = col < width ? col : 0;
      v = *reinterpret_cast<const f32x4*>(base + (int64_t)rc * ld + cc);
    } else {  // narrow / unaligned operand (e.g. the 3-column inputs): scalar, still unconditional
      const float* q = base + (int64_t)rc * ld;
      v.x = q[col + 0 < width ? col + 0 : 0]; v.y = q[col + 1 < width ? col + 1 : 0];
      v.z = q[col + 2 < width ? col + 2 : 0]; v.w = q[col + 3 < width ? col + 3 : 0];
    }
    const bool rok = r < rows;
    v.x = (rok && col + 0 < width) ? v.x : 0.f; v.y = (rok && col + 1 < width) ? v.y : 0.f;
    v.z = (rok && col + 2 < width) ? v.z : 0.f; v.w = (rok && col + 3 < width) ? v.w : 0.f;
    pre[p] = v;
  }
}

// ---------------------------------------------------------------------------------------------------
// xty: C[M, K] = sum over rows of A[row, 0:M]^T B[row, 0:K];  colsum[M] = sum over rows of A[row, 0:M]
// (dW = dz^T a, db = colsum(dz)).  M, K <= 64.  Each wave streams 32-row tiles with the rows on the MFMA
// k axis (2 rows per step: lane (i, h) holds A[row 2s+h][32 tA + i] and B[row 2s+h][32 tB + i]: 128-B
// contiguous per half-wave) and keeps its own partial C; partials are summed afterwards in a fixed order.
// ---------------------------------------------------------------------------------------------------
template <int TM, int TK>
__global__ __launch_bounds__(256) void xty_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
                                                  int64_t ldb, int rows, int M, int K, float* __restrict__ partial,
                                                  int pstride) {
  // Rows arrive as whole 16-B pieces (coalesced), are parked in wave-private LDS tiles (one [32][64] tile per 64
  // columns of an operand) and are read back down the columns (ds_read_b32) as MFMA operands: lane (i, h) of step s
  // needs row 2s+h, column 32t+i.  TM, TK = 32-column blocks of A and B (1, 2 or 4: operands up to 128 columns, so a
  // 256 x 256 weight gradient is four launches that each read their two [rows, 128] slabs once).
  extern __shared__ __attribute__((aligned(16))) float xty_lds[];
  constexpr int NA = (TM + 1) / 2, NB = (TK + 1) / 2;  // 64-column LDS tiles per operand
  const int lane = threadIdx.x & 63;
  const int i = lane & 31;
  const int h = lane >> 5;
  const int c4 = lane & 15, rs = lane >> 4;
  const int wib = threadIdx.x >> 6;
  const int wave = ((int)blockIdx.x * (int)blockDim.x + (int)threadIdx.x) >> 6;
  const int total_waves = ((int)gridDim.x * (int)blockDim.x) >> 6;
  const int num_tiles = (rows + RPW - 1) / RPW;
  float* ta = xty_lds + wib * (NA + NB) * RPW * LDSW;
  float* tb = ta + NA * RPW * LDSW;
  const bool avec = (lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(A) & 15u) == 0);
  const bool bvec = (ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(B) & 15u) == 0);
  f32x16 acc[TM][TK];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int c = 0; c < TK; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
  float csum[TM];
#pragma unroll
  for (int a = 0; a < TM; ++a) csum[a] = 0.f;

  auto load_tile = [&](f32x4 (&pre)[NP], const float* base, int64_t ld, int width, int c0, bool vec, int tile) {
    xty_load_tile(pre, base, ld, width, c0, vec, tile, rows, c4, rs);
  };

  int tile = wave;
  f32x4 pa[NA][NP], pb[NB][NP];
  if (tile < num_tiles) {
#pragma unroll
    for (int u = 0; u < NA; ++u) load_tile(pa[u], A, lda, M, u * KC, avec, tile);
#pragma unroll
    for (int u = 0; u < NB; ++u) load_tile(pb[u], B, ldb, K, u * KC, bvec, tile);
  }
  for (; tile < num_tiles; tile += total_waves) {
    compiler_lds_barrier();
#pragma unroll
    for (int p = 0; p < NP; ++p) {
#pragma unroll
      for (int u = 0; u < NA; ++u) *reinterpret_cast<f32x4*>(ta + u * RPW * LDSW + (p * 4 + rs) * LDSW + c4 * 4) = pa[u][p];
#pragma unroll
      for (int u = 0; u < NB; ++u) *reinterpret_cast<f32x4*>(tb + u * RPW * LDSW + (p * 4 + rs) * LDSW + c4 * 4) = pb[u][p];
    }
    compiler_lds_barrier();
    const int nt = tile + total_waves;  // next tile's rows fly while this one is multiplied (clamped: always legal)
    const int ntc = nt < num_tiles ? nt : num_tiles - 1;
#pragma unroll
    for (int u = 0; u < NA; ++u) load_tile(pa[u], A, lda, M, u * KC, avec, ntc);
#pragma unroll
    for (int u = 0; u < NB; ++u) load_tile(pb[u], B, ldb, K, u * KC, bvec, ntc);
#pragma unroll 2
    for (int s = 0; s < RPW / 2; ++s) {
      float av[TM], bv[TK];
#pragma unroll
      for (int a = 0; a < TM; ++a) {
        av[a] = ta[(a >> 1) * RPW * LDSW + (2 * s + h) * LDSW + 32 * (a & 1) + i];
        csum[a] += av[a];
      }
#pragma unroll
      for (int c = 0; c < TK; ++c) bv[c] = tb[(c >> 1) * RPW * LDSW + (2 * s + h) * LDSW + 32 * (c & 1) + i];
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int c = 0; c < TK; ++c) acc[a][c] = mfma(av[a], bv[c], acc[a][c]);
    }
  }
  // partial layout per wave: [M*K] row-major C, then [M] column sums
  float* dst = partial + (int64_t)wave * pstride;
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int c = 0; c < TK; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int k = 32 * c + i;
        if (m < M && k < K) dst[m * K + k] = acc[a][c][r];
      }
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    const float tot = csum[a] + __shfl_xor(csum[a], 32, 64);
    if (h == 0 && 32 * a + i < M) dst[M * K + 32 * a + i] = tot;
  }
}

// ---------------------------------------------------------------------------------------------------
// xty_wg: the same product for WIDE operands (both above 64 columns, up to 256 x 256) with the row tile SHARED by a
// workgroup.  xty_kernel gives every wave its own rows and its own full [M, K] accumulator: at 128 x 128 that is 256
// accumulator registers (one wave per SIMD: nothing covers the wave's own row loads, and the launch's MFMA time and
// HBM time add up - 1.75 + 1.4 = 3.2 ms measured at c2), and a 256 x 256 gradient is four launches that read every
// row slab twice.  Here the WVM x WVK waves of a workgroup own one (32 WM) x (32 WK) block of C each and multiply the
// SAME 32-row tile: every wave brings in one 64-column slab of it (8 x 16 B per lane), parks it in the workgroup's
// LDS tile set (double buffered: one barrier per tile) and reads its own operand columns down the rows.  Accumulators
// per wave: 64 registers (128 x 128: 4 waves, two workgroups per CU) or 128 (256 x 256: 8 waves), so two waves share
// a SIMD and each row of A and B is read from HBM once per launch.  One partial C per WORKGROUP.
// ---------------------------------------------------------------------------------------------------
template <int WM, int WK, int WVM, int WVK>
__global__ __launch_bounds__(64 * WVM * WVK) void xty_wg_kernel(const float* __restrict__ A, int64_t lda,
                                                                const float* __restrict__ B, int64_t ldb, int rows, int M,
                                                                int K, float* __restrict__ partial, int pstride) {
  static_assert((WM % 2 == 0 || WM == 1) && (WK % 2 == 0 || WK == 1), "a per-wave block may not straddle 64-column LDS tiles");
  extern __shared__ __attribute__((aligned(16))) float xty_lds[];
  constexpr int NW = WVM * WVK;
  constexpr int NA = (WM * WVM + 1) / 2, NB = (WK * WVK + 1) / 2, NT = NA + NB;  // 64-column LDS tiles of A, of B
  constexpr int LPW = (NT + NW - 1) / NW;                            // slabs a wave brings in per row tile
  constexpr int TILE = RPW * LDSW;
  const int lane = threadIdx.x & 63;
  const int i = lane & 31;
  const int h = lane >> 5;
  const int c4 = lane & 15, rs = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int wm = w / WVK, wk = w % WVK;
  const int num_tiles = (rows + RPW - 1) / RPW;
  const bool avec = (lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(A) & 15u) == 0);
  const bool bvec = (ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(B) & 15u) == 0);
  f32x16 acc[WM][WK];
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int c = 0; c < WK; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
  float csum[WM];
#pragma unroll
  for (int a = 0; a < WM; ++a) csum[a] = 0.f;

  auto load_slab = [&](f32x4 (&pre)[NP], int t, int tile) {  // slab t < NA: A columns [64 t, ..), else B columns [64 (t - NA), ..)
    const bool isa = t < NA;
    xty_load_tile(pre, isa ? A : B, isa ? lda : ldb, isa ? M : K, (isa ? t : t - NA) * KC, isa ? avec : bvec, tile, rows, c4, rs);
  };

  int tile = (int)blockIdx.x;
  f32x4 pre[LPW][NP];
  if (tile < num_tiles) {
#pragma unroll
    for (int j = 0; j < LPW; ++j)
      if (w + j * NW < NT) load_slab(pre[j], w + j * NW, tile);
  }
  // this wave's operand columns inside a tile set (loop constants; the two tile sets are NT * TILE floats apart)
  const float* arow = xty_lds + ((wm * WM) >> 1) * TILE + 32 * ((wm * WM) & 1) + h * LDSW + i;
  const float* brow = xty_lds + (NA + ((wk * WK) >> 1)) * TILE + 32 * ((wk * WK) & 1) + h * LDSW + i;
  int cur = 0;
  for (; tile < num_tiles; tile += (int)gridDim.x) {
    float* set = xty_lds + cur * NT * TILE;
#pragma unroll
    for (int j = 0; j < LPW; ++j)
      if (w + j * NW < NT) {
        float* dstt = set + (w + j * NW) * TILE + rs * LDSW + c4 * 4;
#pragma unroll
        for (int p = 0; p < NP; ++p) *reinterpret_cast<f32x4*>(dstt + p * 4 * LDSW) = pre[j][p];
      }
    __syncthreads();  // the tile set is complete; the other set was last read before the previous barrier
    const int nt = tile + (int)gridDim.x;
    const int ntc = nt < num_tiles ? nt : num_tiles - 1;  // clamped: always a legal read, ignored after the last tile
#pragma unroll
    for (int j = 0; j < LPW; ++j)
      if (w + j * NW < NT) load_slab(pre[j], w + j * NW, ntc);
    const float* ar = arow + cur * NT * TILE;
    const float* br = brow + cur * NT * TILE;
    // operand columns one step ahead of the MFMAs that use them (two register sets, fully unrolled: the LDS round trip
    // of step s + 1 runs under the MFMAs of step s)
    float av[2][WM], bv[2][WK];
    auto read_step = [&](int s, int set) {
#pragma unroll
      for (int a = 0; a < WM; ++a) av[set][a] = ar[(a >> 1) * TILE + 2 * s * LDSW + 32 * (a & 1)];
#pragma unroll
      for (int c = 0; c < WK; ++c) bv[set][c] = br[(c >> 1) * TILE + 2 * s * LDSW + 32 * (c & 1)];
    };
    read_step(0, 0);
#pragma unroll
    for (int s = 0; s < RPW / 2; ++s) {
      if (s + 1 < RPW / 2) read_step(s + 1, (s + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);  // keep the reads in front of this step's MFMAs (the scheduler sinks them to their use)
#pragma unroll
      for (int a = 0; a < WM; ++a) csum[a] += av[s & 1][a];
#pragma unroll
      for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int c = 0; c < WK; ++c) acc[a][c] = mfma(av[s & 1][a], bv[s & 1][c], acc[a][c]);
    }
    cur ^= 1;
  }
  // partial layout per workgroup: [M*K] row-major C, then [M] column sums (written by the waves of the first block column)
  float* dst = partial + (int64_t)blockIdx.x * pstride;
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int c = 0; c < WK; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = 32 * (wm * WM + a) + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int k = 32 * (wk * WK + c) + i;
        if (m < M && k < K) dst[m * K + k] = acc[a][c][r];
      }
  if (wk == 0) {
#pragma unroll
    for (int a = 0; a < WM; ++a) {
      const float tot = csum[a] + __shfl_xor(csum[a], 32, 64);
      const int m = 32 * (wm * WM + a) + i;
      if (h == 0 && m < M) dst[M * K + m] = tot;
    }
  }
}

// colsum_pair: partial[w] = [ column sums of G | column sums of G*Y ] over the rows wave w streamed
// (d beta and d gamma of the LayerNorm: G = grad_out, Y = yhat).  width <= 64, 16-B aligned rows.
__global__ __launch_bounds__(256) void colsum_pair_kernel(const float* __restrict__ G, int64_t ldg,
                                                          const float* __restrict__ Y, int64_t ldy, int rows, int width,
                                                          float* __restrict__ partial) {
  const int lane = threadIdx.x & 63;
  const int c4 = lane & 15, rs = lane >> 4;
  const int wave = ((int)blockIdx.x * (int)blockDim.x + (int)threadIdx.x) >> 6;
  const int total_waves = ((int)gridDim.x * (int)blockDim.x) >> 6;
  const int col = c4 * 4 < width ? c4 * 4 : 0;
  f32x4 sg = {0.f, 0.f, 0.f, 0.f}, sgy = {0.f, 0.f, 0.f, 0.f};
  for (int r0 = wave * 16; r0 < rows; r0 += total_waves * 16) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int r = r0 + p * 4 + rs;
      const int rc = r < rows ? r : rows - 1;
      const f32x4 g = *reinterpret_cast<const f32x4*>(G + (int64_t)rc * ldg + col);
      const f32x4 y = *reinterpret_cast<const f32x4*>(Y + (int64_t)rc * ldy + col);
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      const f32x4 gm = r < rows ? g : z;
      sg += gm;
      sgy += gm * y;
    }
  }
  // the four row groups of the wave hold the same columns: fold them
#pragma unroll
  for (int off = 16; off <= 32; off <<= 1) {
    sg.x += __shfl_xor(sg.x, off, 64); sg.y += __shfl_xor(sg.y, off, 64);
    sg.z += __shfl_xor(sg.z, off, 64); sg.w += __shfl_xor(sg.w, off, 64);
    sgy.x += __shfl_xor(sgy.x, off, 64); sgy.y += __shfl_xor(sgy.y, off, 64);
    sgy.z += __shfl_xor(sgy.z, off, 64); sgy.w += __shfl_xor(sgy.w, off, 64);
  }
  if (rs == 0 && c4 * 4 < width) {
    float* dst = partial + (int64_t)wave * 2 * width;
    const float a[4] = {sg.x, sg.y, sg.z, sg.w}, c[4] = {sgy.x, sgy.y, sgy.z, sgy.w};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (c4 * 4 + k < width) { dst[c4 * 4 + k] = a[k]; dst[width + c4 * 4 + k] = c[k]; }
  }
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// schedule shape shared by the support query and the launcher; returns 0 if unsupported
int bwd_shape(const gnc_mlp_desc_t& d, int* nmm_out, int* nadd_out, int* T_out) {
  const int L = d.num_linear;
  if (L < 2 || L > GNC_MAX_LINEAR - 1 || d.activation != GNC_ACT_RELU || d.rows >= INT32_MAX || d.rows < 1) return 0;
  const int H = d.out_dim[0], od = d.out_dim[L - 1];
  if (H > KC || od > KC) return 0;
  for (int l = 1; l < L; ++l)
    if (d.in_dim[l] > KC || d.out_dim[l] > KC) return 0;
  int nmm = 0, nadd = 0;
  for (int s = 0; s < d.num_segments; ++s) {
    const gnc_mlp_segment_t& g = d.seg[s];
    if (g.width > KC || g.ld % 4 != 0 || !al16(g.ptr)) return 0;
    if (g.mode == GNC_SEG_ADD) {
      ++nadd;
    } else {
      if (nadd) return 0;
      ++nmm;
    }
  }
  if (nmm < 1 || nmm > 3 || (nadd != 0 && !(nadd == 2 && nmm == 1))) return 0;
  const int T = tiles_for(H > od ? H : od);
  if (T > 2) return 0;
  *nmm_out = nmm;
  *nadd_out = nadd;
  *T_out = T;
  return 1;
}

template <int HT, int NMM, int NADD, bool SAVED = false>
int launch_bwd(const gnc_mlp_desc_t& d, const BwdArgs& b, int total_chunks, size_t smem, int grid, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    int rc = gnc::check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_backward_data_kernel<HT, NMM, NADD, SAVED>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                            "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    if (rc) return rc;
    attr_set = true;
  }
  const int64_t num_wtiles = gnc::ceil_div(d.rows, RPW);
  mlp_backward_data_kernel<HT, NMM, NADD, SAVED>
      <<<dim3((unsigned)grid), dim3(BNT), smem, stream>>>(d, b, (int)num_wtiles, total_chunks);
  return gnc::check_launch("mlp_backward_data_kernel");
}

// plan of the streamed variant; returns false if the shape is outside it
bool bwd_stream_plan(const gnc_mlp_desc_t& d, bool want_dx, BwdPlan* pl, int* T_out, bool saved = false) {
  const int L = d.num_linear;
  if (L < 2 || L > GNC_MAX_LINEAR - 1 || d.activation != GNC_ACT_RELU || d.rows >= INT32_MAX || d.rows < 1) return false;
  int wmax = 0;
  for (int l = 0; l < L; ++l) {
    if (d.out_dim[l] > wmax) wmax = d.out_dim[l];
    if (l > 0 && d.in_dim[l] > wmax) wmax = d.in_dim[l];
    if (ldw_of(d, l) % 4 != 0 || !al16(d.weight[l])) return false;
  }
  if (wmax > 128) return false;  // four 64-register accumulator sets is what 512 registers hold
  const int T = tiles_for(wmax);
  *pl = BwdPlan{};
  int add_seg[GNC_MAX_SEGMENTS], nadd = 0;
  for (int s = 0; s < d.num_segments; ++s) {
    const gnc_mlp_segment_t& g = d.seg[s];
    if (g.ld % 4 != 0 || !al16(g.ptr)) return false;
    if (g.mode == GNC_SEG_ADD) { add_seg[nadd++] = s; continue; }
    if (g.wcol % 4 != 0) return false;
    for (int c0 = 0; c0 < g.width; c0 += KC) {
      if (pl->num_steps >= SB_MAX_STEPS || pl->num_wchunks >= SB_MAX_WCHUNKS) return false;
      pl->step[pl->num_steps++] = {(short)s, (short)-1, (short)c0, (short)0};
      pl->wc[pl->num_wchunks++] = {(short)0, (short)(g.wcol + c0), (short)(g.wcol + g.width), 0};
    }
  }
  if (pl->num_wchunks == 0) return false;
  const int mm_chunks = pl->num_wchunks;
  std::remove_reference_t<decltype(pl->wc[0])> mm_wc[SB_MAX_WCHUNKS];
  for (int k = 0; k < mm_chunks; ++k) mm_wc[k] = pl->wc[k];
  if (saved) {  // the forward's post-activations are inputs: no chunk of the first Linear or of the hidden layers going forward
    pl->saved = 1;
    pl->num_wchunks = 0;
  }
  for (int a = 0; a < nadd; a += 2) {
    const int s = add_seg[a], s2 = a + 1 < nadd ? add_seg[a + 1] : -1;
    for (int c0 = 0; c0 < d.seg[s].width; c0 += KC) {
      if (pl->num_steps >= SB_MAX_STEPS) return false;
      pl->step[pl->num_steps++] = {(short)s, (short)s2, (short)c0, (short)1};
    }
  }
  auto push_layer = [&](int l) {
    for (int c = 0; c * KC < d.in_dim[l]; ++c) {
      if (pl->num_wchunks >= SB_MAX_WCHUNKS) return false;
      pl->wc[pl->num_wchunks++] = {(short)l, (short)(c * KC), (short)d.in_dim[l], 0};
    }
    return true;
  };
  if (!saved)
    for (int l = 1; l < L - 1; ++l)
      if (!push_layer(l)) return false;        // forward, hidden layers
  if (d.ln_gamma && !push_layer(L - 1)) return false;  // forward of the last Linear only feeds the LayerNorm statistics
  for (int l = L - 2; l >= 0; --l)
    if (!push_layer(l + 1)) return false;      // backward: W_{l+1}^T
  if (want_dx)
    for (int k = 0; k < mm_chunks; ++k) {      // dx: the first Linear's chunks again, in step order
      if (pl->num_wchunks >= SB_MAX_WCHUNKS) return false;
      pl->wc[pl->num_wchunks] = mm_wc[k];
      ++pl->num_wchunks;
    }
  *T_out = T;
  return true;
}

template <int HT, int WAVES, bool SAVED = false>
int launch_bwd_stream(const gnc_mlp_desc_t& d, const BwdArgs& b, const BwdPlan& pl, hipStream_t stream) {
  const size_t smem = ((size_t)2 * HT * 32 * LDSW + (size_t)(d.num_linear + 2) * HT * 32 + (size_t)WAVES * RPW * LDSW +
                       (size_t)WAVES * 2 * HT * 32) * sizeof(float);
  if (smem > 160 * 1024) { gnc::set_error("mlp_backward_stream: LDS budget exceeded"); return GNC_ERR_UNSUPPORTED; }
  static bool attr_set = false;
  if (!attr_set) {
    int rc = gnc::check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_backward_stream_kernel<HT, WAVES, SAVED>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                            "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    if (rc) return rc;
    attr_set = true;
  }
  const int64_t num_tiles = gnc::ceil_div(d.rows, (int64_t)WAVES * RPW);
  const int64_t grid = num_tiles < gnc::num_cu() ? num_tiles : gnc::num_cu();
  mlp_backward_stream_kernel<HT, WAVES, SAVED><<<dim3((unsigned)grid), dim3(WAVES * 64), smem, stream>>>(d, b, pl, (int)num_tiles);
  return gnc::check_launch("mlp_backward_stream_kernel");
}

// waves per workgroup of the 32-row streaming kernel for T 32-column tiles (the dispatch below and the partial-row query)
int bwd_stream_waves(int T, int64_t rows) {
  static const bool w4 = getenv("GNC_BWD_STREAM_W4") != nullptr;  // A/B: the 4-wave (one per SIMD, 512 registers) form at 128
  if (T < 4) return 4;
  if (rows <= (int64_t)2 * RPW * gnc::num_cu()) return 2;  // small batch: see mlp_stream.hip
  return w4 ? 4 : 8;
}

int bwd_grid(int64_t rows) {
  int64_t grid = gnc::ceil_div(gnc::ceil_div(rows, RPW), BWAVES);
  return (int)(grid > gnc::num_cu() ? gnc::num_cu() : grid);
}

}  // namespace

namespace {
// widths 129..256: the 16-row kernel of mlp_backward16.hip (GNC_NO_BACKWARD16=1 switches it off for A/B runs)
bool use_stream16(const gnc_mlp_desc_t& d, bool want_dx) {
  static const bool off = getenv("GNC_NO_BACKWARD16") != nullptr;
  return !off && bwd_stream16_supported(d, want_dx);
}
}  // namespace

extern "C" size_t gnc_sizeof_mlp_bwd_desc(void) { return sizeof(gnc_mlp_bwd_desc_t); }

extern "C" int gnc_mlp_backward_supported(const gnc_mlp_desc_t* fwd) {
  int rc = validate_desc(fwd, false);
  if (rc) return rc;
  int nmm, nadd, T;
  BwdPlan pl;
  if (!(fwd->save_act[0] && fwd->num_linear >= 2 && bwd_col16_supported(*fwd)) && !bwd_shape(*fwd, &nmm, &nadd, &T) &&
      !bwd_stream_plan(*fwd, true, &pl, &T) && !use_stream16(*fwd, true)) {
    gnc::set_error("gnc_mlp_backward: shape outside the HIP backward kernels (needs ReLU, widths <= 256, aligned tables)");
    return GNC_ERR_UNSUPPORTED;
  }
  return GNC_OK;
}

// 1 if the small-batch data kernel (mlp_bwd_col16.hip) serves this backward exactly as the operands lie: it reads neither the
// segment tables nor (without dx) the first Linear's weight, so the caller need not make 16-B-row copies of [N, 3] inputs or
// nn.Linear(3, H) weights.  fwd.save_act must carry the saved post-activations (see gnc_mlp_bwd_desc_t).
extern "C" int gnc_mlp_backward_small_batch_supported(const gnc_mlp_desc_t* fwd) {
  if (!fwd || validate_desc(fwd, false) != GNC_OK || fwd->num_linear < 2 || !fwd->save_act[0]) return 0;
  return bwd_col16_supported(*fwd) ? 1 : 0;
}

extern "C" int gnc_mlp_backward_dx_add_honoured(const gnc_mlp_desc_t* fwd) {
  int nmm, nadd, T;
  BwdPlan pl;
  static const bool off = getenv("GNC_NO_STREAM_DX_FOLD") != nullptr;  // A/B switch: the streaming kernels leave the add to the caller
  if (!fwd || validate_desc(fwd, false) != GNC_OK) return 0;
  if (fwd->save_act[0] && fwd->num_linear >= 2 && (bwd_col16_supported(*fwd) || bwd_col16_persist_supported(*fwd))) return 1;
  if (bwd_shape(*fwd, &nmm, &nadd, &T)) return 1;
  return (!off && (bwd_stream_plan(*fwd, true, &pl, &T) || use_stream16(*fwd, true))) ? 1 : 0;
}


extern "C" int gnc_mlp_backward_f32(const gnc_mlp_bwd_desc_t* bd, void* stream_) {
  if (!bd) { gnc::set_error("gnc_mlp_backward_f32: null descriptor"); return GNC_ERR_INVALID_ARGUMENT; }
  const gnc_mlp_desc_t& d = bd->fwd;
  int rc = validate_desc(&d, false);
  if (rc) return rc;
  if (d.ef_pos) {  // the backward reads the segment's rows (dW_0): a training forward materialises them (gnc_edge_features_f32)
    gnc::set_error("gnc_mlp_backward_f32: fwd.ef_pos (computed edge features) is an inference-only form");
    return GNC_ERR_UNSUPPORTED;
  }
  int nmm = 0, nadd = 0, T = 0;
  BwdPlan pl;
  if (bd->dw_partial[0]) {  // fused data + weight-gradient kernel
    const int fn = fused_shape(d);
    if (fn < 0) { gnc::set_error("gnc_mlp_backward_f32: dw_partial given but the shape is outside the fused kernel"); return GNC_ERR_UNSUPPORTED; }
    GNC_REQUIRE(bd->dw_partial[1] && bd->dw_partial[2], "gnc_mlp_backward_f32: dw_partial[0..2] must all be given");
    GNC_REQUIRE(!d.ln_gamma || bd->ln_partial, "gnc_mlp_backward_f32: the fused kernel needs ln_partial with LayerNorm");
    GNC_REQUIRE(!bd->grad_gather || gnc_mlp_backward_grad_gather_honoured(bd) == 1,
                "gnc_mlp_backward_f32: grad_gather is not honoured for this description (gnc_mlp_backward_grad_gather_honoured)");
    GNC_REQUIRE(bd->grad_out || bd->grad_gather, "gnc_mlp_backward_f32: grad_out is null");
    GNC_REQUIRE(!bd->grad_out || (bd->ld_grad_out % 4 == 0 && al16(bd->grad_out) && bd->ld_grad_out >= d.out_dim[2]),
                "gnc_mlp_backward_f32: grad_out must be 16-B aligned with ld %% 4 == 0");
    GNC_REQUIRE(!bd->dx || bd->ld_dx >= d.in_dim[0], "gnc_mlp_backward_f32: ld_dx < in_dim[0]");
    BwdArgs fb = {};
    fb.gg = bd->grad_gather;
    fb.gg_index = bd->grad_gather_index;
    fb.ld_gg = bd->ld_grad_gather;
    fb.gg_rows = bd->grad_gather_rows;
    fb.g_sum = bd->grad_sum;
    fb.ld_g_sum = bd->ld_grad_sum;
    GNC_REQUIRE(!(bd->grad_gather && bd->grad_out && bd->dx && bd->dx_add_grad_out) ||
                    (bd->grad_sum && al16(bd->grad_sum) && bd->ld_grad_sum % 4 == 0 && bd->ld_grad_sum >= d.out_dim[2]),
                "gnc_mlp_backward_f32: grad_out + grad_gather with dx_add_grad_out needs the grad_sum scratch tensor");
    fb.grad_out = bd->grad_out;
    fb.ld_grad_out = bd->ld_grad_out;
    fb.dz[0] = bd->dz[0];
    fb.dx = bd->dx;
    fb.ld_dx = bd->ld_dx;
    fb.dx_add_grad_out = bd->dx_add_grad_out ? 1 : 0;
    fb.ln_partial = d.ln_gamma ? bd->ln_partial : nullptr;
    FusedOut fo = {};
    for (int l = 0; l < 3; ++l) {
      fo.dw[l] = bd->dw_partial[l];
      fo.M[l] = d.out_dim[l];
      fo.K[l] = l == 0 ? d.seg[0].width : d.in_dim[l];
    }
    if (bd->act_given) {
      GNC_REQUIRE(gnc_mlp_backward_saved_act_honoured(bd) == 1,
                  "gnc_mlp_backward_f32: act_given is not honoured for this description (gnc_mlp_backward_saved_act_honoured)");
      fb.act[0] = bd->act[0];
      fb.act[1] = bd->act[1];
      return launch_fused_saved(d, fb, fo, fn, (hipStream_t)stream_);
    }
    return launch_fused_recompute(d, fb, fo, fn, (hipStream_t)stream_);
  }
  if (bd->act_given && bwd_col16_supported(d)) return launch_bwd_col16(*bd, (hipStream_t)stream_);  // small batches, saved activations
  if (bd->act_given && bwd_col16_persist_supported(d)) return launch_bwd_col16_persist(*bd, (hipStream_t)stream_);  // large batches at 128 features
  GNC_REQUIRE(!bd->grad_gather, "gnc_mlp_backward_f32: grad_gather is only honoured by the fused data + weight-gradient kernel");
  const bool resident = bwd_shape(d, &nmm, &nadd, &T);
  const bool saved = bd->act_given != 0;
  GNC_REQUIRE(!saved || gnc_mlp_backward_saved_act_honoured(bd) == 1,
              "gnc_mlp_backward_f32: act_given is not honoured for this description (gnc_mlp_backward_saved_act_honoured)");
  const bool stream32 = !resident && bwd_stream_plan(d, bd->dx != nullptr, &pl, &T, saved);
  const bool stream16 = !resident && !stream32 && use_stream16(d, bd->dx != nullptr);
  if (!resident && !stream32 && !stream16) {
    gnc::set_error("gnc_mlp_backward_f32: shape outside the HIP backward kernels");
    return GNC_ERR_UNSUPPORTED;
  }
  const int L = d.num_linear;
  GNC_REQUIRE(bd->grad_out && bd->ld_grad_out % 4 == 0 && al16(bd->grad_out) && bd->ld_grad_out >= d.out_dim[L - 1],
              "gnc_mlp_backward_f32: grad_out must be 16-B aligned with ld %% 4 == 0");
  for (int l = 0; l < L; ++l) {
    GNC_REQUIRE(bd->dz[l], "gnc_mlp_backward_f32: dz[%d] is null", l);
    GNC_REQUIRE(l == L - 1 || bd->act[l], "gnc_mlp_backward_f32: act[%d] is null", l);
  }
  const int grid = bwd_grid(d.rows);
  GNC_REQUIRE(!d.ln_gamma || bd->yhat || bd->ln_partial, "gnc_mlp_backward_f32: with LayerNorm either yhat or ln_partial is required");
  GNC_REQUIRE(!bd->dx || bd->ld_dx >= d.in_dim[0], "gnc_mlp_backward_f32: ld_dx < in_dim[0]");

  BwdArgs b = {};
  b.grad_out = bd->grad_out;
  b.ld_grad_out = bd->ld_grad_out;
  for (int l = 0; l < L; ++l) { b.act[l] = bd->act[l]; b.dz[l] = bd->dz[l]; }
  b.dx = bd->dx;
  b.ld_dx = bd->ld_dx;
  b.yhat = d.ln_gamma ? bd->yhat : nullptr;
  // resident kernel: a flag (its last MATMUL segment); streaming kernels: 1 + the index of that segment
  int last_mm = -1;
  for (int sgi = 0; sgi < d.num_segments; ++sgi)
    if (d.seg[sgi].mode == GNC_SEG_MATMUL) last_mm = sgi;
  b.dx_add_grad_out = !bd->dx_add_grad_out ? 0 : resident ? 1 : last_mm + 1;
  GNC_REQUIRE(!bd->dx_add_grad_out || (last_mm >= 0 && d.seg[last_mm].width == d.out_dim[L - 1] && !d.seg[last_mm].index),
              "gnc_mlp_backward_f32: dx_add_grad_out needs a row-ordered last MATMUL segment as wide as the output");
  b.ln_partial = d.ln_gamma ? bd->ln_partial : nullptr;

  if (stream16) return launch_bwd_stream16(d, b, (hipStream_t)stream_, saved);
  if (!resident) {
    hipStream_t st = (hipStream_t)stream_;
    switch (T) {
      case 1: return launch_bwd_stream<1, 4>(d, b, pl, st);
      case 2: return launch_bwd_stream<2, 4>(d, b, pl, st);
      default: {
        const int wv = bwd_stream_waves(T, d.rows);
        if (pl.saved) {
          if (wv == 2) return launch_bwd_stream<4, 2, true>(d, b, pl, st);
          return wv == 4 ? launch_bwd_stream<4, 4, true>(d, b, pl, st) : launch_bwd_stream<4, 8, true>(d, b, pl, st);
        }
        if (wv == 2) return launch_bwd_stream<4, 2>(d, b, pl, st);
        return wv == 4 ? launch_bwd_stream<4, 4>(d, b, pl, st) : launch_bwd_stream<4, 8>(d, b, pl, st);
      }
    }
  }
  const int total_chunks = nmm + (L - 1);
  const size_t smem =
      ((size_t)total_chunks * T * 32 * LDSW + (size_t)(L + 2) * T * 32 + (size_t)BWAVES * RPW * LDSW) * sizeof(float);
  if (smem > 160 * 1024) { gnc::set_error("gnc_mlp_backward_f32: weights do not fit in LDS"); return GNC_ERR_UNSUPPORTED; }
  hipStream_t stream = (hipStream_t)stream_;
#define GNC_BWD(HT_, NMM_, NADD_) return launch_bwd<HT_, NMM_, NADD_>(d, b, total_chunks, smem, grid, stream)
  if (saved) return launch_bwd<2, 2, 0, true>(d, b, total_chunks, smem, grid, stream);  // (the support query admits only this shape)
  if (T == 2) {
    if (nadd == 2) GNC_BWD(2, 1, 2);
    if (nmm == 1) GNC_BWD(2, 1, 0);
    if (nmm == 2) GNC_BWD(2, 2, 0);
    GNC_BWD(2, 3, 0);
  }
  if (nadd == 2) GNC_BWD(1, 1, 2);
  if (nmm == 1) GNC_BWD(1, 1, 0);
  if (nmm == 2) GNC_BWD(1, 2, 0);
  GNC_BWD(1, 3, 0);
#undef GNC_BWD
}

extern "C" int gnc_mlp_backward_saved_act_honoured(const gnc_mlp_bwd_desc_t* bd) {
  static const bool off = getenv("GNC_NO_SAVED_ACT") != nullptr;  // A/B switch: the backward recomputes the forward of every tile
  if (off || !bd) return 0;
  const gnc_mlp_desc_t& d = bd->fwd;
  if (validate_desc(&d, false) != GNC_OK) return 0;
  for (int l = 0; l < d.num_linear - 1; ++l)  // contiguous [rows, out_dim[l]] rows read as 16-B pieces
    if (!bd->act[l] || !fused_al16(bd->act[l]) || d.out_dim[l] % 4 != 0) return 0;
  if (bd->dw_partial[0]) return fused_shape(d) >= 0 ? 1 : 0;  // fused data + weight-gradient kernel (widths <= 64)
  if (d.num_linear >= 2 && bwd_col16_supported(d)) return 1;     // small batches: the column-split data kernel
  if (bwd_col16_persist_supported(d)) return 1;
  // split path: the streaming kernels read them (and gnc_xty_f32 after them); the weights-resident data kernel recomputes
  int nmm, nadd, T;
  BwdPlan pl;
  if (d.num_linear < 2) return 0;
  // weights-resident data kernel: a SAVED instance exists for the node processors' shape (widths 33..64, two MATMUL segments)
  if (bwd_shape(d, &nmm, &nadd, &T)) return (T == 2 && nmm == 2 && nadd == 0) ? 1 : 0;
  // SAVED instances exist for the 65..128 class of the 32-row kernel and for the 16-row kernel (129..256)
  if (bwd_stream_plan(d, bd->dx != nullptr, &pl, &T, true)) return T == 4 ? 1 : 0;
  return use_stream16(d, bd->dx != nullptr) ? 1 : 0;
}

extern "C" int gnc_mlp_backward_grad_gather_honoured(const gnc_mlp_bwd_desc_t* bd) {
  static const bool off = getenv("GNC_NO_GRAD_GATHER_FOLD") != nullptr;  // A/B switch: the caller gathers the rows itself
  if (off || !bd || !bd->grad_gather || !bd->grad_gather_index) return 0;
  if (!bd->dw_partial[0]) {  // split path: only the small-batch data kernel (saved activations) gathers in the launch
    return (bd->act_given && gnc_mlp::validate_desc(&bd->fwd, false) == GNC_OK && (bwd_col16_supported(bd->fwd) || (bwd_col16_persist_supported(bd->fwd) && bd->fwd.in_dim[0] <= 128)) &&
            bd->ld_grad_gather % 4 == 0 && fused_al16(bd->grad_gather) && bd->ld_grad_gather >= bd->fwd.out_dim[bd->fwd.num_linear - 1])
               ? 1 : 0;
  }
  const gnc_mlp_desc_t& d = bd->fwd;
  if (validate_desc(&d, false) != GNC_OK || fused_shape(d) != 2) return 0;
  // the fused kernel gathers with the ids of its second ADD segment (in the edge processor both are the destination)
  if (bd->grad_gather_index != d.seg[2].index || bd->grad_gather_rows != d.seg[2].table_rows) return 0;
  const int64_t bytes = bd->grad_gather_rows * (int64_t)bd->ld_grad_gather * 4;
  if (bd->ld_grad_gather % 4 != 0 || !fused_al16(bd->grad_gather) || bd->ld_grad_gather < d.out_dim[2]) return 0;
  // ids one row past either table (rows beyond the end of the batch) must still be addressable in 32 bits
  if (bytes + 256 > 0xffffffffll || d.seg[2].table_rows * (int64_t)d.seg[2].ld * 4 + 256 > 0xffffffffll) return 0;
  return 1;
}

extern "C" int gnc_mlp_backward_fused_rows(const gnc_mlp_desc_t* fwd) {
  static const bool off = getenv("GNC_NO_FUSED_BACKWARD") != nullptr;  // A/B switch
  if (off || !fwd || gnc_mlp::validate_desc(fwd, false) != GNC_OK || fused_shape(*fwd) < 0) return 0;
  static_assert(kFusedSmem <= 160 * 1024, "fused backward kernel: LDS budget");
  return fused_grid(fwd->rows) * FWAVES;
}

extern "C" int gnc_mlp_backward_ln_partial_rows(const gnc_mlp_desc_t* fwd) {
  if (!fwd || gnc_mlp::validate_desc(fwd, false) != GNC_OK || !fwd->ln_gamma) return 0;
  static const bool off = getenv("GNC_NO_STREAM_LN_SUMS") != nullptr;  // A/B switch: the streaming kernels write y_hat again
  // backward descriptions carry the forward's saved post-activations in fwd.save_act too (same pointers as act[]): with
  // them a small batch runs the column-split data kernel, one partial row per 16-row tile
  if (fwd->save_act[0] && fwd->num_linear >= 2 && bwd_col16_supported(*fwd)) return bwd_col16_ln_partial_rows(fwd->rows);
  if (fwd->save_act[0] && bwd_col16_persist_supported(*fwd)) return bwd_col16_persist_ln_partial_rows(fwd->rows);
  int nmm = 0, nadd = 0, T = 0;
  if (bwd_shape(*fwd, &nmm, &nadd, &T)) return bwd_grid(fwd->rows) * BWAVES;
  if (off) return 0;
  // the streaming kernels: the answer must not depend on whether the caller will ask for dx (the chunk plans do)
  BwdPlan pl;
  int T0 = 0;
  const bool s32 = bwd_stream_plan(*fwd, true, &pl, &T), s32_nodx = bwd_stream_plan(*fwd, false, &pl, &T0);
  if (s32 != s32_nodx || (s32 && T != T0)) return 0;
  if (s32) {  // the 32-row kernel: one row per wave of its grid
    const int wv = bwd_stream_waves(T, fwd->rows);
    const int64_t tiles = gnc::ceil_div(fwd->rows, (int64_t)wv * RPW);
    return (int)(tiles < gnc::num_cu() ? tiles : gnc::num_cu()) * wv;
  }
  if (use_stream16(*fwd, true) && use_stream16(*fwd, false)) return bwd_stream16_ln_partial_rows(fwd->rows);
  return 0;
}

extern "C" int gnc_xty_partials(int64_t rows) {
  int64_t tiles = gnc::ceil_div(rows > 0 ? rows : 1, RPW);
  int64_t waves = tiles < 4 * gnc::num_cu() * 2 ? tiles : 4 * gnc::num_cu() * 2;  // <= 2 blocks of 4 waves per CU
  waves = (waves + 3) / 4 * 4;
  return (int)waves;
}

namespace {
template <int TM, int TK>
int launch_xty(const float* A, int64_t lda, const float* B, int64_t ldb, int rows, int M, int K, float* partial, int waves,
               hipStream_t stream) {
  constexpr int NA = (TM + 1) / 2, NB = (TK + 1) / 2;
  constexpr size_t smem = (size_t)4 * (NA + NB) * RPW * LDSW * sizeof(float);
  static_assert(smem <= 160 * 1024, "xty: LDS budget");
  static bool attr_set = false;
  if (!attr_set) {
    int rc = gnc::check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&xty_kernel<TM, TK>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                            "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    if (rc) return rc;
    attr_set = true;
  }
  xty_kernel<TM, TK><<<dim3((unsigned)(waves / 4)), dim3(256), smem, stream>>>(A, lda, B, ldb, rows, M, K, partial, M * K + M);
  return gnc::check_launch("xty_kernel");
}
inline int blocks32(int w) { return w <= 32 ? 1 : w <= 64 ? 2 : 4; }
}  // namespace

namespace {
// workgroup-shared instances (xty_wg_kernel): which one serves [M, K], 0 = the per-wave kernel
inline int xty_wg_shape(int M, int K) {
  if (M > 256 || K > 256 || M <= 64 || K <= 64) return 0;  // a narrow operand: the per-wave kernel (<= 128 columns each)
  if (M <= 128 && K <= 128) return 1;  // 128 x 128: 4 waves x (64 x 64)
  if (M <= 128) return 2;              // 128 x 256: 8 waves x (64 x 64)
  if (K <= 128) return 3;              // 256 x 128: 8 waves x (64 x 64)
  return 4;                            // 256 x 256: 8 waves x (128 x 64)
}
inline int xty_wg_groups(int shape, int64_t rows) {
  const int64_t tiles = gnc::ceil_div(rows > 0 ? rows : 1, RPW);
  const int64_t cap = (int64_t)gnc::num_cu() * (shape == 1 ? 2 : 1);
  return (int)(tiles < cap ? tiles : cap);
}
template <int WM, int WK, int WVM, int WVK>
int launch_xty_wg(const float* A, int64_t lda, const float* B, int64_t ldb, int rows, int M, int K, float* partial, int groups,
                  hipStream_t stream) {
  constexpr int NT = (WM * WVM + 1) / 2 + (WK * WVK + 1) / 2;
  constexpr size_t smem = (size_t)2 * NT * RPW * LDSW * sizeof(float);
  static_assert(smem <= 160 * 1024, "xty_wg: LDS budget");
  static bool attr_set = false;
  if (!attr_set) {
    int rc = gnc::check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&xty_wg_kernel<WM, WK, WVM, WVK>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                            "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    if (rc) return rc;
    attr_set = true;
  }
  xty_wg_kernel<WM, WK, WVM, WVK><<<dim3((unsigned)groups), dim3(64 * WVM * WVK), smem, stream>>>(A, lda, B, ldb, rows, M, K,
                                                                                                    partial, M * K + M);
  return gnc::check_launch("xty_wg_kernel");
}
}  // namespace

extern "C" int gnc_xty_partials_for(int64_t rows, int32_t M, int32_t K) {
  const int shape = xty_wg_shape(M, K);
  return shape ? xty_wg_groups(shape, rows) : gnc_xty_partials(rows);
}

extern "C" int gnc_xty_f32(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t rows, int32_t M, int32_t K,
                           float* partial, int32_t num_partials, void* stream_) {
  GNC_REQUIRE(rows >= 1 && rows < INT32_MAX && M >= 1 && K >= 1 && M <= 256 && K <= 256, "gnc_xty_f32: need 1 <= M, K <= 256");
  GNC_REQUIRE(A && B && partial && lda >= M && ldb >= K, "gnc_xty_f32: null pointer or leading dimension too small");
  const int shape = xty_wg_shape(M, K);
  GNC_REQUIRE(shape || (M <= 128 && K <= 128), "gnc_xty_f32: a narrow operand (<= 64 columns) next to one above 128: split the wide one");
  const int parts = gnc_xty_partials_for(rows, M, K);
  GNC_REQUIRE(num_partials >= parts, "gnc_xty_f32: partial buffer smaller than gnc_xty_partials_for()");
  hipStream_t stream = (hipStream_t)stream_;
  if (shape == 1) return launch_xty_wg<2, 2, 2, 2>(A, lda, B, ldb, (int)rows, M, K, partial, parts, stream);
  if (shape == 2) return launch_xty_wg<2, 2, 2, 4>(A, lda, B, ldb, (int)rows, M, K, partial, parts, stream);
  if (shape == 3) return launch_xty_wg<2, 2, 4, 2>(A, lda, B, ldb, (int)rows, M, K, partial, parts, stream);
  if (shape == 4) return launch_xty_wg<4, 2, 2, 4>(A, lda, B, ldb, (int)rows, M, K, partial, parts, stream);
  const int tm = blocks32(M), tk = blocks32(K);
#define GNC_XTY(TM_, TK_) if (tm == TM_ && tk == TK_) return launch_xty<TM_, TK_>(A, lda, B, ldb, (int)rows, M, K, partial, parts, stream)
  GNC_XTY(1, 1); GNC_XTY(1, 2); GNC_XTY(1, 4); GNC_XTY(2, 1); GNC_XTY(2, 2); GNC_XTY(2, 4); GNC_XTY(4, 1); GNC_XTY(4, 2);
#undef GNC_XTY
  gnc::set_error("gnc_xty_f32: no instance for M = %d, K = %d", M, K);
  return GNC_ERR_UNSUPPORTED;
}

namespace {
// out[j] = sum over p (a fixed order, reproducible) of partial[p * stride + j] for j < M * K + M; the first
// M * K values go to dW (row-major, leading dimension ld_dw), the last M to db.  A workgroup = 64 columns x 8
// partial-lanes: lane l of a column adds the partials p = l, l + 8, ... (loads of different lanes and of one lane's
// unrolled group fly together: the small-graph regime is latency-bound), the eight lane sums are combined through LDS
// in lane order.
__global__ __launch_bounds__(512) void reduce_partials_kernel(const float* __restrict__ partial, int num_partials, int stride,
                                                              int M, int K, float* __restrict__ dW, int64_t ld_dw,
                                                              float* __restrict__ db) {
  __shared__ float part[8][64];
  const int col = (int)threadIdx.x & 63, pl = (int)threadIdx.x >> 6;
  const int j = (int)blockIdx.x * 64 + col;
  const int mk = M * K;
  const int total = mk + (db ? M : 0);
  float a4[4] = {0.f, 0.f, 0.f, 0.f};
  if (j < total) {
    const float* p = partial + j;
    int q = pl;
    for (; q + 24 < num_partials; q += 32) {
#pragma unroll
      for (int u = 0; u < 4; ++u) a4[u] += p[(int64_t)(q + 8 * u) * stride];
    }
    for (int u = 0; q < num_partials; q += 8, ++u) a4[u] += p[(int64_t)q * stride];
  }
  part[pl][col] = (a4[0] + a4[1]) + (a4[2] + a4[3]);
  __syncthreads();
  if (pl == 0 && j < total) {
    float acc = 0.f;
#pragma unroll
    for (int l = 0; l < 8; ++l) acc += part[l][col];
    if (j < mk) dW[(int64_t)(j / K) * ld_dw + (j % K)] = acc;
    else db[j - mk] = acc;
  }
}
}  // namespace

extern "C" int gnc_reduce_partials_f32(const float* partial, int32_t num_partials, int32_t stride, int32_t M, int32_t K,
                                       float* dW, int64_t ld_dw, float* db, void* stream_) {
  GNC_REQUIRE(partial && dW && num_partials >= 1 && M >= 1 && K >= 1 && stride >= M * K + (db ? M : 0) && ld_dw >= K,
              "gnc_reduce_partials_f32: bad arguments");
  const int total = M * K + (db ? M : 0);
  reduce_partials_kernel<<<dim3((unsigned)((total + 63) / 64)), dim3(512), 0, (hipStream_t)stream_>>>(partial, num_partials, stride,
                                                                                                     M, K, dW, ld_dw, db);
  return gnc::check_launch("reduce_partials_kernel");
}

extern "C" int gnc_colsum_pair_f32(const float* G, int64_t ldg, const float* Y, int64_t ldy, int64_t rows, int32_t width,
                                   float* partial, int32_t num_partials, void* stream_) {
  GNC_REQUIRE(rows >= 1 && rows < INT32_MAX && width >= 1 && width <= 64, "gnc_colsum_pair_f32: need width <= 64");
  GNC_REQUIRE(G && Y && partial && ldg % 4 == 0 && ldy % 4 == 0 && al16(G) && al16(Y) && ldg >= width && ldy >= width,
              "gnc_colsum_pair_f32: rows must be 16-B aligned with ld %% 4 == 0");
  const int waves = gnc_xty_partials(rows);
  GNC_REQUIRE(num_partials >= waves, "gnc_colsum_pair_f32: partial buffer smaller than gnc_xty_partials()");
  colsum_pair_kernel<<<dim3((unsigned)(waves / 4)), dim3(256), 0, (hipStream_t)stream_>>>(G, ldg, Y, ldy, (int)rows, width,
                                                                                          partial);
  return gnc::check_launch("colsum_pair_kernel");
}
