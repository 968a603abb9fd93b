// K4, weights-resident variant of the fused MLP (formulation: see mlp_fused.hip).
//
// When every Linear's weights fit in the 160 KB LDS of a CU (hidden width <= 64: the config-c3
// model) the kernel keeps them there for its whole life: one persistent 8-wave workgroup per
// CU stages all weight chunks once, and from then on the waves never meet at a barrier again.
// Each wave walks its own 32-row tiles; the only per-tile memory traffic is the tile's input
// rows and its output rows.
//
// Per-tile schedule (one "step" = the <=64 columns of one concat segment for the tile's rows):
//   MATMUL steps first  - rows land in registers, go to the wave's LDS tile, feed MFMAs;
//                         the loads of the NEXT step are issued before the MFMAs start;
//   one combined ADD step - the W-split's pre-projected rows (ps[src], pd[dst]) are summed in
//                         registers, staged once and added to the pre-activation; their loads
//                         fly under the last MATMUL step's MFMAs;
//   the first step's rows of the NEXT tile are requested right after that and have the
//   remaining Linear layers, LayerNorm and epilogue (>= 8k cycles) to land.  Gather ids are
//   fetched one tile earlier still, with one coalesced load per segment and tile.
//
// Two things make the register prefetch actually overlap on gfx950 (hipcc 7.2):
//   * gfx9 counts loads and stores in ONE vmcnt and hipcc falls back to `s_waitcnt vmcnt(0)`
//     whenever both kinds are pending, which would drain every prefetch at its first use.  The
//     output stores are therefore issued from inline asm (invisible to that bookkeeping;
//     loads still complete in order among themselves, so the compiler's counted waits stay
//     correct - they can only over-wait).
//   * every load is unconditional: row ids are clamped instead of predicated (rows past the end
//     produce values that are never stored), so there is no exec-masked VMEM in the loop.
// Requirements checked by the launcher (anything else runs the streaming kernel): ReLU,
// all tables / weights / output 16-B aligned with leading dimensions % 4 == 0, every segment
// <= 64 columns, MATMUL segments listed before ADD segments, out width % 4 == 0, rows < 2^31.
#include <stdlib.h>

#include "mlp_device.h"

using namespace gnc_mlp;

namespace {

constexpr int RWAVES = 8;
constexpr int RNT = RWAVES * 64;

// HT/OT: accumulator tiles of the hidden / output width.  NMM: MATMUL steps per tile (1..3).
// NADD: additive segments merged into one step (0 or 2).  RESREG: the residual is the table of
// the LAST MATMUL step (no index): its rows are kept in registers instead of being re-read.
template <int HT, int OT, int NMM, int NADD, bool RESREG>
__global__ __launch_bounds__(RNT) void mlp_resident_kernel(const gnc_mlp_desc_t d, const int num_wtiles,
                                                           const int total_chunks, const int flags) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int WT = HT > OT ? HT : OT;
  constexpr int CH = WT * 32 * LDSW;  // floats per resident weight chunk
  constexpr int PSTRIDE = WT * 32;
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

  // ---- one-time: parameters and every weight chunk into LDS ---------------------------------
  stage_params<RNT>(pbuf, d, PSTRIDE, tid);
  {
    int chunk = 0;
    const int ldw0 = ldw_of(d, 0);
    const bool w0v = (ldw0 % 4 == 0) && ((reinterpret_cast<uintptr_t>(d.weight[0]) & 15u) == 0);
#pragma unroll
    for (int s = 0; s < NMM; ++s, ++chunk)
      stage_weights<WT * 32, RNT>(wres + chunk * CH, d.weight[0], ldw0, d.out_dim[0], d.seg[s].wcol,
                                  d.seg[s].wcol + d.seg[s].width, 16, w0v && (d.seg[s].wcol % 4 == 0), tid);
    for (int l = 1; l < L; ++l) {
      const int ldw = ldw_of(d, l);
      const bool wv = (ldw % 4 == 0) && ((reinterpret_cast<uintptr_t>(d.weight[l]) & 15u) == 0);
      for (int c = 0; c * KC < d.in_dim[l]; ++c, ++chunk)
        stage_weights<WT * 32, RNT>(wres + chunk * CH, d.weight[l], ldw, d.out_dim[l], c * KC, d.in_dim[l], 16, wv, tid);
    }
  }
  __syncthreads();

  SegView sv[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) sv[s] = {d.seg[s].ptr, d.seg[s].index, d.seg[s].ld, d.seg[s].width};

  // ---- per-wave pipeline -----------------------------------------------------------------------
  const int total_waves = (int)gridDim.x * RWAVES;
  const int last_wt = num_wtiles - 1;
  const bool NT_STREAM = (flags & 1) != 0;

  // table row of tile row (lane & 31); clamped so the load is always legal
  auto load_idx = [&](int wt, const SegView& s) -> int {
    const int wtc = wt < last_wt ? wt : last_wt;
    int r = wtc * RPW + (lane & 31);
    r = r < rows ? r : rows - 1;
    return s.index ? s.index[r] : r;
  };
  // the step's rows: lane (rs, c4) gets 16 B of row p*4+rs for p = 0..7; no predication
  auto load_rows = [&](f32x4 (&pre)[NP], const SegView& s, int idxv) {
    const int col = c4 * 4 < s.ld ? c4 * 4 : 0;
    if (s.index == nullptr && NT_STREAM) {  // rows read once, in order: keep them out of L2 / MALL
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int tr = __shfl(idxv, p * 4 + rs, 64);
        pre[p] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(s.ptr + (int64_t)tr * s.ld + col));
      }
    } else {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int tr = __shfl(idxv, p * 4 + rs, 64);
        pre[p] = *reinterpret_cast<const f32x4*>(s.ptr + (int64_t)tr * s.ld + col);
      }
    }
  };
  // registers -> the wave's LDS tile; columns at or beyond the segment width become zero
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

  int wt = (int)blockIdx.x * RWAVES + wave;
  f32x4 cur[NP];                  // rows of the step about to be staged
  f32x4 keep[RESREG ? NP : 1];    // residual rows (copy of the last MATMUL step)
  f32x4 addA[NADD ? NP : 1], addB[NADD ? NP : 1];
  // gather ids: id0 = segment 0 of the NEXT tile, ids[s>=1] = segment s of the CURRENT tile
  int ids[NS];
  load_rows(cur, sv[0], load_idx(wt, sv[0]));
#pragma unroll
  for (int s = 1; s < NS; ++s) ids[s] = load_idx(wt, sv[s]);
  int id0 = load_idx(wt + total_waves, sv[0]);

  while (wt < num_wtiles) {
    const int row0 = wt * RPW;
    const int nwt = wt + total_waves;

    // ------------------------------------------------------------------ first Linear
    f32x16 hid[HT];
    init_bias<HT>(hid, pbuf, h);
#pragma unroll
    for (int s = 0; s < NMM; ++s) {
      stage(cur, sv[s].width);
      if constexpr (RESREG) if (s == NMM - 1) {
#pragma unroll
        for (int p = 0; p < NP; ++p) keep[p] = cur[p];
      }
      // request what comes next before the MFMAs of this step start
      if (s + 1 < NMM) {
        load_rows(cur, sv[s + 1 < NMM ? s + 1 : 0], ids[s + 1 < NMM ? s + 1 : 0]);
        ids[s + 1 < NMM ? s + 1 : 0] = load_idx(nwt, sv[s + 1 < NMM ? s + 1 : 0]);
      } else {
        if constexpr (NADD > 0) {
          load_rows(addA, sv[NMM], ids[NMM]);
          load_rows(addB, sv[NMM + 1], ids[NMM + 1]);
          ids[NMM] = load_idx(nwt, sv[NMM]);
          ids[NMM + 1] = load_idx(nwt, sv[NMM + 1]);
        } else {
          load_rows(cur, sv[0], id0);
          id0 = load_idx(nwt + total_waves, sv[0]);
        }
      }
      mma_chunk_from_lds<HT>(hid, abuf, wres + s * CH, (sv[s].width + 7) >> 3, i, h);
    }
    if constexpr (NADD > 0) {
      f32x4 sum[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) sum[p] = addA[p] + addB[p];
      stage(sum, sv[NMM].width);
      load_rows(cur, sv[0], id0);  // next tile's first step: the rest of this tile to land
      id0 = load_idx(nwt + total_waves, sv[0]);
      add_tile_from_lds<HT>(hid, abuf, i, h);
    }

    if (L == 1) {  // plain projection
      if (d.ln_gamma) layer_norm_tiles<HT>(hid, pbuf + L * PSTRIDE, pbuf + (L + 1) * PSTRIDE, out_dim, d.ln_eps, h);
      compiler_lds_barrier();
      tiles_to_lds<HT>(hid, abuf, i, h);
    } else {
      relu_tiles<HT>(hid);
      // ---------------------------------------------------------------- hidden layers 1 .. L-2
      for (int l = 1; l < L - 1; ++l) {
        f32x16 nxt[HT];
        init_bias<HT>(nxt, pbuf + l * PSTRIDE, h);
        mma_chunk_from_regs<HT, HT>(nxt, hid, wres + (NMM + l - 1) * CH, 0, d.in_dim[l], i, h);
        relu_tiles<HT>(nxt);
#pragma unroll
        for (int t = 0; t < HT; ++t) hid[t] = nxt[t];
      }
      // ---------------------------------------------------------------- last Linear, LayerNorm
      f32x16 o[OT];
      init_bias<OT>(o, pbuf + (L - 1) * PSTRIDE, h);
      mma_chunk_from_regs<HT, OT>(o, hid, wres + (NMM + L - 2) * CH, 0, d.in_dim[L - 1], i, h);
      if (d.ln_gamma) layer_norm_tiles<OT>(o, pbuf + L * PSTRIDE, pbuf + (L + 1) * PSTRIDE, out_dim, d.ln_eps, h);
      compiler_lds_barrier();
      tiles_to_lds<OT>(o, abuf, i, h);
    }
    compiler_lds_barrier();
    // The compiler's counted vmcnt waits do not know about the asm stores below; make it collect the
    // next tile's prefetched rows (issued >= one Linear ago) BEFORE the stores join the queue, so that
    // no later wait for them has to sit out the stores as well.
#pragma unroll
    for (int p = 0; p < NP; ++p) asm volatile("" ::"v"(cur[p]));

    // ------------------------------------------------------------------ epilogue: whole rows out
    {
      const int col = c4 * 4;
      const bool col_ok = col < out_dim;
      const bool vec_out = (out_dim % 4 == 0) && (d.ld_out % 4 == 0) && ((reinterpret_cast<uintptr_t>(d.out) & 15u) == 0);
      f32x4 outv[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        outv[p] = *reinterpret_cast<const f32x4*>(abuf + (p * 4 + rs) * LDSW + col);
        if constexpr (RESREG) outv[p] += keep[p];
      }
      if (!RESREG && d.residual) {  //  // rows this tile has just read: L2 hits; unconditional, clamped
        const int rc = col < d.ld_residual ? col : 0;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          int r = row0 + p * 4 + rs;
          r = r < rows ? r : rows - 1;
          outv[p] += *reinterpret_cast<const f32x4*>(d.residual + (int64_t)r * d.ld_residual + rc);
        }
      }
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int r = row0 + p * 4 + rs;
        store_row_piece(d.out + (int64_t)(r < rows ? r : rows - 1) * d.ld_out, col, outv[p], r < rows && col_ok, out_dim,
                        vec_out);
      }
    }
    compiler_lds_barrier();
    wt = nwt;
  }
}

template <int HT, int OT, int NMM, int NADD, bool RESREG>
int launch(const gnc_mlp_desc_t& d, int total_chunks, size_t smem, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    int rc = gnc::check_hip(
        hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_resident_kernel<HT, OT, NMM, NADD, RESREG>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
        "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    if (rc) return rc;
    attr_set = true;
  }
  const int64_t num_wtiles = gnc::ceil_div(d.rows, RPW);
  int64_t grid = gnc::ceil_div(num_wtiles, RWAVES);
  if (grid > gnc::kNumCU) grid = gnc::kNumCU;  // one persistent workgroup per CU
  static const int flags = getenv("GNC_MLP_NT") ? atoi(getenv("GNC_MLP_NT")) : 0;  // bit 0: nontemporal streamed rows
  mlp_resident_kernel<HT, OT, NMM, NADD, RESREG>
      <<<dim3((unsigned)grid), dim3(RNT), smem, stream>>>(d, (int)num_wtiles, total_chunks, flags);
  return gnc::check_launch("mlp_resident_kernel");
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

int gnc_mlp::launch_resident(const gnc_mlp_desc_t& d, int T, bool narrow_out, hipStream_t stream, bool* launched) {
  *launched = false;
  static const bool disabled = getenv("GNC_MLP_NO_RESIDENT") != nullptr;  // A/B switch for benchmarking
  if (disabled) return GNC_OK;
  if (T > 2 || d.rows >= INT32_MAX) return GNC_OK;
  const int L = d.num_linear;
  const int od = d.out_dim[L - 1];
  if (L > 1 && d.activation != GNC_ACT_RELU) return GNC_OK;
  // shape of the step schedule: MATMUL segments first, then exactly 0 or 2 ADD segments
  int nmm = 0, nadd = 0;
  for (int s = 0; s < d.num_segments; ++s) {
    const gnc_mlp_segment_t& g = d.seg[s];
    if (g.width > KC || g.ld % 4 != 0 || !al16(g.ptr)) return GNC_OK;
    if (g.mode == GNC_SEG_ADD) {
      ++nadd;
    } else {
      if (nadd) return GNC_OK;  // a MATMUL segment after an ADD one
      ++nmm;
    }
  }
  if (nmm < 1 || nmm > 3 || (nadd != 0 && !(nadd == 2 && nmm == 1))) return GNC_OK;
  if (d.residual && (d.ld_residual % 4 != 0 || !al16(d.residual))) return GNC_OK;
  if (L > 1 && d.in_dim[1] > KC) return GNC_OK;

  const int total_chunks = nmm + (L - 1);
  const size_t floats = (size_t)total_chunks * T * 32 * LDSW + (size_t)(L + 2) * T * 32 + (size_t)RWAVES * RPW * LDSW;
  const size_t smem = floats * sizeof(float);
  if (smem > 160 * 1024) return GNC_OK;

  const gnc_mlp_segment_t& lm = d.seg[nmm - 1];
  const bool resreg = d.residual && d.residual == lm.ptr && !lm.index && lm.ld == d.ld_residual && lm.width == od;

#define GNC_RES(HT_, NMM_, NADD_)                                                             \
  do {                                                                                        \
    *launched = true;                                                                         \
    return resreg ? launch<HT_, HT_, NMM_, NADD_, true>(d, total_chunks, smem, stream)        \
                  : launch<HT_, HT_, NMM_, NADD_, false>(d, total_chunks, smem, stream);      \
  } while (0)
  if (narrow_out) {  // out width <= 32 (the decoder): one output tile
    if (resreg || nadd || nmm != 1) return GNC_OK;
    *launched = true;
    return T == 2 ? launch<2, 1, 1, 0, false>(d, total_chunks, smem, stream)
                  : launch<1, 1, 1, 0, false>(d, total_chunks, smem, stream);
  }
  if (T == 2) {
    if (nadd == 2) GNC_RES(2, 1, 2);
    if (nmm == 1) GNC_RES(2, 1, 0);
    if (nmm == 2) GNC_RES(2, 2, 0);
    GNC_RES(2, 3, 0);
  }
  if (nadd == 2) GNC_RES(1, 1, 2);
  if (nmm == 1) GNC_RES(1, 1, 0);
  if (nmm == 2) GNC_RES(1, 2, 0);
  GNC_RES(1, 3, 0);
#undef GNC_RES
}
