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
//                         registers, staged once and added to the pre-activation; their loads are
//                         issued from INSIDE the last MATMUL step's MFMA loop, one 4-row piece per
//                         k-group (a wave's LDS / VMEM / scalar instructions ride in the shadow of its
//                         own MFMAs; in front of the loop they would crawl whenever the SIMD mate is
//                         in an MFMA phase);
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
//   * every load is unconditional: row-ordered tables are read through per-tile buffer windows
//     (rows past the end read as 0, rows past the end are never stored - the hardware bounds check
//     does both), gather ids are clamped; there is no exec-masked VMEM in the loop.
// fp32 MFMA time and VALU time ADD UP on a gfx950 SIMD (tools/hw_probe.hip), so the loop keeps the
// vector unit out of address arithmetic: full tiles use one loop-invariant descriptor over the whole
// table with the piece offsets in the instruction's SGPR offset field, per-lane offsets are loop
// constants, gathered rows cost one v_add_u32 each.
// Optional epilogue (template flag AGG, gnc_mlp_desc_t.agg_out): per-destination sums of the output rows,
// bit-identical to K1 (see the comment at the kernel).
// Requirements checked by the launcher (anything else runs the streaming kernel): ReLU,
// all tables / weights / output 16-B aligned with leading dimensions % 4 == 0, every segment
// <= 64 columns, MATMUL segments listed before ADD segments, rows < 2^31, gathered tables with a stated
// size (table_rows) below 4 GiB.
#include <stdlib.h>

#include "mlp_device.h"

using namespace gnc_mlp;

// Phase probe (build with `make PROBE=1`): every wave sums the shader-clock cycles (s_memtime) it spends
// in each phase of its tile loop and leaves them, its total and the 100 MHz wall clock in gnc_phase_probe;
// tools/profile_mlp.py reads them back.  Off in the shipped library.
#ifdef GNC_PHASE_PROBE
__device__ unsigned long long gnc_phase_probe[4096 * 12];
extern "C" int gnc_phase_probe_read(void* dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(gnc_phase_probe), bytes);
}
#define PROBE_BEGIN() unsigned long long pr_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pr_tp, pr_tn; \
  const unsigned long long pr_c0 = __builtin_readcyclecounter(), pr_w0 = wall_clock64()
#define PROBE_TILE() pr_tp = __builtin_readcyclecounter()
#define PROBE(k) do { pr_tn = __builtin_readcyclecounter(); pr_acc[k] += pr_tn - pr_tp; pr_tp = pr_tn; } while (0)
#define PROBE_END() do { if ((threadIdx.x & 63) == 0) {                                                      \
    unsigned long long* o = gnc_phase_probe + (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 12;    \
    for (int k = 0; k < 8; ++k) o[k] = pr_acc[k];                                                            \
    o[8] = __builtin_readcyclecounter() - pr_c0; o[9] = wall_clock64() - pr_w0; } } while (0)
#else
#define PROBE_BEGIN() do {} while (0)
#define PROBE_TILE() do {} while (0)
#define PROBE(k) do {} while (0)
#define PROBE_END() do {} while (0)
#endif

namespace {

constexpr int RWAVES = 8;
constexpr int RNT = RWAVES * 64;

// HT/OT: accumulator tiles of the hidden / output width.  NMM: MATMUL steps per tile (1..3).
// NADD: additive segments merged into one step (0 or 2).  RESREG: the residual is the table of
// the LAST MATMUL step (no index): its rows are kept in registers instead of being re-read.
// AGG: fused aggregation epilogue (gnc_mlp_desc_t.agg_out): every wave walks a CONTIGUOUS range of tiles and
// carries the running sum of the destination in progress from tile to tile in one register (lane = feature),
// so a destination's rows are added in ascending order exactly as K1 does; only the first and the last
// destination of a wave's range (which may continue in a neighbour's range) are left to gnc_agg_fixup_f32.
// SAVE: training forward (gnc_mlp_desc_t.save_act): the hidden layers' post-activations are also written.  A template
// flag, not a run-time test: the mere presence of the branch cost the inference forward 1.3 % at c3 (same-box A/B).
// FULL: every width of the launch is exactly 64 (segments, hidden layers, output), three Linear layers, LayerNorm, vector
// output rows - the c3 edge processor.  The general instance keeps ~40 loop-invariant lane masks (feature / column < width)
// and the dimensions themselves alive across the tile loop; here they are compile-time constants.
// EF = 1: K6 as the prologue (gnc_mlp_desc_t.ef_pos): the rows of the only segment are computed from the positions of the
// edge's endpoints (two coalesced id loads one tile ahead, two 8-B gathers per row and tile) instead of being read.
// DUAL: TWO single-Linear projections of the same rows in one launch (the W-split's node-side products x Ws^T and x Wd^T,
// models/GNN.py:58-61): the rows are read and staged once, both weight chunks are resident, the second matrix / output travel in
// d.weight[1] / d.save_act[0] (a description built by gnc_mlp_dual_projection_f32, never by a caller).  No bias, no LayerNorm.
// EF = 2: the only segment is a contiguous [rows, 3] table (the reference's node features, models/GNN.py:305): one 12-B load
// per row and tile (lane & 31 = tile row) where it lies - no zero-padded [rows, 4] copy in front of the launch.
template <int HT, int OT, int NMM, int NADD, bool RESREG, bool AGG = false, bool SAVE = false, bool FULL = false, int EF = 0, bool DUAL = false>
__global__ __launch_bounds__(RNT) void mlp_resident_kernel(const gnc_mlp_desc_t d, const int num_wtiles,
                                                           const int total_chunks) {
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
  const int L = FULL ? 3 : d.num_linear;
  const int out_dim = FULL ? KC : d.out_dim[L - 1];
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
    if constexpr (DUAL) {
      const int ldw1 = ldw_of(d, 1);
      const bool w1v = (ldw1 % 4 == 0) && ((reinterpret_cast<uintptr_t>(d.weight[1]) & 15u) == 0);
      stage_weights<WT * 32, RNT>(wres + CH, d.weight[1], ldw1, d.out_dim[0], d.seg[0].wcol, d.seg[0].wcol + d.seg[0].width, 16,
                                  w1v && (d.seg[0].wcol % 4 == 0), tid);
    }
  }
  __syncthreads();

  SegView sv[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) sv[s] = {d.seg[s].ptr, d.seg[s].index, d.seg[s].ld, FULL ? KC : d.seg[s].width,
                                          (uint32_t)(d.seg[s].table_rows * d.seg[s].ld * 4)};  // launcher: < 4 GiB

  // ---- per-wave pipeline -----------------------------------------------------------------------
  const int total_waves = (int)gridDim.x * RWAVES;
  const int last_wt = num_wtiles - 1;

  // Gathered segments: table row of tile row (lane & 31), fetched one tile ahead and NOT touched until the
  // gather is issued (any use makes the compiler wait for it and, vmcnt being in order, for every load issued
  // before it); clamped so the id load is always legal.  Row-ordered segments need no ids at all.
  auto row_offset = [&](int wt, const SegView& s) -> uint32_t {
    if (s.index == nullptr) return 0;
    const int wtc = wt < last_wt ? wt : last_wt;
    int r = wtc * RPW + (lane & 31);
    r = r < rows ? r : rows - 1;
    return (uint32_t)s.index[r];
  };
  // the step's rows: lane (rs, c4) gets 16 B of row p*4+rs for p = 0..7; no predication anywhere.
  //   row-ordered table: one window per tile, piece offsets in the SGPR offset field, ONE loop-constant
  //                      per-lane byte offset; the last tile moves the window per piece so that rows past
  //                      the end read as 0 (load_tile_rows);
  //   gathered table:    the row's offset arrives by ds_bpermute, one v_lshl_add_u64 per access.
  auto load_rows = [&](f32x4 (&pre)[NP], const SegView& s, int wt, uint32_t off) {
    const int col = c4 * 4 < s.ld ? c4 * 4 : 0;
    if (s.index == nullptr) {
      load_tile_rows(pre, s.ptr, s.ld, (int64_t)wt * RPW, rows, (uint32_t)(rs * s.ld + col) * 4u);
    } else {
      const __amdgpu_buffer_rsrc_t w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(s.ptr), 0, (int)s.bytes, 0x00020000);
      const uint32_t row_bytes = off * (uint32_t)(s.ld * 4);  // byte offset of this lane's tile row, once per tile
#pragma unroll
      for (int p = 0; p < NP; ++p) pre[p] = window_load(w, (uint32_t)__shfl((int)row_bytes, p * 4 + rs, 64) + (uint32_t)(col * 4));
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

  // tile schedule: strided over the grid, or (AGG) one contiguous range [wt, wt_end) per wave
  const int gwave = (int)blockIdx.x * RWAVES + wave;
  const int stride = AGG ? 1 : total_waves;
  int wt = gwave, wt_end = num_wtiles;
  if constexpr (AGG) {
    const int q = num_wtiles / total_waves, rem = num_wtiles - q * total_waves;
    wt = gwave * q + (gwave < rem ? gwave : rem);
    wt_end = wt + q + (gwave < rem ? 1 : 0);
  }
  f32x4 cur[NP];                  // rows of the step about to be staged
  f32x4 addA[NADD ? NP : 1], addB[NADD ? NP : 1];
  // row offsets of gathered segments: off0 = segment 0 of the NEXT tile, offs[s>=1] = segment s of the CURRENT tile
  uint32_t offs[NS];
  // EF: endpoint ids of the tile after next (lane & 31 = tile row), endpoint positions of the next tile
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  int ef_s = 0, ef_d = 0;
  f32x2 ef_ps = {0.f, 0.f}, ef_pd = {0.f, 0.f};
  const __amdgpu_buffer_rsrc_t ef_win =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(EF == 1 ? d.ef_pos : nullptr), 0, EF == 1 ? (int)(uint32_t)(d.ef_nodes * 8) : 0, 0x00020000);
  auto ef_ids = [&](int t) {
    const int tc = t < last_wt ? t : last_wt;
    int r = tc * RPW + (lane & 31);
    r = r < rows ? r : rows - 1;
    ef_s = d.ef_src[r];
    ef_d = d.ef_dst[r];
  };
  auto ef_fetch = [&]() {  // launcher: ef_nodes * 8 < 4 GiB; an id outside the table reads as (0, 0)
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    ef_ps = __builtin_bit_cast(f32x2, (u32x2)__builtin_amdgcn_raw_buffer_load_b64(ef_win, (uint32_t)ef_s * 8u, 0, 0));
    ef_pd = __builtin_bit_cast(f32x2, (u32x2)__builtin_amdgcn_raw_buffer_load_b64(ef_win, (uint32_t)ef_d * 8u, 0, 0));
  };
  auto ef_stage = [&]() {  // [dx, dy, |dx| + |dy|, 0 | 0 0 0 0]: the 8 columns the first Linear's one k-group reads
    compiler_lds_barrier();
    const float dx = ef_pd.x - ef_ps.x, dy = ef_pd.y - ef_ps.y;
    f32x4 v = {dx, dy, __builtin_fabsf(dx) + __builtin_fabsf(dy), 0.f};
    if (h) v = f32x4{0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(abuf + i * LDSW + 4 * h) = v;
    compiler_lds_barrier();
  };
  // EF = 2: the next tile's [*, 3] rows
  typedef float f32x3 __attribute__((ext_vector_type(3)));
  f32x3 n3_row = {0.f, 0.f, 0.f};
  const __amdgpu_buffer_rsrc_t n3_win = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(EF == 2 ? d.seg[0].ptr : nullptr), 0, EF == 2 ? (int)(uint32_t)(d.rows * 12) : 0, 0x00020000);
  auto n3_fetch = [&](int t) {  // launcher: rows * 12 < 4 GiB; rows past the end read as zeros
    typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
    const int tc = t < last_wt ? t : last_wt;
    n3_row = __builtin_bit_cast(f32x3, (u32x3)__builtin_amdgcn_raw_buffer_load_b96(n3_win, (uint32_t)(tc * RPW + (lane & 31)) * 12u, 0, 0));
  };
  auto n3_stage = [&]() {
    compiler_lds_barrier();
    f32x4 v = {n3_row.x, n3_row.y, n3_row.z, 0.f};
    if (h) v = f32x4{0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(abuf + i * LDSW + 4 * h) = v;
    compiler_lds_barrier();
  };
  uint32_t off0 = 0;
  if constexpr (EF == 1) {
    ef_ids(wt);
    ef_fetch();
    ef_ids(wt + stride);
  } else if constexpr (EF == 2) {
    n3_fetch(wt);
  } else {
    load_rows(cur, sv[0], wt, row_offset(wt, sv[0]));
  }
#pragma unroll
  for (int s = 1; s < NS; ++s) offs[s] = row_offset(wt, sv[s]);
  if constexpr (EF == 0) off0 = row_offset(wt + stride, sv[0]);

  const int col_out = c4 * 4;
  const bool vec_out = FULL || ((out_dim % 4 == 0) && (d.ld_out % 4 == 0) && ((reinterpret_cast<uintptr_t>(d.out) & 15u) == 0));
  const uint32_t out_lane_off = (uint32_t)(rs * d.ld_out + col_out) * 4u;
  const uint32_t res_lane_off = (uint32_t)(rs * d.ld_residual + (col_out < d.ld_residual ? col_out : 0)) * 4u;

  // fused aggregation state: destination ids of the tile's rows (lane & 31), running sum (lane = feature)
  auto agg_ids = [&](int t) -> int {
    int r = t * RPW + (lane & 31);
    r = r < rows ? r : rows - 1;
    return d.agg_index[r];
  };
  int aid = 0, aid_next = 0;
  float agg_acc = 0.f;
  int agg_cur = -1, agg_first_dst = -1;  // wave-uniform
  bool agg_first = true;
  if constexpr (AGG) aid = agg_ids(wt < last_wt ? wt : last_wt);

  PROBE_BEGIN();
  while (wt < wt_end) {
    PROBE_TILE();
    const int row0 = wt * RPW;
    const int nwt = wt + stride;
    if constexpr (AGG) aid_next = agg_ids(nwt < last_wt ? nwt : last_wt);

    // ------------------------------------------------------------------ first Linear
    f32x16 hid[HT];
    f32x16 res[RESREG ? HT : 1];  // residual rows in accumulator layout (read back from the staged tile)
    init_bias<HT>(hid, pbuf, h);
#pragma unroll
    for (int s = 0; s < NMM; ++s) {
      bool interleaved = false;
      if constexpr (EF == 1) ef_stage();
      else if constexpr (EF == 2) n3_stage();
      else stage(cur, sv[s].width);
      PROBE(0);  // wait for the step's rows + staging
      // request what comes next before the MFMAs of this step start
      if (s + 1 < NMM) {
        load_rows(cur, sv[s + 1 < NMM ? s + 1 : 0], wt, offs[s + 1 < NMM ? s + 1 : 0]);
        offs[s + 1 < NMM ? s + 1 : 0] = row_offset(nwt, sv[s + 1 < NMM ? s + 1 : 0]);
      } else {
        if constexpr (NADD > 0) {
          if (sv[s].width == KC && sv[NMM].index && sv[NMM + 1].index) {
            // Gathered ADD rows requested from INSIDE the first Linear's MFMA loop: LDS / VMEM / scalar
            // instructions of a wave ride in the shadow of its own MFMAs (tools/hw_probe.hip: up to ~4 per MFMA for
            // free), whereas issued in front of the loop they crawl at ~1 per 37 cycles whenever the SIMD mate
            // is streaming MFMAs.  One 4-row piece of each table per k-group; the row offsets travel one group
            // ahead by ds_bpermute.
            const SegView &sa = sv[NMM], &sb = sv[NMM + 1];
            const __amdgpu_buffer_rsrc_t wa = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sa.ptr), 0, (int)sa.bytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t wb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sb.ptr), 0, (int)sb.bytes, 0x00020000);
            const uint32_t cola = (uint32_t)(c4 * 4 < sa.ld ? c4 * 16 : 0), colb = (uint32_t)(c4 * 4 < sb.ld ? c4 * 16 : 0);
            const int rba = (int)(offs[NMM] * (uint32_t)(sa.ld * 4)), rbb = (int)(offs[NMM + 1] * (uint32_t)(sb.ld * 4));
            int na = __shfl(rba, rs, 64), nb = __shfl(rbb, rs, 64);
            const float* wch = wres + s * CH;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
              const f32x4 b = *reinterpret_cast<const f32x4*>(abuf + i * LDSW + 8 * g + 4 * h);
#pragma unroll
              for (int t = 0; t < HT; ++t) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(wch + (32 * t + i) * LDSW + 8 * g + 4 * h);
                hid[t] = mfma(a.x, b.x, hid[t]);
                hid[t] = mfma(a.y, b.y, hid[t]);
                hid[t] = mfma(a.z, b.z, hid[t]);
                hid[t] = mfma(a.w, b.w, hid[t]);
              }
              const uint32_t oa = (uint32_t)na + cola, ob = (uint32_t)nb + colb;
              if (g + 1 < 8) {
                na = __shfl(rba, (g + 1) * 4 + rs, 64);
                nb = __shfl(rbb, (g + 1) * 4 + rs, 64);
              }
              addA[g] = window_load(wa, oa);
              addB[g] = window_load(wb, ob);
            }
            offs[NMM] = row_offset(nwt, sv[NMM]);
            offs[NMM + 1] = row_offset(nwt, sv[NMM + 1]);
            interleaved = true;
          } else {
            load_rows(addA, sv[NMM], wt, offs[NMM]);
            load_rows(addB, sv[NMM + 1], wt, offs[NMM + 1]);
            offs[NMM] = row_offset(nwt, sv[NMM]);
            offs[NMM + 1] = row_offset(nwt, sv[NMM + 1]);
          }
        } else if constexpr (EF == 1) {
          ef_fetch();  // the next tile's endpoints (ids fetched one tile ago), then the ids of the tile after it
          ef_ids(nwt + stride);
        } else if constexpr (EF == 2) {
          n3_fetch(nwt);
        } else {
          load_rows(cur, sv[0], nwt, off0);
          off0 = row_offset(nwt + stride, sv[0]);
        }
      }
      PROBE(7);  // issue of the next rows' loads
      if (!interleaved) mma_chunk_from_lds<HT>(hid, abuf, wres + s * CH, (sv[s].width + 7) >> 3, i, h);
      if constexpr (RESREG) if (s == NMM - 1) {
#pragma unroll
        for (int t = 0; t < HT; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(abuf + i * LDSW + 32 * t + 8 * q + 4 * h);
            res[t][4 * q + 0] = v.x; res[t][4 * q + 1] = v.y; res[t][4 * q + 2] = v.z; res[t][4 * q + 3] = v.w;
          }
      }
      PROBE(1);  // load issue + first-Linear MFMAs
    }
    if constexpr (NADD > 0) {
      f32x4 sum[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) sum[p] = addA[p] + addB[p];
      stage(sum, sv[NMM].width);
      load_rows(cur, sv[0], nwt, off0);  // next tile's first step: the rest of this tile to land
      off0 = row_offset(nwt + stride, sv[0]);
      add_tile_from_lds<HT>(hid, abuf, i, h);
      PROBE(2);  // wait for the gathered rows + ADD step
    }

    f32x16 hidb[DUAL ? HT : 1];
    if constexpr (DUAL) {  // the second projection of the staged rows (same tile, the other resident weight chunk)
      init_bias<HT>(hidb, pbuf, h);  // (no bias: zeros)
      mma_chunk_from_lds<HT>(hidb, abuf, wres + CH, (sv[0].width + 7) >> 3, i, h);
    }
    if (L == 1) {  // plain projection
      if (d.ln_gamma) layer_norm_tiles<HT>(hid, pbuf + L * PSTRIDE, pbuf + (L + 1) * PSTRIDE, out_dim, d.ln_eps, h);
      if constexpr (RESREG) {
#pragma unroll
        for (int t = 0; t < HT; ++t) hid[t] += res[t];
      }
      compiler_lds_barrier();
      tiles_to_lds<HT>(hid, abuf, i, h);
    } else {
      relu_tiles<HT>(hid);
      // training forward (gnc_mlp_desc_t.save_act): the post-activations leave as whole rows through the wave's LDS tile,
      // which is idle between the first Linear and the epilogue (16-B pieces straight from the accumulator layout - 32 B
      // per row and instruction - measured +0.55 ms per c3 edge launch, four times the requests at the L2)
      auto save_rows = [&](const f32x16 (&acc)[HT], int l) {
        compiler_lds_barrier();
        tiles_to_lds<HT>(acc, abuf, i, h);
        compiler_lds_barrier();
        const int fl = fresh_lane();
        store_staged_rows<NP, true>(abuf, d.save_act[l], d.out_dim[l], d.out_dim[l], row0, rows, fl & 15, fl >> 4);
        compiler_lds_barrier();
      };
      if constexpr (SAVE) save_rows(hid, 0);
      // ---------------------------------------------------------------- hidden layers 1 .. L-2
      for (int l = 1; l < L - 1; ++l) {
        f32x16 nxt[HT];
        init_bias<HT>(nxt, pbuf + l * PSTRIDE, h);
        mma_chunk_from_regs<HT, HT>(nxt, hid, wres + (NMM + l - 1) * CH, 0, FULL ? KC : d.in_dim[l], i, h);
        relu_tiles<HT>(nxt);
        if constexpr (SAVE) save_rows(nxt, l);
#pragma unroll
        for (int t = 0; t < HT; ++t) hid[t] = nxt[t];
      }
      PROBE(3);  // hidden Linears
      // ---------------------------------------------------------------- last Linear, LayerNorm
      f32x16 o[OT];
      init_bias<OT>(o, pbuf + (L - 1) * PSTRIDE, h);
      mma_chunk_from_regs<HT, OT>(o, hid, wres + (NMM + L - 2) * CH, 0, FULL ? KC : d.in_dim[L - 1], i, h);
      PROBE(4);  // last Linear
      if (FULL || d.ln_gamma) layer_norm_tiles<OT>(o, pbuf + L * PSTRIDE, pbuf + (L + 1) * PSTRIDE, out_dim, d.ln_eps, h);
      if constexpr (RESREG) {
#pragma unroll
        for (int t = 0; t < OT && t < HT; ++t) o[t] += res[t < HT ? t : 0];
      }
      compiler_lds_barrier();
      tiles_to_lds<OT>(o, abuf, i, h);
      PROBE(5);  // LayerNorm + residual + transpose out
    }
    compiler_lds_barrier();
    if constexpr (AGG) {
      // The tile's final rows (residual included) sit in the wave's LDS tile.  bnd bit r = "row r starts a new
      // destination"; everything that steers the walk is wave-uniform (SALU), the vector unit only adds.
      const int valid = rows - row0 < RPW ? rows - row0 : RPW;
      int prv = __shfl_up(aid, 1, 64);
      prv = lane == 0 ? agg_cur : prv;
      const unsigned long long vmask = valid >= 32 ? 0xffffffffull : ((1ull << valid) - 1ull);
      const unsigned long long bnd = __ballot(aid != prv) & vmask;
      // all 32 row values first (one batch of LDS reads): the walk below has a scalar branch per row, and a
      // read inside it would cost a full LDS round trip every time
      float rowv[RPW];
#pragma unroll
      for (int r = 0; r < RPW; ++r) rowv[r] = abuf[r * LDSW + lane];
      auto agg_row = [&](int r) {  // r and everything tested here are wave-uniform
        if ((bnd >> r) & 1ull) {
          if (agg_cur >= 0) {
            if (agg_first) {  // the range's first destination may have begun in the previous range: fix-up's
              agg_first = false;
              agg_first_dst = agg_cur;
            } else if (lane < out_dim) {
              float* dstp = d.agg_out + (int64_t)agg_cur * d.ld_agg;
              asm volatile("s_nop 4\n\tglobal_store_dword %0, %1, %2\n\ts_nop 0" ::"v"(lane * 4), "v"(agg_acc), "s"(dstp) : "memory");
            }
          }
          agg_cur = __builtin_amdgcn_readlane(aid, r);
          agg_acc = 0.f;
        }
        agg_acc += rowv[r];
      };
#pragma unroll
      for (int r4 = 0; r4 < RPW; r4 += 4) {
        if (r4 + 4 <= valid && ((bnd >> r4) & 0xfull) == 0) {  // four rows of the destination in progress: no tests
          agg_acc += rowv[r4 + 0];
          agg_acc += rowv[r4 + 1];
          agg_acc += rowv[r4 + 2];
          agg_acc += rowv[r4 + 3];
        } else {
#pragma unroll
          for (int r = r4; r < r4 + 4; ++r)
            if (r < valid) agg_row(r);
        }
      }
      aid = aid_next;
    }
    // The compiler's counted vmcnt waits do not know about the asm stores below; make it collect the
    // next tile's prefetched rows (issued >= one Linear ago) BEFORE the stores join the queue, so that
    // no later wait for them has to sit out the stores as well.
    if constexpr (EF == 1) {
      asm volatile("" ::"v"(ef_ps), "v"(ef_pd), "v"(ef_s), "v"(ef_d));
    } else if constexpr (EF == 2) {
      asm volatile("" ::"v"(n3_row.x), "v"(n3_row.y), "v"(n3_row.z));
    } else {
#pragma unroll
      for (int p = 0; p < NP; ++p) asm volatile("" ::"v"(cur[p]));
    }
    // ... and the gather ids fetched for the next tile: their first use comes after the stores, where a wait for
    // them (vmcnt is in order) would sit out the stores' HBM round trip
#pragma unroll
    for (int s = 1; s < NS; ++s) asm volatile("" ::"v"(offs[s]));
    asm volatile("" ::"v"(off0));
    if constexpr (AGG) asm volatile("" ::"v"(aid_next));

    // ------------------------------------------------------------------ epilogue: whole rows out
    {
      f32x4 outv[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) outv[p] = *reinterpret_cast<const f32x4*>(abuf + (p * 4 + rs) * LDSW + col_out);
      if (!RESREG && d.residual) {  // rows this tile has just read: L2 hits; rows past the end read as 0
        f32x4 rv[NP];
        load_tile_rows(rv, d.residual, d.ld_residual, row0, rows, res_lane_off);
#pragma unroll
        for (int p = 0; p < NP; ++p) outv[p] += rv[p];
      }
      if (vec_out) {  // rows past the end are dropped by the window's bounds check
        if (col_out < out_dim) {
          if (row0 + RPW <= rows) {
            uint32_t r0b;
            const __amdgpu_buffer_rsrc_t ow = full_tile_window(d.out, row0, rows, d.ld_out, &r0b);
#pragma unroll
            for (int p = 0; p < NP; ++p) hidden_window_store_s<true>(outv[p], out_lane_off + r0b, ow, (uint32_t)(p * 16 * d.ld_out));
          } else {
#pragma unroll
            for (int p = 0; p < NP; ++p) hidden_window_store(outv[p], out_lane_off, row_window(d.out, row0 + 4 * p, rows, d.ld_out));
          }
        }
      } else {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const int r = row0 + p * 4 + rs;
          store_row_piece(d.out + (int64_t)(r < rows ? r : rows - 1) * d.ld_out, col_out, outv[p], r < rows && col_out < out_dim,
                          out_dim, false);
        }
      }
    }
    if constexpr (DUAL) {  // the second output: through the wave's LDS tile like the first (launcher: whole 16-B pieces)
      compiler_lds_barrier();
      tiles_to_lds<HT>(hidb, abuf, i, h);
      compiler_lds_barrier();
      f32x4 outb[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) outb[p] = *reinterpret_cast<const f32x4*>(abuf + (p * 4 + rs) * LDSW + col_out);
      if (col_out < out_dim) {
        if (row0 + RPW <= rows) {
          uint32_t r0b;
          const __amdgpu_buffer_rsrc_t ow = full_tile_window(d.save_act[0], row0, rows, d.ld_out, &r0b);
#pragma unroll
          for (int p = 0; p < NP; ++p) hidden_window_store_s<true>(outb[p], out_lane_off + r0b, ow, (uint32_t)(p * 16 * d.ld_out));
        } else {
#pragma unroll
          for (int p = 0; p < NP; ++p) hidden_window_store(outb[p], out_lane_off, row_window(d.save_act[0], row0 + 4 * p, rows, d.ld_out));
        }
      }
    }
    compiler_lds_barrier();
    PROBE(6);  // epilogue: rows out
    wt = nwt;
  }
  if constexpr (AGG) {  // the range's first and last destination (possibly the same, possibly none) go to the fix-up
    if (lane == 0) {
      d.agg_fix[2 * gwave] = agg_first ? agg_cur : agg_first_dst;
      d.agg_fix[2 * gwave + 1] = agg_cur;
    }
  }
  PROBE_END();
}

template <int HT, int OT, int NMM, int NADD, bool RESREG, bool AGG = false, bool SAVE = false, bool FULL = false, int EF = 0, bool DUAL = false>
int launch(const gnc_mlp_desc_t& d, int total_chunks, size_t smem, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    int rc = gnc::check_hip(
        hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_resident_kernel<HT, OT, NMM, NADD, RESREG, AGG, SAVE, FULL, EF, DUAL>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
        "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    if (rc) return rc;
    attr_set = true;
  }
  const int64_t num_wtiles = gnc::ceil_div(d.rows, RPW);
  int64_t grid = gnc::ceil_div(num_wtiles, RWAVES);
  if (grid > gnc::num_cu()) grid = gnc::num_cu();  // one persistent workgroup per CU
  if constexpr (AGG) grid = gnc::num_cu();        // agg_fix has two entries for every wave of the full grid
  mlp_resident_kernel<HT, OT, NMM, NADD, RESREG, AGG, SAVE, FULL, EF, DUAL>
      <<<dim3((unsigned)grid), dim3(RNT), smem, stream>>>(d, (int)num_wtiles, total_chunks);
  return gnc::check_launch("mlp_resident_kernel");
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

extern "C" int gnc_mlp_agg_fix_len(void) { return 2 * gnc::num_cu() * RWAVES; }

// Two projections of the same rows (x wa^T -> out_a, x wb^T -> out_b) for a large batch at widths <= 64: see DUAL above.
int gnc_mlp::launch_resident_dual(const float* x, int64_t ld_x, int64_t rows, const float* wa, int64_t ld_wa, const float* wb,
                                  int64_t ld_wb, int in_dim, int out_dim, float* out_a, float* out_b, int64_t ld_out,
                                  hipStream_t stream, bool* launched) {
  *launched = false;
  static const bool disabled = getenv("GNC_MLP_NO_RESIDENT") != nullptr;
  if (disabled || rows < 1 || rows >= INT32_MAX || in_dim < 1 || in_dim > KC || out_dim <= 32 || out_dim > KC || out_dim % 4 != 0 ||
      ld_x % 4 != 0 || ld_x < in_dim || ld_out % 4 != 0 || ld_out < out_dim || ld_wa < in_dim || ld_wb < in_dim || ld_wa > INT32_MAX ||
      ld_wb > INT32_MAX || ld_x > INT32_MAX || ld_out > INT32_MAX || !al16(x) || !al16(out_a) || !al16(out_b))
    return GNC_OK;
  gnc_mlp_desc_t d = {};
  d.num_segments = 1;
  d.num_linear = 1;
  d.activation = GNC_ACT_RELU;
  d.seg[0].ptr = x;
  d.seg[0].width = in_dim;
  d.seg[0].ld = (int32_t)ld_x;
  d.seg[0].mode = GNC_SEG_MATMUL;
  d.weight[0] = wa;
  d.ld_weight[0] = (int32_t)ld_wa;
  d.weight[1] = wb;  // DUAL: the second matrix and output ride in fields a single-Linear description leaves unused
  d.ld_weight[1] = (int32_t)ld_wb;
  d.in_dim[0] = in_dim;
  d.out_dim[0] = out_dim;
  d.in_dim[1] = in_dim;
  d.out = out_a;
  d.save_act[0] = out_b;
  d.ld_out = (int32_t)ld_out;
  d.rows = rows;
  const int total_chunks = 2;
  const size_t floats = (size_t)total_chunks * 2 * 32 * LDSW + (size_t)(1 + 2) * 2 * 32 + (size_t)RWAVES * RPW * LDSW;
  *launched = true;
  return launch<2, 2, 1, 0, false, false, false, false, 0, true>(d, total_chunks, floats * sizeof(float), stream);
}

int gnc_mlp::launch_resident(const gnc_mlp_desc_t& d, int T, bool narrow_out, hipStream_t stream, bool* launched,
                             bool probe_only) {
  *launched = false;
  static const bool disabled = getenv("GNC_MLP_NO_RESIDENT") != nullptr;  // A/B switch for benchmarking
  if (disabled) return GNC_OK;
  if (T > 2 || d.rows >= INT32_MAX) return GNC_OK;
  const int L = d.num_linear;
  const int od = d.out_dim[L - 1];
  if (L > 1 && d.activation != GNC_ACT_RELU) return GNC_OK;
  // shape of the step schedule: MATMUL segments first, then exactly 0 or 2 ADD segments
  int nmm = 0, nadd = 0;
  bool n3 = false;
  for (int s = 0; s < d.num_segments; ++s) {
    const gnc_mlp_segment_t& g = d.seg[s];
    if (d.ef_pos && s == 0) continue;  // computed rows: nothing is read through this segment's table
    // a contiguous [rows, 3] table as the only input (the node features, models/GNN.py:305): read where it lies (EF = 2)
    if (s == 0 && d.num_segments == 1 && !g.index && g.mode == GNC_SEG_MATMUL && g.width == 3 && g.ld == 3 &&
        (reinterpret_cast<uintptr_t>(g.ptr) & 3u) == 0 && d.rows * 12 <= 0xffffffffll && !d.residual && !d.save_act[0] && !d.agg_out &&
        !narrow_out && L >= 2 && T == 2) {
      n3 = true;
      ++nmm;
      continue;
    }
    if (g.width > KC || g.ld % 4 != 0 || !al16(g.ptr)) return GNC_OK;
    // gathers go through a buffer window over the whole table: its size must be stated and below 4 GiB
    if (g.index && (g.table_rows <= 0 || g.table_rows * (int64_t)g.ld * 4 > 0xffffffffll)) return GNC_OK;
    if (g.mode == GNC_SEG_ADD) {
      ++nadd;
    } else {
      if (nadd) return GNC_OK;  // a MATMUL segment after an ADD one
      ++nmm;
    }
  }
  if (d.ef_pos) {  // K6 prologue: the edge encoder's shape only (one computed segment, no residual, inference)
    ++nmm;
    const gnc_mlp_segment_t& g = d.seg[0];
    if (d.num_segments != 1 || g.mode != GNC_SEG_MATMUL || g.index || d.ef_space_dim != 2 || g.width != 3 || !d.ef_src || !d.ef_dst ||
        d.ef_nodes <= 0 || d.ef_nodes * 8 > 0xffffffffll || d.residual || d.save_act[0] || d.agg_out || narrow_out || L < 2 || T != 2)
      return GNC_OK;
  }
  if (nmm < 1 || nmm > 3 || (nadd != 0 && !(nadd == 2 && nmm == 1))) return GNC_OK;
  if (d.residual && (d.ld_residual % 4 != 0 || !al16(d.residual))) return GNC_OK;
  if (L > 1 && d.in_dim[1] > KC) return GNC_OK;
  if (d.save_act[0]) {  // saved post-activations: whole rows of 16-B pieces
    if (L < 2) return GNC_OK;
    for (int l = 0; l < L - 1; ++l)
      if (!d.save_act[l] || d.out_dim[l] % 4 != 0 || !al16(d.save_act[l])) return GNC_OK;
  }

  const int total_chunks = nmm + (L - 1);
  const size_t floats = (size_t)total_chunks * T * 32 * LDSW + (size_t)(L + 2) * T * 32 + (size_t)RWAVES * RPW * LDSW;
  const size_t smem = floats * sizeof(float);
  if (smem > 160 * 1024) return GNC_OK;

  const gnc_mlp_segment_t& lm = d.seg[nmm - 1];
  const bool resreg = d.residual && d.residual == lm.ptr && !lm.index && lm.ld == d.ld_residual && lm.width == od;

  if (narrow_out && (resreg || nadd || nmm != 1)) return GNC_OK;  // out width <= 32 (the decoder): plain shapes only
  // every width exactly 64, three Linear layers, LayerNorm, whole 16-B output rows: the FULL instances (c3's edge processor)
  static const bool no_full = getenv("GNC_MLP_NO_FULL64") != nullptr;  // A/B switch
  bool full64 = !no_full && L == 3 && od == KC && d.ln_gamma && d.ld_out % 4 == 0 && al16(d.out);
  for (int s = 0; s < d.num_segments; ++s) full64 = full64 && d.seg[s].width == KC;
  for (int l = 0; l < L; ++l) full64 = full64 && d.out_dim[l] == KC && (l == 0 || d.in_dim[l] == KC);
  if (d.ef_pos) {
    *launched = true;
    if (probe_only) return GNC_OK;
    return launch<2, 2, 1, 0, false, false, false, false, 1>(d, total_chunks, smem, stream);
  }
  if (n3) {
    *launched = true;
    if (probe_only) return GNC_OK;
    return launch<2, 2, 1, 0, false, false, false, false, 2>(d, total_chunks, smem, stream);
  }
  if (d.save_act[0]) {
    // training forward: SAVE instances exist for the shapes whose K8 kernel reads the saved tensors (the fused data +
    // weight-gradient kernel: widths 33..64, ONE row-ordered MATMUL segment, 0 or 2 gathered ADD segments)
    if (T != 2 || narrow_out || nmm > 2 || L < 2) return GNC_OK;
    if (nmm == 2 && (nadd != 0 || d.agg_out)) return GNC_OK;  // two MATMUL segments: the node processors (split K8 path)
    if (d.agg_out && !(nadd == 2 && resreg && d.agg_index && d.agg_fix && d.ld_agg >= od)) return GNC_OK;
    if (nadd == 2 && !resreg) return GNC_OK;
    *launched = true;
    if (probe_only) return GNC_OK;
    if (nmm == 2)
      return resreg ? launch<2, 2, 2, 0, true, false, true>(d, total_chunks, smem, stream)
                    : launch<2, 2, 2, 0, false, false, true>(d, total_chunks, smem, stream);
    if (d.agg_out) return full64 ? launch<2, 2, 1, 2, true, true, true, true>(d, total_chunks, smem, stream)
                                 : launch<2, 2, 1, 2, true, true, true>(d, total_chunks, smem, stream);
    if (nadd == 2) return launch<2, 2, 1, 2, true, false, true>(d, total_chunks, smem, stream);
    return resreg ? launch<2, 2, 1, 0, true, false, true>(d, total_chunks, smem, stream)
                  : launch<2, 2, 1, 0, false, false, true>(d, total_chunks, smem, stream);
  }
  if (d.agg_out) {  // fused aggregation epilogue: the W-split edge processor shape only
    if (!(nadd == 2 && nmm == 1 && resreg && !narrow_out && d.agg_index && d.agg_fix && d.ld_agg >= od)) return GNC_OK;
    *launched = true;
    if (probe_only) return GNC_OK;
    if (T == 2 && full64) return launch<2, 2, 1, 2, true, true, false, true>(d, total_chunks, smem, stream);
    return T == 2 ? launch<2, 2, 1, 2, true, true>(d, total_chunks, smem, stream)
                  : launch<1, 1, 1, 2, true, true>(d, total_chunks, smem, stream);
  }
  if (probe_only) {
    *launched = true;
    return GNC_OK;
  }
#define GNC_RES(HT_, NMM_, NADD_)                                                             \
  do {                                                                                        \
    *launched = true;                                                                         \
    return resreg ? launch<HT_, HT_, NMM_, NADD_, true>(d, total_chunks, smem, stream)        \
                  : launch<HT_, HT_, NMM_, NADD_, false>(d, total_chunks, smem, stream);      \
  } while (0)
  if (narrow_out) {  // one output tile
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
