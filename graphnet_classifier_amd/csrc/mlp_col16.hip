// K4, small-batch variant ("column split") for hidden widths <= 128: the reference's own regime, one ~1000-node graph per call
// (main.py:60, utils/train_model.py:35-45, utils/inference.py:59).
//
// The throughput kernels give a wave a whole row tile: at 128 features that is 3 x 8 x 32 = 768 DEPENDENT 16x16x4 MFMAs per
// 16 rows (10 us) behind a weight stream that is staged through LDS chunk by chunk (two barriers each) - 17..40 us per launch
// however few rows there are, and a forward is 15 such launches in a row.  Here a workgroup owns ONE 16-row tile and its
// waves split the OUTPUT FEATURES of every Linear: wave w computes features [16w, 16w + 16) with 32 MFMAs per layer,
//   * its 16 weight rows come straight from L2 into registers (W[16w + i][16cb + 4g .. +3] is exactly the A operand of
//     k block cb).  All weight chunks of the tile, the biases and the LayerNorm parameters are requested at the top of the
//     tile, in the order they are needed, and consumed where they are needed - no LDS staging of weights;
//   * the layer's input rows are shared through LDS ([16][K + 4] floats: the B operand of k block cb is one ds_read_b128),
//     hidden layers ping-pong between two such tiles: ONE barrier per layer;
//   * the accumulator layout of the 16-row kernels (register r on lane (j, g) = feature 16w + 4g + r of row j) makes every
//     row-ordered operand a plain 16-B access per lane: W-split addends, residual, saved activations and the output rows
//     need no transposes;
//   * LayerNorm sums cross the waves through 2 x NT x 16 floats of LDS (two-pass, two barriers).
// A launch of this kernel is a few microseconds long, so its INSTRUCTION COUNT in front of the first MFMA is what it costs:
// the launcher digests the generic description into a ColPlan (per weight chunk: pointer, stride, window; per staging pass:
// the one segment that covers it; ...) and the kernel never looks at gnc_mlp_desc_t.
// No aggregation epilogue here: at this size the fix-up and zero-fill launches the epilogue needs cost more than K1 itself
// (scatter_sum_csr_small: 4.8 us), so gnc_mlp_agg_supported() says no for these launches and the caller runs K1 on the rows.
// Rounding differs from the streaming kernels (two interleaved accumulators per output tile): callers compare with
// tolerances, not bit for bit, across kernel families.
#include <stdlib.h>

#include <type_traits>

#include "mlp_device16.h"

using namespace gnc_mlp;

namespace {

constexpr int CK = 128;        // k columns of one register-resident weight chunk
constexpr int NKB = CK / 16;   // its 16-wide k blocks
constexpr int NQ = 4;          // weight chunks of one tile: ceil(in_dim[0] / CK) + num_linear - 1 <= NQ
constexpr int NSET = 3;        // register sets: chunk q lives in set q % NSET (chunk 3 is requested when chunk 0 has been consumed)
constexpr int MAXPASS = 8;     // staging passes of the narrowest workgroup (4 waves: 64 columns per pass, NQ * CK columns)

struct ColChunk {   // one weight chunk: 16 rows per wave x CK columns
  const float* w;   // weight[l] + first column of the chunk
  int ldw;
  int nrec;         // bytes from w to the end of the matrix (0: nothing to read): rows / columns past it read as 0
  int kvalid;       // columns of the chunk inside the matrix (< CK: the others are cleared at use)
};
struct ColPass {    // one staging pass: the segment that covers columns [pass * PW, pass * PW + width) of the virtual concat
  const float* ptr;  // table + the pass's first column of the segment
  const int32_t* index;
  uint32_t table_rows;
  int ld;
  int width;        // 0: dead pass
};
struct ColAdd {     // W-split addend: [*, H] rows added to the first Linear's pre-activation
  const float* ptr;
  const int32_t* index;
  uint32_t table_rows;
  int ld;
};
struct ColSeg {     // element-by-element staging (tables whose rows are not 16-B pieces)
  const float* ptr;
  const int32_t* index;
  uint32_t table_rows;
  int width, ld, wcol;
};
struct ColPlan {
  int rows, num_tiles, L, H, od, ldx;
  int n0, nq;        // chunks of the first Linear / of the tile
  int k0pad;         // in_dim[0] rounded up to 16
  int k_in;          // in_dim[0]
  int narrow_w0;     // weight[0] is read by dwords (in_dim[0] <= 16, rows not made of 16-B pieces)
  int nrec_narrow;   // ... through a window of this many bytes (ch[0].nrec is 0 then: the 16-B loads of set 0 read nothing)
  int vec_stage;     // every MATMUL segment is made of 16-B pieces and every staging pass lies inside one segment
  int nadd, nseg;
  int res_xcol;      // >= 0: the residual is this column block of the staged input tile
  int vec_out;       // output rows are 16-B pieces
  int out_dim[NQ], in_dim[NQ];
  ColChunk ch[NQ];
  ColPass ps[MAXPASS];
  ColAdd ad[GNC_MAX_SEGMENTS];
  ColSeg sg[GNC_MAX_SEGMENTS];
  const float* bias[NQ];
  const float* gamma;
  const float* beta;
  float eps;
  const float* residual;
  int ld_res;
  float* out;
  int ld_out;
  float* save[NQ];
  const float* w_alt;  // gridDim.y == 2 (two projections of the same rows in one launch): blockIdx.y == 1 reads this first-chunk
  float* out_alt;      // weight and writes here
};

// Shape traits: what the launcher knows about a launch at compile time.  A launch of this kernel is a few microseconds long
// and most of that used to be scalar bookkeeping in front of the first MFMA (which segment covers which pass, which chunk is
// alive, feature masks, ~110 plan fields fighting for ~100 SGPRs); the default model (models/GNN.py:220-259, D = 128) issues
// five shapes over and over, so those get instances in which all of it is a constant.  0 / -1 = decided at run time.
struct ShapeAny {   // everything from the plan
  static constexpr int L = 0, N0 = 0, NADD = -1, LN = -1, RES = -1, STAGE = -1, NARROW = -1;
  static constexpr bool FULLH = false, FULLK0 = false, ROWIDS = true, TRW = false;
};
struct ShapeProj {  // one Linear 128 -> 128 over row-ordered rows: the W-split's node-side products
  static constexpr int L = 1, N0 = 1, NADD = 0, LN = 0, RES = 0, STAGE = 1, NARROW = 0;
  static constexpr bool FULLH = true, FULLK0 = true, ROWIDS = false, TRW = false;
};
struct ShapeProjT {  // one Linear over [a | b] (256) read through the TRANSPOSED weight: out = a W[:, :128] + b W[:, 128:256] (the
  // node-side gradient of the W-split: dx = d(ps) Ws + d(pd) Wd on the nn.Linear matrix as it lies)
  static constexpr int L = 1, N0 = 2, NADD = 0, LN = 0, RES = 0, STAGE = 1, NARROW = 0;
  static constexpr bool FULLH = true, FULLK0 = true, ROWIDS = false, TRW = true;
};
struct ShapeEdge {  // edge processor, W-split: e (128) + two gathered addends, 3 Linears, LayerNorm, residual e
  static constexpr int L = 3, N0 = 1, NADD = 2, LN = 1, RES = 1, STAGE = 1, NARROW = 0;
  static constexpr bool FULLH = true, FULLK0 = true, ROWIDS = false, TRW = false;
};
struct ShapeNode {  // node processor: [x | agg] (256), 3 Linears, LayerNorm, residual x
  static constexpr int L = 3, N0 = 2, NADD = 0, LN = 1, RES = 1, STAGE = 1, NARROW = 0;
  static constexpr bool FULLH = true, FULLK0 = true, ROWIDS = false, TRW = false;
};
struct ShapeEnc {   // encoders: <= 16 raw columns (rows not made of 16-B pieces), 3 Linears, LayerNorm
  static constexpr int L = 3, N0 = 1, NADD = 0, LN = 1, RES = 0, STAGE = 0, NARROW = 1;
  static constexpr bool FULLH = true, FULLK0 = false, ROWIDS = false, TRW = false;
};
struct ShapeDec {   // decoder: 128 -> 128 -> 128 -> few, no LayerNorm
  static constexpr int L = 3, N0 = 1, NADD = 0, LN = 0, RES = 0, STAGE = 1, NARROW = 0;
  static constexpr bool FULLH = false, FULLK0 = true, ROWIDS = false, TRW = false;
};

// 16 readable bytes for the loads that have nothing to read (kept unconditional: see the tile loop)
__device__ __attribute__((aligned(16))) float gnc_col_dummy[4];

// Phase probe (`make probe_col16`, tools/profile_col16.py): shader-clock stamps of every wave at the phase boundaries of its
// first tile.  Off in the shipped library.
#ifdef GNC_PHASE_PROBE
__device__ unsigned long long gnc_col_probe[1024 * 8 * 12];
extern "C" int gnc_col_probe_read(void* dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(gnc_col_probe), bytes);
}
#define CPROBE(k) do { if (blockIdx.x < 1024 && it == 0 && (threadIdx.x & 63) == 0)                                   \
    gnc_col_probe[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 12 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define CPROBE(k) do {} while (0)
#endif

template <int NT, bool SAVE, bool IDX, class S = ShapeAny>
__global__ __launch_bounds__(NT * 64) void mlp_col16_kernel(const ColPlan p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NTV = NT * 64;
  constexpr int LDH = NT * 16 + 4;
  constexpr int CPP = NTV / 16;               // staging: 16-B pieces of a row per pass
  constexpr int NPASS = NQ * CK / (CPP * 4);  // 4 (8 waves) or 8 (4 waves)
  const int tid0 = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int rows = p.rows;
  const int L = S::L ? S::L : p.L;
  const int H = S::FULLH ? NT * 16 : p.H, od = S::FULLH ? NT * 16 : p.od;
  const int n0 = S::N0 ? S::N0 : p.n0;
  const int ldx = S::N0 ? S::N0 * CK + 4 : p.ldx;
  const int nadd = S::NADD >= 0 ? S::NADD : p.nadd;
  const bool narrow_w0 = S::NARROW >= 0 ? S::NARROW != 0 : p.narrow_w0 != 0;
  const bool vec_stage = S::STAGE >= 0 ? S::STAGE != 0 : p.vec_stage != 0;
  const bool has_ln = S::LN >= 0 ? S::LN != 0 : p.gamma != nullptr;
  const int k0pad = S::FULLK0 ? n0 * CK : p.k0pad;
  float* xbuf = lds;                                    // [16][ldx]: the first Linear's input rows (virtual concat)
  float* hbuf = xbuf + 16 * ldx;                        // [2][16][LDH]: hidden rows, ping-pong
  float* lnbuf = hbuf + 2 * 16 * LDH;                   // [2][NT][16]: per-wave row sums of the LayerNorm
#ifdef GNC_PHASE_PROBE
  const unsigned long long probe_t0 = __builtin_readcyclecounter();
#endif
  if constexpr (!S::FULLK0) {  // the padding columns [in_dim, end of the last chunk) of the input tile are zero for good
    const int padw = ldx - 4 - p.k_in;
    for (int u = tid0; u < 16 * padw; u += NTV) {
      const int r = u / padw, c = u - r * padw;
      xbuf[r * ldx + p.k_in + c] = 0.f;
    }
  }

  // Weight rows [16w, 16w + 16) of a chunk through a bounds-checked window that ends with the matrix (rows past it read as
  // 0); every load is unconditional and addressed by one lane offset + an immediate.  A dead chunk has a window of 0 bytes:
  // its loads are still issued - every path through the tile then has the SAME number of loads in flight, which is what
  // lets hipcc wait for "all but the n youngest" instead of for everything - but read nothing.  Nothing here may touch the
  // loaded registers: columns past the matrix are cleared by wmask at use.
  auto wload = [&](f32x4 (&wr)[NKB], const ColChunk& c, int j, int g) {
    const __amdgpu_buffer_rsrc_t win = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(c.w), 0, c.nrec, 0x00020000);
    if constexpr (S::TRW) {  // transposed chunk: component s of block cb = W[16cb + 4g + s][16w + j] (see mlp_bwd_col16.hip)
      const uint32_t off0 = (uint32_t)(4 * g * c.ldw + 16 * w + j) * 4u;
      const uint32_t rowb = (uint32_t)c.ldw * 4u;
#pragma unroll
      for (int cb = 0; cb < NKB; ++cb) {
        const uint32_t so = (uint32_t)(16 * cb) * rowb;
        wr[cb].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(win, off0, so, 0));
        wr[cb].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(win, off0, so + rowb, 0));
        wr[cb].z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(win, off0, so + 2 * rowb, 0));
        wr[cb].w = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(win, off0, so + 3 * rowb, 0));
      }
      return;
    }
    const uint32_t off0 = (uint32_t)((16 * w + j) * c.ldw + 4 * g) * 4u;
#pragma unroll
    for (int cb = 0; cb < NKB; ++cb) wr[cb] = window_load(win, off0 + (uint32_t)(cb * 64));
  };
  auto wmask = [&](f32x4 (&wr)[NKB], int kvalid, int g) {
    if (kvalid < CK) {  // wave-uniform: a partial chunk
#pragma unroll
      for (int cb = 0; cb < NKB; ++cb) {
        const int k = 16 * cb + 4 * g;
        wr[cb].x = k + 0 < kvalid ? wr[cb].x : 0.f; wr[cb].y = k + 1 < kvalid ? wr[cb].y : 0.f;
        wr[cb].z = k + 2 < kvalid ? wr[cb].z : 0.f; wr[cb].w = k + 3 < kvalid ? wr[cb].w : 0.f;
      }
    }
  };
  // the first Linear of an encoder: in_dim <= 16 with rows that are not 16-B pieces (nn.Linear(3, H)): one k block, by dwords
  auto wload_narrow = [&](f32x4& v, int j, int g) {
    const __amdgpu_buffer_rsrc_t win = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.ch[0].w), 0, p.nrec_narrow, 0x00020000);
    const uint32_t off0 = (uint32_t)((16 * w + j) * p.ch[0].ldw + 4 * g) * 4u;
    v.x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(win, off0 + 0u, 0, 0));
    v.y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(win, off0 + 4u, 0, 0));
    v.z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(win, off0 + 8u, 0, 0));
    v.w = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(win, off0 + 12u, 0, 0));
  };
  // ... its one k block replaces block 0 of set 0 (whose own loads read nothing); a dword past a row's end is the next row's
  // first and is cleared here, at use
  auto wmask_narrow = [&](f32x4 (&wr)[NKB], const f32x4& v, int g) {
    const int kin = p.k_in, k = 4 * g;
    wr[0].x = k + 0 < kin ? v.x : 0.f; wr[0].y = k + 1 < kin ? v.y : 0.f;
    wr[0].z = k + 2 < kin ? v.z : 0.f; wr[0].w = k + 3 < kin ? v.w : 0.f;
  };
  // acc0 / acc1 += W chunk x rows: k blocks [0, nkb) of the chunk, `src` = &tile[j][first column of the chunk + 4g]
  auto mma = [&](auto nb_, f32x4& acc0, f32x4& acc1, const f32x4 (&wr)[NKB], const float* src, int nkb) {
    constexpr int NB = decltype(nb_)::value;
#pragma unroll
    for (int c0 = 0; c0 < NB; c0 += 4) {  // four B fragments in flight
      f32x4 b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) b[u] = *reinterpret_cast<const f32x4*>(src + 16 * (c0 + u));  // in bounds: tiles hold whole chunks
#pragma unroll
      for (int u = 0; u < 4; u += 2) {
        if (c0 + u < nkb) {  // wave-uniform: k blocks past the layer's width are skipped two at a time (zeros x zeros otherwise)
          acc0 = mfma16(wr[c0 + u].x, b[u].x, acc0);
          acc1 = mfma16(wr[c0 + u + 1].x, b[u + 1].x, acc1);
          acc0 = mfma16(wr[c0 + u].y, b[u].y, acc0);
          acc1 = mfma16(wr[c0 + u + 1].y, b[u + 1].y, acc1);
          acc0 = mfma16(wr[c0 + u].z, b[u].z, acc0);
          acc1 = mfma16(wr[c0 + u + 1].z, b[u + 1].z, acc1);
          acc0 = mfma16(wr[c0 + u].w, b[u].w, acc0);
          acc1 = mfma16(wr[c0 + u + 1].w, b[u + 1].w, acc1);
        }
      }
    }
  };
  // this lane's four per-feature parameters (a bias, gamma, beta): one bounds-checked 16-B load, features past `n` read as 0;
  // requested with the weights, consumed where they are needed (q == nullptr: a window of 0 bytes)
  auto param4 = [&](const float* q, int n, int f0) -> f32x4 {
    const __amdgpu_buffer_rsrc_t win = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(q), 0, q ? n * 4 : 0, 0x00020000);
    return window_load(win, (uint32_t)f0 * 4u);
  };

  int it = 0;
  for (int t = blockIdx.x; t < p.num_tiles; t += gridDim.x, ++it) {
#ifdef GNC_PHASE_PROBE
    if (blockIdx.x < 1024 && it == 0 && (threadIdx.x & 63) == 0) gnc_col_probe[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 12] = probe_t0;
#endif
    CPROBE(1);  // kernel prologue done
    // lane roles from an id the compiler cannot see through: everything derived from it (masks, offsets, addresses) then
    // lives inside the tile loop - which runs once for a small batch - instead of being hoisted out of it and carried
    // (and spilled) across it
    const int lane = fresh_lane();
    const int tid = w * 64 + lane;
    const int j = lane & 15;  // data row of the B operand / accumulator column; weight row of the A operand
    const int g = lane >> 4;
    const int f0 = 16 * w + 4 * g;  // first of this lane's four features
    const int row0 = t * R16;
    const int myrow = row0 + j < rows ? row0 + j : rows - 1;  // clamped: rows past the end are computed and dropped
    const bool row_live = row0 + j < rows;
    const int sr = tid / CPP, sc4 = tid % CPP;  // staging: row sr of the tile, 16-B piece sc4 of every pass
    const int srow = row0 + sr < rows ? row0 + sr : rows - 1;

    // Request order = arrival order (one in-order counter per wave): row ids, the first weight chunk and the parameters, the
    // rows the ids name, the other weight chunks - an id asked for behind the weights could not be used before the last
    // weight had landed.  (Loads are unconditional - a segment without ids reads element 0 of a dummy - because hipcc waits
    // for a load that sits in a conditional block at the end of that block.)
    int id_add[GNC_MAX_SEGMENTS], id_row[NPASS];
#pragma unroll
    for (int s = 0; s < GNC_MAX_SEGMENTS; ++s) {
      id_add[s] = myrow;
      if (S::NADD >= 0 && s >= S::NADD) continue;
      if constexpr (IDX) {  // (a launch without gathered segments - all but the edge processor's - asks for its rows at once)
        const int32_t* ip = p.ad[s].index;
        id_add[s] = (ip ? ip : reinterpret_cast<const int32_t*>(gnc_col_dummy))[ip ? myrow : 0];
        id_add[s] = ip ? id_add[s] : myrow;
      }
    }
    constexpr int LIVE_PASSES = (S::N0 && S::FULLK0) ? S::N0 * CK / (CPP * 4) : NPASS;  // (FULLK0: the concat fills whole passes)
#pragma unroll
    for (int s = 0; s < LIVE_PASSES; ++s) {
      id_row[s] = srow;
      if constexpr (IDX && S::ROWIDS) {  // (the W-split edge shape gathers only its addends)
        const int32_t* ip = p.ps[s].index;
        id_row[s] = (ip ? ip : reinterpret_cast<const int32_t*>(gnc_col_dummy))[ip ? srow : 0];
        id_row[s] = ip ? id_row[s] : srow;
      }
    }

    // chunk q of the tile: q < n0: columns [q CK, ..) of the first Linear; else Linear q - n0 + 1 (widths <= 128: one chunk)
    f32x4 wr[NSET][NKB], wnar = {0.f, 0.f, 0.f, 0.f};
    if constexpr (S::NARROW != 0) wload_narrow(wnar, j, g);
    if constexpr (S::NARROW != 1) {
      if (S::L == 1 && blockIdx.y != 0) {
        ColChunk c = p.ch[0];
        c.w = p.w_alt;
        wload(wr[0], c, j, g);
      } else {
        wload(wr[0], p.ch[0], j, g);
      }
    }
    else {
#pragma unroll
      for (int cb = 0; cb < NKB; ++cb) wr[0][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 bias[NQ];
#pragma unroll
    for (int l = 0; l < NQ; ++l) {
      bias[l] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (S::L && l >= S::L) continue;
      bias[l] = param4(p.bias[l], S::FULLH ? NT * 16 : p.out_dim[l], f0);
    }
    f32x4 ln_g = {0.f, 0.f, 0.f, 0.f}, ln_b = {0.f, 0.f, 0.f, 0.f};
    if constexpr (S::LN != 0) {
      ln_g = param4(p.gamma, od, f0);
      ln_b = param4(p.beta, od, f0);
    }
    // residual rows that are not part of the staged input tile
    f32x4 resv = {0.f, 0.f, 0.f, 0.f};
    if constexpr (S::RES < 0 || S::RES == 2) {
      const bool live = p.residual != nullptr && p.res_xcol < 0 && f0 < od;
      resv = *reinterpret_cast<const f32x4*>(live ? p.residual + (int64_t)myrow * p.ld_res + f0 : gnc_col_dummy);
    }
    // W-split addends: rows already in the hidden width, in the accumulator layout (summed after the first Linear)
    f32x4 av[GNC_MAX_SEGMENTS];
#pragma unroll
    for (int s = 0; s < GNC_MAX_SEGMENTS; ++s) {
      av[s] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (S::NADD >= 0 && s >= S::NADD) { id_add[s] = 0; continue; }
      const bool ok = s < nadd && (uint32_t)id_add[s] < p.ad[s].table_rows && (S::FULLH || f0 < H);
      av[s] = *reinterpret_cast<const f32x4*>(ok ? p.ad[s].ptr + (int64_t)id_add[s] * p.ad[s].ld + f0 : gnc_col_dummy);
      id_add[s] = ok ? 1 : 0;
    }
    // ... the remaining weight sets are requested behind the tile's rows (which wait for their ids, but are needed first)
    auto wload_rest = [&]() {
#pragma unroll
      for (int q = 1; q < NSET; ++q) {
        if (S::L && S::N0 && q >= S::N0 + S::L - 1) continue;  // (dead chunks of a run-time shape are windows of 0 bytes)
        wload(wr[q], p.ch[q], j, g);
      }
    };
    // the virtual concat's rows: column c of the tile = column c - wcol of the segment that covers it
    if (vec_stage) {
      // thread (sr, sc4) takes piece sc4 of each CPP-piece pass; all passes are requested before the first is written
      f32x4 sv[LIVE_PASSES];
      int sok[LIVE_PASSES];
#pragma unroll
      for (int s = 0; s < LIVE_PASSES; ++s) {
        const bool here = S::FULLK0 || sc4 * 4 < p.ps[s].width;
        const bool ok = here && (uint32_t)id_row[s] < p.ps[s].table_rows;
        sv[s] = *reinterpret_cast<const f32x4*>(ok ? p.ps[s].ptr + (int64_t)id_row[s] * p.ps[s].ld + sc4 * 4 : gnc_col_dummy);
        sok[s] = here ? (ok ? 1 : 2) : 0;  // 2: covered, but the id is outside the table: zeros
      }
      wload_rest();
#pragma unroll
      for (int s = 0; s < LIVE_PASSES; ++s)
        if (sok[s]) *reinterpret_cast<f32x4*>(xbuf + sr * ldx + (s * CPP + sc4) * 4) = sok[s] == 1 ? sv[s] : f32x4{0.f, 0.f, 0.f, 0.f};
    } else {  // rows that are not 16-B pieces (the reference's [N, 3] inputs): element by element
      wload_rest();
#pragma unroll
      for (int s = 0; s < GNC_MAX_SEGMENTS; ++s) {
        if (S::NARROW == 1 && s >= 1) continue;  // (the encoder shape: one table)
        if (s < p.nseg) {
          const float* base = p.sg[s].ptr;
          const int32_t* ip = p.sg[s].index;
          const int width = p.sg[s].width, ld = p.sg[s].ld, wcol = p.sg[s].wcol;
          if constexpr (S::NARROW == 1) {  // <= 16 columns: thread (r, c) = (tid / 16, tid % 16), one element each
            const int r = tid >> 4, c = tid & 15;
            if (tid < 256 && c < width) {
              const int rr = row0 + r < rows ? row0 + r : rows - 1;
              const int id = ip ? ip[rr] : rr;
              const bool ok = (uint32_t)id < p.sg[s].table_rows;
              const float v = base[(int64_t)(ok ? id : 0) * ld + c];
              xbuf[r * ldx + wcol + c] = ok ? v : 0.f;
            }
            continue;
          }
          for (int u = tid; u < 16 * width; u += NTV) {
            const int r = u / width, c = u - r * width;
            const int rr = row0 + r < rows ? row0 + r : rows - 1;
            const int id = ip ? ip[rr] : rr;
            const bool ok = (uint32_t)id < p.sg[s].table_rows;
            const float v = base[(int64_t)(ok ? id : 0) * ld + c];
            xbuf[r * ldx + wcol + c] = ok ? v : 0.f;
          }
        }
      }
    }
    CPROBE(2);  // rows staged (ids, rows waited for)
    __syncthreads();
    CPROBE(3);

    // ---------------------------------------------------------------- first Linear: chunks of the virtual concat
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if (q < n0) {
        if (q == 0 && narrow_w0) wmask_narrow(wr[0], wnar, g);
        else if (!S::FULLK0) wmask(wr[q % NSET], p.ch[q].kvalid, g);
        const int nkb = (k0pad - q * CK) >> 4;
        mma(std::integral_constant<int, NKB>{}, acc0, acc1, wr[q % NSET], xbuf + j * ldx + q * CK + 4 * g, nkb < NKB ? nkb : NKB);
      }
      if (q == 0 && !(S::L && S::N0 && S::N0 + S::L - 1 <= NSET)) wload(wr[0], p.ch[NSET], j, g);  // chunk 3 -> set 0
    }
    CPROBE(4);  // first Linear issued
    f32x4 o = (acc0 + acc1) + bias[0];
#pragma unroll
    for (int s = 0; s < GNC_MAX_SEGMENTS; ++s)
      if (id_add[s]) o += av[s];

    // ---------------------------------------------------------------- hidden and output Linears
    for (int l = 1; l < L; ++l) {
      o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
      float* hb = hbuf + ((l - 1) & 1) * 16 * LDH;
      *reinterpret_cast<f32x4*>(hb + j * LDH + f0) = o;
      if constexpr (SAVE) {
        if (row_live && (S::FULLH || f0 < H)) *reinterpret_cast<f32x4*>(p.save[l - 1] + (int64_t)(row0 + j) * H + f0) = o;
      }
      __syncthreads();
      acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
      acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
      if (S::FULLH || 16 * w < p.out_dim[l]) {
        const int q = n0 + l - 1;  // 1..NQ-1, wave-uniform: one branch per register set (static register names)
        const int nkb = S::FULLH ? NT : (p.in_dim[l] + 15) >> 4;
#define GNC_COL_HIDDEN(Q_)                                                                          \
  if (q == Q_) {                                                                                     \
    if (!S::FULLH) wmask(wr[Q_ % NSET], p.ch[Q_].kvalid, g);                                           \
    mma(std::integral_constant<int, NT>{}, acc0, acc1, wr[Q_ % NSET], hb + j * LDH + 4 * g, nkb);      \
  }
        GNC_COL_HIDDEN(1) GNC_COL_HIDDEN(2) GNC_COL_HIDDEN(3)
#undef GNC_COL_HIDDEN
      }
      o = (acc0 + acc1) + (l == 1 ? bias[1] : l == 2 ? bias[2] : bias[3]);
    }

    CPROBE(5);  // hidden / output Linears
    // ---------------------------------------------------------------- LayerNorm across the waves (two-pass)
    if (has_ln) {
      const float inv_n = 1.f / (float)od;
      float s = add_quarters((o.x + o.y) + (o.z + o.w));  // features >= od hold exact zeros (zero weight rows, zero bias)
      if (g == 0) lnbuf[w * 16 + j] = s;
      __syncthreads();
      s = 0.f;
#pragma unroll
      for (int ww = 0; ww < NT; ++ww) s += lnbuf[ww * 16 + j];
      const float mean = s * inv_n;
      o.x = (S::FULLH || f0 + 0 < od) ? o.x - mean : 0.f; o.y = (S::FULLH || f0 + 1 < od) ? o.y - mean : 0.f;
      o.z = (S::FULLH || f0 + 2 < od) ? o.z - mean : 0.f; o.w = (S::FULLH || f0 + 3 < od) ? o.w - mean : 0.f;
      float v = add_quarters((o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w));
      if (g == 0) lnbuf[NT * 16 + w * 16 + j] = v;
      __syncthreads();
      v = 0.f;
#pragma unroll
      for (int ww = 0; ww < NT; ++ww) v += lnbuf[NT * 16 + ww * 16 + j];
      const float rstd = __frsqrt_rn(v * inv_n + p.eps);
      o.x = o.x * (ln_g.x * rstd) + ln_b.x; o.y = o.y * (ln_g.y * rstd) + ln_b.y;
      o.z = o.z * (ln_g.z * rstd) + ln_b.z; o.w = o.w * (ln_g.w * rstd) + ln_b.w;
    }

    CPROBE(6);  // LayerNorm
    // ---------------------------------------------------------------- residual, output rows
    if constexpr (S::RES == 1) {
      o += *reinterpret_cast<const f32x4*>(xbuf + j * ldx + p.res_xcol + f0);  // rows of the staged input
    } else if constexpr (S::RES != 0) {
      if (p.residual && f0 < od) {
        if (p.res_xcol >= 0) o += *reinterpret_cast<const f32x4*>(xbuf + j * ldx + p.res_xcol + f0);
        else o += resv;
      }
    }
    if (row_live && (S::FULLH || f0 < od)) {
      float* op = ((S::L == 1 && blockIdx.y != 0) ? p.out_alt : p.out) + (int64_t)(row0 + j) * p.ld_out + f0;
      if (S::FULLH || p.vec_out) {
        *reinterpret_cast<f32x4*>(op) = o;
      } else {
        op[0] = o.x;
        if (f0 + 1 < od) op[1] = o.y;
        if (f0 + 2 < od) op[2] = o.z;
        if (f0 + 3 < od) op[3] = o.w;
      }
    }

    CPROBE(7);  // rows stored
    CPROBE(8);
    // More tiles than workgroups: the next tile's staging must not overwrite the input tile under a wave that is still reading
    // it - the residual is read from it after the last barrier of the tile, and a lone Linear has no barrier behind its MFMAs
    // at all.  (Never taken in the one-tile-per-workgroup regime.)
    if (t + (int)gridDim.x < p.num_tiles) __syncthreads();
  }
}

template <int NT, bool SAVE, bool IDX, class S = ShapeAny>
int launch_col(const ColPlan& p, hipStream_t stream) {
  const size_t smem = ((size_t)16 * p.ldx + 2 * 16 * (NT * 16 + 4) + 2 * NT * 16) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    int rc = gnc::check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_col16_kernel<NT, SAVE, IDX, S>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                            "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    if (rc) return rc;
    attr_set = true;
  }
  const int cap = 2 * gnc::num_cu();
  const int grid = p.num_tiles < cap ? p.num_tiles : cap;
  mlp_col16_kernel<NT, SAVE, IDX, S><<<dim3((unsigned)grid, p.w_alt ? 2u : 1u), dim3(NT * 64), smem, stream>>>(p);
  return gnc::check_launch("mlp_col16_kernel");
}

template <int NT>
int launch_col_nt(const ColPlan& p, bool save, bool idx, hipStream_t stream) {
  if (idx) return save ? launch_col<NT, true, true>(p, stream) : launch_col<NT, false, true>(p, stream);
  return save ? launch_col<NT, true, false>(p, stream) : launch_col<NT, false, false>(p, stream);
}

// the default model's launch shapes (ShapeProj .. ShapeDec): 0 = none of them
enum { SHAPE_ANY = 0, SHAPE_PROJ, SHAPE_EDGE, SHAPE_NODE, SHAPE_ENC, SHAPE_DEC };
int shape_of(const ColPlan& p, int nt, bool save, bool idx, bool row_ids) {
  static const bool off = getenv("GNC_COL16_NO_SHAPES") != nullptr;  // A/B: run-time shapes only
  if (off || nt != 8 || p.H != 128 || row_ids) return SHAPE_ANY;
  const bool ln = p.gamma != nullptr, res = p.residual != nullptr;
  if (p.L == 1 && p.k_in == 128 && p.nadd == 0 && !ln && !res && p.vec_stage && !p.narrow_w0 && !idx && !save && p.vec_out)
    return SHAPE_PROJ;
  if (p.L != 3) return SHAPE_ANY;
  if (p.od == 128 && p.k_in == 128 && p.nadd == 2 && ln && p.res_xcol >= 0 && p.vec_stage && !p.narrow_w0 && idx && p.vec_out)
    return SHAPE_EDGE;
  if (p.od == 128 && p.k_in == 256 && p.nadd == 0 && ln && p.res_xcol >= 0 && p.vec_stage && !p.narrow_w0 && !idx && p.vec_out)
    return SHAPE_NODE;
  if (p.od == 128 && p.k_in <= 16 && p.nseg == 1 && p.nadd == 0 && ln && !res && !p.vec_stage && p.narrow_w0 && !idx && p.vec_out)
    return SHAPE_ENC;
  if (p.k_in == 128 && p.nadd == 0 && !ln && !res && p.vec_stage && !p.narrow_w0 && !idx) return SHAPE_DEC;
  return SHAPE_ANY;
}

bool al16h(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; }
uint32_t rows32(int64_t r) { return r > INT32_MAX ? (uint32_t)INT32_MAX : (uint32_t)r; }

}  // namespace

// Rows up to which the column-split kernel serves a description (GNC_COL16_MAX_ROWS overrides; 0 switches it off): 128 x the
// number of CUs (32,768: `tools/latency_sizes.py` - it beats the throughput kernels up to there, 0.60 vs 0.73 ms per replayed
// forward of a 128 x 128 pixel graph)
int64_t gnc_mlp::col16_max_rows() {
  static const int64_t v = [] {
    const char* e = getenv("GNC_COL16_MAX_ROWS");
    return e ? (int64_t)atoll(e) : (int64_t)8 * R16 * gnc::num_cu();
  }();
  return v;
}

// ... and up to which the small-batch BACKWARD kernels do (data kernel, one-launch weight gradients): above it the
// register-resident data kernel / the partial-sum products take over (GNC_BWD_COL16_MAX_ROWS overrides)
int64_t gnc_mlp::bwd_col16_max_rows() {
  static const int64_t v = [] {
    const char* e = getenv("GNC_BWD_COL16_MAX_ROWS");
    const int64_t lim = e ? (int64_t)atoll(e) : (int64_t)2 * R16 * gnc::num_cu();
    return lim < col16_max_rows() ? lim : col16_max_rows();
  }();
  return v;
}

int gnc_mlp::launch_col16(const gnc_mlp_desc_t& d, hipStream_t stream, bool* launched, bool probe_only) {
  *launched = false;
  if (d.rows > col16_max_rows() || d.rows >= INT32_MAX) return GNC_OK;
  const int L = d.num_linear;
  if (L > 1 && d.activation != GNC_ACT_RELU) return GNC_OK;
  const int H = d.out_dim[0], od = d.out_dim[L - 1];
  static const bool narrow = getenv("GNC_COL16_D64") != nullptr;  // A/B: also widths <= 64 (weights-resident kernel otherwise)
  if (H > 128 || od > 128 || (!narrow && H <= 64 && od <= 64)) return GNC_OK;
  const int n0 = (d.in_dim[0] + CK - 1) / CK;
  if (n0 + L - 1 > NQ) return GNC_OK;  // (the default model: 1 + 2 everywhere, 2 + 2 for the node processor's [x | agg])
  const int wmax = H > od ? H : od;
  const int nt = wmax <= 64 ? 4 : 8;
  const int pw = nt * 16;  // columns of one staging pass

  ColPlan p = {};
  p.rows = (int)d.rows;
  p.num_tiles = (int)gnc::ceil_div(d.rows, (int64_t)R16);
  p.L = L; p.H = H; p.od = od;
  p.n0 = n0; p.nq = n0 + L - 1;
  p.k_in = d.in_dim[0];
  p.k0pad = (d.in_dim[0] + 15) & ~15;
  p.ldx = n0 * CK + 4;  // whole chunks: the B fragments of a chunk are read unconditionally
  for (int l = 0; l < L; ++l) {
    const int ldw = ldw_of(d, l);
    if ((int64_t)d.out_dim[l] * ldw * 4 > 0x7fffffffll) return GNC_OK;
    p.out_dim[l] = d.out_dim[l];
    p.in_dim[l] = d.in_dim[l];
    p.bias[l] = d.bias[l];
    if (ldw % 4 == 0 && al16h(d.weight[l])) continue;  // rows of 16-B pieces (a width that is not a multiple of 4 is masked)
    if (l == 0 && d.in_dim[0] <= 16) p.narrow_w0 = 1;
    else return GNC_OK;
  }
  for (int q = 0; q < p.nq; ++q) {
    const int l = q < n0 ? 0 : q - n0 + 1, kbase = q < n0 ? q * CK : 0;
    const int ldw = ldw_of(d, l);
    ColChunk& c = p.ch[q];
    c.w = d.weight[l] + kbase;
    c.ldw = ldw;
    c.nrec = ((d.out_dim[l] - 1) * ldw + d.in_dim[l] - kbase) * 4;
    c.kvalid = d.in_dim[l] - kbase;
  }
  if (p.narrow_w0) {  // the dword loader reads chunk 0; the 16-B loads of set 0 read nothing
    p.nrec_narrow = p.ch[0].nrec;
    p.ch[0].nrec = 0;
  }
  // staging: the vector path needs every pass inside one MATMUL segment made of 16-B pieces
  p.vec_stage = 1;
  for (int s = 0; s < d.num_segments; ++s) {
    const gnc_mlp_segment_t& sg = d.seg[s];
    if (sg.mode == GNC_SEG_ADD) {
      if (sg.width != H || H % 4 != 0 || sg.ld % 4 != 0 || !al16h(sg.ptr) || p.nadd >= GNC_MAX_SEGMENTS) return GNC_OK;
      p.ad[p.nadd++] = {sg.ptr, sg.index, sg.index ? rows32(sg.table_rows) : (uint32_t)INT32_MAX, sg.ld};
      continue;
    }
    if (sg.wcol < 0 || sg.wcol + sg.width > d.in_dim[0]) return GNC_OK;
    p.sg[p.nseg++] = {sg.ptr, sg.index, sg.index ? rows32(sg.table_rows) : (uint32_t)INT32_MAX, sg.width, sg.ld, sg.wcol};
    if (sg.ld % 4 != 0 || sg.width % 4 != 0 || sg.wcol % pw != 0 || !al16h(sg.ptr)) p.vec_stage = 0;
  }
  if (p.vec_stage) {
    for (int s = 0; s < p.nseg; ++s) {
      const ColSeg& sg = p.sg[s];
      for (int c = 0; c < sg.width; c += pw) {
        ColPass& ps = p.ps[(sg.wcol + c) / pw];
        ps.ptr = sg.ptr + c;
        ps.index = sg.index;
        ps.table_rows = sg.table_rows;
        ps.ld = sg.ld;
        ps.width = sg.width - c < pw ? sg.width - c : pw;
      }
    }
  }
  p.gamma = d.ln_gamma;
  p.beta = d.ln_gamma ? d.ln_beta : nullptr;
  p.eps = d.ln_eps;
  p.res_xcol = -1;
  if (d.residual) {
    if (d.ld_residual % 4 != 0 || od % 4 != 0 || !al16h(d.residual)) return GNC_OK;
    p.residual = d.residual;
    p.ld_res = d.ld_residual;
    for (int s = 0; s < p.nseg; ++s)  // the residual is one of the staged row-ordered tables: its rows are in the input tile
      if (p.vec_stage && !p.sg[s].index && p.sg[s].ptr == d.residual && p.sg[s].ld == d.ld_residual && p.sg[s].width >= od)
        p.res_xcol = p.sg[s].wcol;
  }
  p.out = d.out;
  p.ld_out = d.ld_out;
  p.vec_out = (d.ld_out % 4 == 0 && od % 4 == 0 && al16h(d.out)) ? 1 : 0;
  const bool save = d.save_act[0] != nullptr;
  if (save) {
    if (L < 2 || H % 4 != 0) return GNC_OK;
    for (int l = 0; l < L - 1; ++l) {
      if (!d.save_act[l] || !al16h(d.save_act[l])) return GNC_OK;
      p.save[l] = d.save_act[l];
    }
  }
  if (d.agg_out) return GNC_OK;  // (never asked for: gnc_mlp_agg_supported() sends these launches to K1)
  bool idx = false;
  for (int s = 0; s < d.num_segments; ++s) idx = idx || d.seg[s].index != nullptr;
  bool row_ids = false;  // a gathered MATMUL segment (the concat form of the edge processor)
  for (int s = 0; s < p.nseg; ++s) row_ids = row_ids || p.sg[s].index != nullptr;
  *launched = true;
  if (probe_only) return GNC_OK;
  switch (shape_of(p, nt, save, idx, row_ids)) {
    case SHAPE_PROJ: return launch_col<8, false, false, ShapeProj>(p, stream);
    case SHAPE_EDGE: return save ? launch_col<8, true, true, ShapeEdge>(p, stream) : launch_col<8, false, true, ShapeEdge>(p, stream);
    case SHAPE_NODE: return save ? launch_col<8, true, false, ShapeNode>(p, stream) : launch_col<8, false, false, ShapeNode>(p, stream);
    case SHAPE_ENC: return save ? launch_col<8, true, false, ShapeEnc>(p, stream) : launch_col<8, false, false, ShapeEnc>(p, stream);
    case SHAPE_DEC: return save ? launch_col<8, true, false, ShapeDec>(p, stream) : launch_col<8, false, false, ShapeDec>(p, stream);
    default: break;
  }
  return nt == 4 ? launch_col_nt<4>(p, save, idx, stream) : launch_col_nt<8>(p, save, idx, stream);
}

extern "C" int64_t gnc_mlp_small_batch_max_rows(void) { return gnc_mlp::col16_max_rows(); }

// Two projections of the same rows in ONE launch (the W-split's node-side products x Ws^T, x Wd^T: models/GNN.py:58-61 once per
// node): grid.y = 2 over the projection shape of the small-batch kernel.  GNC_ERR_UNSUPPORTED (nothing launched) outside that
// shape - the caller then issues two gnc_mlp_forward_f32 launches.
extern "C" int gnc_mlp_dual_projection_f32(const float* x, int64_t ld_x, int64_t rows, const float* wa, int64_t ld_wa, const float* wb,
                                           int64_t ld_wb, int32_t in_dim, int32_t out_dim, float* out_a, float* out_b, int64_t ld_out,
                                           void* stream) {
  static const bool off = getenv("GNC_NO_DUAL_PROJECTION") != nullptr;  // A/B switch
  if (!off && x && wa && wb && out_a && out_b && rows > gnc_mlp::col16_max_rows() && out_dim <= 64) {
    // a large batch at widths <= 64: the weights-resident kernel's DUAL instance (rows read once, both matrices resident)
    bool launched = false;
    const int rc = gnc_mlp::launch_resident_dual(x, ld_x, rows, wa, ld_wa, wb, ld_wb, in_dim, out_dim, out_a, out_b, ld_out,
                                                 (hipStream_t)stream, &launched);
    if (rc || launched) return rc;
  }
  if (off || !x || !wa || !wb || !out_a || !out_b || rows < 1 || rows > gnc_mlp::col16_max_rows() || in_dim != 128 || out_dim != 128 ||
      ld_wa != ld_wb || ld_wa % 4 != 0 || ld_x % 4 != 0 || ld_out % 4 != 0 || !al16h(x) || !al16h(wa) || !al16h(wb) || !al16h(out_a) ||
      !al16h(out_b) || (int64_t)128 * ld_wa * 4 > 0x7fffffffll) {
    gnc::set_error("gnc_mlp_dual_projection_f32: outside the small-batch projection shape (128 -> 128, 16-B rows) and the large-batch one (widths 33..64)");
    return GNC_ERR_UNSUPPORTED;
  }
  ColPlan p = {};
  p.rows = (int)rows;
  p.num_tiles = (int)gnc::ceil_div(rows, (int64_t)R16);
  p.L = 1; p.H = 128; p.od = 128;
  p.n0 = 1; p.nq = 1;
  p.k_in = 128; p.k0pad = 128; p.ldx = CK + 4;
  p.out_dim[0] = 128; p.in_dim[0] = 128;
  p.ch[0] = {wa, (int)ld_wa, (int)((127 * ld_wa + 128) * 4), 128};
  p.w_alt = wb;
  p.vec_stage = 1;
  p.nseg = 1;
  p.sg[0] = {x, nullptr, (uint32_t)INT32_MAX, 128, (int)ld_x, 0};
  p.ps[0] = {x, nullptr, (uint32_t)INT32_MAX, (int)ld_x, 128};
  p.res_xcol = -1;
  p.out = out_a;
  p.out_alt = out_b;
  p.ld_out = (int)ld_out;
  p.vec_out = 1;
  return launch_col<8, false, false, ShapeProj>(p, (hipStream_t)stream);
}

// out = a W[:, 0:128] + b W[:, 128:256] for W [128, ld_w] as nn.Linear holds it (read transposed, no copy): the node-side
// gradient of the W-split first Linear, dx = d(ps) Ws + d(pd) Wd (models/GNN.py:58-61 under autograd), in one launch for a small
// batch.  GNC_ERR_UNSUPPORTED (nothing launched) outside that shape.
extern "C" int gnc_mlp_projection_t2_f32(const float* a, int64_t ld_a, const float* b, int64_t ld_b, int64_t rows, const float* wmat,
                                         int64_t ld_w, int32_t hidden, int32_t dn, float* out, int64_t ld_out, void* stream) {
  static const bool off = getenv("GNC_NO_PROJECTION_T2") != nullptr;  // A/B switch
  if (off || !a || !b || !wmat || !out || rows < 1 || rows > gnc_mlp::col16_max_rows() || hidden != 128 || dn != 128 || ld_a % 4 != 0 ||
      ld_b % 4 != 0 || ld_out % 4 != 0 || ld_w < 2 * dn || !al16h(a) || !al16h(b) || !al16h(out) || (int64_t)128 * ld_w * 4 > 0x7fffffffll) {
    gnc::set_error("gnc_mlp_projection_t2_f32: outside the small-batch shape (two [rows, 128] tables, W [128, >= 256])");
    return GNC_ERR_UNSUPPORTED;
  }
  ColPlan p = {};
  p.rows = (int)rows;
  p.num_tiles = (int)gnc::ceil_div(rows, (int64_t)R16);
  p.L = 1; p.H = 128; p.od = 128;
  p.n0 = 2; p.nq = 2;
  p.k_in = 256; p.k0pad = 256; p.ldx = 2 * CK + 4;
  p.out_dim[0] = 128; p.in_dim[0] = 256;
  // chunk q: contraction over the 128 rows of W, output columns [128 q, 128 q + 128): window from W + 128 q to the matrix's end
  p.ch[0] = {wmat, (int)ld_w, (int)((127 * ld_w + 128) * 4), 128};
  p.ch[1] = {wmat + 128, (int)ld_w, (int)((127 * ld_w + 128) * 4), 128};
  p.vec_stage = 1;
  p.nseg = 2;
  p.sg[0] = {a, nullptr, (uint32_t)INT32_MAX, 128, (int)ld_a, 0};
  p.sg[1] = {b, nullptr, (uint32_t)INT32_MAX, 128, (int)ld_b, 128};
  p.ps[0] = {a, nullptr, (uint32_t)INT32_MAX, (int)ld_a, 128};
  p.ps[1] = {b, nullptr, (uint32_t)INT32_MAX, (int)ld_b, 128};
  p.res_xcol = -1;
  p.out = out;
  p.ld_out = (int)ld_out;
  p.vec_out = 1;
  return launch_col<8, false, false, ShapeProjT>(p, (hipStream_t)stream);
}
