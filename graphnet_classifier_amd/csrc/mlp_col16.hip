// K4, small-batch variant ("column split") for hidden widths <= 128: the reference's own regime, one ~1000-node graph per call
// (main.py:60, utils/train_model.py:35-45, utils/inference.py:59).
//
// The throughput kernels give a wave a whole row tile: at 128 features that is 3 x 8 x 32 = 768 DEPENDENT 16x16x4 MFMAs per
// 16 rows (10 us) behind a weight stream that is staged through LDS chunk by chunk (two barriers each) - 17..40 us per launch
// however few rows there are, and a forward is 15 such launches in a row.  Here a workgroup owns ONE 16-row tile and its
// waves split the OUTPUT FEATURES of every Linear: wave w computes features [16w, 16w + 16) with 32 MFMAs per layer,
//   * its 16 weight rows come straight from L2 into registers (W[16w + i][16cb + 4g .. +3] is exactly the A operand of
//     k block cb; the next layer's rows are requested before the current layer's MFMAs start) - no LDS staging of weights;
//   * the layer's input rows are shared through LDS ([16][K + 4] floats: the B operand of k block cb is one ds_read_b128),
//     hidden layers ping-pong between two such tiles: ONE barrier per layer;
//   * the accumulator layout of the 16-row kernels (register r on lane (j, g) = feature 16w + 4g + r of row j) makes every
//     row-ordered operand a plain 16-B access per lane: W-split addends, residual, saved activations and the output rows
//     need no transposes;
//   * LayerNorm sums cross the waves through 2 x NT x 16 floats of LDS (two-pass, two barriers).
// Aggregation epilogue (AGG): the finished rows go through the idle hidden tile and `out_dim` lanes walk the 16 rows in order
// (bit-identical to K1: sums start at 0.0 and add rows in ascending order); the first / last destination of a tile are left
// to gnc_agg_fixup_f32 when the neighbouring tile holds more of their rows (agg_fix[2 tile], [2 tile + 1]; -1 otherwise).
// Rounding differs from the streaming kernels (two interleaved accumulators per output tile): callers compare with
// tolerances, not bit for bit, across kernel families.
#include <stdlib.h>

#include <type_traits>

#include "mlp_device16.h"

using namespace gnc_mlp;

namespace {

constexpr int CK = 128;        // k columns of one register-resident weight chunk
constexpr int NKB = CK / 16;   // its 16-wide k blocks
constexpr int MAX_K0 = 1024;   // widest virtual concat the LDS input tile is sized for

struct ColArgs {
  int num_tiles;
  int k0pad;    // in_dim[0] rounded up to 16
  int ldx;      // row stride of the input tile (floats)
  int fix_len;  // entries of agg_fix
  int narrow_w0;  // weight[0] is read by dwords (in_dim[0] <= 16, rows not made of 16-B pieces)
};

__device__ __forceinline__ bool al16d(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <int NT, bool AGG, bool SAVE>
__global__ __launch_bounds__(NT * 64) void mlp_col16_kernel(const gnc_mlp_desc_t d, const ColArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NTV = NT * 64;
  constexpr int LDH = NT * 16 + 4;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15;  // data row of the B operand / accumulator column; weight row of the A operand
  const int g = lane >> 4;
  const int L = d.num_linear;
  const int rows = (int)d.rows;
  const int H = d.out_dim[0];
  const int od = d.out_dim[L - 1];
  const int ldx = a.ldx;
  float* xbuf = lds;                                    // [16][ldx]: the first Linear's input rows (virtual concat)
  float* hbuf = xbuf + 16 * ldx;                        // [2][16][LDH]: hidden rows, ping-pong
  float* lnbuf = hbuf + 2 * 16 * LDH;                   // [2][NT][16]: per-wave row sums of the LayerNorm
  int* idbuf0 = reinterpret_cast<int*>(lnbuf + 2 * NT * 16);  // [2][18]: destination of row0 - 1, the tile's rows, row0 + 16
  const int f0 = 16 * w + 4 * g;                        // first of this lane's four features

  for (int u = tid; u < 16 * ldx; u += NTV) xbuf[u] = 0.f;  // the padding columns [in_dim, k0pad) stay zero for good
  if constexpr (AGG) {
    for (int e = 2 * a.num_tiles + (int)blockIdx.x * NTV + tid; e < a.fix_len; e += (int)gridDim.x * NTV) d.agg_fix[e] = -1;
  }
  __syncthreads();

  // weight rows [16w, 16w + 16) of Linear l, k columns [kbase, kbase + CK), through a bounds-checked window that ends with the
  // matrix (rows past it read as 0); every load is unconditional and addressed by one lane offset + an immediate.  Columns
  // in [kin, kpad) of a matrix whose width is not a multiple of 16 are cleared by selects (launcher: 16-B aligned rows).
  auto wload = [&](f32x4 (&wr)[NKB], int l, int kbase) {
    const int ldw = ldw_of(d, l);
    const int kin = d.in_dim[l];
    const __amdgpu_buffer_rsrc_t win = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(d.weight[l]), 0, ((d.out_dim[l] - 1) * ldw + kin) * 4, 0x00020000);
    const uint32_t off0 = (uint32_t)((16 * w + j) * ldw + kbase + 4 * g) * 4u;
#pragma unroll
    for (int cb = 0; cb < NKB; ++cb) wr[cb] = window_load(win, off0 + (uint32_t)(cb * 64));
    if (kbase + CK > kin) {  // wave-uniform: a partial chunk
#pragma unroll
      for (int cb = 0; cb < NKB; ++cb) {
        const int k = kbase + 16 * cb + 4 * g;
        wr[cb].x = k + 0 < kin ? wr[cb].x : 0.f; wr[cb].y = k + 1 < kin ? wr[cb].y : 0.f;
        wr[cb].z = k + 2 < kin ? wr[cb].z : 0.f; wr[cb].w = k + 3 < kin ? wr[cb].w : 0.f;
      }
    }
  };
  // the first Linear of an encoder: in_dim <= 16 with rows that are not 16-B pieces (nn.Linear(3, H)): one k block, by dwords
  auto wload_narrow = [&](f32x4 (&wr)[NKB]) {
    const int ldw = ldw_of(d, 0);
    const int kin = d.in_dim[0];
    const __amdgpu_buffer_rsrc_t win = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(d.weight[0]), 0, ((d.out_dim[0] - 1) * ldw + kin) * 4, 0x00020000);
    const uint32_t off0 = (uint32_t)((16 * w + j) * ldw + 4 * g) * 4u;
    const int k = 4 * g;
    f32x4 v;
    v.x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(win, off0 + 0u, 0, 0));
    v.y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(win, off0 + 4u, 0, 0));
    v.z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(win, off0 + 8u, 0, 0));
    v.w = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(win, off0 + 12u, 0, 0));
    v.x = k + 0 < kin ? v.x : 0.f; v.y = k + 1 < kin ? v.y : 0.f;
    v.z = k + 2 < kin ? v.z : 0.f; v.w = k + 3 < kin ? v.w : 0.f;
    wr[0] = v;
#pragma unroll
    for (int cb = 1; cb < NKB; ++cb) wr[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  // acc0 / acc1 += W chunk x rows: k blocks [0, nkb) of the chunk, `src` = &tile[j][kbase + 4g]
  auto mma = [&](auto nb_, f32x4& acc0, f32x4& acc1, const f32x4 (&wr)[NKB], const float* src, int nkb) {
    constexpr int NB = decltype(nb_)::value;
    f32x4 b[NB];
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) b[cb] = *reinterpret_cast<const f32x4*>(src + 16 * cb);  // in bounds: tiles hold whole chunks
#pragma unroll
    for (int cb = 0; cb < NB; cb += 2) {
      if (cb < nkb) {  // wave-uniform: k blocks past the layer's width are skipped two at a time (zeros x zeros otherwise)
        acc0 = mfma16(wr[cb].x, b[cb].x, acc0);
        acc1 = mfma16(wr[cb + 1].x, b[cb + 1].x, acc1);
        acc0 = mfma16(wr[cb].y, b[cb].y, acc0);
        acc1 = mfma16(wr[cb + 1].y, b[cb + 1].y, acc1);
        acc0 = mfma16(wr[cb].z, b[cb].z, acc0);
        acc1 = mfma16(wr[cb + 1].z, b[cb + 1].z, acc1);
        acc0 = mfma16(wr[cb].w, b[cb].w, acc0);
        acc1 = mfma16(wr[cb + 1].w, b[cb + 1].w, acc1);
      }
    }
  };
  // four per-feature parameters (bias, gamma, beta) of this lane; features past `n` read as 0
  auto param4 = [&](const float* p, int n) -> f32x4 {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (p) {
      v.x = f0 + 0 < n ? p[f0 + 0] : 0.f; v.y = f0 + 1 < n ? p[f0 + 1] : 0.f;
      v.z = f0 + 2 < n ? p[f0 + 2] : 0.f; v.w = f0 + 3 < n ? p[f0 + 3] : 0.f;
    }
    return v;
  };

  int it = 0;
  for (int t = blockIdx.x; t < a.num_tiles; t += gridDim.x, ++it) {
    int* idbuf = idbuf0 + (it & 1) * 18;  // by tile parity: the walkers of the previous tile may still be reading theirs
    const int row0 = t * R16;
    const int myrow = row0 + j < rows ? row0 + j : rows - 1;  // clamped: rows past the end are computed and dropped
    const bool row_live = row0 + j < rows;

    f32x4 wc[NKB], wn[NKB];
    if (a.narrow_w0) wload_narrow(wc);
    else wload(wc, 0, 0);

    // W-split addends: rows already in the hidden width, summed straight into the accumulator layout
    f32x4 addv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < GNC_MAX_SEGMENTS; ++s) {
      if (s < d.num_segments && d.seg[s].mode == GNC_SEG_ADD) {
        const int32_t* ip = d.seg[s].index;
        const int id = ip ? ip[myrow] : myrow;
        const bool ok = (ip == nullptr || (uint64_t)(int64_t)id < (uint64_t)d.seg[s].table_rows) && f0 < H;
        const f32x4 v = *reinterpret_cast<const f32x4*>(d.seg[s].ptr + (int64_t)(ok ? id : 0) * d.seg[s].ld + (ok ? f0 : 0));
        addv += ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    // the virtual concat's rows: column wcol + c of the tile = column c of the segment
#pragma unroll
    for (int s = 0; s < GNC_MAX_SEGMENTS; ++s) {
      if (s < d.num_segments && d.seg[s].mode == GNC_SEG_MATMUL) {
        const float* base = d.seg[s].ptr;
        const int32_t* ip = d.seg[s].index;
        const int width = d.seg[s].width, ld = d.seg[s].ld, wcol = d.seg[s].wcol;
        if (ld % 4 == 0 && width % 4 == 0 && wcol % 4 == 0 && al16d(base)) {
          const int w4 = width >> 2;
          for (int u = tid; u < 16 * w4; u += NTV) {
            const int r = u / w4, c4 = u - r * w4;
            const int rr = row0 + r < rows ? row0 + r : rows - 1;
            const int id = ip ? ip[rr] : rr;
            const bool ok = ip == nullptr || (uint64_t)(int64_t)id < (uint64_t)d.seg[s].table_rows;
            const f32x4 v = *reinterpret_cast<const f32x4*>(base + (int64_t)(ok ? id : 0) * ld + c4 * 4);
            *reinterpret_cast<f32x4*>(xbuf + r * ldx + wcol + c4 * 4) = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
          }
        } else {
          for (int u = tid; u < 16 * width; u += NTV) {
            const int r = u / width, c = u - r * width;
            const int rr = row0 + r < rows ? row0 + r : rows - 1;
            const int id = ip ? ip[rr] : rr;
            const bool ok = ip == nullptr || (uint64_t)(int64_t)id < (uint64_t)d.seg[s].table_rows;
            const float v = base[(int64_t)(ok ? id : 0) * ld + c];
            xbuf[r * ldx + wcol + c] = ok ? v : 0.f;
          }
        }
      }
    }
    if constexpr (AGG) {
      if (tid < 18) {
        const int e = row0 - 1 + tid;
        idbuf[tid] = (e >= 0 && e < rows) ? d.agg_index[e] : -1;
      }
    }
    __syncthreads();

    // ---------------------------------------------------------------- first Linear: chunks of the virtual concat
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int kb = 0; kb < a.k0pad; kb += CK) {
      const bool lastc = kb + CK >= a.k0pad;
      if (!lastc) wload(wn, 0, kb + CK);
      else if (L > 1) wload(wn, 1, 0);
      const int nkb = (a.k0pad - kb) >> 4;
      mma(std::integral_constant<int, NKB>{}, acc0, acc1, wc, xbuf + j * ldx + kb + 4 * g, nkb < NKB ? nkb : NKB);
#pragma unroll
      for (int cb = 0; cb < NKB; ++cb) wc[cb] = wn[cb];
    }
    f32x4 o = (acc0 + acc1) + param4(d.bias[0], d.out_dim[0]) + addv;

    // ---------------------------------------------------------------- hidden and output Linears
    for (int l = 1; l < L; ++l) {
      o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
      float* hb = hbuf + ((l - 1) & 1) * 16 * LDH;
      *reinterpret_cast<f32x4*>(hb + j * LDH + f0) = o;
      if constexpr (SAVE) {
        if (row_live && f0 < H) *reinterpret_cast<f32x4*>(d.save_act[l - 1] + (int64_t)(row0 + j) * H + f0) = o;
      }
      __syncthreads();
      if (l + 1 < L) wload(wn, l + 1, 0);
      acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
      acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
      if (16 * w < d.out_dim[l]) mma(std::integral_constant<int, NT>{}, acc0, acc1, wc, hb + j * LDH + 4 * g, (d.in_dim[l] + 15) >> 4);
      o = (acc0 + acc1) + param4(d.bias[l], d.out_dim[l]);
#pragma unroll
      for (int cb = 0; cb < NKB; ++cb) wc[cb] = wn[cb];
    }

    // ---------------------------------------------------------------- LayerNorm across the waves (two-pass)
    if (d.ln_gamma) {
      const float inv_n = 1.f / (float)od;
      float s = add_quarters((o.x + o.y) + (o.z + o.w));  // features >= od hold exact zeros (zero weight rows, zero bias)
      if (g == 0) lnbuf[w * 16 + j] = s;
      __syncthreads();
      s = 0.f;
#pragma unroll
      for (int ww = 0; ww < NT; ++ww) s += lnbuf[ww * 16 + j];
      const float mean = s * inv_n;
      o.x = f0 + 0 < od ? o.x - mean : 0.f; o.y = f0 + 1 < od ? o.y - mean : 0.f;
      o.z = f0 + 2 < od ? o.z - mean : 0.f; o.w = f0 + 3 < od ? o.w - mean : 0.f;
      float v = add_quarters((o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w));
      if (g == 0) lnbuf[NT * 16 + w * 16 + j] = v;
      __syncthreads();
      v = 0.f;
#pragma unroll
      for (int ww = 0; ww < NT; ++ww) v += lnbuf[NT * 16 + ww * 16 + j];
      const float rstd = __frsqrt_rn(v * inv_n + d.ln_eps);
      const f32x4 gm = param4(d.ln_gamma, od), bt = param4(d.ln_beta, od);
      o.x = o.x * (gm.x * rstd) + bt.x; o.y = o.y * (gm.y * rstd) + bt.y;
      o.z = o.z * (gm.z * rstd) + bt.z; o.w = o.w * (gm.w * rstd) + bt.w;
    }

    // ---------------------------------------------------------------- residual, output rows
    if (d.residual && f0 < od) o += *reinterpret_cast<const f32x4*>(d.residual + (int64_t)myrow * d.ld_residual + f0);
    if (row_live && f0 < od) {
      float* op = d.out + (int64_t)(row0 + j) * d.ld_out + f0;
      if (d.ld_out % 4 == 0 && od % 4 == 0 && al16d(d.out)) {
        *reinterpret_cast<f32x4*>(op) = o;
      } else {
        op[0] = o.x;
        if (f0 + 1 < od) op[1] = o.y;
        if (f0 + 2 < od) op[2] = o.z;
        if (f0 + 3 < od) op[3] = o.w;
      }
    }

    if constexpr (AGG) {
      float* ob = hbuf + ((L - 1) & 1) * 16 * LDH;  // idle since the previous layer but one
      *reinterpret_cast<f32x4*>(ob + j * LDH + f0) = o;
      __syncthreads();
      if (tid < od) {
        const int valid = rows - row0 < R16 ? rows - row0 : R16;
        const int first = idbuf[1];
        const bool cut_first = idbuf[0] == first;
        float acc = 0.f;
        int cur = first;
        bool cur_is_first = true;
        for (int r = 0; r < valid; ++r) {
          const int id = idbuf[1 + r];
          if (id != cur) {
            if (!(cur_is_first && cut_first)) d.agg_out[(int64_t)cur * d.ld_agg + tid] = acc;
            cur = id;
            cur_is_first = false;
            acc = 0.f;
          }
          acc += ob[r * LDH + tid];
        }
        const bool cut_last = idbuf[1 + valid] == cur;
        if (!cut_last && !(cur_is_first && cut_first)) d.agg_out[(int64_t)cur * d.ld_agg + tid] = acc;
        if (tid == 0) {
          d.agg_fix[2 * t] = cut_first ? first : -1;
          d.agg_fix[2 * t + 1] = cut_last ? cur : -1;
        }
      }
    }
  }
}

template <int NT, bool AGG, bool SAVE>
int launch_col(const gnc_mlp_desc_t& d, const ColArgs& a, hipStream_t stream) {
  const size_t smem = ((size_t)16 * a.ldx + 2 * 16 * (NT * 16 + 4) + 2 * NT * 16 + 48) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    int rc = gnc::check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_col16_kernel<NT, AGG, SAVE>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                            "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    if (rc) return rc;
    attr_set = true;
  }
  const int cap = 2 * gnc::num_cu();
  const int grid = a.num_tiles < cap ? a.num_tiles : cap;
  mlp_col16_kernel<NT, AGG, SAVE><<<dim3((unsigned)grid), dim3(NT * 64), smem, stream>>>(d, a);
  return gnc::check_launch("mlp_col16_kernel");
}

template <int NT>
int launch_col_nt(const gnc_mlp_desc_t& d, const ColArgs& a, bool agg, bool save, hipStream_t stream) {
  if (agg) return save ? launch_col<NT, true, true>(d, a, stream) : launch_col<NT, true, false>(d, a, stream);
  return save ? launch_col<NT, false, true>(d, a, stream) : launch_col<NT, false, false>(d, a, stream);
}

bool al16h(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

// Rows up to which the column-split kernel serves a description (GNC_COL16_MAX_ROWS overrides; 0 switches it off).
int64_t gnc_mlp::col16_max_rows() {
  static const int64_t v = [] {
    const char* e = getenv("GNC_COL16_MAX_ROWS");
    return e ? (int64_t)atoll(e) : (int64_t)2 * R16 * gnc::num_cu();
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
  if (d.in_dim[0] > MAX_K0) return GNC_OK;
  bool narrow_w0 = false;
  for (int l = 0; l < L; ++l) {
    const int ldw = ldw_of(d, l);
    if ((int64_t)d.out_dim[l] * ldw * 4 > 0x7fffffffll) return GNC_OK;
    if (ldw % 4 == 0 && al16h(d.weight[l])) continue;  // rows of 16-B pieces (a width that is not a multiple of 4 is masked)
    if (l == 0 && d.in_dim[0] <= 16) narrow_w0 = true;
    else return GNC_OK;
  }
  for (int s = 0; s < d.num_segments; ++s) {
    const gnc_mlp_segment_t& sg = d.seg[s];
    if (sg.mode == GNC_SEG_ADD) {
      if (sg.width != H || H % 4 != 0 || sg.ld % 4 != 0 || !al16h(sg.ptr)) return GNC_OK;
    } else if (sg.wcol < 0 || sg.wcol + sg.width > d.in_dim[0]) {
      return GNC_OK;
    }
  }
  if (d.residual && (d.ld_residual % 4 != 0 || od % 4 != 0 || !al16h(d.residual))) return GNC_OK;
  const bool save = d.save_act[0] != nullptr;
  if (save) {
    if (L < 2 || H % 4 != 0) return GNC_OK;
    for (int l = 0; l < L - 1; ++l)
      if (!d.save_act[l] || !al16h(d.save_act[l])) return GNC_OK;
  }
  ColArgs a;
  a.num_tiles = (int)gnc::ceil_div(d.rows, (int64_t)R16);
  a.k0pad = (d.in_dim[0] + 15) & ~15;
  const int wmax = H > od ? H : od;
  const int nt = wmax <= 64 ? 4 : 8;
  const int k0c = (a.k0pad + CK - 1) / CK * CK;  // whole chunks: the B fragments of a chunk are read unconditionally
  a.ldx = k0c + 4;
  a.fix_len = gnc_mlp_agg_fix_len();
  a.narrow_w0 = narrow_w0 ? 1 : 0;
  const bool agg = d.agg_out != nullptr;
  if (agg && (!d.agg_index || !d.agg_fix || d.ld_agg < od || 2 * a.num_tiles > a.fix_len)) return GNC_OK;
  *launched = true;
  if (probe_only) return GNC_OK;
  return nt == 4 ? launch_col_nt<4>(d, a, agg, save, stream) : launch_col_nt<8>(d, a, agg, save, stream);
}
