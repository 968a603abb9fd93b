// K4: fused MLP (concat/gather prologue -> Linear/act chain on fp32 MFMA -> LayerNorm ->
// residual) for the GraphNet encoders, edge/node processors and decoder.
//
// Replaces models/MLP.py:45-47 plus the cat/residual around it (models/GNN.py:58-62, :100-102).
//
// Formulation.  Everything is computed TRANSPOSED:  Y^T[feature, row] = W[feature, k] * X^T[k, row],
// i.e. the nn.Linear weight (already [out, in] row-major) is the MFMA A operand and the
// activations are the B operand.  With v_mfma_f32_32x32x2_f32 the accumulator of a 32x32 tile
// keeps the data row on the LANE (lane & 31) and the feature index in the REGISTER
// (feature = (r&3) + 8*(r>>2) + 4*(lane>>5)).  A following Linear sums over that feature
// index, so accumulator register r of one layer is, as it stands, the B operand of MFMA
// step r of the next layer: the hidden activations never leave registers - no LDS
// transpose, no HBM round trip between the Linear layers.  The matching A operand for step
// (q = r>>2, s = r&3) is W[n][8q + 4h + s], four consecutive floats = one ds_read_b128.
//
// This file: the STREAMING variant (any width up to 256).  Workgroup = 4 waves (one per
// SIMD); each wave owns 32 data rows of a 128-row tile and ALL features (HT accumulator tiles
// of 32 features).  Weights stream L2 -> LDS in [features][64 k] chunks shared by the four
// waves (2 barriers per chunk); the first layer's input rows (gathered by index when the
// segment has one) are staged per wave into a private [32][64] LDS tile with whole-row 16-B
// loads.  LDS rows are padded to 68 floats (4*odd) which makes every ds_read_b128 fragment
// read conflict-free.  mlp_resident.hip holds the variant for widths whose weights fit in LDS.
#include <stdlib.h>

#include "mlp_device.h"

using namespace gnc_mlp;

namespace {

constexpr int WAVES = 4;
constexpr int NT = WAVES * 64;
constexpr int RPB = RPW * WAVES;  // data rows per workgroup tile

// Wave-private: abuf[j][0 .. ncol4*4) = seg.ptr[row(j)][c0 + ..] for the wave's 32 rows.
__device__ __forceinline__ void stage_rows(float* abuf, const gnc_mlp_segment_t& seg, int64_t row0, int64_t rows,
                                           int c0, int ncol4, bool vec_ok, int lane) {
  const int c4 = lane & 15;
  const int rs = lane >> 4;
  if (c4 < ncol4) {
#pragma unroll
    for (int p = 0; p < RPW / 4; ++p) {
      const int j = p * 4 + rs;
      const int64_t r = row0 + j;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (r < rows) {
        const int64_t tr = seg.index ? (int64_t)seg.index[r] : r;
        // an id outside the stated table reads zeros, as in the window-addressed kernels (never a fault)
        if (!seg.index || (uint64_t)tr < (uint64_t)seg.table_rows)
          v = load4_guarded(seg.ptr + tr * seg.ld + c0 + c4 * 4, c0 + c4 * 4, seg.width, vec_ok);
      }
      *reinterpret_cast<f32x4*>(abuf + j * LDSW + c4 * 4) = v;
    }
  }
}

// dst[TO] += W_l * src[TI]   (src = previous layer's activations, in registers)
template <int TI, int TO>
__device__ __forceinline__ void linear_from_regs(f32x16 (&dst)[TO], const f32x16 (&src)[TI], float* wbuf,
                                                 const float* __restrict__ W, int ldw, int in_dim, int out_dim,
                                                 bool vec_ok, int tid, int i, int h) {
  constexpr int NCH = (TI + 1) / 2;  // 64-wide k chunks covering TI tiles
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if (c * KC < in_dim) {  // workgroup-uniform
      stage_weights<TO * 32, NT>(wbuf, W, ldw, out_dim, c * KC, in_dim, 16, vec_ok, tid);
      __syncthreads();
      mma_chunk_from_regs<TI, TO>(dst, src, wbuf, c, in_dim, i, h);
      __syncthreads();
    }
  }
}

template <int HT, int OT>
__global__ __launch_bounds__(NT) void mlp_fused_kernel(const gnc_mlp_desc_t d, const int64_t num_tiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int WT = HT > OT ? HT : OT;
  constexpr int PSTRIDE = WT * 32;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int i = lane & 31;  // feature row of the A operand == data row of the B operand / accumulator column
  const int h = lane >> 5;
  float* wbuf = lds;
  float* abuf = lds + (WT * 32 + wave * RPW) * LDSW;
  float* pbuf = lds + (WT * 32 + RPB) * LDSW;  // [num_linear + 2][PSTRIDE]: biases, gamma, beta (zero padded)
  const int L = d.num_linear;
  const int out_dim = d.out_dim[L - 1];

  stage_params<NT>(pbuf, d, PSTRIDE, tid);
  __syncthreads();

  unsigned wvec = 0;  // bit l: layer l's weight rows can be read with 16-B loads
  for (int l = 0; l < L; ++l)
    if ((ldw_of(d, l) % 4 == 0) && ((reinterpret_cast<uintptr_t>(d.weight[l]) & 15u) == 0)) wvec |= 1u << l;

  const bool res_staged = residual_is_staged(d, out_dim);

  for (int64_t tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
    const int64_t row0 = tile * RPB + wave * RPW;

    // ------------------------------------------------------------------ layer 0
    f32x16 hid[HT];
    init_bias<HT>(hid, pbuf, h);
    {
      for (int s = 0; s < d.num_segments; ++s) {
        const gnc_mlp_segment_t seg = d.seg[s];
        const bool avec = (seg.ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(seg.ptr) & 15u) == 0);
        if (seg.mode == GNC_SEG_ADD) {  // rows already in the hidden width: gather, add (wave-private, no barrier)
          for (int c0 = 0; c0 < seg.width; c0 += KC) {
            const int kc = seg.width - c0 < KC ? seg.width - c0 : KC;
            wave_lds_fence();
            stage_rows(abuf, seg, row0, d.rows, c0, 16, avec, lane);
            (void)kc;
            wave_lds_fence();
            add_rows_from_lds<HT>(hid, abuf, c0 / 32, i, h);
          }
          wave_lds_fence();
          continue;
        }
        const int koff = seg.wcol;
        const bool w0vec = (wvec & 1u) && (koff % 4 == 0);
        for (int c0 = 0; c0 < seg.width; c0 += KC) {
          const int kc = seg.width - c0 < KC ? seg.width - c0 : KC;
          const int kc8 = (kc + 7) >> 3;
          stage_weights<HT * 32, NT>(wbuf, d.weight[0], ldw_of(d, 0), d.out_dim[0], koff + c0, koff + seg.width,
                                     kc8 * 2, w0vec, tid);
          stage_rows(abuf, seg, row0, d.rows, c0, kc8 * 2, avec, lane);
          __syncthreads();
          mma_chunk_from_lds<HT>(hid, abuf, wbuf, kc8, i, h);
          __syncthreads();
        }
      }
    }
    if (L == 1) {  // plain projection: the first Linear is also the last
      if (d.ln_gamma) layer_norm_tiles<HT>(hid, pbuf + L * PSTRIDE, pbuf + (L + 1) * PSTRIDE, out_dim, d.ln_eps, h);
      store_tiles<HT>(hid, abuf, d, row0, out_dim, lane, i, h, res_staged);
      continue;
    }
    activate_tiles<HT>(hid, d.activation, d.act_param);

    // ------------------------------------------------------------------ hidden layers 1 .. L-2
    for (int l = 1; l < L - 1; ++l) {
      f32x16 nxt[HT];
      init_bias<HT>(nxt, pbuf + l * PSTRIDE, h);
      linear_from_regs<HT, HT>(nxt, hid, wbuf, d.weight[l], ldw_of(d, l), d.in_dim[l], d.out_dim[l], (wvec >> l) & 1u, tid,
                               i, h);
      activate_tiles<HT>(nxt, d.activation, d.act_param);
#pragma unroll
      for (int t = 0; t < HT; ++t) hid[t] = nxt[t];
    }

    // ------------------------------------------------------------------ last Linear, LayerNorm, store
    f32x16 o[OT];
    init_bias<OT>(o, pbuf + (L - 1) * PSTRIDE, h);
    linear_from_regs<HT, OT>(o, hid, wbuf, d.weight[L - 1], ldw_of(d, L - 1), d.in_dim[L - 1], out_dim,
                             (wvec >> (L - 1)) & 1u, tid, i, h);
    if (d.ln_gamma) layer_norm_tiles<OT>(o, pbuf + L * PSTRIDE, pbuf + (L + 1) * PSTRIDE, out_dim, d.ln_eps, h);
    store_tiles<OT>(o, abuf, d, row0, out_dim, lane, i, h, res_staged);
  }
}

size_t lds_bytes(int WT, int L) { return ((size_t)(WT * 32 + RPB) * LDSW + (size_t)(L + 2) * WT * 32) * sizeof(float); }

template <int HT, int OT>
int launch(const gnc_mlp_desc_t& d, hipStream_t stream) {
  constexpr int WT = HT > OT ? HT : OT;
  const size_t smem = lds_bytes(WT, d.num_linear);
  static bool attr_set = false;  // per instantiation; set once, before the first launch
  if (!attr_set) {
    int rc = gnc::check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fused_kernel<HT, OT>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                            "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    if (rc) return rc;
    attr_set = true;
  }
  const int64_t num_tiles = gnc::ceil_div(d.rows, RPB);
  const int per_cu = (int)((160 * 1024) / smem) < 1 ? 1 : (int)((160 * 1024) / smem);
  int64_t grid = (int64_t)gnc::num_cu() * (per_cu > 3 ? 3 : per_cu);
  if (grid > num_tiles) grid = num_tiles;
  mlp_fused_kernel<HT, OT><<<dim3((unsigned)grid), dim3(NT), smem, stream>>>(d, num_tiles);
  return gnc::check_launch("mlp_fused_kernel");
}

}  // namespace

int gnc_mlp::validate_desc(const gnc_mlp_desc_t* d, bool check_ptrs) {
  if (!d) { gnc::set_error("gnc_mlp: null descriptor"); return GNC_ERR_INVALID_ARGUMENT; }
  if (d->num_segments < 1 || d->num_segments > GNC_MAX_SEGMENTS || d->num_linear < 1 || d->num_linear > GNC_MAX_LINEAR) {
    gnc::set_error("gnc_mlp: need 1..%d segments and 1..%d Linear layers (got %d, %d)", GNC_MAX_SEGMENTS,
                   GNC_MAX_LINEAR, d->num_segments, d->num_linear);
    return GNC_ERR_UNSUPPORTED;
  }
  if (d->activation < GNC_ACT_RELU || d->activation > GNC_ACT_ELU) {
    gnc::set_error("gnc_mlp: unknown activation id %d", d->activation);
    return GNC_ERR_UNSUPPORTED;
  }
  int64_t in0 = 0;
  for (int s = 0; s < d->num_segments; ++s) {
    if (d->seg[s].width < 1 || d->seg[s].ld < d->seg[s].width) {
      gnc::set_error("gnc_mlp: segment %d has width %d, ld %d", s, d->seg[s].width, d->seg[s].ld);
      return GNC_ERR_INVALID_ARGUMENT;
    }
    if (check_ptrs && d->rows > 0 && !d->seg[s].ptr) { gnc::set_error("gnc_mlp: segment %d null", s); return GNC_ERR_INVALID_ARGUMENT; }
    if (d->seg[s].index && (d->seg[s].table_rows < 0 || (d->seg[s].table_rows == 0 && d->rows > 0))) {
      gnc::set_error("gnc_mlp: gathered segment %d must state its table (table_rows = %lld)", s, (long long)d->seg[s].table_rows);
      return GNC_ERR_INVALID_ARGUMENT;
    }
    if (d->seg[s].mode == GNC_SEG_ADD) {
      if (d->seg[s].width != d->out_dim[0]) {
        gnc::set_error("gnc_mlp: additive segment %d has width %d, first Linear has %d outputs", s, d->seg[s].width, d->out_dim[0]);
        return GNC_ERR_INVALID_ARGUMENT;
      }
      continue;
    }
    if (d->seg[s].mode != GNC_SEG_MATMUL) { gnc::set_error("gnc_mlp: segment %d has unknown mode %d", s, d->seg[s].mode); return GNC_ERR_INVALID_ARGUMENT; }
    if (d->seg[s].wcol < 0 || d->seg[s].wcol + d->seg[s].width > d->in_dim[0]) {
      gnc::set_error("gnc_mlp: segment %d covers weight columns [%d, %d) outside [0, %d)", s, d->seg[s].wcol, d->seg[s].wcol + d->seg[s].width, d->in_dim[0]);
      return GNC_ERR_INVALID_ARGUMENT;
    }
    in0 += d->seg[s].width;
  }
  if (in0 != d->in_dim[0]) {
    gnc::set_error("gnc_mlp: matmul segment widths sum to %lld but in_dim[0] is %d", (long long)in0, d->in_dim[0]);
    return GNC_ERR_INVALID_ARGUMENT;
  }
  const int H = d->out_dim[0];
  for (int l = 0; l < d->num_linear; ++l) {
    if (d->in_dim[l] < 1 || d->out_dim[l] < 1) { gnc::set_error("gnc_mlp: layer %d has empty dims", l); return GNC_ERR_INVALID_ARGUMENT; }
    if (d->ld_weight[l] != 0 && d->ld_weight[l] < d->in_dim[l]) { gnc::set_error("gnc_mlp: ld_weight[%d] < in_dim", l); return GNC_ERR_INVALID_ARGUMENT; }
    if (l > 0 && d->in_dim[l] != d->out_dim[l - 1]) {
      gnc::set_error("gnc_mlp: layer %d in_dim %d != layer %d out_dim %d", l, d->in_dim[l], l - 1, d->out_dim[l - 1]);
      return GNC_ERR_INVALID_ARGUMENT;
    }
    if (l < d->num_linear - 1 && d->out_dim[l] != H) {
      gnc::set_error("gnc_mlp: hidden layers must share one width (layer %d: %d vs %d)", l, d->out_dim[l], H);
      return GNC_ERR_UNSUPPORTED;
    }
    if (d->out_dim[l] > 256) { gnc::set_error("gnc_mlp: width %d > 256 not instantiated", d->out_dim[l]); return GNC_ERR_UNSUPPORTED; }
    if (check_ptrs && !d->weight[l]) { gnc::set_error("gnc_mlp: weight %d null", l); return GNC_ERR_INVALID_ARGUMENT; }
  }
  const int od = d->out_dim[d->num_linear - 1];
  if (d->ld_out < od || (d->residual && d->ld_residual < od)) {
    gnc::set_error("gnc_mlp: ld_out/ld_residual smaller than out_dim %d", od);
    return GNC_ERR_INVALID_ARGUMENT;
  }
  if (d->rows < 0) { gnc::set_error("gnc_mlp: negative rows"); return GNC_ERR_INVALID_ARGUMENT; }
  if (check_ptrs && d->rows > 0 && !d->out) { gnc::set_error("gnc_mlp: out null"); return GNC_ERR_INVALID_ARGUMENT; }
  return GNC_OK;
}

extern "C" size_t gnc_sizeof_mlp_desc(void) { return sizeof(gnc_mlp_desc_t); }

extern "C" int gnc_mlp_supported(const gnc_mlp_desc_t* desc) { return validate_desc(desc, false); }

// shape-only answer for the fused aggregation epilogue: would the weights-resident kernel take this description
// with agg_out set?  (pointer fields are used for alignment / aliasing tests only, never dereferenced)
extern "C" int gnc_mlp_agg_supported(const gnc_mlp_desc_t* desc) {
  int rc = validate_desc(desc, false);
  if (rc) return rc;
  const int L = desc->num_linear;
  int T = tiles_for(desc->out_dim[0]);
  const int od = desc->out_dim[L - 1];
  const bool narrow_out = od <= 32;
  if (!narrow_out && tiles_for(od) > T) T = tiles_for(od);
  gnc_mlp_desc_t probe = *desc;
  int32_t dummy_i = 0;
  float dummy_f = 0.f;
  if (!probe.agg_out) probe.agg_out = &dummy_f;
  if (!probe.agg_index) probe.agg_index = &dummy_i;
  if (!probe.agg_fix) probe.agg_fix = &dummy_i;
  if (probe.ld_agg < od) probe.ld_agg = od;
  bool ok = false;
  {  // small batches: the column-split kernel serves the launch WITHOUT the epilogue, and K1 on its rows afterwards
     // (scatter_sum_csr_small) is cheaper than the epilogue's fix-up + zero-fill launches
    gnc_mlp_desc_t plain = *desc;
    plain.agg_out = nullptr;
    bool small = false;
    rc = launch_col16(plain, nullptr, &small, true);
    if (rc) return rc;
    if (small) {
      gnc::set_error("gnc_mlp_agg_supported: a small batch: run gnc_scatter_sum_csr_f32 on the output rows");
      return GNC_ERR_UNSUPPORTED;
    }
  }
  rc = launch_resident(probe, T, narrow_out, nullptr, &ok, true);
  if (rc) return rc;
  if (!ok && T == 4) {  // 65..128 features: the streaming kernel carries the epilogue too
    rc = launch_stream(probe, T, narrow_out, nullptr, &ok, true);
    if (rc) return rc;
  }
  if (!ok && T == 8) {  // 129..256 features: the 16-row streaming kernel
    rc = launch_stream16(probe, nullptr, &ok, true);
    if (rc) return rc;
  }
  if (!ok) {
    gnc::set_error("gnc_mlp_agg_supported: needs the weights-resident W-split shape (1 MATMUL + 2 ADD segments, residual = "
                   "the MATMUL segment, widths 33..64) or a 65..128-wide description of the streaming kernel");
    return GNC_ERR_UNSUPPORTED;
  }
  return GNC_OK;
}

// shape-only answer for the training forward's saved post-activations: the weights-resident kernel writes them
namespace {
// Small batches at 65..128 features (one ~1000-node graph per call, main.py:60): a 32-row wave tile is 768 dependent
// v_mfma_f32_32x32x2 steps (49 k cycles, 20 us) whatever the batch; 16-row tiles on v_mfma_f32_16x16x4 in 2-wave
// workgroups halve that chain and spread the rows over twice as many SIMDs.  GNC_NO_SMALL16=1 switches back for A/B runs.
bool small16(const gnc_mlp_desc_t& d, int T) {
  static const bool off = getenv("GNC_NO_SMALL16") != nullptr;
  return !off && T == 4 && d.rows <= (int64_t)2 * 16 * gnc::num_cu();
}
}  // namespace

extern "C" int gnc_mlp_save_act_supported(const gnc_mlp_desc_t* desc) {
  int rc = validate_desc(desc, false);
  if (rc) return rc;
  const int L = desc->num_linear;
  int T = tiles_for(desc->out_dim[0]);
  const int od = desc->out_dim[L - 1];
  const bool narrow_out = od <= 32;
  if (!narrow_out && tiles_for(od) > T) T = tiles_for(od);
  gnc_mlp_desc_t probe = *desc;
  alignas(16) static float dummy_f[4];
  for (int l = 0; l < L - 1; ++l)
    if (!probe.save_act[l]) probe.save_act[l] = dummy_f;
  bool ok = false;
  if (L >= 2) {  // the dispatch chain of gnc_mlp_forward_f32: column split, resident, 16-row streaming (129..256), 32-row streaming
    rc = launch_col16(probe, nullptr, &ok, true);
    if (rc) return rc;
    if (!ok) rc = launch_resident(probe, T, narrow_out, nullptr, &ok, true);
    if (rc) return rc;
    static const bool s16_128 = getenv("GNC_STREAM16_D128") != nullptr;
    if (!ok && (T == 8 || (T == 4 && s16_128 && !probe.agg_out) || small16(probe, T))) {
      rc = launch_stream16(probe, nullptr, &ok, true);
      if (rc) return rc;
    }
    if (!ok) {
      rc = launch_stream(probe, T, narrow_out, nullptr, &ok, true);
      if (rc) return rc;
    }
  }
  if (!ok) {
    gnc::set_error("gnc_mlp_save_act_supported: needs >= 2 Linear layers with ReLU, hidden widths that are multiples of 4 "
                   "(at most 256) and aligned tables (the generic kernel does not write them)");
    return GNC_ERR_UNSUPPORTED;
  }
  return GNC_OK;
}

// ABI 19: 0 when the kernel that will serve this description reads its operands where they lie: WEIGHT matrices of any row
// stride and alignment (nn.Linear(3, H).weight is [H, 3]) - the column-split kernel and the weights-resident kernel, which
// stages every matrix into LDS once per workgroup with guarded scalar loads where 16-B pieces are not possible - and, as the
// only segment, a contiguous [rows, 3] table (the weights-resident kernel's 12-B row loads).  The streaming kernels want
// rows of 16-B pieces: for them the caller hands over zero-padded copies.
extern "C" int gnc_mlp_operands_in_place_supported(const gnc_mlp_desc_t* desc) {
  int rc = validate_desc(desc, false);
  if (rc) return rc;
  const int L = desc->num_linear;
  int T = tiles_for(desc->out_dim[0]);
  const int od = desc->out_dim[L - 1];
  const bool narrow_out = od <= 32;
  if (!narrow_out && tiles_for(od) > T) T = tiles_for(od);
  bool ok = false;
  if (desc->rows > 0) {
    rc = launch_col16(*desc, nullptr, &ok, true);
    if (rc) return rc;
    if (!ok) rc = launch_resident(*desc, T, narrow_out, nullptr, &ok, true);
    if (rc) return rc;
  }
  if (!ok) {
    gnc::set_error("gnc_mlp_operands_in_place_supported: this description runs on a kernel that reads rows of 16-B pieces");
    return GNC_ERR_UNSUPPORTED;
  }
  return GNC_OK;
}

// K6 prologue (gnc_mlp_desc_t.ef_pos, ABI 19): served by the weights-resident kernel, and not for a small batch (there the
// column-split kernel on a materialised [rows, 3] table is the faster route)
extern "C" int gnc_mlp_edge_features_supported(const gnc_mlp_desc_t* desc) {
  int rc = validate_desc(desc, false);
  if (rc) return rc;
  const int L = desc->num_linear;
  int T = tiles_for(desc->out_dim[0]);
  const int od = desc->out_dim[L - 1];
  const bool narrow_out = od <= 32;
  if (!narrow_out && tiles_for(od) > T) T = tiles_for(od);
  gnc_mlp_desc_t probe = *desc;
  alignas(16) static float dummy_f[4];
  static int32_t dummy_i[1];
  if (!probe.ef_pos) probe.ef_pos = dummy_f;
  if (!probe.ef_src) probe.ef_src = dummy_i;
  if (!probe.ef_dst) probe.ef_dst = dummy_i;
  bool ok = false;
  {
    gnc_mlp_desc_t plain = *desc;
    plain.ef_pos = nullptr;
    if (!plain.seg[0].ptr) plain.seg[0].ptr = dummy_f;
    bool small = false;
    rc = launch_col16(plain, nullptr, &small, true);
    if (rc) return rc;
    if (!small && desc->rows > 0) {
      rc = launch_resident(probe, T, narrow_out, nullptr, &ok, true);
      if (rc) return rc;
    }
  }
  if (!ok) {
    gnc::set_error("gnc_mlp_edge_features_supported: needs the edge encoder's shape on the weights-resident kernel (one computed "
                   "segment of width 3 = space_dim 2, >= 2 Linear layers of widths 33..64, ReLU, no residual / save_act / agg_out) "
                   "and a batch above the small-batch limit");
    return GNC_ERR_UNSUPPORTED;
  }
  return GNC_OK;
}

// shape + alignment answer: 0 when the small-batch (column-split) kernel serves this description exactly as given - it is
// the only kernel that reads tables and weights whose rows are not 16-B pieces (3-column inputs, [H, 3] weights) in place
extern "C" int gnc_mlp_small_batch_supported(const gnc_mlp_desc_t* desc) {
  int rc = validate_desc(desc, false);
  if (rc) return rc;
  bool ok = false;
  if (desc->rows > 0) {
    rc = launch_col16(*desc, nullptr, &ok, true);
    if (rc) return rc;
  }
  if (!ok) {
    gnc::set_error("gnc_mlp_small_batch_supported: not a small-batch description (rows, widths <= 128, ReLU)");
    return GNC_ERR_UNSUPPORTED;
  }
  return GNC_OK;
}

extern "C" int gnc_mlp_forward_f32(const gnc_mlp_desc_t* desc, void* stream_) {
  int rc = validate_desc(desc, true);
  if (rc) return rc;
  if (desc->rows == 0) return GNC_OK;
  hipStream_t stream = (hipStream_t)stream_;
  const int L = desc->num_linear;
  int T = tiles_for(desc->out_dim[0]);
  const int od = desc->out_dim[L - 1];
  const bool narrow_out = od <= 32;
  if (!narrow_out && tiles_for(od) > T) T = tiles_for(od);

  bool launched = false;
  if (desc->ef_pos) {  // K6 prologue (ABI 19): the weights-resident kernel only, and only where the query says so
    rc = gnc_mlp_edge_features_supported(desc);
    if (rc) return rc;
    rc = launch_resident(*desc, T, narrow_out, stream, &launched);
    if (!rc && !launched) {
      gnc::set_error("gnc_mlp_forward_f32: ef_pos given but the weights-resident kernel did not take the launch");
      rc = GNC_ERR_UNSUPPORTED;
    }
    return rc;
  }
  rc = launch_col16(*desc, stream, &launched);  // small batches at 65..128 features: column-split workgroups
  if (rc || launched) return rc;
  rc = launch_resident(*desc, T, narrow_out, stream, &launched);  // weights-resident variant (decides by LDS fit)
  if (rc || launched) return rc;
  if (desc->agg_out && T != 4 && T != 8) {
    gnc::set_error("gnc_mlp_forward_f32: the fused aggregation epilogue is not available for this description "
                   "(gnc_mlp_agg_supported)");
    return GNC_ERR_UNSUPPORTED;
  }
  static const bool s16_128 = getenv("GNC_STREAM16_D128") != nullptr;  // A/B: 16-row kernel also for 65..128 features
  // 129..256 features: 16-row tiles on the 16x16x4 MFMA; 65..128 features: its 2-wave instances for small batches (small16)
  if (T == 8 || (T == 4 && s16_128 && !desc->agg_out) || small16(*desc, T)) {
    rc = launch_stream16(*desc, stream, &launched);
    if (rc || launched) return rc;
    if (desc->agg_out && T == 8) {
      gnc::set_error("gnc_mlp_forward_f32: the fused aggregation epilogue is not available for this description "
                     "(gnc_mlp_agg_supported)");
      return GNC_ERR_UNSUPPORTED;
    }
  }
  rc = launch_stream(*desc, T, narrow_out, stream, &launched);  // wide layers: double-buffered weight stream
  if (rc || launched) return rc;
  if (desc->agg_out) {
    gnc::set_error("gnc_mlp_forward_f32: the fused aggregation epilogue is not available for this description "
                   "(gnc_mlp_agg_supported)");
    return GNC_ERR_UNSUPPORTED;
  }
  if (desc->save_act[0]) {
    gnc::set_error("gnc_mlp_forward_f32: save_act is not available for this description (gnc_mlp_save_act_supported)");
    return GNC_ERR_UNSUPPORTED;
  }

  switch (T * 2 + (narrow_out ? 1 : 0)) {
    case 1 * 2 + 0: case 1 * 2 + 1: return launch<1, 1>(*desc, stream);
    case 2 * 2 + 0: return launch<2, 2>(*desc, stream);
    case 2 * 2 + 1: return launch<2, 1>(*desc, stream);
    case 4 * 2 + 0: return launch<4, 4>(*desc, stream);
    case 4 * 2 + 1: return launch<4, 1>(*desc, stream);
    case 8 * 2 + 0: return launch<8, 8>(*desc, stream);
    case 8 * 2 + 1: return launch<8, 1>(*desc, stream);
  }
  gnc::set_error("gnc_mlp_forward_f32: no kernel for hidden width %d / out width %d", desc->out_dim[0], od);
  return GNC_ERR_UNSUPPORTED;
}
