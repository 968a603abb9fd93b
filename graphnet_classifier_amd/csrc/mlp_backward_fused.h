// K8, fused data + weight-gradient kernel of the MLP backward (widths <= 64, three Linear layers): kernel template, its
// launcher template and the host-side shape test.  Instantiated in two translation units so that they compile side by
// side: mlp_backward_fused_rc.hip (the tile's forward is recomputed from the inputs) and mlp_backward_fused_sv.hip (the
// forward's saved post-activations are read, gnc_mlp_bwd_desc_t.act_given); gnc_mlp_backward_f32 (mlp_backward.hip)
// picks one through the two launch functions declared at the end.
#pragma once
#include <stdlib.h>

#include "mlp_backward_common.h"

// Phase probe (`make probe_bwd`): per-wave shader-clock cycles (s_memtime) per phase of the tile loop, read back by
// tools/profile_bwd.py.  A translation unit that wants it defines GNC_BWD_PROBE_SYMBOL (its own __device__ buffer) before
// including this header; off in the shipped library.
#if defined(GNC_PHASE_PROBE) && defined(GNC_BWD_PROBE_SYMBOL)
#define BPROBE_BEGIN() unsigned long long pr_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pr_tp, pr_tn; \
  const unsigned long long pr_c0 = __builtin_readcyclecounter(), pr_w0 = wall_clock64()
#define BPROBE_TILE() pr_tp = __builtin_readcyclecounter()
#define BPROBE(k) do { pr_tn = __builtin_readcyclecounter(); pr_acc[k] += pr_tn - pr_tp; pr_tp = pr_tn; } while (0)
#define BPROBE_END() do { if ((threadIdx.x & 63) == 0) {                                                        \
    unsigned long long* o = GNC_BWD_PROBE_SYMBOL + (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 12;  \
    for (int k = 0; k < 8; ++k) o[k] = pr_acc[k];                                                               \
    o[8] = __builtin_readcyclecounter() - pr_c0; o[9] = wall_clock64() - pr_w0; } } while (0)
#else
#define BPROBE_BEGIN() do {} while (0)
#define BPROBE_TILE() do {} while (0)
#define BPROBE(k) do {} while (0)
#define BPROBE_END() do {} while (0)
#endif

namespace gnc_mlp {
struct FusedOut {
  float* dw[3];   // per-wave partials [waves, M_l * K_l + M_l]: [dW_l row-major | db_l]
  int M[3], K[3];
};
// nadd = fused_shape(d) (0 / 2); the variant (gathered output gradient or not) follows from b.gg / b.grad_out
int launch_fused_recompute(const gnc_mlp_desc_t& d, const BwdArgs& b, const FusedOut& fo, int nadd, hipStream_t stream);
int launch_fused_saved(const gnc_mlp_desc_t& d, const BwdArgs& b, const FusedOut& fo, int nadd, hipStream_t stream);
}  // namespace gnc_mlp

namespace {

inline bool fused_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---------------------------------------------------------------------------------------------------
// Fused variant for the shape that dominates training (widths 33..64, three Linear layers, ONE row-ordered
// MATMUL segment that may also be the residual, plus 0 or 2 gathered ADD segments): the weight gradients are
// formed in the same kernel.  dW_l = dz_l^T a_{l-1} needs both operands of a tile with the rows on the MFMA k
// axis; both sit in the wave's LDS tiles anyway, so the outer products are accumulated in registers for the whole
// life of the wave (3 x 64 registers) and leave as ONE row of partials per wave and layer (same layout and fixed-
// order reduction as gnc_xty_f32).  Nothing but dz_0 (the ADD segments' gradient), dx and the partials is written:
// the a_l / dz_l / y_hat tensors of the split path (6 x [rows, 64] written here, read back by xty / colsum) do not
// exist.  4 waves per CU, one per SIMD, with the 512-register budget that takes.
// ---------------------------------------------------------------------------------------------------
constexpr int FWAVES = 4;
constexpr int FNT = FWAVES * 64;


// The weight-gradient accumulators of the fused kernel live in AccVGPRs for the whole life of the wave and are touched
// by nothing but these MFMAs, issued from inline asm with the accumulator pinned to the "a" register class.  Reason: a
// kernel that needs more than 256 registers makes hipcc put EVERY MFMA result into AccVGPRs, and the chain's working
// sets (which the vector unit post-processes: ReLU, masks, LayerNorm) then shuttle through v_accvgpr_read / _write -
// ~1,100 static moves in the tile loop, each a vector-unit slot that fp32 MFMA time does not hide on gfx950.  With the
// 192 accumulator registers out of the compiler's sight the rest of the kernel fits the 256 architectural VGPRs.
// Hazards the compiler no longer tracks for these instructions: a dependent MFMA on the same accumulator needs 2 wait
// states (the four accumulators of a step rotate, so 3 other MFMAs sit in between; `volatile` keeps that order), and
// the accumulators are read by VALU only after the tile loop, behind an explicit s_nop block (agpr_mfma_fence).
__device__ __forceinline__ void mfma_agpr(f32x16& acc, float a, float b) {
  asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void agpr_mfma_fence() { asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7" ::: "memory"); }

// ... and because hipcc selects the AccVGPR form for every MFMA *builtin* of a function that owns AccVGPRs at all, the
// chain's MFMAs of the fused kernel are issued from inline asm too, in VGPR form ("v" class): results land where the
// vector unit reads them.  What the hazard recognizer would have done for them is done by hand:
//   * XDL (16-pass) result -> VALU / LDS / VMEM read of it: 18 wait states.  Every chain ends in chain_fence(), an
//     s_nop block that also takes the accumulators as in/out operands, so no consumer can be scheduled above it;
//   * a dependent MFMA on the same accumulator (SrcC == vDst, same opcode) may follow back to back (hipcc emits that
//     itself for the builtin form);
//   * operands are VALU or LDS results: no hazard beyond the lgkmcnt wait the compiler still places, since they are
//     ordinary asm inputs.
__device__ __forceinline__ void mfma_vgpr(f32x16& acc, float a, float b) {
  asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_vgpr_from_zero(f32x16& acc, float a, float b) {
  asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void chain_fence(f32x16 (&acc)[2]) {
  asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[0]), "+v"(acc[1]));
}
// asm-issued counterparts of mma_chunk_from_lds / mma_chunk_from_regs (mlp_device.h) and mma_transposed_from_regs for
// two accumulator tiles (widths <= 64)
__device__ __forceinline__ void fmma_chunk_from_lds(f32x16 (&acc)[2], const float* abuf, const float* wbuf, int kc8, int i, int h) {
#pragma unroll 2
  for (int g = 0; g < kc8; ++g) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(abuf + i * LDSW + 8 * g + 4 * h);
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(wbuf + i * LDSW + 8 * g + 4 * h);
    const f32x4 a1 = *reinterpret_cast<const f32x4*>(wbuf + (32 + i) * LDSW + 8 * g + 4 * h);
    mfma_vgpr(acc[0], a0.x, b.x); mfma_vgpr(acc[1], a1.x, b.x);
    mfma_vgpr(acc[0], a0.y, b.y); mfma_vgpr(acc[1], a1.y, b.y);
    mfma_vgpr(acc[0], a0.z, b.z); mfma_vgpr(acc[1], a1.z, b.z);
    mfma_vgpr(acc[0], a0.w, b.w); mfma_vgpr(acc[1], a1.w, b.w);
  }
}
__device__ __forceinline__ void fmma_chunk_from_regs(f32x16 (&dst)[2], const f32x16 (&src)[2], const float* wbuf, int in_dim, int i,
                                                     int h) {
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    if (g * 8 < in_dim) {
      const int ts = g >> 2, q = g & 3;
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(wbuf + i * LDSW + 8 * g + 4 * h);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(wbuf + (32 + i) * LDSW + 8 * g + 4 * h);
      mfma_vgpr(dst[0], a0.x, src[ts][4 * q + 0]); mfma_vgpr(dst[1], a1.x, src[ts][4 * q + 0]);
      mfma_vgpr(dst[0], a0.y, src[ts][4 * q + 1]); mfma_vgpr(dst[1], a1.y, src[ts][4 * q + 1]);
      mfma_vgpr(dst[0], a0.z, src[ts][4 * q + 2]); mfma_vgpr(dst[1], a1.z, src[ts][4 * q + 2]);
      mfma_vgpr(dst[0], a0.w, src[ts][4 * q + 3]); mfma_vgpr(dst[1], a1.w, src[ts][4 * q + 3]);
    }
  }
}
// dst = W^T-tile * src (see mma_transposed_from_regs): the first MFMA of each output tile starts from the constant 0;
// the two column reads of step k + 2 are issued before the MFMAs of step k (two steps = 256 MFMA cycles of cover)
__device__ __forceinline__ void fmma_transposed_from_regs(f32x16 (&dst)[2], const f32x16 (&src)[2], const float* wbuf, int i, int h) {
  auto row_of = [&](int k) { return 32 * (k >> 4) + (k & 3) + 8 * ((k & 15) >> 2) + 4 * h; };  // k = 16 t + r
  float w0[3], w1[3];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    w0[k] = wbuf[row_of(k) * LDSW + i];
    w1[k] = wbuf[row_of(k) * LDSW + 32 + i];
  }
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    if (k + 2 < 32) {
      w0[(k + 2) % 3] = wbuf[row_of(k + 2) * LDSW + i];
      w1[(k + 2) % 3] = wbuf[row_of(k + 2) * LDSW + 32 + i];
    }
    const float bsrc = src[k >> 4][k & 15];
    if (k == 0) {
      mfma_vgpr_from_zero(dst[0], w0[0], bsrc);
      mfma_vgpr_from_zero(dst[1], w1[0], bsrc);
    } else {
      mfma_vgpr(dst[0], w0[k % 3], bsrc);
      mfma_vgpr(dst[1], w1[k % 3], bsrc);
    }
  }
}

// acc[a][c] += sum over the tile's 32 rows of tm[row][32a + .] (x) tk[row][32c + .]; csum[a] += column sums of tm
__device__ __forceinline__ void xty_tile(f32x16 (&acc)[2][2], float (&csum)[2], const float* tm, const float* tk, int i, int h,
                                         bool both = true) {  // both (wave-uniform) = false: tk holds at most 32 columns
  // software-pipelined by one step: the four column reads of step s + 1 are in flight under the four MFMAs of step s
  float av[2], bv[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) av[a] = tm[h * LDSW + 32 * a + i];
#pragma unroll
  for (int c = 0; c < 2; ++c) bv[c] = tk[h * LDSW + 32 * c + i];
#pragma unroll 4
  for (int s = 0; s < RPW / 2; ++s) {
    float na[2], nb[2];
    const int sn = s + 1 < RPW / 2 ? s + 1 : s;  // the last step re-reads its own row (never used)
#pragma unroll
    for (int a = 0; a < 2; ++a) na[a] = tm[(2 * sn + h) * LDSW + 32 * a + i];
#pragma unroll
    for (int c = 0; c < 2; ++c) nb[c] = tk[(2 * sn + h) * LDSW + 32 * c + i];
#pragma unroll
    for (int a = 0; a < 2; ++a) csum[a] += av[a];
#pragma unroll
    for (int a = 0; a < 2; ++a) mfma_agpr(acc[a][0], av[a], bv[0]);
    if (both) {  // the encoders' 3-column inputs: half the MFMAs of their first Linear's weight gradient
#pragma unroll
      for (int a = 0; a < 2; ++a) mfma_agpr(acc[a][1], av[a], bv[1]);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) av[a] = na[a];
#pragma unroll
    for (int c = 0; c < 2; ++c) bv[c] = nb[c];
  }
}

// GG: 0 = the output gradient is grad_out's rows; 1 = those rows + the gathered rows b.gg[id of the second ADD segment]
// (the backward of the scatter-sum that consumed the output rows, folded into this launch); 2 = the gathered rows alone
// (grad_out is null).  Compile-time, so that every load of the tile loop stays unconditional.
// SAVED: the forward kept its post-activations (gnc_mlp_desc_t.save_act -> b.act[0], b.act[1], contiguous rows): the tile
// reads them (a1 parked in the wave's LDS tile one tile ahead, a0 requested at the top of the tile and used two phases
// later) instead of recomputing the first two Linear layers - 448 instead of 576 MFMAs per tile, no gathered rows.
template <int NADD, int GG = 0, bool SAVED = false>
__global__ __launch_bounds__(FNT) void mlp_backward_fused_kernel(const gnc_mlp_desc_t d, const BwdArgs b, const FusedOut fo,
                                                                 const int num_wtiles) {
  static_assert(GG == 0 || NADD == 2, "a gathered output gradient shares the ids of the second ADD segment");
  constexpr bool HAS_G = GG != 2;  // a row-ordered grad_out exists
  constexpr bool GATHER_IN = NADD > 0 && !SAVED;   // the gathered projections are inputs of the recomputed first Linear
  constexpr bool NEED_IDS = GATHER_IN || GG != 0;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int HT = 2;
  constexpr int CH = HT * 32 * LDSW;
  constexpr int PSTRIDE = HT * 32;
  constexpr int L = 3;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31;
  const int h = lane >> 5;
  const int c4 = lane & 15;
  const int rs = lane >> 4;
  const int out_dim = d.out_dim[L - 1];
  const int rows = (int)d.rows;
  float* wres = lds;                                   // 3 weight chunks: W0 (its MATMUL columns), W1, W2
  float* pbuf = lds + 3 * CH;                          // biases 0..2, gamma, beta
  // The wave's three row tiles.  Their offsets are made opaque to the compiler: left alone it addresses all three from
  // ONE base register plus constants beyond the 16-bit offset field of the DS instructions (the tiles sit above 64 KB),
  // which costs a v_add_u32 per LDS access in the column-read loops; with one base per tile every access is base +
  // immediate.
  int te_o = 3 * CH + (L + 2) * PSTRIDE + wave * 3 * RPW * LDSW, ta_o = te_o + RPW * LDSW, tb_o = te_o + 2 * RPW * LDSW;
  float* te = lds + te_o;                              // the tile's input rows (kept for dW0)
  float* ta = lds + ta_o;                              // working tile A
  float* tb = lds + tb_o;                              // working tile B

  stage_params<FNT>(pbuf, d, PSTRIDE, tid);
  {
    const int ldw0 = ldw_of(d, 0);
    const bool w0v = (ldw0 % 4 == 0) && ((reinterpret_cast<uintptr_t>(d.weight[0]) & 15u) == 0) && (d.seg[0].wcol % 4 == 0);
    stage_weights<HT * 32, FNT>(wres, d.weight[0], ldw0, d.out_dim[0], d.seg[0].wcol, d.seg[0].wcol + d.seg[0].width, 16, w0v, tid);
    for (int l = 1; l < L; ++l) {
      const int ldw = ldw_of(d, l);
      const bool wv = (ldw % 4 == 0) && ((reinterpret_cast<uintptr_t>(d.weight[l]) & 15u) == 0);
      stage_weights<HT * 32, FNT>(wres + l * CH, d.weight[l], ldw, d.out_dim[l], 0, d.in_dim[l], 16, wv, tid);
    }
  }
  __syncthreads();

  f32x16 dW0[2][2], dW1[2][2], dW2[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) dW0[a][c][r] = dW1[a][c][r] = dW2[a][c][r] = 0.f;
  float cs0[2] = {0.f, 0.f}, cs1[2] = {0.f, 0.f}, cs2[2] = {0.f, 0.f};
  f32x4 sum_g = {0.f, 0.f, 0.f, 0.f}, sum_gy = {0.f, 0.f, 0.f, 0.f};

  const gnc_mlp_segment_t& s0 = d.seg[0];
  const uint32_t e_off = (uint32_t)(rs * s0.ld + (c4 * 4 < s0.ld ? c4 * 4 : 0)) * 4u;
  const uint32_t g_off = (uint32_t)(rs * b.ld_grad_out + (c4 * 4 < out_dim ? c4 * 4 : 0)) * 4u;
  const int total_waves = (int)gridDim.x * FWAVES;

  auto to_tile = [&](float* t, const f32x4 (&pre)[NP], int width) {  // whole-row pieces -> LDS tile, zero beyond `width`
    compiler_lds_barrier();
    const int fl = fresh_lane();
    const int c4 = fl & 15, rs = fl >> 4;
    const int c = c4 * 4;
    if (width >= KC) {  // wave-uniform: the full-width case carries no masks
#pragma unroll
      for (int p = 0; p < NP; ++p) *reinterpret_cast<f32x4*>(t + (p * 4 + rs) * LDSW + c) = pre[p];
    } else {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        f32x4 v = pre[p];
        v.x = c + 0 < width ? v.x : 0.f; v.y = c + 1 < width ? v.y : 0.f;
        v.z = c + 2 < width ? v.z : 0.f; v.w = c + 3 < width ? v.w : 0.f;
        *reinterpret_cast<f32x4*>(t + (p * 4 + rs) * LDSW + c) = v;
      }
    }
    compiler_lds_barrier();
  };
  auto acc_to_tile = [&](float* t, const f32x16 (&acc)[HT]) {
    compiler_lds_barrier();
    tiles_to_lds<HT>(acc, t, i, h);
    compiler_lds_barrier();
  };

  // Load schedule.  Everything a tile reads from memory - its e rows, its grad_out rows, the two gathered projection
  // rows - is requested during the PREVIOUS tile's backward phases, at the points where the accumulator sets of the
  // chain have died and registers are free (after layer 2: e and grad_out; after layer 1: the gathers), and is
  // parked in the wave's LDS tiles at the END of that tile: no load latency is exposed inside a tile and no row
  // piece is carried in registers across the loop's back edge or the forward recompute.  Gather ids travel two
  // tiles ahead.
  int ids_nxt[2] = {0, 0}, ids_cur[2] = {0, 0};
  auto load_ids = [&](int wt_) {
    if constexpr (NEED_IDS) {
      int r = wt_ * RPW + (lane & 31);
      const bool inside = r < rows;
      r = inside ? r : rows - 1;
      if constexpr (GATHER_IN) ids_nxt[0] = d.seg[1].index[r];
      const int id1 = d.seg[2].index[r];
      // with a gathered gradient (same ids) a row past the end must contribute ZEROS to grad_out, not the last row's
      // gradient: its id points one row past the table, where the buffer window reads 0
      ids_nxt[1] = (GG != 0 && !inside) ? (int)d.seg[2].table_rows : id1;
    }
  };
  // rows b.gg[id] of the gathered part of the output gradient (first out_dim columns), as whole-row pieces
  auto gather_gg = [&](f32x4 (&dst)[NP], int id) {
    const __amdgpu_buffer_rsrc_t w =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(b.gg), 0, (int)(uint32_t)(b.gg_rows * b.ld_gg * 4), 0x00020000);
    const int fl = fresh_lane();
    const int rb = id * (b.ld_gg * 4);
    const uint32_t col = (uint32_t)((fl & 15) * 4 < out_dim ? (fl & 15) * 16 : 0);
#pragma unroll
    for (int p = 0; p < NP; ++p) dst[p] = window_load(w, (uint32_t)__shfl(rb, p * 4 + (fl >> 4), 64) + col);
  };
  const uint32_t a_off0 = (uint32_t)(rs * d.out_dim[0] + (c4 * 4 < d.out_dim[0] ? c4 * 4 : 0)) * 4u;  // saved a0 / a1 rows
  const uint32_t a_off1 = (uint32_t)(rs * d.out_dim[1] + (c4 * 4 < d.out_dim[1] ? c4 * 4 : 0)) * 4u;
  constexpr int NPA = (NADD > 0 || SAVED) ? NP : 1;   // pa: the gathered projections' sum, or (SAVED) the saved a1 rows
  constexpr int NPB = GATHER_IN ? NP : 1;
  // the next tile's rows for the ta tile: the two gathered projections, or the saved a1 rows
  auto gather_rows_of = [&](f32x4 (&pa)[NPA], f32x4 (&pb)[NPB], int row_first) {
    if constexpr (SAVED) load_tile_rows(pa, b.act[1], d.out_dim[1], row_first, rows, a_off1);
    if constexpr (GATHER_IN) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const gnc_mlp_segment_t& sg = d.seg[1 + k];
        const __amdgpu_buffer_rsrc_t w =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sg.ptr), 0, (int)(uint32_t)(sg.table_rows * sg.ld * 4), 0x00020000);
        const int rb = ids_cur[k] * (sg.ld * 4);
        const uint32_t col = (uint32_t)(c4 * 4 < sg.ld ? c4 * 16 : 0);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const f32x4 v = window_load(w, (uint32_t)__shfl(rb, p * 4 + rs, 64) + col);
          if (k == 0) pa[p] = v; else pb[p] = v;
        }
      }
    }
  };
  const int wt0 = (int)blockIdx.x * FWAVES + wave;
  f32x4 pe[NP], pg[NP] = {}, pa[NPA], pb[NPB];  // pg stays zero without a row-ordered grad_out
  load_ids(wt0);
  ids_cur[0] = ids_nxt[0]; ids_cur[1] = ids_nxt[1];
  load_ids(wt0 + total_waves);
  load_tile_rows(pe, s0.ptr, s0.ld, wt0 * RPW, rows, e_off);   // past-the-end tiles read zeros
  if constexpr (HAS_G) load_tile_rows(pg, b.grad_out, b.ld_grad_out, wt0 * RPW, rows, g_off);
  gather_rows_of(pa, pb, wt0 * RPW);
  // park a tile's rows in the wave's LDS tiles: e -> te, grad_out -> tb, the sum of the gathered projections -> ta
  auto park = [&](bool with_e) {
    if (with_e) to_tile(te, pe, s0.width);
    if constexpr (HAS_G) to_tile(tb, pg, out_dim);
    if constexpr (SAVED) to_tile(ta, pa, d.out_dim[1]);
    else if constexpr (NADD > 0) to_tile(ta, pa, d.seg[1].width);  // pa holds the SUM of both gathered rows by now
  };
  auto sum_gathered = [&]() {
    if constexpr (GATHER_IN) {
#pragma unroll
      for (int p = 0; p < NP; ++p) pa[p] += pb[p];
    }
  };
  sum_gathered();
  park(true);
  BPROBE_BEGIN();
  for (int wt = wt0; wt < num_wtiles; wt += total_waves) {
    BPROBE_TILE();
    const int row0 = wt * RPW;
    const int nrow0 = (wt + total_waves) * RPW;  // may lie past the end: windows return zeros, ids are clamped
    int id_this = 0;  // this tile's ids of the second ADD segment (ids_cur moves on to the next tile's)
    f32x4 pq[GG != 0 ? NP : 1];
    if constexpr (NEED_IDS) {
      if constexpr (GG != 0) id_this = ids_cur[1];
      ids_cur[0] = ids_nxt[0]; ids_cur[1] = ids_nxt[1];
      load_ids(wt + 2 * total_waves);
      // the gathered part of this tile's output gradient lands under the forward recompute / the third Linear
      if constexpr (GG != 0) gather_gg(pq, id_this);
    }
    f32x16 a0[HT], a1[HT];
    f32x4 p0[SAVED ? NP : 1];  // SAVED: this tile's saved a0 rows, in flight until layer 1
    if constexpr (SAVED) {
      load_tile_rows(p0, b.act[0], d.out_dim[0], row0, rows, a_off0);
      tile_from_lds<HT>(a1, ta, i, h);  // the saved a1 rows (parked one tile ahead) in accumulator layout
      BPROBE(0);
      BPROBE(1);
    } else {
      // -------------------------------------------------------------- forward recompute
      init_bias<HT>(a0, pbuf, h);
      fmma_chunk_from_lds(a0, te, wres, (s0.width + 7) >> 3, i, h);
      chain_fence(a0);
      if constexpr (NADD > 0) add_tile_from_lds<HT>(a0, ta, i, h);
      relu_tiles<HT>(a0);
      BPROBE(0);  // first Linear + additive rows + ReLU
      init_bias<HT>(a1, pbuf + PSTRIDE, h);
      fmma_chunk_from_regs(a1, a0, wres + CH, d.in_dim[1], i, h);
      chain_fence(a1);
      relu_tiles<HT>(a1);
      BPROBE(1);  // second Linear + ReLU
    }
    // the third Linear (for the LayerNorm statistics).  SAVED: before the output gradient is touched, so that the
    // gathered part of it (requested at the top of the tile) has these 64 MFMAs to land; the recomputing variants keep it
    // behind the gradient tile (their gather has had the whole forward recompute)
    f32x16 y[HT];
    auto third_linear = [&]() {
      init_bias<HT>(y, pbuf + 2 * PSTRIDE, h);
      fmma_chunk_from_regs(y, a1, wres + 2 * CH, d.in_dim[2], i, h);
      chain_fence(y);
    };
    if constexpr (SAVED) {
      if (d.ln_gamma) third_linear();
    }
    // ---------------------------------------------------------------- grad of the pre-LayerNorm output
    if constexpr (GG != 0) {  // grad_out's rows (parked in tb) + the gathered rows; GG == 2: the gathered rows alone
      compiler_lds_barrier();
      const int fl = fresh_lane();
      const int c = (fl & 15) * 4;
      float* tp = tb + (fl >> 4) * LDSW + c;
      if (out_dim < KC) {  // wave-uniform: the full-width case carries no masks
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          f32x4 v = pq[p];
          v.x = c + 0 < out_dim ? v.x : 0.f; v.y = c + 1 < out_dim ? v.y : 0.f;
          v.z = c + 2 < out_dim ? v.z : 0.f; v.w = c + 3 < out_dim ? v.w : 0.f;
          pq[p] = v;
        }
      }
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        if constexpr (HAS_G) pq[p] += *reinterpret_cast<const f32x4*>(tp + p * 4 * LDSW);
        *reinterpret_cast<f32x4*>(tp + p * 4 * LDSW) = pq[p];
      }
      compiler_lds_barrier();
      // both parts exist and the residual path of dx needs their sum again at the end of the tile: the summed rows go
      // out to a scratch tensor here and come back (L2 / MALL) where grad_out's rows alone would - re-gathering there
      // would cost a register set the kernel does not have (measured: 104 B of scratch, +0.65 ms per launch)
      if constexpr (GG == 1) {
        if (b.dx && b.dx_add_grad_out) store_row_pieces(pq, b.g_sum, b.ld_g_sum, out_dim, row0, rows, fl & 15, fl >> 4);
      }
    }
    f32x16 g[HT];
    tile_from_lds<HT>(g, tb, i, h);
    if (d.ln_gamma) {
      if constexpr (!SAVED) third_linear();
      layer_norm_backward_tiles<HT>(y, g, pbuf + L * PSTRIDE, out_dim, d.ln_eps, h);
      acc_to_tile(ta, y);  // y_hat as whole rows next to grad_out's rows (still in tb): the LayerNorm parameter sums
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const f32x4 yh = *reinterpret_cast<const f32x4*>(ta + (p * 4 + rs) * LDSW + c4 * 4);
        const f32x4 gp = *reinterpret_cast<const f32x4*>(tb + (p * 4 + rs) * LDSW + c4 * 4);
        sum_g += gp;
        sum_gy = __builtin_elementwise_fma(gp, yh, sum_gy);
      }
    } else {
#pragma unroll
      for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) g[t][r] = (feat_of(t, r, h) < out_dim) ? g[t][r] : 0.f;
    }
    BPROBE(2);  // third Linear + LayerNorm backward + parameter sums
    // ---------------------------------------------------------------- layer 2: dW2 += dz2^T a1, da1 = W2^T dz2
    acc_to_tile(tb, g);
    acc_to_tile(ta, a1);
    xty_tile(dW2, cs2, tb, ta, i, h);
    BPROBE(3);  // transposes + dW2
    {
      f32x16 da[HT];
      fmma_transposed_from_regs(da, g, wres + 2 * CH, i, h);
      chain_fence(da);
      mask_by_positive<HT>(da, a1);
#pragma unroll
      for (int t = 0; t < HT; ++t) g[t] = da[t];
    }
    BPROBE(4);  // da1 + mask
    // a1 is dead: the next tile's e and grad_out rows are requested here and land under layers 1 and 0
    load_tile_rows(pe, s0.ptr, s0.ld, nrow0, rows, e_off);
    if constexpr (HAS_G) load_tile_rows(pg, b.grad_out, b.ld_grad_out, nrow0, rows, g_off);
    // ---------------------------------------------------------------- layer 1
    acc_to_tile(tb, g);
    if constexpr (SAVED) to_tile(ta, p0, d.out_dim[0]);  // the saved a0 rows as they are
    else acc_to_tile(ta, a0);
    xty_tile(dW1, cs1, tb, ta, i, h);
    {
      f32x16 da[HT];
      fmma_transposed_from_regs(da, g, wres + CH, i, h);
      chain_fence(da);
      if constexpr (SAVED) tile_from_lds<HT>(a0, ta, i, h);  // ... and in accumulator layout for the ReLU derivative
      mask_by_positive<HT>(da, a0);
#pragma unroll
      for (int t = 0; t < HT; ++t) g[t] = da[t];
    }
    BPROBE(5);  // load issue + layer 1 (transposes, dW1, da0, mask)
    // a0 is dead: the next tile's gathered rows, and this tile's grad_out rows again for the residual path of dx
    gather_rows_of(pa, pb, nrow0);
    // ---------------------------------------------------------------- layer 0: dW0 += dz0^T e, dz0 out, dx
    acc_to_tile(tb, g);
    xty_tile(dW0, cs0, tb, te, i, h, NADD != 0 || s0.width > 32);
    to_tile(te, pe, s0.width);  // this tile's e rows are done with: the next tile's (requested two layers ago) move in
    sum_gathered();  // the next tile's gathered rows have landed under dW0: one register set instead of two from here on
    f32x4 gres[NP];  // requested as late as the dx product still covers (L2 / MALL hits): 32 registers less under dW0
    if (b.dx && b.dx_add_grad_out) {
      if constexpr (GG == 2) {
        gather_gg(gres, id_this);
      } else {
        // GG == 1: the summed rows this wave stored at the top of the tile (tens of thousands of cycles ago; the explicit
        // wait makes the read-after-write independent of that distance - every load requested so far has long landed)
        if constexpr (GG == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const float* gsrc = GG == 1 ? b.g_sum : b.grad_out;
        const int gld = GG == 1 ? b.ld_g_sum : b.ld_grad_out;
        const int flg = fresh_lane();
        load_tile_rows(gres, gsrc, gld, row0, rows, (uint32_t)((flg >> 4) * gld + ((flg & 15) * 4 < out_dim ? (flg & 15) * 4 : 0)) * 4u);
      }
    }
    f32x16 dxs[HT];
    if (b.dx) {
      fmma_transposed_from_regs(dxs, g, wres, i, h);
      chain_fence(dxs);
      acc_to_tile(ta, dxs);
    }
    BPROBE(6);  // gather issue + layer 0 (transpose, dW0, dx)
    // everything requested above is collected before the asm stores join the memory queue
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      asm volatile("" ::"v"(pg[p]));
      if constexpr (NPA > 1) asm volatile("" ::"v"(pa[p]));
    }
    asm volatile("" ::"v"(ids_nxt[0]), "v"(ids_nxt[1]));
    // the residual path of dx BEFORE any store is issued: its grad_out rows are the youngest load, the compiler can
    // only wait for them with vmcnt(0), and behind a store that wait would sit out the store's HBM round trip
    // (measured with the phase probe: 4.7k of a tile's 52k cycles)
    const int fle = fresh_lane();
    const int c4e = fle & 15, rse = fle >> 4;
    {  // all row pieces of both outputs out of the LDS tiles first, the residual added in registers, then the stores
      f32x4 oz[NP], ox[NP];
      compiler_lds_barrier();
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        oz[p] = *reinterpret_cast<const f32x4*>(tb + (p * 4 + rse) * LDSW + c4e * 4);
        ox[p] = *reinterpret_cast<const f32x4*>(ta + (p * 4 + rse) * LDSW + c4e * 4);
      }
      if (b.dx && b.dx_add_grad_out) {
#pragma unroll
        for (int p = 0; p < NP; ++p) ox[p] += gres[p];
      }
      compiler_lds_barrier();
      if (b.dz[0]) store_row_pieces(oz, b.dz[0], d.out_dim[0], d.out_dim[0], row0, rows, c4e, rse);
      if (b.dx) store_row_pieces(ox, b.dx + s0.wcol, b.ld_dx, s0.width, row0, rows, c4e, rse);
    }
    compiler_lds_barrier();
    park(false);  // the rest of the next tile's rows (requested above, long landed) into the tiles this one has finished with
    BPROBE(7);  // collect loads + stores + park
  }
  BPROBE_END();

  // ------------------------------------------------------------------ one row of partials per wave and layer
  agpr_mfma_fence();  // the last asm-issued MFMAs have written their accumulators before anything reads them
  const int gw = (int)blockIdx.x * FWAVES + wave;
  auto write_partial = [&](int l, const f32x16 (&acc)[2][2], const float (&cs)[2]) {
    const int M = fo.M[l], K = fo.K[l];
    float* dst = fo.dw[l] + (int64_t)gw * (M * K + M);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
          const int k = 32 * c + i;
          if (m < M && k < K) dst[m * K + k] = acc[a][c][r];
        }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const float tot = add_halves(cs[a]);
      if (h == 0 && 32 * a + i < M) dst[M * K + 32 * a + i] = tot;
    }
  };
  write_partial(0, dW0, cs0);
  write_partial(1, dW1, cs1);
  write_partial(2, dW2, cs2);
  if (b.ln_partial) {
    float* dst = b.ln_partial + (int64_t)gw * 2 * out_dim;
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

// shape of the fused data + weight-gradient kernel; returns the number of ADD segments (0 / 2) or -1
int fused_shape(const gnc_mlp_desc_t& d) {
  if (d.num_linear != 3 || d.activation != GNC_ACT_RELU || d.rows < 1 || d.rows >= INT32_MAX - (1 << 22)) return -1;
  for (int l = 0; l < 3; ++l)
    if (d.out_dim[l] > KC || (l > 0 && d.in_dim[l] > KC)) return -1;
  if (d.num_segments != 1 && d.num_segments != 3) return -1;
  const gnc_mlp_segment_t& s0 = d.seg[0];
  if (s0.mode != GNC_SEG_MATMUL || s0.index || s0.width > KC || s0.ld % 4 != 0 || !fused_al16(s0.ptr)) return -1;
  for (int s = 1; s < d.num_segments; ++s) {
    const gnc_mlp_segment_t& g = d.seg[s];
    if (g.mode != GNC_SEG_ADD || !g.index || g.width > KC || g.ld % 4 != 0 || !fused_al16(g.ptr)) return -1;
    if (g.table_rows <= 0 || g.table_rows * (int64_t)g.ld * 4 > 0xffffffffll) return -1;
  }
  return d.num_segments - 1;
}
int fused_grid(int64_t rows) {
  const int64_t grid = gnc::ceil_div(gnc::ceil_div(rows, RPW), FWAVES);
  return (int)(grid > gnc::num_cu() ? gnc::num_cu() : grid);
}
constexpr size_t kFusedSmem = ((size_t)3 * 2 * 32 * LDSW + (size_t)5 * 64 + (size_t)FWAVES * 3 * RPW * LDSW) * sizeof(float);

template <int NADD, int GG, bool SAVED>
int launch_fused(const gnc_mlp_desc_t& d, const BwdArgs& b, const FusedOut& fo, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    int rc = gnc::check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_backward_fused_kernel<NADD, GG, SAVED>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                            "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    if (rc) return rc;
    attr_set = true;
  }
  const int64_t num_wtiles = gnc::ceil_div(d.rows, RPW);
  mlp_backward_fused_kernel<NADD, GG, SAVED><<<dim3((unsigned)fused_grid(d.rows)), dim3(FNT), kFusedSmem, stream>>>(d, b, fo, (int)num_wtiles);
  return gnc::check_launch("mlp_backward_fused_kernel");
}

// the variant of one translation unit (SAVED fixed) that serves this call
template <bool SAVED>
int launch_fused_variant(const gnc_mlp_desc_t& d, const BwdArgs& b, const FusedOut& fo, int nadd, hipStream_t stream) {
  if (b.gg) return b.grad_out ? launch_fused<2, 1, SAVED>(d, b, fo, stream) : launch_fused<2, 2, SAVED>(d, b, fo, stream);
  return nadd == 2 ? launch_fused<2, 0, SAVED>(d, b, fo, stream) : launch_fused<0, 0, SAVED>(d, b, fo, stream);
}
}  // namespace
