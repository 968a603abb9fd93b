// Device-side helpers shared by the fused-MLP kernels (mlp_fused.hip: weights streamed through
// LDS; mlp_resident.hip: weights resident in LDS).  The formulation is described at the top of
// mlp_fused.hip.
#pragma once
#include "gnc_common.h"

namespace gnc_mlp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KC = 64;    // k-chunk: columns of one staged weight / activation block
constexpr int LDSW = 68;  // padded LDS row in floats (4 * odd): conflict-free ds_read_b128 fragments
constexpr int RPW = 32;   // data rows per wave (the N dimension of the 32x32 MFMA tile)

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ void wave_lds_fence() {
  // LDS traffic of one wave is processed in issue order; this only stops the compiler from
  // moving LDS accesses across the point and waits for outstanding LDS returns.
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

__device__ __forceinline__ float activate(float x, int act, float p) {
  switch (act) {
    case GNC_ACT_RELU: return x < 0.f ? 0.f : x;  // NaN propagates like torch.clamp_min
    case GNC_ACT_IDENTITY: return x;
    case GNC_ACT_TANH: return tanhf(x);
    case GNC_ACT_SIGMOID: return 1.f / (1.f + expf(-x));
    case GNC_ACT_SILU: return x / (1.f + expf(-x));
    case GNC_ACT_GELU: return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f));
    case GNC_ACT_LEAKY_RELU: return x < 0.f ? x * p : x;
    case GNC_ACT_ELU: return x > 0.f ? x : p * (expf(x) - 1.f);
    default: return x;
  }
}

template <int T>
__device__ __forceinline__ void activate_tiles(f32x16 (&acc)[T], int act, float p) {
  if (act == GNC_ACT_RELU) {  // hot case kept branch-free
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = acc[t][r] < 0.f ? 0.f : acc[t][r];
  } else {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = activate(acc[t][r], act, p);
  }
}

// feature index held by accumulator register r of tile t on lane half h
__device__ __forceinline__ constexpr int feat_of(int t, int r, int h) { return 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h; }

// 4 floats at p[0..3], elements at column >= limit read as 0; vector load when allowed.
__device__ __forceinline__ f32x4 load4_guarded(const float* p, int col, int limit, bool vec_ok) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (vec_ok && col + 4 <= limit) {
    v = *reinterpret_cast<const f32x4*>(p);
  } else {
    if (col + 0 < limit) v.x = p[0];
    if (col + 1 < limit) v.y = p[1];
    if (col + 2 < limit) v.z = p[2];
    if (col + 3 < limit) v.w = p[3];
  }
  return v;
}

// Workgroup-wide: wbuf[n][0 .. ncol4*4) = W[n][kbase + ..] for n < TROWS, zero outside the
// matrix.  16 lanes cover one 64-float row, so a pass of NT threads covers NT/16 weight rows.
template <int TROWS, int NT>
__device__ __forceinline__ void stage_weights(float* wbuf, const float* __restrict__ W, int ldw, int out_dim,
                                              int kbase, int klimit, int ncol4, bool vec_ok, int tid) {
  constexpr int RPP = NT / 16;
  static_assert(TROWS % RPP == 0, "weight rows must be a multiple of the rows staged per pass");
  const int c4 = tid & 15;
  const int r0 = tid >> 4;
  if (c4 < ncol4) {
#pragma unroll
    for (int p = 0; p < TROWS / RPP; ++p) {
      const int n = p * RPP + r0;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (n < out_dim) v = load4_guarded(W + (int64_t)n * ldw + kbase + c4 * 4, kbase + c4 * 4, klimit, vec_ok);
      *reinterpret_cast<f32x4*>(wbuf + n * LDSW + c4 * 4) = v;
    }
  }
}

// biases, LayerNorm gamma and beta, zero padded to PSTRIDE floats per row, into LDS
template <int NT>
__device__ __forceinline__ void stage_params(float* pbuf, const gnc_mlp_desc_t& d, int pstride, int tid) {
  const int L = d.num_linear;
  const int out_dim = d.out_dim[L - 1];
  for (int idx = tid; idx < (L + 2) * pstride; idx += NT) {
    const int l = idx / pstride, n = idx - l * pstride;
    float v = 0.f;
    if (l < L) {
      if (d.bias[l] && n < d.out_dim[l]) v = d.bias[l][n];
    } else if (d.ln_gamma && n < out_dim) {
      v = (l == L) ? d.ln_gamma[n] : (d.ln_beta ? d.ln_beta[n] : 0.f);
    }
    pbuf[idx] = v;
  }
}

template <int T>
__device__ __forceinline__ void init_bias(f32x16 (&acc)[T], const float* pb, int h) {
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(pb + 32 * t + 8 * q + 4 * h);
      acc[t][4 * q + 0] = b.x; acc[t][4 * q + 1] = b.y; acc[t][4 * q + 2] = b.z; acc[t][4 * q + 3] = b.w;
    }
}

// acc[t] += W_chunk[32t + i][8g + 4h + s] * X[i][8g + 4h + s] for g < kc8: one staged 64-column
// chunk of the first Linear; A fragments from wbuf, B fragments from the wave's row tile.
template <int T>
__device__ __forceinline__ void mma_chunk_from_lds(f32x16 (&acc)[T], const float* abuf, const float* wbuf, int kc8,
                                                   int i, int h) {
#pragma unroll(T >= 4 ? 1 : 2)
  for (int g = 0; g < kc8; ++g) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(abuf + i * LDSW + 8 * g + 4 * h);
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(wbuf + (32 * t + i) * LDSW + 8 * g + 4 * h);
      acc[t] = mfma(a.x, b.x, acc[t]);
      acc[t] = mfma(a.y, b.y, acc[t]);
      acc[t] = mfma(a.z, b.z, acc[t]);
      acc[t] = mfma(a.w, b.w, acc[t]);
    }
  }
}

// acc[t][r] += X[i][feature(t, r, h)]: a staged [32][<=64] tile of rows that are already in the
// hidden width is added to the pre-activation (GNC_SEG_ADD); `t0` = first accumulator tile the
// 64 staged columns belong to.
template <int T>
__device__ __forceinline__ void add_rows_from_lds(f32x16 (&acc)[T], const float* abuf, int t0, int i, int h) {
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      if (t == t0 + tt) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(abuf + i * LDSW + 32 * tt + 8 * q + 4 * h);
          acc[t][4 * q + 0] += v.x; acc[t][4 * q + 1] += v.y; acc[t][4 * q + 2] += v.z; acc[t][4 * q + 3] += v.w;
        }
      }
    }
  }
}

// dst[t] += W_chunk * src for the 64-column chunk `c` of a Linear whose input is the previous
// layer's accumulators (register r of tile ts holds feature 32*ts + 8*(r>>2) + 4h + (r&3)).
template <int TI, int TO>
__device__ __forceinline__ void mma_chunk_from_regs(f32x16 (&dst)[TO], const f32x16 (&src)[TI], const float* wbuf,
                                                    int c, int in_dim, int i, int h) {
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const int ts = 2 * c + (g >> 2);
    if (ts < TI && c * KC + g * 8 < in_dim) {
      const int q = g & 3;
#pragma unroll
      for (int t = 0; t < TO; ++t) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(wbuf + (32 * t + i) * LDSW + 8 * g + 4 * h);
        dst[t] = mfma(a.x, src[ts < TI ? ts : 0][4 * q + 0], dst[t]);
        dst[t] = mfma(a.y, src[ts < TI ? ts : 0][4 * q + 1], dst[t]);
        dst[t] = mfma(a.z, src[ts < TI ? ts : 0][4 * q + 2], dst[t]);
        dst[t] = mfma(a.w, src[ts < TI ? ts : 0][4 * q + 3], dst[t]);
      }
      // wide layers: stop hipcc from hoisting every A fragment of the unrolled chunk to the top
      // (64 x ds_read_b128 = 256 live registers at 8 tiles); one k-group of fragments at a time
      if constexpr (TO >= 4) __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// x(lane) + x(lane ^ 32) in every lane, on the VALU: gfx950's v_permlane32_swap exchanges the upper half of
// one register with the lower half of another (no LDS round trip like ds_bpermute / __shfl_xor).
__device__ __forceinline__ float add_halves(float x) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);  // lower-half value + upper-half value, in that order
}

// x summed over the four lanes {l, l^16, l^32, l^48} (the 16-row MFMA layout keeps a row's features on those),
// on the VALU: v_permlane16_swap exchanges odd 16-lane rows of one register with even rows of another.
__device__ __forceinline__ float add_quarters(float x) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return add_halves(__uint_as_float(r[0]) + __uint_as_float(r[1]));
}

// LayerNorm over the out_dim features of each data row: registers + one cross-half exchange.
// fp32 MFMA and VALU instructions do NOT overlap on a gfx950 SIMD (tools/hw_probe.hip: one 32x32x2 MFMA
// + k VALU = 64 + ~3..5k cycles), so the instruction count here is kernel time: packed fp32 math, no
// per-feature selects (features >= out_dim hold exact zeros - zero weight rows, zero bias - so they drop out
// of the sum; their deviations are cleared before the variance when the width is not 32*OT), 1/n as a
// multiplier and v_rsq_f32 (1 ulp): ~85 VALU per 32 rows instead of ~310.
template <int OT>
__device__ __forceinline__ void layer_norm_tiles(f32x16 (&o)[OT], const float* pg, const float* pbt, int out_dim,
                                                 float eps, int h) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const float inv_n = 1.f / (float)out_dim;  // loop-invariant for the caller's tile loop
  f32x2 sa = {0.f, 0.f}, sb = {0.f, 0.f};
#pragma unroll
  for (int t = 0; t < OT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      sa += f32x2{o[t][4 * q + 0], o[t][4 * q + 1]};
      sb += f32x2{o[t][4 * q + 2], o[t][4 * q + 3]};
    }
  const float mean = add_halves((sa.x + sa.y) + (sb.x + sb.y)) * inv_n;
  const f32x2 m2 = {mean, mean};
#pragma unroll
  for (int t = 0; t < OT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x2 da = f32x2{o[t][4 * q + 0], o[t][4 * q + 1]} - m2;
      const f32x2 db = f32x2{o[t][4 * q + 2], o[t][4 * q + 3]} - m2;
      o[t][4 * q + 0] = da.x; o[t][4 * q + 1] = da.y; o[t][4 * q + 2] = db.x; o[t][4 * q + 3] = db.y;
    }
  if (out_dim != OT * 32) {  // wave-uniform and rare: clear the deviations of the padding features
#pragma unroll
    for (int t = 0; t < OT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[t][r] = (feat_of(t, r, h) < out_dim) ? o[t][r] : 0.f;
  }
  f32x2 va = {0.f, 0.f}, vb = {0.f, 0.f};
#pragma unroll
  for (int t = 0; t < OT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x2 da = {o[t][4 * q + 0], o[t][4 * q + 1]}, db = {o[t][4 * q + 2], o[t][4 * q + 3]};
      va = __builtin_elementwise_fma(da, da, va);
      vb = __builtin_elementwise_fma(db, db, vb);
    }
  const float var = add_halves((va.x + va.y) + (vb.x + vb.y)) * inv_n;
  const float rstd = __frsqrt_rn(var + eps);
  const f32x2 r2 = {rstd, rstd};
#pragma unroll
  for (int t = 0; t < OT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 gm = *reinterpret_cast<const f32x4*>(pg + 32 * t + 8 * q + 4 * h);
      const f32x4 bt = *reinterpret_cast<const f32x4*>(pbt + 32 * t + 8 * q + 4 * h);
      const f32x2 ya = __builtin_elementwise_fma(f32x2{o[t][4 * q + 0], o[t][4 * q + 1]}, f32x2{gm.x, gm.y} * r2,
                                                 f32x2{bt.x, bt.y});
      const f32x2 yb = __builtin_elementwise_fma(f32x2{o[t][4 * q + 2], o[t][4 * q + 3]}, f32x2{gm.z, gm.w} * r2,
                                                 f32x2{bt.z, bt.w});
      o[t][4 * q + 0] = ya.x; o[t][4 * q + 1] = ya.y; o[t][4 * q + 2] = yb.x; o[t][4 * q + 3] = yb.y;
    }
}

// LayerNorm backward on the accumulator layout, same instruction diet as layer_norm_tiles:
//   in : y = pre-LayerNorm output of the row (features >= out_dim exact zeros), g = grad wrt the LayerNorm output
//   out: y = y_hat (normalised, pre-affine; zeros beyond out_dim), g = grad wrt the pre-LayerNorm output
// (d gamma = colsum(grad_out * y_hat), d beta = colsum(grad_out) are formed elsewhere from y_hat.)
template <int HT>
__device__ __forceinline__ void layer_norm_backward_tiles(f32x16 (&y)[HT], f32x16 (&g)[HT], const float* pg, int out_dim,
                                                          float eps, int h) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const float inv_n = 1.f / (float)out_dim;
  const bool padded = out_dim != HT * 32;  // wave-uniform, rare
  f32x2 sa = {0.f, 0.f}, sb = {0.f, 0.f};
#pragma unroll
  for (int t = 0; t < HT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      sa += f32x2{y[t][4 * q + 0], y[t][4 * q + 1]};
      sb += f32x2{y[t][4 * q + 2], y[t][4 * q + 3]};
    }
  const float mean = add_halves((sa.x + sa.y) + (sb.x + sb.y)) * inv_n;
  const f32x2 m2v = {mean, mean};
#pragma unroll
  for (int t = 0; t < HT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x2 da = f32x2{y[t][4 * q + 0], y[t][4 * q + 1]} - m2v;
      const f32x2 db = f32x2{y[t][4 * q + 2], y[t][4 * q + 3]} - m2v;
      y[t][4 * q + 0] = da.x; y[t][4 * q + 1] = da.y; y[t][4 * q + 2] = db.x; y[t][4 * q + 3] = db.y;
    }
  if (padded) {
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool ok = feat_of(t, r, h) < out_dim;
        y[t][r] = ok ? y[t][r] : 0.f;
        g[t][r] = ok ? g[t][r] : 0.f;
      }
  }
  f32x2 va = {0.f, 0.f}, vb = {0.f, 0.f};
#pragma unroll
  for (int t = 0; t < HT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x2 da = {y[t][4 * q + 0], y[t][4 * q + 1]}, db = {y[t][4 * q + 2], y[t][4 * q + 3]};
      va = __builtin_elementwise_fma(da, da, va);
      vb = __builtin_elementwise_fma(db, db, vb);
    }
  const float rstd = __frsqrt_rn(add_halves((va.x + va.y) + (vb.x + vb.y)) * inv_n + eps);
  const f32x2 r2 = {rstd, rstd};
  f32x2 m1a = {0.f, 0.f}, m1b = {0.f, 0.f}, m2a = {0.f, 0.f}, m2b = {0.f, 0.f};
#pragma unroll
  for (int t = 0; t < HT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 gm = *reinterpret_cast<const f32x4*>(pg + 32 * t + 8 * q + 4 * h);  // gamma, zero padded
      const f32x2 ya = f32x2{y[t][4 * q + 0], y[t][4 * q + 1]} * r2, yb = f32x2{y[t][4 * q + 2], y[t][4 * q + 3]} * r2;
      const f32x2 ga = f32x2{g[t][4 * q + 0], g[t][4 * q + 1]} * f32x2{gm.x, gm.y};
      const f32x2 gb = f32x2{g[t][4 * q + 2], g[t][4 * q + 3]} * f32x2{gm.z, gm.w};
      m1a += ga; m1b += gb;
      m2a = __builtin_elementwise_fma(ga, ya, m2a);
      m2b = __builtin_elementwise_fma(gb, yb, m2b);
      y[t][4 * q + 0] = ya.x; y[t][4 * q + 1] = ya.y; y[t][4 * q + 2] = yb.x; y[t][4 * q + 3] = yb.y;
      g[t][4 * q + 0] = ga.x; g[t][4 * q + 1] = ga.y; g[t][4 * q + 2] = gb.x; g[t][4 * q + 3] = gb.y;
    }
  const float m1 = add_halves((m1a.x + m1a.y) + (m1b.x + m1b.y)) * inv_n;
  const float m2 = add_halves((m2a.x + m2a.y) + (m2b.x + m2b.y)) * inv_n;
  const f32x2 nm1 = {-m1, -m1}, nm2 = {-m2, -m2};
#pragma unroll
  for (int t = 0; t < HT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      // rstd * (g - m1 - y_hat * m2)
      const f32x2 ua = __builtin_elementwise_fma(f32x2{y[t][4 * q + 0], y[t][4 * q + 1]}, nm2, f32x2{g[t][4 * q + 0], g[t][4 * q + 1]} + nm1) * r2;
      const f32x2 ub = __builtin_elementwise_fma(f32x2{y[t][4 * q + 2], y[t][4 * q + 3]}, nm2, f32x2{g[t][4 * q + 2], g[t][4 * q + 3]} + nm1) * r2;
      g[t][4 * q + 0] = ua.x; g[t][4 * q + 1] = ua.y; g[t][4 * q + 2] = ub.x; g[t][4 * q + 3] = ub.y;
    }
  if (padded) {
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) g[t][r] = (feat_of(t, r, h) < out_dim) ? g[t][r] : 0.f;
  }
}

// Epilogue: transpose the output tiles through the wave's private LDS tile, add the residual
// and store whole rows (16 lanes x 16 B per row).
// true when the residual rows are exactly what the last staging step left in the wave's tile
__device__ __forceinline__ bool residual_is_staged(const gnc_mlp_desc_t& d, int out_dim) {
  const gnc_mlp_segment_t& ls = d.seg[d.num_segments - 1];
  return d.residual && d.residual == ls.ptr && !ls.index && ls.width == out_dim && ls.ld == d.ld_residual &&
         out_dim <= KC;
}

template <int OT>
__device__ __forceinline__ void store_tiles(f32x16 (&o)[OT], float* abuf, const gnc_mlp_desc_t& d, int64_t row0,
                                            int out_dim, int lane, int i, int h, bool res_staged) {
  if (res_staged) {  // residual straight from the staged input rows, in accumulator layout (no global re-read)
#pragma unroll
    for (int t = 0; t < OT && t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(abuf + i * LDSW + 32 * t + 8 * q + 4 * h);
        o[t][4 * q + 0] += v.x; o[t][4 * q + 1] += v.y; o[t][4 * q + 2] += v.z; o[t][4 * q + 3] += v.w;
      }
  }
  const bool ovec = (d.ld_out % 4 == 0) && ((reinterpret_cast<uintptr_t>(d.out) & 15u) == 0);
  const bool rvec = d.residual && (d.ld_residual % 4 == 0) && ((reinterpret_cast<uintptr_t>(d.residual) & 15u) == 0);
  constexpr int OCH = (OT + 1) / 2;
#pragma unroll
  for (int cc = 0; cc < OCH; ++cc) {
    if (cc * KC < out_dim) {
      wave_lds_fence();
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int t = 2 * cc + tt;
        if (t < OT) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 v = {o[t < OT ? t : 0][4 * q + 0], o[t < OT ? t : 0][4 * q + 1], o[t < OT ? t : 0][4 * q + 2],
                       o[t < OT ? t : 0][4 * q + 3]};
            *reinterpret_cast<f32x4*>(abuf + i * LDSW + 32 * tt + 8 * q + 4 * h) = v;
          }
        }
      }
      wave_lds_fence();
      const int c4 = lane & 15;
      const int rs = lane >> 4;
      const int col = cc * KC + c4 * 4;
      if (col < out_dim) {
#pragma unroll
        for (int p = 0; p < RPW / 4; ++p) {
          const int j = p * 4 + rs;
          const int64_t r = row0 + j;
          if (r < d.rows) {
            f32x4 v = *reinterpret_cast<const f32x4*>(abuf + j * LDSW + c4 * 4);
            if (d.residual && !res_staged) {
              const f32x4 rv = load4_guarded(d.residual + r * d.ld_residual + col, col, out_dim, rvec);
              v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
            }
            float* po = d.out + r * d.ld_out + col;
            if (ovec && col + 4 <= out_dim) {
              *reinterpret_cast<f32x4*>(po) = v;
            } else {
              if (col + 0 < out_dim) po[0] = v.x;
              if (col + 1 < out_dim) po[1] = v.y;
              if (col + 2 < out_dim) po[2] = v.z;
              if (col + 3 < out_dim) po[3] = v.w;
            }
          }
        }
      }
    }
  }
  wave_lds_fence();
}

// ---- helpers of the FAST (ReLU, aligned, unconditional-load) kernels ---------------------------------
constexpr int NP = RPW / 4;  // row groups per staging pass: 16 lanes x 16 B per row, 4 rows per instruction

// Lane id recomputed where it is needed (2 VALU), opaque to the compiler: the staging roles derived from it (c4, rs,
// byte offsets, LDS addresses) then have short live ranges instead of being hoisted out of the tile loop, carried
// across it and - in the kernels that sit at their register limit - spilled to scratch and reloaded inside it.
__device__ __forceinline__ int fresh_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

__device__ __forceinline__ void compiler_lds_barrier() {
  // LDS operations of one wave execute in issue order, so a later ds_read sees an earlier
  // ds_write of another lane without any wait; only the COMPILER must not reorder them.
  asm volatile("" ::: "memory");
}

// 16-B store of the lanes whose bit is set in `mask`, issued behind the compiler's back.
__device__ __forceinline__ void hidden_store_b128(float* p, f32x4 v, unsigned long long mask) {
  unsigned long long saved;
  asm volatile(
      "s_and_saveexec_b64 %0, %1\n\t"
      "global_store_dwordx4 %2, %3, off\n\t"
      "s_mov_b64 exec, %0\n\t"
      "s_nop 1"
      : "=&s"(saved)
      : "s"(mask), "v"(p), "v"(v)
      : "memory");
}

// one float of the lanes in `mask`, same reason as hidden_store_b128
__device__ __forceinline__ void hidden_store_b32(float* p, float v, unsigned long long mask) {
  unsigned long long saved;
  asm volatile(
      "s_and_saveexec_b64 %0, %1\n\t"
      "global_store_dword %2, %3, off\n\t"
      "s_mov_b64 exec, %0\n\t"
      "s_nop 1"
      : "=&s"(saved)
      : "s"(mask), "v"(p), "v"(v)
      : "memory");
}

// ---- row windows: buffer addressing ------------------------------------------------------------------
// A "window" is a raw buffer descriptor (4 SGPRs, built by the scalar unit) that starts at row `row0` of a
// row-major table and ends where the table ends.  Accesses name a per-lane BYTE offset inside the window,
// so the vector unit spends nothing on addresses (flat global_load needs a 64-bit v_mad / v_lshl_add per
// access, and fp32 MFMA time and VALU time add up on gfx950), and the hardware bounds check replaces the
// tail handling: loads past the table's end return 0, stores past it are dropped.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_window(const float* base, int64_t row0, int64_t rows, int ld) {
  int64_t rem = (rows - row0) * (int64_t)ld * 4;  // bytes from the window's first row to the end of the table
  rem = rem < 0 ? 0 : (rem > 0xffffffffll ? 0xffffffffll : rem);
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + row0 * ld), 0, (int)(uint32_t)rem, 0x00020000);
}

__device__ __forceinline__ f32x4 window_load(__amdgpu_buffer_rsrc_t w, uint32_t byte_off) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  return __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(w, byte_off, 0, 0));
}

// same, for rows that are read exactly once (nontemporal: keeps the stream out of L2 / MALL)
__device__ __forceinline__ f32x4 window_load_stream(__amdgpu_buffer_rsrc_t w, uint32_t byte_off) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  return __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(w, byte_off, 0, 2 /* nt */));
}

// The forward kernels store their output tiles nontemporal (STREAM_OUT): measured on the same box, c3 forward
// 8.98 -> 8.83 ms; the K8 emits, which the next kernels read back at once, lose with it and keep the default.

// 16-B store into a window, issued behind the compiler's back (same reason as hidden_store_b128)
template <bool NT = false>
__device__ __forceinline__ void hidden_window_store(f32x4 v, uint32_t byte_off, __amdgpu_buffer_rsrc_t w) {
  // s_nop 4 first: the descriptor SGPRs may just have been restored by v_readlane (SGPR spill), and a VALU
  // write of an SGPR needs 5 wait states before a VMEM instruction reads it - the hazard recognizer does not
  // look inside inline asm.  s_nop 1 after: the store data registers may be overwritten right away.
  if constexpr (NT)
    asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, 0 offen nt\n\ts_nop 1" : : "v"(v), "v"(byte_off), "s"(w) : "memory");
  else
    asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" : : "v"(v), "v"(byte_off), "s"(w) : "memory");
}

// Descriptor for a FULL tile (every row inside the table), as cheap as it gets: below 4 GiB a window over the
// whole table - loop-invariant, no per-tile 64-bit arithmetic - plus the tile's first row as a 32-bit byte offset
// for the caller's lane offset (one s_mul + one v_add per tile); larger tables get a window at the tile's first
// row.  Every instruction of a wave that is not an MFMA crawls at ~1 per 37 cycles while its SIMD mate streams
// MFMAs (tools/hw_probe.hip), so the scalar instructions of the tile loop count like vector ones.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t full_tile_window(const float* base, int64_t row0, int64_t rows, int ld,
                                                                   uint32_t* row0_bytes) {
  const int64_t tbytes = rows * (int64_t)ld * 4;
  if (tbytes <= 0xffffffffll) {
    *row0_bytes = (uint32_t)row0 * (uint32_t)(ld * 4);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)(uint32_t)tbytes, 0x00020000);
  }
  *row0_bytes = 0;
  return row_window(base, row0, rows, ld);
}

// A tile's eight 4-row pieces through windows.  Full tiles: ONE window, the piece offset rides in the
// instruction's SGPR offset (1 SALU per piece, no VGPR, but that field is not bounds-checked).  The last tile
// and anything beyond the table: one window per piece, so the bounds check sees the row.
template <bool STREAM = false, int PIECES = 8>
__device__ __forceinline__ void load_tile_rows(f32x4 (&pre)[PIECES], const float* base, int ld, int64_t row0, int64_t rows,
                                               uint32_t lane_off) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  if (row0 + 4 * PIECES <= rows) {
    uint32_t r0b;
    const __amdgpu_buffer_rsrc_t w = full_tile_window(base, row0, rows, ld, &r0b);
    const uint32_t voff = lane_off + r0b;
#pragma unroll
    for (int p = 0; p < PIECES; ++p)
      pre[p] = __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(w, voff, p * 16 * ld, STREAM ? 2 : 0));
  } else {
#pragma unroll
    for (int p = 0; p < PIECES; ++p)
      pre[p] = __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(row_window(base, row0 + 4 * p, rows, ld),
                                                                                     lane_off, 0, STREAM ? 2 : 0));
  }
}

template <bool NT = false>
__device__ __forceinline__ void hidden_window_store_s(f32x4 v, uint32_t byte_off, __amdgpu_buffer_rsrc_t w, uint32_t soff) {
  // s_nop 4: VALU-written SGPR (v_readlane restore) -> VMEM read needs 5 wait states, see hidden_window_store
  if constexpr (NT)
    asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen nt\n\ts_nop 1" : : "v"(v), "v"(byte_off), "s"(w), "s"(soff) : "memory");
  else
    asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" : : "v"(v), "v"(byte_off), "s"(w), "s"(soff) : "memory");
}

// one output row group (declared below)
__device__ __forceinline__ void store_row_piece(float* rowp, int col, f32x4 v, bool row_ok, int out_dim, bool vec_out);

// The wave's staged [4*PIECES][64] LDS tile -> columns [0, width) of rows row0.. of dst ([rows, ld]; dst may point at a
// column offset inside a wider tensor), optionally + the same rows of `add` ([rows, ld_add], first 64 columns).
// Vector case (width % 4 == 0, ld % 4 == 0, 16-B aligned): windows as in load_tile_rows, no VALU per store,
// rows past the end dropped by the bounds check; otherwise the masked scalar path.
template <int PIECES = NP, bool STREAM_OUT = false>
__device__ __forceinline__ void store_staged_rows(const float* abuf, float* dst, int ld, int width, int row0, int rows,
                                                  int c4, int rs, const float* add = nullptr, int ld_add = 0) {
  const int col = c4 * 4;
  const bool vec = (width % 4 == 0) && (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0);
  const float* src = abuf + rs * LDSW + col;
  if (add) {  // rows this tile has read before: L2 hits; rows past the end read as 0
    f32x4 v[PIECES];
    const uint32_t aoff = (uint32_t)(rs * ld_add + (col < width ? col : 0)) * 4u;  // lanes beyond `width` are not stored
    load_tile_rows<false, PIECES>(v, add, ld_add, row0, rows, aoff);
    if (vec) {
      const uint32_t off = (uint32_t)(rs * ld + col) * 4u;
      if (col < width) {
        if (row0 + 4 * PIECES <= rows) {
          uint32_t r0b;
          const __amdgpu_buffer_rsrc_t w = full_tile_window(dst, row0, rows, ld, &r0b);
#pragma unroll
          for (int p = 0; p < PIECES; ++p) {
            const f32x4 o = v[p] + *reinterpret_cast<const f32x4*>(src + p * 4 * LDSW);
            hidden_window_store_s<STREAM_OUT>(o, off + r0b, w, (uint32_t)(p * 16 * ld));
          }
        } else {
#pragma unroll
          for (int p = 0; p < PIECES; ++p)
            hidden_window_store(v[p] + *reinterpret_cast<const f32x4*>(src + p * 4 * LDSW), off, row_window(dst, row0 + 4 * p, rows, ld));
        }
      }
    } else {
#pragma unroll
      for (int p = 0; p < PIECES; ++p) {
        const int r = row0 + p * 4 + rs;
        store_row_piece(dst + (int64_t)(r < rows ? r : rows - 1) * ld, col, v[p] + *reinterpret_cast<const f32x4*>(src + p * 4 * LDSW),
                        r < rows && col < width, width, false);
      }
    }
    return;
  }
  if (vec) {  // four pieces at a time: their LDS reads fly together (one round trip per group, not per piece)
    const uint32_t off = (uint32_t)(rs * ld + col) * 4u;
    if (col < width) {
      if (row0 + 4 * PIECES <= rows) {
        uint32_t r0b;
        const __amdgpu_buffer_rsrc_t w = full_tile_window(dst, row0, rows, ld, &r0b);
        constexpr int GRP = PIECES < 4 ? PIECES : 4;
#pragma unroll
        for (int p0 = 0; p0 < PIECES; p0 += GRP) {
          f32x4 o[GRP];
#pragma unroll
          for (int p = 0; p < GRP; ++p) o[p] = *reinterpret_cast<const f32x4*>(src + (p0 + p) * 4 * LDSW);
#pragma unroll
          for (int p = 0; p < GRP; ++p) hidden_window_store_s<STREAM_OUT>(o[p], off + r0b, w, (uint32_t)((p0 + p) * 16 * ld));
        }
      } else {
#pragma unroll
        for (int p = 0; p < PIECES; ++p)
          hidden_window_store(*reinterpret_cast<const f32x4*>(src + p * 4 * LDSW), off, row_window(dst, row0 + 4 * p, rows, ld));
      }
    }
  } else {
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
      const int r = row0 + p * 4 + rs;
      store_row_piece(dst + (int64_t)(r < rows ? r : rows - 1) * ld, col, *reinterpret_cast<const f32x4*>(src + p * 4 * LDSW),
                      r < rows && col < width, width, false);
    }
  }
}

// Whole-row pieces that are already in registers (lane (rs, c4): 16 B of row p*4+rs for p < PIECES) -> columns
// [0, width) of rows row0.. of dst: the register form of store_staged_rows.  Reading all pieces of a tile from LDS
// first and issuing the stores back to back costs ~300 cycles per 32-row tile where the one-piece-at-a-time form
// (LDS read, wait, s_nop, store, eight times) measured ~1,200 in the fused K8 kernel's phase probe.
template <int PIECES = NP, bool STREAM_OUT = false>
__device__ __forceinline__ void store_row_pieces(const f32x4 (&v)[PIECES], float* dst, int ld, int width, int row0, int rows,
                                                 int c4, int rs) {
  const int col = c4 * 4;
  const bool vec = (width % 4 == 0) && (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0);
  if (vec) {
    const uint32_t off = (uint32_t)(rs * ld + col) * 4u;
    if (col < width) {
      if (row0 + 4 * PIECES <= rows) {
        uint32_t r0b;
        const __amdgpu_buffer_rsrc_t w = full_tile_window(dst, row0, rows, ld, &r0b);
#pragma unroll
        for (int p = 0; p < PIECES; ++p) hidden_window_store_s<STREAM_OUT>(v[p], off + r0b, w, (uint32_t)(p * 16 * ld));
      } else {
#pragma unroll
        for (int p = 0; p < PIECES; ++p) hidden_window_store(v[p], off, row_window(dst, row0 + 4 * p, rows, ld));
      }
    }
  } else {
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
      const int r = row0 + p * 4 + rs;
      store_row_piece(dst + (int64_t)(r < rows ? r : rows - 1) * ld, col, v[p], r < rows && col < width, width, false);
    }
  }
}

// one output row group: 16-B store when the row layout allows it, else up to four masked scalar stores
__device__ __forceinline__ void store_row_piece(float* rowp, int col, f32x4 v, bool row_ok, int out_dim, bool vec_out) {
  if (vec_out) {
    const bool ok = row_ok && col < out_dim;
    hidden_store_b128(rowp + (ok ? col : 0), v, __ballot(ok));
  } else {
    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool ok = row_ok && col + k < out_dim;
      hidden_store_b32(rowp + (ok ? col + k : 0), e[k], __ballot(ok));
    }
  }
}

template <int T>
__device__ __forceinline__ void relu_tiles(f32x16 (&acc)[T]) {
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = fmaxf(acc[t][r], 0.f);
}

template <int T>
__device__ __forceinline__ void add_tile_from_lds(f32x16 (&acc)[T], const float* abuf, int i, int h) {
#pragma unroll
  for (int t = 0; t < T && t < 2; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(abuf + i * LDSW + 32 * t + 8 * q + 4 * h);
      acc[t][4 * q + 0] += v.x; acc[t][4 * q + 1] += v.y; acc[t][4 * q + 2] += v.z; acc[t][4 * q + 3] += v.w;
    }
}

template <int T>
__device__ __forceinline__ void tiles_to_lds(const f32x16 (&o)[T], float* abuf, int i, int h) {
#pragma unroll
  for (int t = 0; t < T && t < 2; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v = {o[t][4 * q + 0], o[t][4 * q + 1], o[t][4 * q + 2], o[t][4 * q + 3]};
      *reinterpret_cast<f32x4*>(abuf + i * LDSW + 32 * t + 8 * q + 4 * h) = v;
    }
}

// tiles [T0, T0+2) of the accumulator <-> the 64 staged columns
template <int T, int TA, int TT>
__device__ __forceinline__ void add_one_tile_from_lds(f32x16 (&acc)[T], const float* abuf, int i, int h) {
  if constexpr (TA < T) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(abuf + i * LDSW + 32 * TT + 8 * q + 4 * h);
      acc[TA][4 * q + 0] += v.x; acc[TA][4 * q + 1] += v.y; acc[TA][4 * q + 2] += v.z; acc[TA][4 * q + 3] += v.w;
    }
  }
}
template <int T, int T0>
__device__ __forceinline__ void add_chunk_from_lds(f32x16 (&acc)[T], const float* abuf, int i, int h) {
  add_one_tile_from_lds<T, T0, 0>(acc, abuf, i, h);
  add_one_tile_from_lds<T, T0 + 1, 1>(acc, abuf, i, h);
}

template <int T, int TA, int TT>
__device__ __forceinline__ void one_tile_to_lds(const f32x16 (&o)[T], float* abuf, int i, int h) {
  if constexpr (TA < T) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v = {o[TA][4 * q + 0], o[TA][4 * q + 1], o[TA][4 * q + 2], o[TA][4 * q + 3]};
      *reinterpret_cast<f32x4*>(abuf + i * LDSW + 32 * TT + 8 * q + 4 * h) = v;
    }
  }
}
template <int T, int T0>
__device__ __forceinline__ void chunk_to_lds(const f32x16 (&o)[T], float* abuf, int i, int h) {
  one_tile_to_lds<T, T0, 0>(o, abuf, i, h);
  one_tile_to_lds<T, T0 + 1, 1>(o, abuf, i, h);
}

struct SegView {  // wave-uniform view of one segment
  const float* ptr;
  const int32_t* index;
  int ld;
  int width;
  uint32_t bytes;  // gathered tables: size of the table (bound of the buffer window), < 4 GiB
};


// outputs of the K8 data kernels (mlp_backward.hip, mlp_backward16.hip), see gnc_mlp_bwd_desc_t
struct BwdArgs {
  const float* grad_out;
  int ld_grad_out;
  float* act[GNC_MAX_LINEAR];  // [rows, H] post-activation outputs of layers 0 .. L-2
  float* dz[GNC_MAX_LINEAR];   // [rows, width_l] grad wrt the pre-activation of layer l (l = L-1: pre-LayerNorm)
  float* dx;                   // nullable: [rows, in_dim0] grad wrt the MATMUL part of the input
  int ld_dx;
  float* yhat;                 // [rows, out_dim] normalised pre-affine output (only with LayerNorm)
  int dx_add_grad_out;         // add grad_out rows to dx (the residual path of a segment that is also the residual)
  float* ln_partial;           // nullable: [waves, 2 * out_dim] per-wave [colsum(grad_out) | colsum(grad_out * yhat)]
  // nullable: the gradient of output row r is (grad_out ? grad_out[r] : 0) + gg[gg_index[r]] (gnc_mlp_bwd_desc_t.grad_gather)
  const float* gg;
  const int* gg_index;
  int ld_gg;
  int64_t gg_rows;
  float* g_sum;  // [rows, ld_g_sum] scratch for the summed rows (both parts given and dx_add_grad_out)
  int ld_g_sum;
};

// smallest of {1,2,4,8} accumulator tiles (32 features each) covering `width`
__host__ __device__ inline int ldw_of(const gnc_mlp_desc_t& d, int l) { return d.ld_weight[l] ? d.ld_weight[l] : d.in_dim[l]; }

inline int tiles_for(int width) {
  const int t = (width + 31) / 32;
  return t <= 1 ? 1 : t <= 2 ? 2 : t <= 4 ? 4 : 8;
}

// host-side validation shared by both launchers (mlp_fused.hip)
int validate_desc(const gnc_mlp_desc_t* d, bool check_ptrs);
// resident-weights variant (mlp_resident.hip): returns GNC_ERR_UNSUPPORTED when the weights
// do not fit in LDS, in which case the caller falls through to the streaming kernel
int launch_resident_dual(const float* x, int64_t ld_x, int64_t rows, const float* wa, int64_t ld_wa, const float* wb, int64_t ld_wb,
                         int in_dim, int out_dim, float* out_a, float* out_b, int64_t ld_out, hipStream_t stream, bool* launched);
int launch_resident(const gnc_mlp_desc_t& d, int T, bool narrow_out, hipStream_t stream, bool* launched,
                    bool probe_only = false);
// streaming FAST variant (mlp_stream.hip) for widths whose weights do not fit in LDS; same contract
int launch_stream(const gnc_mlp_desc_t& d, int T, bool narrow_out, hipStream_t stream, bool* launched,
                  bool probe_only = false);
// widths 129..256 on v_mfma_f32_16x16x4_f32, 16 rows per wave (mlp_stream16.hip); same contract
int launch_stream16(const gnc_mlp_desc_t& d, hipStream_t stream, bool* launched, bool probe_only = false);
// small batches (rows <= col16_max_rows()) at up to 128 features: column-split workgroups, mlp_col16.hip
int launch_col16(const gnc_mlp_desc_t& d, hipStream_t stream, bool* launched, bool probe_only = false);
int64_t col16_max_rows();
int64_t bwd_col16_max_rows();
// K8 data kernel for small batches with saved post-activations (mlp_bwd_col16.hip): shape query / partial rows / launch
bool bwd_col16_supported(const gnc_mlp_desc_t& d);
int bwd_col16_ln_partial_rows(int64_t rows);
int launch_bwd_col16(const gnc_mlp_bwd_desc_t& bd, hipStream_t stream);
// ... and its register-resident variant for large batches at exactly 128 features (3 Linears, LayerNorm, one row-ordered table)
bool bwd_col16_persist_supported(const gnc_mlp_desc_t& d);
int bwd_col16_persist_ln_partial_rows(int64_t rows);
int launch_bwd_col16_persist(const gnc_mlp_bwd_desc_t& bd, hipStream_t stream);
// K8 data kernel for widths 129..256 on 16-row tiles (mlp_backward16.hip): shape query / launch
bool bwd_stream16_supported(const gnc_mlp_desc_t& d, bool want_dx);
int bwd_stream16_ln_partial_rows(int64_t rows);
int launch_bwd_stream16(const gnc_mlp_desc_t& d, const BwdArgs& b, hipStream_t stream, bool saved = false);

}  // namespace gnc_mlp
