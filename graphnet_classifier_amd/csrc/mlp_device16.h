// Device helpers of the 16-row (v_mfma_f32_16x16x4_f32) kernels: mlp_stream16.hip (forward, widths 129..256) and
// mlp_backward16.hip (its backward).  Layout: accumulator register r of tile t on lane (j = lane & 15, g = lane >> 4)
// holds feature 16t + 4g + r of data row j.
#pragma once
#include "mlp_device.h"

namespace gnc_mlp {

constexpr int R16 = 16;         // data rows per wave
constexpr int NP16 = R16 / 4;   // staging passes (4 rows per wave instruction)
constexpr int W16 = 8;          // waves per workgroup
constexpr int NT16 = W16 * 64;
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// NTL = number of 16-feature tiles (16 for 256 features)
template <int NTL>
__device__ __forceinline__ void init_bias16(f32x4 (&acc)[NTL], const float* pb, int g) {
#pragma unroll
  for (int t = 0; t < NTL; ++t) acc[t] = *reinterpret_cast<const f32x4*>(pb + 16 * t + 4 * g);
}

template <int NTL>
__device__ __forceinline__ void relu16(f32x4 (&acc)[NTL]) {
#pragma unroll
  for (int t = 0; t < NTL; ++t) {
    acc[t].x = fmaxf(acc[t].x, 0.f); acc[t].y = fmaxf(acc[t].y, 0.f);
    acc[t].z = fmaxf(acc[t].z, 0.f); acc[t].w = fmaxf(acc[t].w, 0.f);
  }
}

// acc[T0 + cb] += the staged rows' columns [16 cb, 16 cb + 16) for cb = 0..3: one 64-column slab of an additive segment.
// T0 is a template argument (the plans list additive steps slab-major and the kernels walk the slabs with a static
// index): with a run-time tile index every accumulator would go through a v_cndmask per component - ~320 vector
// instructions per slab instead of 16.
template <int NTL, int T0>
__device__ __forceinline__ void add_slab16(f32x4 (&acc)[NTL], const float* abuf, int i, int g) {
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
    if (T0 + cb < NTL) acc[T0 + cb < NTL ? T0 + cb : 0] += *reinterpret_cast<const f32x4*>(abuf + i * LDSW + 16 * cb + 4 * g);
}
// acc[t] += W_chunk[16t + i][16cb + 4g + s] * X[j][16cb + 4g + s]: one staged 64-column chunk of the first Linear
template <int NTL>
__device__ __forceinline__ void mma16_chunk_from_lds(f32x4 (&acc)[NTL], const float* abuf, const float* wbuf, int kc16,
                                                     int i, int g) {
  for (int cb = 0; cb < kc16; ++cb) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(abuf + i * LDSW + 16 * cb + 4 * g);
    const float bs[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int th = 0; th < NTL; th += 4) {  // four tiles at a time: independent accumulators between dependent steps
      f32x4 a[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] = *reinterpret_cast<const f32x4*>(wbuf + (16 * (th + u) + i) * LDSW + 16 * cb + 4 * g);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float av = s == 0 ? a[u].x : s == 1 ? a[u].y : s == 2 ? a[u].z : a[u].w;
          acc[th + u] = mfma16(av, bs[s], acc[th + u]);
        }
      __builtin_amdgcn_sched_barrier(0);  // one group of A fragments in flight at a time (register budget)
    }
  }
}

// dst[t] += W_chunk * src for the 64-column chunk c of a Linear fed by the previous layer's accumulators
template <int NTI, int NTO, int C>
__device__ __forceinline__ void mma16_chunk_from_regs(f32x4 (&dst)[NTO], const f32x4 (&src)[NTI], const float* wbuf,
                                                      int in_dim, int i, int g) {
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    const int ts = 4 * C + cb;
    if (ts < NTI && 16 * ts < in_dim) {
      const f32x4 sv = src[ts < NTI ? ts : 0];
      const float bs[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
      for (int th = 0; th < NTO; th += 4) {
        f32x4 a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          a[u] = *reinterpret_cast<const f32x4*>(wbuf + (16 * (th + u < NTO ? th + u : 0) + i) * LDSW + 16 * cb + 4 * g);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (th + u < NTO) {
              const float av = s == 0 ? a[u].x : s == 1 ? a[u].y : s == 2 ? a[u].z : a[u].w;
              dst[th + u] = mfma16(av, bs[s], dst[th + u]);
            }
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

// Same instruction diet as layer_norm_tiles (mlp_device.h): packed fp32 math, features >= out_dim hold exact
// zeros and drop out of the sums without selects, 1/n as a multiplier, v_rsq_f32, cross-lane sums on the VALU.
template <int NTL>
__device__ __forceinline__ void layer_norm16(f32x4 (&o)[NTL], const float* pg, const float* pbt, int out_dim, float eps,
                                             int g) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const float inv_n = 1.f / (float)out_dim;
  f32x2 sa = {0.f, 0.f}, sb = {0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NTL; ++t) {
    sa += f32x2{o[t].x, o[t].y};
    sb += f32x2{o[t].z, o[t].w};
  }
  const float mean = add_quarters((sa.x + sa.y) + (sb.x + sb.y)) * inv_n;
  const f32x4 m4 = {mean, mean, mean, mean};
#pragma unroll
  for (int t = 0; t < NTL; ++t) o[t] -= m4;
  if (out_dim != NTL * 16) {  // wave-uniform and rare: clear the deviations of the padding features
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
      const int f = 16 * t + 4 * g;
      o[t].x = f + 0 < out_dim ? o[t].x : 0.f; o[t].y = f + 1 < out_dim ? o[t].y : 0.f;
      o[t].z = f + 2 < out_dim ? o[t].z : 0.f; o[t].w = f + 3 < out_dim ? o[t].w : 0.f;
    }
  }
  f32x2 va = {0.f, 0.f}, vb = {0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NTL; ++t) {
    const f32x2 da = {o[t].x, o[t].y}, db = {o[t].z, o[t].w};
    va = __builtin_elementwise_fma(da, da, va);
    vb = __builtin_elementwise_fma(db, db, vb);
  }
  const float rstd = __frsqrt_rn(add_quarters((va.x + va.y) + (vb.x + vb.y)) * inv_n + eps);
  const f32x4 r4 = {rstd, rstd, rstd, rstd};
#pragma unroll
  for (int t = 0; t < NTL; ++t) {
    const f32x4 gm = *reinterpret_cast<const f32x4*>(pg + 16 * t + 4 * g);
    const f32x4 bt = *reinterpret_cast<const f32x4*>(pbt + 16 * t + 4 * g);
    o[t] = __builtin_elementwise_fma(o[t], gm * r4, bt);
  }
}


}  // namespace gnc_mlp
