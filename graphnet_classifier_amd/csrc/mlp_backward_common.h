// Helpers shared by the K8 backward kernels (mlp_backward.hip, mlp_backward_fused.h): transposed products, ReLU masks,
// accumulator tiles from the wave's LDS tile.  Every translation unit gets its own copy (anonymous namespace).
#pragma once
#include "mlp_device.h"

using namespace gnc_mlp;

namespace {

// dst[t'] += W^T-tile * src: contraction over the feature index n of src (register r of tile t holds
// n = 32t + (r&3) + 8(r>>2) + 4h); A operand = W[n][32t' + i] read down the column of the [n][k] chunk.
// FROM_ZERO: dst = product (the first MFMA of every tile takes the constant 0 as its C operand, so no register set
// has to be cleared first)
template <int TI, int TO, bool FROM_ZERO = false>
__device__ __forceinline__ void mma_transposed_from_regs(f32x16 (&dst)[TO], const f32x16 (&src)[TI], const float* wbuf,
                                                         int i, int h) {
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < TI; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
      for (int tp = 0; tp < TO; ++tp) {
        const float a = wbuf[n * LDSW + 32 * tp + i];
        dst[tp] = mfma(a, src[t][r], (FROM_ZERO && t == 0 && r == 0) ? zero : dst[tp]);
      }
    }
}

template <int T>
__device__ __forceinline__ void zero_tiles(f32x16 (&acc)[T]) {
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
}

// bit (16t + r) = acc[t][r] > 0 (T <= 2); ReLU applied in place
template <int T>
__device__ __forceinline__ unsigned relu_tiles_mask(f32x16 (&acc)[T]) {
  unsigned m = 0;
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bool pos = acc[t][r] > 0.f;
      m |= pos ? (1u << (16 * t + r)) : 0u;
      acc[t][r] = pos ? acc[t][r] : 0.f;
    }
  return m;
}

template <int T>
__device__ __forceinline__ void apply_mask(f32x16 (&acc)[T], unsigned m) {
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = (m >> (16 * t + r)) & 1u ? acc[t][r] : 0.f;
}

// acc = acc where the (post-ReLU) activation is positive, else 0: the ReLU derivative taken from the activation
// itself when that is still in registers (2 VALU per value, no bit masks to pack and unpack)
template <int T>
__device__ __forceinline__ void mask_by_positive(f32x16 (&acc)[T], const f32x16 (&act)[T]) {
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = act[t][r] > 0.f ? acc[t][r] : 0.f;
}

template <int T>
__device__ __forceinline__ void tile_from_lds(f32x16 (&acc)[T], const float* abuf, int i, int h) {
#pragma unroll
  for (int t = 0; t < T && t < 2; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(abuf + i * LDSW + 32 * t + 8 * q + 4 * h);
      acc[t][4 * q + 0] = v.x; acc[t][4 * q + 1] = v.y; acc[t][4 * q + 2] = v.z; acc[t][4 * q + 3] = v.w;
    }
}

}  // namespace
