// Read-out classifier of ONE graph (models/GNN.py:312-325, called at :340): logits = fc3(relu(fc2(relu(fc1(y))))) over the flattened
// node outputs y [F] - three matrix-vector products that PyTorch-ROCm runs as 3 GEMM + 2 clamp + 3 copy launches forward and a
// dozen backward: 8 of the 25 kernel nodes of a per-sample forward replay.  One launch each way here.
//
// Forward: workgroup r forms row r of fc1 (a coalesced dot product over F, reduced in a fixed order), stores h1[r] and takes a
// ticket; the workgroup that draws the last ticket runs fc2 and fc3 on the finished h1 (64 KB of weights at most) and resets the
// ticket counter (replay-safe).  Backward: every workgroup recomputes the tiny upper part (dz3 -> dz2 -> dz1) in LDS - 4 K
// multiply-adds - and then owns a 256-wide slice of F: dW1[:, f] = dz1 * y[f] and dy[f] = W1[:, f] . dz1; workgroup 0 also leaves
// dW3, db3, dW2, db2, db1.  Everything is summed in a fixed order (bitwise reproducible).
#include <stdlib.h>

#include "gnc_common.h"

namespace {

// nn.ReLU keeps a NaN (fmaxf would swallow it: a poisoned forward must not turn into plausible logits)
__device__ __forceinline__ float relu_keep_nan(float v) { return v < 0.f ? 0.f : v; }


constexpr int RT = 256;       // threads per workgroup
constexpr int RMAXH = 1024;   // fc1 / fc2 widths the LDS arrays are sized for
constexpr int RMAXC = 64;     // classes

__device__ __forceinline__ float block_sum(float v, float* red, int tid) {  // fixed order: lane tree, then the 4 waves in order
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  const float s = ((red[0] + red[1]) + red[2]) + red[3];
  __syncthreads();
  return s;
}

__global__ __launch_bounds__(RT) void readout_fwd_kernel(const float* __restrict__ y, int64_t F, const float* __restrict__ w1, int64_t ld1,
                                                         const float* __restrict__ b1, int H1, const float* __restrict__ w2, int64_t ld2,
                                                         const float* __restrict__ b2, int H2, const float* __restrict__ w3, int64_t ld3,
                                                         const float* __restrict__ b3, int C, float* __restrict__ h1, float* __restrict__ h2,
                                                         float* __restrict__ logits, unsigned* __restrict__ ticket) {
  __shared__ float red[4];
  __shared__ float sh1[RMAXH], sh2[RMAXH];
  __shared__ int last;
  const int tid = threadIdx.x, r = blockIdx.x;
  const float* wr = w1 + (int64_t)r * ld1;
  float acc = 0.f;
  if (F % 4 == 0 && ld1 % 4 == 0 && ((reinterpret_cast<uintptr_t>(w1) | reinterpret_cast<uintptr_t>(y)) & 15u) == 0) {
    for (int64_t f = (int64_t)tid * 4; f < F; f += RT * 4) {
      const float4 a = *reinterpret_cast<const float4*>(wr + f), v = *reinterpret_cast<const float4*>(y + f);
      acc += (a.x * v.x + a.y * v.y) + (a.z * v.z + a.w * v.w);
    }
  } else {
    for (int64_t f = tid; f < F; f += RT) acc += wr[f] * y[f];
  }
  const float s = block_sum(acc, red, tid);
  if (tid == 0) {
    h1[r] = relu_keep_nan(s + (b1 ? b1[r] : 0.f));
    __threadfence();
    last = atomicAdd(ticket, 1u) == (unsigned)(gridDim.x - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  for (int k = tid; k < H1; k += RT) sh1[k] = __builtin_nontemporal_load(h1 + k);  // written by the other workgroups: not from a stale line
  __syncthreads();
  // fc2: one wave per output row at a time (4 waves), lanes stride the row
  for (int o = tid >> 6; o < H2; o += RT / 64) {
    float a = 0.f;
    for (int k = tid & 63; k < H1; k += 64) a += w2[(int64_t)o * ld2 + k] * sh1[k];
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) a += __shfl_down(a, sft, 64);
    if ((tid & 63) == 0) {
      const float v = relu_keep_nan(a + (b2 ? b2[o] : 0.f));
      sh2[o] = v;
      h2[o] = v;
    }
  }
  __syncthreads();
  for (int o = tid >> 6; o < C; o += RT / 64) {
    float a = 0.f;
    for (int k = tid & 63; k < H2; k += 64) a += w3[(int64_t)o * ld3 + k] * sh2[k];
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) a += __shfl_down(a, sft, 64);
    if ((tid & 63) == 0) logits[o] = a + (b3 ? b3[o] : 0.f);
  }
  if (tid == 0) *ticket = 0u;  // ready for the next launch (a hipGraph replay runs this kernel again on the same counter)
}

__global__ __launch_bounds__(RT) void readout_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y, int64_t F,
                                                         const float* __restrict__ w1, int64_t ld1, int H1, const float* __restrict__ w2,
                                                         int64_t ld2, int H2, const float* __restrict__ w3, int64_t ld3, int C,
                                                         const float* __restrict__ h1, const float* __restrict__ h2, float* __restrict__ dw1,
                                                         float* __restrict__ db1, float* __restrict__ dw2, float* __restrict__ db2,
                                                         float* __restrict__ dw3, float* __restrict__ db3, float* __restrict__ dy) {
  __shared__ float sg[RMAXC], dz2[RMAXH], dz1[RMAXH], sh1[RMAXH], sh2[RMAXH];
  const int tid = threadIdx.x;
  for (int k = tid; k < C; k += RT) sg[k] = g[k];
  for (int k = tid; k < H2; k += RT) sh2[k] = h2[k];
  for (int k = tid; k < H1; k += RT) sh1[k] = h1[k];
  __syncthreads();
  // dz2[k] = (sum_c w3[c][k] g[c]) * [h2[k] > 0];   dz1[k] = (sum_o w2[o][k] dz2[o]) * [h1[k] > 0]   (ascending order of the sums)
  for (int k = tid; k < H2; k += RT) {
    float a = 0.f;
    for (int c = 0; c < C; ++c) a += w3[(int64_t)c * ld3 + k] * sg[c];
    dz2[k] = sh2[k] > 0.f ? a : 0.f;
  }
  __syncthreads();
  for (int k = tid; k < H1; k += RT) {
    float a = 0.f;
    for (int o = 0; o < H2; ++o) a += w2[(int64_t)o * ld2 + k] * dz2[o];
    dz1[k] = sh1[k] > 0.f ? a : 0.f;
  }
  __syncthreads();
  if (blockIdx.x == 0) {  // the small gradients
    for (int idx = tid; idx < C * H2; idx += RT) dw3[idx] = sg[idx / H2] * sh2[idx % H2];
    for (int idx = tid; idx < H2 * H1; idx += RT) dw2[idx] = dz2[idx / H1] * sh1[idx % H1];
    for (int k = tid; k < C; k += RT) db3[k] = sg[k];
    for (int k = tid; k < H2; k += RT) db2[k] = dz2[k];
    for (int k = tid; k < H1; k += RT) db1[k] = dz1[k];
  }
  // this workgroup's slice of F
  const int64_t f = (int64_t)blockIdx.x * RT + tid;
  if (f >= F) return;
  const float yf = y[f];
  float a = 0.f;
  for (int r = 0; r < H1; ++r) {
    const float d = dz1[r];
    dw1[(int64_t)r * F + f] = d * yf;
    a += w1[(int64_t)r * ld1 + f] * d;
  }
  if (dy) dy[f] = a;
}

}  // namespace

extern "C" int gnc_readout_forward_f32(const float* y, int64_t F, const float* w1, int64_t ld1, const float* b1, int32_t H1, const float* w2,
                                       int64_t ld2, const float* b2, int32_t H2, const float* w3, int64_t ld3, const float* b3, int32_t C,
                                       float* h1, float* h2, float* logits, uint32_t* ticket, void* stream) {
  GNC_REQUIRE(y && w1 && w2 && w3 && h1 && h2 && logits && ticket, "gnc_readout_forward_f32: null pointer");
  GNC_REQUIRE(F >= 1 && H1 >= 1 && H1 <= RMAXH && H2 >= 1 && H2 <= RMAXH && C >= 1 && C <= RMAXC && ld1 >= F && ld2 >= H1 && ld3 >= H2,
              "gnc_readout_forward_f32: widths up to %d / %d / %d classes", RMAXH, RMAXH, RMAXC);
  readout_fwd_kernel<<<dim3((unsigned)H1), dim3(RT), 0, (hipStream_t)stream>>>(y, F, w1, ld1, b1, H1, w2, ld2, b2, H2, w3, ld3, b3, C, h1, h2,
                                                                              logits, ticket);
  return gnc::check_launch("readout_fwd_kernel");
}

extern "C" int gnc_readout_backward_f32(const float* grad_logits, const float* y, int64_t F, const float* w1, int64_t ld1, int32_t H1,
                                        const float* w2, int64_t ld2, int32_t H2, const float* w3, int64_t ld3, int32_t C, const float* h1,
                                        const float* h2, float* dw1, float* db1, float* dw2, float* db2, float* dw3, float* db3, float* dy,
                                        void* stream) {
  GNC_REQUIRE(grad_logits && y && w1 && w2 && w3 && h1 && h2 && dw1 && db1 && dw2 && db2 && dw3 && db3, "gnc_readout_backward_f32: null pointer");
  GNC_REQUIRE(F >= 1 && H1 >= 1 && H1 <= RMAXH && H2 >= 1 && H2 <= RMAXH && C >= 1 && C <= RMAXC && ld1 >= F && ld2 >= H1 && ld3 >= H2,
              "gnc_readout_backward_f32: widths up to %d / %d / %d classes", RMAXH, RMAXH, RMAXC);
  const int64_t blocks = gnc::ceil_div(F, (int64_t)RT);
  GNC_REQUIRE(blocks < (1ll << 31), "gnc_readout_backward_f32: F too large");
  readout_bwd_kernel<<<dim3((unsigned)blocks), dim3(RT), 0, (hipStream_t)stream>>>(grad_logits, y, F, w1, ld1, H1, w2, ld2, H2, w3, ld3, C, h1,
                                                                                  h2, dw1, db1, dw2, db2, dw3, db3, dy);
  return gnc::check_launch("readout_bwd_kernel");
}
