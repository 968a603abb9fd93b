// Fused Adam over ONE flat parameter buffer and ONE flat gradient buffer.
//
// Replaces `optimizer.step()` of utils/train_model.py:42 (`optim.Adam(model.parameters(), lr=1e-3)`, :9): the
// reference's optimizer walks 76 parameter tensors (PyTorch's foreach path still needs ~10 launches per step
// and one host-side scalar per tensor); here every parameter is a view of one flat fp32 buffer, every gradient a
// view of another, so the whole update is one HBM-bound elementwise pass: 4 reads + 3 writes of 4 B per element
// (2.73 MB of parameters at the reference defaults -> 19 MB per step, ~3 us at HBM rate: launch-bound).
//
// The step counter lives on the DEVICE so that the launch sequence is identical on every step and can be replayed
// from a captured hipGraph: a one-thread kernel advances it and derives the two bias-correction scalars in double
// precision (as Python does for torch.optim.Adam), the element kernel reads them.
//
// Arithmetic order follows torch.optim.Adam (amsgrad off, maximize off), so that training runs agree with the
// reference's to fp32 rounding:
//   g   = grad + weight_decay * p                         (weight_decay == 0 in the reference)
//   m   = m + (1 - beta1) * (g - m)                       exp_avg.lerp_(g, 1 - beta1)
//   v   = v * beta2 + (1 - beta2) * g * g                 exp_avg_sq.mul_(beta2).addcmul_(g, g, 1 - beta2)
//   den = sqrt(v) / sqrt(1 - beta2^t) + eps
//   p   = p - (lr / (1 - beta1^t)) * (m / den)            param.addcdiv_(m, den, -step_size)
#include <math.h>

#include "gnc_common.h"

namespace {

__global__ void adam_tick_kernel(int64_t* __restrict__ step, float* __restrict__ scal, float lr, float beta1, float beta2) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const int64_t t = *step + 1;
    *step = t;
    const double bc1 = 1.0 - pow((double)beta1, (double)t);
    const double bc2 = 1.0 - pow((double)beta2, (double)t);
    scal[0] = (float)((double)lr / bc1);  // step_size
    scal[1] = (float)sqrt(bc2);           // bias_correction2_sqrt
  }
}

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float step_size, float bc2_sqrt, float beta1,
                                         float beta2, float eps, float wd) {
  // one rounding per ATen kernel of the foreach path (lerp_, mul_, addcmul_, sqrt, div_, add_, addcdiv_); the
  // multiply-adds inside one of those kernels are fused there too
  if (wd != 0.f) g = fmaf(wd, p, g);
  m = fmaf(1.f - beta1, g - m, m);
  v = fmaf(__fmul_rn(1.f - beta2, g), g, __fmul_rn(v, beta2));
  const float den = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), bc2_sqrt), eps);
  p = fmaf(-step_size, __fdiv_rn(m, den), p);
}

__global__ __launch_bounds__(256) void adam_flat_kernel(float* __restrict__ param, const float* __restrict__ grad,
                                                        float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq, int64_t n,
                                                        const float* __restrict__ scal, float beta1, float beta2, float eps,
                                                        float wd) {
  const float step_size = scal[0], bc2_sqrt = scal[1];
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const f4 pv = reinterpret_cast<f4*>(param)[i], gv = reinterpret_cast<const f4*>(grad)[i];
    const f4 mv = reinterpret_cast<f4*>(exp_avg)[i], vv = reinterpret_cast<f4*>(exp_avg_sq)[i];
    float pe[4] = {pv.x, pv.y, pv.z, pv.w}, me[4] = {mv.x, mv.y, mv.z, mv.w}, ve[4] = {vv.x, vv.y, vv.z, vv.w};
    const float ge[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) adam_one(pe[k], ge[k], me[k], ve[k], step_size, bc2_sqrt, beta1, beta2, eps, wd);
    const f4 p = {pe[0], pe[1], pe[2], pe[3]}, m = {me[0], me[1], me[2], me[3]}, v = {ve[0], ve[1], ve[2], ve[3]};
    reinterpret_cast<f4*>(param)[i] = p;
    reinterpret_cast<f4*>(exp_avg)[i] = m;
    reinterpret_cast<f4*>(exp_avg_sq)[i] = v;
  }
  // tail (n % 4 elements)
  const int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) adam_one(param[i], grad[i], exp_avg[i], exp_avg_sq[i], step_size, bc2_sqrt, beta1, beta2, eps, wd);
}

}  // namespace

extern "C" int gnc_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                                 float beta1, float beta2, float eps, float weight_decay, int64_t* step, float* scratch2,
                                 void* stream_) {
  GNC_REQUIRE(n >= 0, "gnc_adam_step_f32: negative size");
  if (n == 0) return GNC_OK;
  GNC_REQUIRE(param && grad && exp_avg && exp_avg_sq && step && scratch2, "gnc_adam_step_f32: null pointer");
  GNC_REQUIRE(gnc::aligned16(param) && gnc::aligned16(grad) && gnc::aligned16(exp_avg) && gnc::aligned16(exp_avg_sq),
              "gnc_adam_step_f32: flat buffers must be 16-B aligned");
  GNC_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f && lr >= 0.f,
              "gnc_adam_step_f32: hyper-parameters out of range");
  hipStream_t stream = (hipStream_t)stream_;
  adam_tick_kernel<<<1, 64, 0, stream>>>(step, scratch2, lr, beta1, beta2);
  int rc = gnc::check_launch("adam_tick_kernel");
  if (rc) return rc;
  int64_t blocks = gnc::ceil_div(gnc::ceil_div(n, 4), 256);
  const int64_t cap = (int64_t)gnc::num_cu() * 8;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  adam_flat_kernel<<<dim3((unsigned)blocks), 256, 0, stream>>>(param, grad, exp_avg, exp_avg_sq, n, scratch2, beta1, beta2, eps,
                                                               weight_decay);
  return gnc::check_launch("adam_flat_kernel");
}
