// Row-wise / elementwise pieces of the backward of an MLP with an activation OTHER than ReLU (models/MLP.py:21 accepts
// any nn.<Name>; the reference's entry points use ReLU only, for which the fused K8 kernels exist).  With these three
// kernels the layer-by-layer backward of such an MLP runs entirely on this library - single-Linear K4 launches for the
// products, gnc_xty_f32 for the weight gradients - instead of on PyTorch-ROCm ops (no fallback):
//   gnc_activation_f32            a  = act(z)                        (recomputing the hidden activations)
//   gnc_activation_backward_f32   dz = da * act'(z)
//   gnc_layer_norm_backward_f32   dy, y_hat of  out = LayerNorm(y) * gamma + beta   (d gamma / d beta: gnc_colsum_pair_f32)
// HBM-bound, one pass each; nothing here is on the path of a ReLU model.
#include "gnc_common.h"

namespace {

__device__ __forceinline__ float act_fwd(float x, int act, float p) {
  switch (act) {
    case GNC_ACT_RELU: return x < 0.f ? 0.f : x;
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

// d act(x) / dx, the formulas torch.autograd uses for the nn.<Name> modules
__device__ __forceinline__ float act_grad(float x, int act, float p) {
  switch (act) {
    case GNC_ACT_RELU: return x > 0.f ? 1.f : 0.f;
    case GNC_ACT_IDENTITY: return 1.f;
    case GNC_ACT_TANH: { const float t = tanhf(x); return 1.f - t * t; }
    case GNC_ACT_SIGMOID: { const float s = 1.f / (1.f + expf(-x)); return s * (1.f - s); }
    case GNC_ACT_SILU: { const float s = 1.f / (1.f + expf(-x)); return s * (1.f + x * (1.f - s)); }
    case GNC_ACT_GELU: {
      const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
      const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
      return cdf + x * pdf;
    }
    case GNC_ACT_LEAKY_RELU: return x > 0.f ? 1.f : p;
    case GNC_ACT_ELU: return x > 0.f ? 1.f : p * expf(x);
    default: return 1.f;
  }
}

template <bool BACKWARD>
__global__ __launch_bounds__(gnc::kBlock) void activation_kernel(const float* __restrict__ z, int64_t ldz, const float* __restrict__ da,
                                                                 int64_t ldd, int64_t rows, int width, int act, float p,
                                                                 float* __restrict__ out, int64_t ldo) {
  const int64_t total = rows * width;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / width;
    const int c = (int)(t - r * width);
    const float x = z[r * ldz + c];
    out[r * ldo + c] = BACKWARD ? da[r * ldd + c] * act_grad(x, act, p) : act_fwd(x, act, p);
  }
}

// one wave per row: y_hat = (y - mean) * rstd, dy = rstd * (g gamma - mean(g gamma) - y_hat mean(g gamma y_hat))
__global__ __launch_bounds__(gnc::kBlock) void layer_norm_backward_kernel(const float* __restrict__ y, int64_t ldy,
                                                                          const float* __restrict__ gamma, const float* __restrict__ g,
                                                                          int64_t ldg, int64_t rows, int width, float eps,
                                                                          float* __restrict__ dy, int64_t ldd, float* __restrict__ yhat,
                                                                          int64_t ldh) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  auto wave_sum = [](float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
  };
  const float inv_n = 1.f / (float)width;
  for (int64_t r = wave; r < rows; r += nwaves) {
    const float* yr = y + r * ldy;
    const float* gr = g + r * ldg;
    float s = 0.f;
    for (int c = lane; c < width; c += 64) s += yr[c];
    const float mean = wave_sum(s) * inv_n;
    float v = 0.f;
    for (int c = lane; c < width; c += 64) {
      const float dlt = yr[c] - mean;
      v += dlt * dlt;
    }
    const float rstd = 1.f / sqrtf(wave_sum(v) * inv_n + eps);
    float m1 = 0.f, m2 = 0.f;
    for (int c = lane; c < width; c += 64) {
      const float yh = (yr[c] - mean) * rstd;
      const float gg = gr[c] * gamma[c];
      m1 += gg;
      m2 += gg * yh;
    }
    m1 = wave_sum(m1) * inv_n;
    m2 = wave_sum(m2) * inv_n;
    for (int c = lane; c < width; c += 64) {
      const float yh = (yr[c] - mean) * rstd;
      const float gg = gr[c] * gamma[c];
      dy[r * ldd + c] = rstd * (gg - m1 - yh * m2);
      yhat[r * ldh + c] = yh;
    }
  }
}

int grid_for(int64_t n) {
  int64_t g = gnc::ceil_div(n > 0 ? n : 1, gnc::kBlock);
  const int64_t cap = gnc::num_cu() * 16;
  return (int)(g < cap ? g : cap);
}

}  // namespace

extern "C" int gnc_activation_f32(const float* z, int64_t ld_z, int64_t rows, int32_t width, int32_t activation, float act_param,
                                  float* out, int64_t ld_out, void* stream) {
  GNC_REQUIRE(rows >= 0 && width >= 0 && ld_z >= width && ld_out >= width, "gnc_activation_f32: bad sizes");
  GNC_REQUIRE(activation >= GNC_ACT_RELU && activation <= GNC_ACT_ELU, "gnc_activation_f32: unknown activation %d", activation);
  if (rows == 0 || width == 0) return GNC_OK;
  GNC_REQUIRE(z && out, "gnc_activation_f32: null pointer");
  activation_kernel<false><<<grid_for(rows * width), gnc::kBlock, 0, (hipStream_t)stream>>>(z, ld_z, nullptr, 0, rows, width, activation,
                                                                                           act_param, out, ld_out);
  return gnc::check_launch("activation_kernel");
}

extern "C" int gnc_activation_backward_f32(const float* z, int64_t ld_z, const float* grad_act, int64_t ld_grad, int64_t rows,
                                           int32_t width, int32_t activation, float act_param, float* grad_z, int64_t ld_out,
                                           void* stream) {
  GNC_REQUIRE(rows >= 0 && width >= 0 && ld_z >= width && ld_grad >= width && ld_out >= width, "gnc_activation_backward_f32: bad sizes");
  GNC_REQUIRE(activation >= GNC_ACT_RELU && activation <= GNC_ACT_ELU, "gnc_activation_backward_f32: unknown activation %d", activation);
  if (rows == 0 || width == 0) return GNC_OK;
  GNC_REQUIRE(z && grad_act && grad_z, "gnc_activation_backward_f32: null pointer");
  activation_kernel<true><<<grid_for(rows * width), gnc::kBlock, 0, (hipStream_t)stream>>>(z, ld_z, grad_act, ld_grad, rows, width,
                                                                                          activation, act_param, grad_z, ld_out);
  return gnc::check_launch("activation_kernel");
}

extern "C" int gnc_layer_norm_backward_f32(const float* y, int64_t ld_y, const float* gamma, const float* grad_out, int64_t ld_grad,
                                           int64_t rows, int32_t width, float eps, float* grad_y, int64_t ld_gy, float* yhat,
                                           int64_t ld_yhat, void* stream) {
  GNC_REQUIRE(rows >= 0 && width >= 1 && ld_y >= width && ld_grad >= width && ld_gy >= width && ld_yhat >= width,
              "gnc_layer_norm_backward_f32: bad sizes");
  if (rows == 0) return GNC_OK;
  GNC_REQUIRE(y && gamma && grad_out && grad_y && yhat, "gnc_layer_norm_backward_f32: null pointer");
  layer_norm_backward_kernel<<<grid_for(rows * 64), gnc::kBlock, 0, (hipStream_t)stream>>>(y, ld_y, gamma, grad_out, ld_grad, rows, width,
                                                                                          eps, grad_y, ld_gy, yhat, ld_yhat);
  return gnc::check_launch("layer_norm_backward_kernel");
}
