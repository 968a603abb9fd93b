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
// Geometry.  Workgroup = 4 waves (one per SIMD); each wave owns 32 data rows of a 128-row
// tile and ALL features (HT accumulator tiles of 32 features).  Weights stream L2 -> LDS in
// [features][64 k] chunks shared by the four waves (2 barriers per chunk); the first layer's
// input rows (gathered by index when the segment has one) are staged per wave into a
// private [32][64] LDS tile with whole-row 16-B loads.  LDS rows are padded to 68 floats
// (4*odd) which makes every ds_read_b128 fragment read conflict-free.  The LayerNorm
// reduction is over registers plus one cross-half lane exchange; the epilogue transposes
// through the wave's private tile so residual loads and output stores are whole rows.
#include "gnc_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KC = 64;      // k-chunk per staging step
constexpr int LDSW = 68;    // padded LDS row (floats)
constexpr int RPW = 32;     // data rows per wave
constexpr int WAVES = 4;
constexpr int RPB = RPW * WAVES;  // data rows per workgroup tile

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
// matrix.  16 lanes cover one 64-float row, so a 256-thread pass covers 16 weight rows.
template <int TROWS>
__device__ __forceinline__ void stage_weights(float* wbuf, const float* __restrict__ W, int ldw, int out_dim,
                                              int kbase, int klimit, int ncol4, bool vec_ok, int tid) {
  const int c4 = tid & 15;
  const int r0 = tid >> 4;
  if (c4 < ncol4) {
#pragma unroll
    for (int p = 0; p < TROWS / 16; ++p) {
      const int n = p * 16 + r0;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (n < out_dim) v = load4_guarded(W + (int64_t)n * ldw + kbase + c4 * 4, kbase + c4 * 4, klimit, vec_ok);
      *reinterpret_cast<f32x4*>(wbuf + n * LDSW + c4 * 4) = v;
    }
  }
}

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
        v = load4_guarded(seg.ptr + tr * seg.ld + c0 + c4 * 4, c0 + c4 * 4, seg.width, vec_ok);
      }
      *reinterpret_cast<f32x4*>(abuf + j * LDSW + c4 * 4) = v;
    }
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

// dst[TO] (+)= W_l * src[TI]   (src = previous layer's activations, in registers)
template <int TI, int TO>
__device__ __forceinline__ void linear_from_regs(f32x16 (&dst)[TO], const f32x16 (&src)[TI], float* wbuf,
                                                 const float* __restrict__ W, int in_dim, int out_dim, bool vec_ok,
                                                 int tid, int i, int h) {
  constexpr int NCH = (TI + 1) / 2;  // 64-wide k chunks covering TI tiles
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if (c * KC < in_dim) {  // workgroup-uniform
      stage_weights<TO * 32>(wbuf, W, in_dim, out_dim, c * KC, in_dim, 16, vec_ok, tid);
      __syncthreads();
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
        }
      }
      __syncthreads();
    }
  }
}

template <int HT, int OT>
__global__ __launch_bounds__(256) void mlp_fused_kernel(const gnc_mlp_desc_t d, const int64_t num_tiles) {
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

  for (int idx = tid; idx < (L + 2) * PSTRIDE; idx += 256) {
    const int l = idx / PSTRIDE, n = idx - l * PSTRIDE;
    float v = 0.f;
    if (l < L) {
      if (d.bias[l] && n < d.out_dim[l]) v = d.bias[l][n];
    } else if (d.ln_gamma && n < out_dim) {
      v = (l == L) ? d.ln_gamma[n] : (d.ln_beta ? d.ln_beta[n] : 0.f);
    }
    pbuf[idx] = v;
  }
  __syncthreads();

  // vector-load eligibility of each operand (workgroup-uniform)
  unsigned wvec = 0;  // bit l: layer l's weight rows can be read with 16-B loads
  for (int l = 0; l < L; ++l)
    if ((d.in_dim[l] % 4 == 0) && ((reinterpret_cast<uintptr_t>(d.weight[l]) & 15u) == 0)) wvec |= 1u << l;

  for (int64_t tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
    const int64_t row0 = tile * RPB + wave * RPW;

    // ------------------------------------------------------------------ layer 0
    f32x16 hid[HT];
    init_bias<HT>(hid, pbuf, h);
    {
      int koff = 0;
      for (int s = 0; s < d.num_segments; ++s) {
        const gnc_mlp_segment_t seg = d.seg[s];
        const bool avec = (seg.ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(seg.ptr) & 15u) == 0);
        const bool w0vec = (wvec & 1u) && (koff % 4 == 0);
        for (int c0 = 0; c0 < seg.width; c0 += KC) {
          const int kc = seg.width - c0 < KC ? seg.width - c0 : KC;
          const int kc8 = (kc + 7) >> 3;
          stage_weights<HT * 32>(wbuf, d.weight[0], d.in_dim[0], d.out_dim[0], koff + c0, koff + seg.width, kc8 * 2,
                                 w0vec, tid);
          stage_rows(abuf, seg, row0, d.rows, c0, kc8 * 2, avec, lane);
          __syncthreads();
#pragma unroll 2
          for (int g = 0; g < kc8; ++g) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(abuf + i * LDSW + 8 * g + 4 * h);
#pragma unroll
            for (int t = 0; t < HT; ++t) {
              const f32x4 a = *reinterpret_cast<const f32x4*>(wbuf + (32 * t + i) * LDSW + 8 * g + 4 * h);
              hid[t] = mfma(a.x, b.x, hid[t]);
              hid[t] = mfma(a.y, b.y, hid[t]);
              hid[t] = mfma(a.z, b.z, hid[t]);
              hid[t] = mfma(a.w, b.w, hid[t]);
            }
          }
          __syncthreads();
        }
        koff += seg.width;
      }
    }
    activate_tiles<HT>(hid, d.activation, d.act_param);

    // ------------------------------------------------------------------ hidden layers 1 .. L-2
    for (int l = 1; l < L - 1; ++l) {
      f32x16 nxt[HT];
      init_bias<HT>(nxt, pbuf + l * PSTRIDE, h);
      linear_from_regs<HT, HT>(nxt, hid, wbuf, d.weight[l], d.in_dim[l], d.out_dim[l], (wvec >> l) & 1u, tid, i, h);
      activate_tiles<HT>(nxt, d.activation, d.act_param);
#pragma unroll
      for (int t = 0; t < HT; ++t) hid[t] = nxt[t];
    }

    // ------------------------------------------------------------------ last Linear
    f32x16 o[OT];
    init_bias<OT>(o, pbuf + (L - 1) * PSTRIDE, h);
    linear_from_regs<HT, OT>(o, hid, wbuf, d.weight[L - 1], d.in_dim[L - 1], out_dim, (wvec >> (L - 1)) & 1u, tid, i, h);

    // ------------------------------------------------------------------ LayerNorm over the features of each data row
    if (d.ln_gamma) {
      float s = 0.f;
#pragma unroll
      for (int t = 0; t < OT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += (feat_of(t, r, h) < out_dim) ? o[t][r] : 0.f;
      s += __shfl_xor(s, 32, 64);
      const float mean = s / (float)out_dim;
      float v = 0.f;
#pragma unroll
      for (int t = 0; t < OT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float dlt = o[t][r] - mean;
          v += (feat_of(t, r, h) < out_dim) ? dlt * dlt : 0.f;
        }
      v += __shfl_xor(v, 32, 64);
      const float rstd = 1.f / sqrtf(v / (float)out_dim + d.ln_eps);
      const float* pg = pbuf + L * PSTRIDE;
      const float* pbt = pbuf + (L + 1) * PSTRIDE;
#pragma unroll
      for (int t = 0; t < OT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 gm = *reinterpret_cast<const f32x4*>(pg + 32 * t + 8 * q + 4 * h);
          const f32x4 bt = *reinterpret_cast<const f32x4*>(pbt + 32 * t + 8 * q + 4 * h);
          o[t][4 * q + 0] = (o[t][4 * q + 0] - mean) * rstd * gm.x + bt.x;
          o[t][4 * q + 1] = (o[t][4 * q + 1] - mean) * rstd * gm.y + bt.y;
          o[t][4 * q + 2] = (o[t][4 * q + 2] - mean) * rstd * gm.z + bt.z;
          o[t][4 * q + 3] = (o[t][4 * q + 3] - mean) * rstd * gm.w + bt.w;
        }
    }

    // ------------------------------------------------------------------ epilogue: transpose through the wave's tile,
    // add the residual and store whole rows
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
              if (d.residual) {
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
}

int tiles_for(int width) {  // smallest of {1,2,4,8} covering width/32
  const int t = (width + 31) / 32;
  return t <= 1 ? 1 : t <= 2 ? 2 : t <= 4 ? 4 : 8;
}

size_t lds_bytes(int WT, int L) { return ((size_t)(WT * 32 + RPB) * LDSW + (size_t)(L + 2) * WT * 32) * sizeof(float); }

int validate(const gnc_mlp_desc_t* d, bool check_ptrs) {
  if (!d) { gnc::set_error("gnc_mlp: null descriptor"); return GNC_ERR_INVALID_ARGUMENT; }
  if (d->num_segments < 1 || d->num_segments > GNC_MAX_SEGMENTS || d->num_linear < 2 || d->num_linear > GNC_MAX_LINEAR) {
    gnc::set_error("gnc_mlp: need 1..%d segments and 2..%d Linear layers (got %d, %d)", GNC_MAX_SEGMENTS,
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
    in0 += d->seg[s].width;
  }
  if (in0 != d->in_dim[0]) {
    gnc::set_error("gnc_mlp: segment widths sum to %lld but in_dim[0] is %d", (long long)in0, d->in_dim[0]);
    return GNC_ERR_INVALID_ARGUMENT;
  }
  const int H = d->out_dim[0];
  for (int l = 0; l < d->num_linear; ++l) {
    if (d->in_dim[l] < 1 || d->out_dim[l] < 1) { gnc::set_error("gnc_mlp: layer %d has empty dims", l); return GNC_ERR_INVALID_ARGUMENT; }
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
  int64_t grid = (int64_t)gnc::kNumCU * (per_cu > 3 ? 3 : per_cu);
  if (grid > num_tiles) grid = num_tiles;
  mlp_fused_kernel<HT, OT><<<dim3((unsigned)grid), dim3(256), smem, stream>>>(d, num_tiles);
  return gnc::check_launch("mlp_fused_kernel");
}

}  // namespace

extern "C" int gnc_mlp_supported(const gnc_mlp_desc_t* desc) { return validate(desc, false); }

extern "C" int gnc_mlp_forward_f32(const gnc_mlp_desc_t* desc, void* stream_) {
  int rc = validate(desc, true);
  if (rc) return rc;
  if (desc->rows == 0) return GNC_OK;
  hipStream_t stream = (hipStream_t)stream_;
  const int L = desc->num_linear;
  int T = tiles_for(desc->out_dim[0]);
  const int od = desc->out_dim[L - 1];
  const bool narrow_out = od <= 32;
  if (!narrow_out && tiles_for(od) > T) T = tiles_for(od);
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
