// K8, small-batch variant of the data kernel (saved post-activations, widths <= 128): the backward sibling of mlp_col16.hip
// for the reference's one-graph-per-step training loop (utils/train_model.py:35-45).
//
// One workgroup per 16-row tile, the waves split the OUTPUT FEATURES of every product (wave w: features [16w, 16w + 16)):
//   z_last = a_{L-2} W_{L-1}^T + b          (only with LayerNorm: its statistics; the forward's weights as they lie)
//   dz_{L-1} = LayerNorm backward of g      (g = grad_out [+ gathered rows]; d gamma / d beta partial sums per tile)
//   dz_{l-1} = (dz_l W_l) * [a_{l-1} > 0]   for l = L-1 .. 1
//   dx       = dz_0 W_0 (+ g on the residual's columns)
// The transposed products read the SAME nn.Linear matrices: the A operand of k block cb is W[16cb + 4g + s][16w + i], one
// dword per MFMA - 32 dword loads per lane and layer instead of 8 16-B loads, the same bytes and, at 4 bytes per lane, the
// same time in the texture path; no transposed copy of the weights exists anywhere.  Every weight chunk, parameter and
// saved-activation row of the tile is requested at the top of the tile in the order it is needed (mlp_col16.hip explains
// why the loads are unconditional); three register sets hold the chunks in flight.
// Rows of dz_l / dx leave in the accumulator layout (16 B per lane); the weight gradients are formed from them and the saved
// activations by gnc_xty_f32 afterwards, as with the streaming data kernels.
#include <stdlib.h>

#include <type_traits>

#include "mlp_device16.h"

using namespace gnc_mlp;

namespace {

constexpr int CK = 128;
constexpr int NKB = CK / 16;
constexpr int NTR = 4;   // transposed chunks of one tile: (num_linear - 1) + ceil(in_dim[0] / CK) <= NTR
constexpr int NSET = 3;

struct BChunk {
  const float* w;   // first element the chunk's window starts at
  int ldw;
  int nrec;         // window bytes (0: dead chunk)
  int nk;           // contraction length (rows of the matrix for a transposed chunk, columns for the forward one)
};
struct BPlan {
  int rows, num_tiles, L, H, od, n0;
  int has_ln, has_g, has_gg, vec_dx;
  int dx_add_chunk;     // chunk of dx that also receives g (the residual's columns), -1: none
  int k_in;             // in_dim[0] (columns of dx)
  int out_dim[NTR], in_dim[NTR];
  BChunk fw;            // W_{L-1} as the forward reads it (LayerNorm statistics)
  BChunk tr[NTR];       // transposed chunks in order of use: W_{L-1} .. W_1, then the column chunks of W_0
  const float* bias_last;
  const float* gamma;
  float eps;
  const float* grad_out;
  int ld_g;
  const float* gg;
  int ld_gg;
  const int32_t* gg_index;
  uint32_t gg_rows;
  const float* act[NTR];
  float* dz[NTR];
  float* dx;
  int ld_dx;
  float* ln_partial;    // [num_tiles][2 * od]
};

__device__ __attribute__((aligned(16))) float gnc_bcol_dummy[4];

template <int NT>
__global__ __launch_bounds__(NT * 64) void mlp_bwd_col16_kernel(const BPlan p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NTV = NT * 64;
  constexpr int LDH = NT * 16 + 4;
  constexpr int CPP = NTV / 16;
  const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int L = p.L, rows = p.rows, H = p.H, od = p.od;
  float* hbuf = lds;                      // [2][16][LDH]: rows of dz_l shared between the waves, ping-pong
  float* abuf = hbuf + 2 * 16 * LDH;      // [16][LDH]: a_{L-2} of the tile (LayerNorm statistics)
  float* lnbuf = abuf + 16 * LDH;         // [4][NT][16]

  auto wload_fw = [&](f32x4 (&wr)[NKB], const BChunk& c, int j, int g) {
    const __amdgpu_buffer_rsrc_t win = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(c.w), 0, c.nrec, 0x00020000);
    const uint32_t off0 = (uint32_t)((16 * w + j) * c.ldw + 4 * g) * 4u;
#pragma unroll
    for (int cb = 0; cb < NKB; ++cb) wr[cb] = window_load(win, off0 + (uint32_t)(cb * 64));
  };
  // transposed chunk: component s of block cb = W[16cb + 4g + s][16w + i]
  auto wload_tr = [&](f32x4 (&wr)[NKB], const BChunk& c, int i, int g) {
    const __amdgpu_buffer_rsrc_t win = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(c.w), 0, c.nrec, 0x00020000);
    const uint32_t off0 = (uint32_t)(4 * g * c.ldw + 16 * w + i) * 4u;
    const uint32_t rowb = (uint32_t)c.ldw * 4u;
#pragma unroll
    for (int cb = 0; cb < NKB; ++cb) {
      const uint32_t so = (uint32_t)(16 * cb) * rowb;
      wr[cb].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(win, off0, so, 0));
      wr[cb].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(win, off0, so + rowb, 0));
      wr[cb].z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(win, off0, so + 2 * rowb, 0));
      wr[cb].w = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(win, off0, so + 3 * rowb, 0));
    }
  };
  auto mma = [&](f32x4& acc0, f32x4& acc1, const f32x4 (&wr)[NKB], const float* src, int nkb) {
#pragma unroll
    for (int c0 = 0; c0 < NT; c0 += 4) {
      f32x4 b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) b[u] = *reinterpret_cast<const f32x4*>(src + 16 * (c0 + u));
#pragma unroll
      for (int u = 0; u < 4; u += 2) {
        if (c0 + u < nkb) {
          acc0 = mfma16(wr[c0 + u].x, b[u].x, acc0);
          acc1 = mfma16(wr[c0 + u + 1].x, b[u + 1].x, acc1);
          acc0 = mfma16(wr[c0 + u].y, b[u].y, acc0);
          acc1 = mfma16(wr[c0 + u + 1].y, b[u + 1].y, acc1);
          acc0 = mfma16(wr[c0 + u].z, b[u].z, acc0);
          acc1 = mfma16(wr[c0 + u + 1].z, b[u + 1].z, acc1);
          acc0 = mfma16(wr[c0 + u].w, b[u].w, acc0);
          acc1 = mfma16(wr[c0 + u + 1].w, b[u + 1].w, acc1);
        }
      }
    }
  };
  // forward chunk: columns past the matrix width are cleared at use (a partial last 16-B piece, the next row's first floats)
  auto wmask_fw = [&](f32x4 (&wr)[NKB], int kvalid, int g) {
    if (kvalid < CK) {
#pragma unroll
      for (int cb = 0; cb < NKB; ++cb) {
        const int k = 16 * cb + 4 * g;
        wr[cb].x = k + 0 < kvalid ? wr[cb].x : 0.f; wr[cb].y = k + 1 < kvalid ? wr[cb].y : 0.f;
        wr[cb].z = k + 2 < kvalid ? wr[cb].z : 0.f; wr[cb].w = k + 3 < kvalid ? wr[cb].w : 0.f;
      }
    }
  };
  auto param4 = [&](const float* q, int n, int f0) -> f32x4 {
    const __amdgpu_buffer_rsrc_t win = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(q), 0, q ? n * 4 : 0, 0x00020000);
    return window_load(win, (uint32_t)f0 * 4u);
  };
  // x summed over the 16 lanes of a row group (the 16 data rows of one feature quartet), result in every lane
  auto sum_rows = [&](float x) -> float {
    x += __shfl_xor(x, 8, 64);
    x += __shfl_xor(x, 4, 64);
    x += __shfl_xor(x, 2, 64);
    x += __shfl_xor(x, 1, 64);
    return x;
  };

  for (int t = blockIdx.x; t < p.num_tiles; t += gridDim.x) {
    const int lane = fresh_lane();
    const int tid = w * 64 + lane;
    const int j = lane & 15;
    const int g = lane >> 4;
    const int f0 = 16 * w + 4 * g;
    const int row0 = t * R16;
    const int myrow = row0 + j < rows ? row0 + j : rows - 1;
    const bool row_live = row0 + j < rows;
    const int sr = tid / CPP, sc4 = tid % CPP;
    const int srow = row0 + sr < rows ? row0 + sr : rows - 1;

    // ---- requests, in the order of use
    int gid;
    {
      const int32_t* ip = p.gg_index;
      gid = (ip ? ip : reinterpret_cast<const int32_t*>(gnc_bcol_dummy))[ip ? myrow : 0];
    }
    f32x4 wr[NSET][NKB];
    wload_fw(wr[0], p.fw, j, g);
    const f32x4 gamma4 = param4(p.gamma, od, f0), bias4 = param4(p.bias_last, od, f0);
    // (grad_out through a bounds-checked window: rows narrower than 16 B - the decoder's [rows, 1] - are read where they lie,
    // the components past the output width are cleared below, a read past the last row returns zeros)
    f32x4 gv;
    {
      const __amdgpu_buffer_rsrc_t gw = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(p.grad_out), 0, p.has_g ? (int)(((int64_t)(rows - 1) * p.ld_g + od) * 4) : 0, 0x00020000);
      gv = window_load(gw, (uint32_t)((int64_t)myrow * p.ld_g + f0) * 4u);
    }
    f32x4 sa = *reinterpret_cast<const f32x4*>((p.has_ln && sc4 * 4 < H) ? p.act[L - 2] + (int64_t)srow * H + sc4 * 4 : gnc_bcol_dummy);
    f32x4 am[NTR - 1];  // a_l rows in the accumulator layout: the ReLU masks
#pragma unroll
    for (int l = 0; l < NTR - 1; ++l)
      am[l] = *reinterpret_cast<const f32x4*>((l < L - 1 && f0 < H) ? p.act[l] + (int64_t)myrow * H + f0 : gnc_bcol_dummy);
    wload_tr(wr[1], p.tr[0], j, g);
    f32x4 ggv;
    {
      const bool ok = p.has_gg && (uint32_t)gid < p.gg_rows && f0 < od;
      ggv = *reinterpret_cast<const f32x4*>(ok ? p.gg + (int64_t)gid * p.ld_gg + f0 : gnc_bcol_dummy);
      ggv = ok ? ggv : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    wload_tr(wr[2], p.tr[1], j, g);

    if (p.has_ln && sc4 * 4 < NT * 16) *reinterpret_cast<f32x4*>(abuf + sr * LDH + sc4 * 4) = sc4 * 4 < H ? sa : f32x4{0.f, 0.f, 0.f, 0.f};
    gv = p.has_g ? gv : f32x4{0.f, 0.f, 0.f, 0.f};
    gv += ggv;
    gv.x = f0 + 0 < od ? gv.x : 0.f; gv.y = f0 + 1 < od ? gv.y : 0.f;  // (a row of grad_out may be wider than the output)
    gv.z = f0 + 2 < od ? gv.z : 0.f; gv.w = f0 + 3 < od ? gv.w : 0.f;
    if (!row_live) gv = f32x4{0.f, 0.f, 0.f, 0.f};  // rows past the end contribute to no sum and are never stored
    __syncthreads();

    // ---- LayerNorm backward (statistics from the recomputed last pre-activation)
    f32x4 dz = gv;
    if (p.has_ln) {
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      if (16 * w < od) {
        wmask_fw(wr[0], p.fw.nk, g);
        mma(acc0, acc1, wr[0], abuf + j * LDH + 4 * g, (p.fw.nk + 15) >> 4);
      }
      f32x4 z = (acc0 + acc1) + bias4;
      const float inv_n = 1.f / (float)od;
      float s = add_quarters((z.x + z.y) + (z.z + z.w));
      if (g == 0) lnbuf[w * 16 + j] = s;
      __syncthreads();
      s = 0.f;
#pragma unroll
      for (int ww = 0; ww < NT; ++ww) s += lnbuf[ww * 16 + j];
      const float mean = s * inv_n;
      z.x = f0 + 0 < od ? z.x - mean : 0.f; z.y = f0 + 1 < od ? z.y - mean : 0.f;
      z.z = f0 + 2 < od ? z.z - mean : 0.f; z.w = f0 + 3 < od ? z.w - mean : 0.f;
      float v = add_quarters((z.x * z.x + z.y * z.y) + (z.z * z.z + z.w * z.w));
      if (g == 0) lnbuf[NT * 16 + w * 16 + j] = v;
      __syncthreads();
      v = 0.f;
#pragma unroll
      for (int ww = 0; ww < NT; ++ww) v += lnbuf[NT * 16 + ww * 16 + j];
      const float rstd = __frsqrt_rn(v * inv_n + p.eps);
      const f32x4 yh = z * rstd;
      // d beta / d gamma partial sums of the tile: sums over its rows, one row of ln_partial
      if (p.ln_partial) {
        f32x4 sb, sg;
        sb.x = sum_rows(gv.x); sb.y = sum_rows(gv.y); sb.z = sum_rows(gv.z); sb.w = sum_rows(gv.w);
        sg.x = sum_rows(gv.x * yh.x); sg.y = sum_rows(gv.y * yh.y); sg.z = sum_rows(gv.z * yh.z); sg.w = sum_rows(gv.w * yh.w);
        if (j == 0 && f0 < od) {
          float* lp = p.ln_partial + (int64_t)t * 2 * od;
          if (od % 4 == 0) {
            *reinterpret_cast<f32x4*>(lp + f0) = sb;
            *reinterpret_cast<f32x4*>(lp + od + f0) = sg;
          } else {
            lp[f0] = sb.x; lp[od + f0] = sg.x;
            if (f0 + 1 < od) { lp[f0 + 1] = sb.y; lp[od + f0 + 1] = sg.y; }
            if (f0 + 2 < od) { lp[f0 + 2] = sb.z; lp[od + f0 + 2] = sg.z; }
            if (f0 + 3 < od) { lp[f0 + 3] = sb.w; lp[od + f0 + 3] = sg.w; }
          }
        }
      }
      const f32x4 dy = gv * gamma4;  // features >= od: gamma reads as 0
      float s1 = add_quarters((dy.x + dy.y) + (dy.z + dy.w));
      float s2 = add_quarters((dy.x * yh.x + dy.y * yh.y) + (dy.z * yh.z + dy.w * yh.w));
      if (g == 0) {
        lnbuf[2 * NT * 16 + w * 16 + j] = s1;
        lnbuf[3 * NT * 16 + w * 16 + j] = s2;
      }
      __syncthreads();
      s1 = 0.f;
      s2 = 0.f;
#pragma unroll
      for (int ww = 0; ww < NT; ++ww) {
        s1 += lnbuf[2 * NT * 16 + ww * 16 + j];
        s2 += lnbuf[3 * NT * 16 + ww * 16 + j];
      }
      const float m1 = s1 * inv_n, m2 = s2 * inv_n;
      dz.x = f0 + 0 < od ? rstd * (dy.x - m1 - yh.x * m2) : 0.f; dz.y = f0 + 1 < od ? rstd * (dy.y - m1 - yh.y * m2) : 0.f;
      dz.z = f0 + 2 < od ? rstd * (dy.z - m1 - yh.z * m2) : 0.f; dz.w = f0 + 3 < od ? rstd * (dy.w - m1 - yh.w * m2) : 0.f;
    }
    wload_tr(wr[0], p.tr[2], j, g);  // set 0 is free: chunk 2

    auto store_rows = [&](float* base, int ld, int width, const f32x4& v, bool vec) {
      if (!row_live || f0 >= width) return;
      float* op = base + (int64_t)(row0 + j) * ld + f0;
      if (vec) {
        *reinterpret_cast<f32x4*>(op) = v;
      } else {
        op[0] = v.x;
        if (f0 + 1 < width) op[1] = v.y;
        if (f0 + 2 < width) op[2] = v.z;
        if (f0 + 3 < width) op[3] = v.w;
      }
    };
    store_rows(p.dz[L - 1], od, od, dz, od % 4 == 0);

    // ---- hidden layers, top down: dz_{l-1} = (dz_l W_l) * [a_{l-1} > 0]
    int c = 0;  // transposed chunk in use
    for (int l = L - 1; l >= 1; --l, ++c) {
      float* hb = hbuf + (c & 1) * 16 * LDH;
      *reinterpret_cast<f32x4*>(hb + j * LDH + f0) = dz;
      __syncthreads();
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      const int nkb = (p.out_dim[l] + 15) >> 4;
      if (16 * w < H) {
        if (c == 0) mma(acc0, acc1, wr[1], hb + j * LDH + 4 * g, nkb);
        else if (c == 1) mma(acc0, acc1, wr[2], hb + j * LDH + 4 * g, nkb);
        else mma(acc0, acc1, wr[0], hb + j * LDH + 4 * g, nkb);
      }
      if (c == 0) wload_tr(wr[1], p.tr[3], j, g);  // set 1 is free: chunk 3
      const f32x4 a = l - 1 == 0 ? am[0] : l - 1 == 1 ? am[1] : am[2];
      const f32x4 d = acc0 + acc1;
      dz.x = (f0 + 0 < H && a.x > 0.f) ? d.x : 0.f; dz.y = (f0 + 1 < H && a.y > 0.f) ? d.y : 0.f;
      dz.z = (f0 + 2 < H && a.z > 0.f) ? d.z : 0.f; dz.w = (f0 + 3 < H && a.w > 0.f) ? d.w : 0.f;
      store_rows(p.dz[l - 1], H, H, dz, true);
    }

    // ---- dx = dz_0 W_0, one 128-column chunk at a time
    if (p.dx) {
      float* hb = hbuf + (c & 1) * 16 * LDH;
      *reinterpret_cast<f32x4*>(hb + j * LDH + f0) = dz;
      __syncthreads();
      const int nkb = (H + 15) >> 4;
      for (int q = 0; q < p.n0; ++q, ++c) {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        if (c == 1) mma(acc0, acc1, wr[2], hb + j * LDH + 4 * g, nkb);
        else if (c == 2) mma(acc0, acc1, wr[0], hb + j * LDH + 4 * g, nkb);
        else mma(acc0, acc1, wr[1], hb + j * LDH + 4 * g, nkb);  // chunk 3 (or chunk 0 of a one-Linear tail: not reachable, L >= 2)
        f32x4 d = acc0 + acc1;
        if (q == p.dx_add_chunk) d += gv;
        const int width = p.k_in - q * CK < CK ? p.k_in - q * CK : CK;
        store_rows(p.dx + q * CK, p.ld_dx, width, d, p.vec_dx != 0);
      }
    }
    __syncthreads();  // the next tile's first writes (abuf, hbuf) wait for this tile's readers
  }
}

// ---- large batches at exactly 128 features (the c2 configuration): the same tile program with the weights RESIDENT IN
// REGISTERS.  One 8-wave workgroup per CU walks its tiles; W_2 (both as the forward reads it and transposed), W_1^T and the
// dx chunk of W_0^T are loaded once (4 x 32 registers per lane), the rows of the NEXT tile (grad_out, the saved activations,
// the gather id) are requested while the current one is computed, and a tile's last result (dx) is stored from the top of the
// next tile, so that the wait for the prefetched rows does not sit behind fresh stores.  Per tile nothing but its rows moves:
// 16 x 128 x (g, a_0, a_1 in; dz_2, dz_1, dz_0, dx out).  LayerNorm parameter sums: one partial row per workgroup.
// Shape (checked by bwd_col16_persist_supported): 3 Linears, hidden = out = 128, LayerNorm, saved activations, in_dim[0] = 128
// when dx is wanted.
template <int N0, bool GG>
__global__ __launch_bounds__(512) void mlp_bwd_col16_persist_kernel(const BPlan p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NT = 8, D = 128;
  constexpr int LDH = NT * 16 + 4;
  constexpr int CPP = 32;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int f0 = 16 * w + 4 * g;
  const int sr = tid / CPP, sc4 = tid % CPP;
  const int rows = p.rows;
  float* hbuf = lds;
  float* abuf = hbuf + 2 * 16 * LDH;
  float* lnbuf = abuf + 16 * LDH;

  auto wload_fw = [&](f32x4 (&wr)[NKB], const BChunk& c) {
    const __amdgpu_buffer_rsrc_t win = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(c.w), 0, c.nrec, 0x00020000);
    const uint32_t off0 = (uint32_t)((16 * w + j) * c.ldw + 4 * g) * 4u;
#pragma unroll
    for (int cb = 0; cb < NKB; ++cb) wr[cb] = window_load(win, off0 + (uint32_t)(cb * 64));
  };
  auto wload_tr = [&](f32x4 (&wr)[NKB], const BChunk& c) {
    const __amdgpu_buffer_rsrc_t win = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(c.w), 0, c.nrec, 0x00020000);
    const uint32_t off0 = (uint32_t)(4 * g * c.ldw + 16 * w + j) * 4u;
    const uint32_t rowb = (uint32_t)c.ldw * 4u;
#pragma unroll
    for (int cb = 0; cb < NKB; ++cb) {
      const uint32_t so = (uint32_t)(16 * cb) * rowb;
      wr[cb].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(win, off0, so, 0));
      wr[cb].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(win, off0, so + rowb, 0));
      wr[cb].z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(win, off0, so + 2 * rowb, 0));
      wr[cb].w = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(win, off0, so + 3 * rowb, 0));
    }
  };
  auto mma = [&](f32x4& acc0, f32x4& acc1, const f32x4 (&wr)[NKB], const float* src) {
#pragma unroll
    for (int c0 = 0; c0 < NKB; c0 += 4) {
      f32x4 b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) b[u] = *reinterpret_cast<const f32x4*>(src + 16 * (c0 + u));
#pragma unroll
      for (int u = 0; u < 4; u += 2) {
        acc0 = mfma16(wr[c0 + u].x, b[u].x, acc0);
        acc1 = mfma16(wr[c0 + u + 1].x, b[u + 1].x, acc1);
        acc0 = mfma16(wr[c0 + u].y, b[u].y, acc0);
        acc1 = mfma16(wr[c0 + u + 1].y, b[u + 1].y, acc1);
        acc0 = mfma16(wr[c0 + u].z, b[u].z, acc0);
        acc1 = mfma16(wr[c0 + u + 1].z, b[u + 1].z, acc1);
        acc0 = mfma16(wr[c0 + u].w, b[u].w, acc0);
        acc1 = mfma16(wr[c0 + u + 1].w, b[u + 1].w, acc1);
      }
    }
  };
  auto sum_rows = [&](float x) -> float {
    x += __shfl_xor(x, 8, 64);
    x += __shfl_xor(x, 4, 64);
    x += __shfl_xor(x, 2, 64);
    x += __shfl_xor(x, 1, 64);
    return x;
  };

  f32x4 wfw[NKB], wt0[NKB], wt1[NKB], wdx[N0 ? NKB : 1];
  wload_fw(wfw, p.fw);
  wload_tr(wt0, p.tr[0]);
  wload_tr(wt1, p.tr[1]);
  if constexpr (N0 > 0) wload_tr(wdx, p.tr[2]);
  // a second dx chunk (in_dim[0] in 129..256: the node processor's [x | agg]) does not fit the register file: its fragments
  // wait in LDS, in the order the lanes read them back (64 KB, wave-private slices: no barrier)
  float* wl2 = lnbuf + 4 * NT * 16 + (size_t)w * NKB * 64 * 4;
  if constexpr (N0 > 1) {
    f32x4 tmp[NKB];
    wload_tr(tmp, p.tr[3]);
#pragma unroll
    for (int cb = 0; cb < NKB; ++cb) *reinterpret_cast<f32x4*>(wl2 + (cb * 64 + lane) * 4) = tmp[cb];
  }
  const f32x4 gamma4 = *reinterpret_cast<const f32x4*>(p.gamma + f0);
  const f32x4 bias4 = p.bias_last ? *reinterpret_cast<const f32x4*>(p.bias_last + f0) : f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 sum_b = {0.f, 0.f, 0.f, 0.f}, sum_g = {0.f, 0.f, 0.f, 0.f};

  // rows of a tile, requested one tile ahead
  f32x4 r_gv = {0.f, 0.f, 0.f, 0.f}, r_gg = {0.f, 0.f, 0.f, 0.f}, r_sa, r_a0, r_a1;
  int r_gid = 0;
  auto request_rows = [&](int tt) {
    const int r0 = tt * R16;
    const int mr = r0 + j < rows ? r0 + j : rows - 1;
    const int sw = r0 + sr < rows ? r0 + sr : rows - 1;
    if constexpr (GG) r_gid = p.gg_index[mr];
    if (p.has_g) r_gv = *reinterpret_cast<const f32x4*>(p.grad_out + (int64_t)mr * p.ld_g + f0);
    r_sa = *reinterpret_cast<const f32x4*>(p.act[1] + (int64_t)sw * D + sc4 * 4);
    r_a0 = *reinterpret_cast<const f32x4*>(p.act[0] + (int64_t)mr * D + f0);
    r_a1 = *reinterpret_cast<const f32x4*>(p.act[1] + (int64_t)mr * D + f0);
  };
  auto request_gather = [&]() {
    if constexpr (GG) {
      const bool ok = (uint32_t)r_gid < p.gg_rows;
      r_gg = *reinterpret_cast<const f32x4*>(p.gg + (int64_t)(ok ? r_gid : 0) * p.ld_gg + f0);
      r_gg = ok ? r_gg : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  int t = blockIdx.x;
  request_rows(t);
  request_gather();
  f32x4 dx_pending = {0.f, 0.f, 0.f, 0.f}, dx_pending1 = {0.f, 0.f, 0.f, 0.f};
  int dx_row = -1;  // row of dx_pending (-1: nothing pending)
  auto store_dx_chunk = [&](const f32x4& v, int c0) {  // (rows of 16-B pieces when vec_dx)
    if (dx_row < 0 || c0 + f0 >= p.k_in) return;
    float* op = p.dx + (int64_t)dx_row * p.ld_dx + c0 + f0;
    if (p.vec_dx) {
      *reinterpret_cast<f32x4*>(op) = v;
    } else {
      op[0] = v.x;
      if (c0 + f0 + 1 < p.k_in) op[1] = v.y;
      if (c0 + f0 + 2 < p.k_in) op[2] = v.z;
      if (c0 + f0 + 3 < p.k_in) op[3] = v.w;
    }
  };
  auto store_dx = [&]() {
    store_dx_chunk(dx_pending, 0);
    if constexpr (N0 > 1) store_dx_chunk(dx_pending1, CK);
  };

  for (; t < p.num_tiles; t += gridDim.x) {
    const int row0 = t * R16;
    const bool row_live = row0 + j < rows;
    // ---- this tile's rows leave the prefetch registers; the next tile's are requested
    f32x4 gv = r_gv + r_gg;
    if (!row_live) gv = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 a0 = r_a0, a1 = r_a1;
    *reinterpret_cast<f32x4*>(abuf + sr * LDH + sc4 * 4) = r_sa;
    if constexpr (N0 > 0) store_dx();
    const int tn = t + (int)gridDim.x < p.num_tiles ? t + (int)gridDim.x : p.num_tiles - 1;
    request_rows(tn);
    __syncthreads();

    // ---- LayerNorm statistics from the recomputed last pre-activation, LayerNorm backward
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    mma(acc0, acc1, wfw, abuf + j * LDH + 4 * g);
    f32x4 z = (acc0 + acc1) + bias4;
    constexpr float inv_n = 1.f / (float)D;
    float s = add_quarters((z.x + z.y) + (z.z + z.w));
    if (g == 0) lnbuf[w * 16 + j] = s;
    __syncthreads();
    s = 0.f;
#pragma unroll
    for (int ww = 0; ww < NT; ++ww) s += lnbuf[ww * 16 + j];
    const float mean = s * inv_n;
    z -= f32x4{mean, mean, mean, mean};
    float v = add_quarters((z.x * z.x + z.y * z.y) + (z.z * z.z + z.w * z.w));
    if (g == 0) lnbuf[NT * 16 + w * 16 + j] = v;
    __syncthreads();
    v = 0.f;
#pragma unroll
    for (int ww = 0; ww < NT; ++ww) v += lnbuf[NT * 16 + ww * 16 + j];
    const float rstd = __frsqrt_rn(v * inv_n + p.eps);
    const f32x4 yh = z * rstd;
    sum_b += gv;
    sum_g += gv * yh;
    const f32x4 dy = gv * gamma4;
    float s1 = add_quarters((dy.x + dy.y) + (dy.z + dy.w));
    float s2 = add_quarters((dy.x * yh.x + dy.y * yh.y) + (dy.z * yh.z + dy.w * yh.w));
    if (g == 0) {
      lnbuf[2 * NT * 16 + w * 16 + j] = s1;
      lnbuf[3 * NT * 16 + w * 16 + j] = s2;
    }
    __syncthreads();
    s1 = 0.f;
    s2 = 0.f;
#pragma unroll
    for (int ww = 0; ww < NT; ++ww) {
      s1 += lnbuf[2 * NT * 16 + ww * 16 + j];
      s2 += lnbuf[3 * NT * 16 + ww * 16 + j];
    }
    const float m1 = s1 * inv_n, m2 = s2 * inv_n;
    f32x4 dz = (dy - f32x4{m1, m1, m1, m1} - yh * m2) * rstd;
    request_gather();  // the next tile's gathered rows: its ids have landed by now
    if (row_live) *reinterpret_cast<f32x4*>(p.dz[2] + (int64_t)(row0 + j) * D + f0) = dz;

    // ---- dz_1 = (dz_2 W_2) * [a_1 > 0], dz_0 = (dz_1 W_1) * [a_0 > 0]
    *reinterpret_cast<f32x4*>(hbuf + j * LDH + f0) = dz;
    __syncthreads();
    acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
    acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
    mma(acc0, acc1, wt0, hbuf + j * LDH + 4 * g);
    dz = acc0 + acc1;
    dz.x = a1.x > 0.f ? dz.x : 0.f; dz.y = a1.y > 0.f ? dz.y : 0.f; dz.z = a1.z > 0.f ? dz.z : 0.f; dz.w = a1.w > 0.f ? dz.w : 0.f;
    if (row_live) *reinterpret_cast<f32x4*>(p.dz[1] + (int64_t)(row0 + j) * D + f0) = dz;
    float* hb1 = hbuf + 16 * LDH;
    *reinterpret_cast<f32x4*>(hb1 + j * LDH + f0) = dz;
    __syncthreads();
    acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
    acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
    mma(acc0, acc1, wt1, hb1 + j * LDH + 4 * g);
    dz = acc0 + acc1;
    dz.x = a0.x > 0.f ? dz.x : 0.f; dz.y = a0.y > 0.f ? dz.y : 0.f; dz.z = a0.z > 0.f ? dz.z : 0.f; dz.w = a0.w > 0.f ? dz.w : 0.f;
    if (row_live) *reinterpret_cast<f32x4*>(p.dz[0] + (int64_t)(row0 + j) * D + f0) = dz;

    // ---- dx = dz_0 W_0 (+ g: the residual's gradient); stored from the top of the next tile
    if constexpr (N0 > 0) {
      *reinterpret_cast<f32x4*>(hbuf + j * LDH + f0) = dz;  // buffer 0: its readers passed the barrier above
      __syncthreads();
      acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
      acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
      mma(acc0, acc1, wdx, hbuf + j * LDH + 4 * g);
      dx_pending = acc0 + acc1;
      if (p.dx_add_chunk == 0) dx_pending += gv;
      dx_row = row_live ? row0 + j : -1;
      if constexpr (N0 > 1) {  // second chunk: A fragments from the wave's LDS slice, four k blocks at a time
        acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
        acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c0 = 0; c0 < NKB; c0 += 4) {
          f32x4 a[4], bf[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            a[u] = *reinterpret_cast<const f32x4*>(wl2 + ((c0 + u) * 64 + lane) * 4);
            bf[u] = *reinterpret_cast<const f32x4*>(hbuf + j * LDH + 4 * g + 16 * (c0 + u));
          }
#pragma unroll
          for (int u = 0; u < 4; u += 2) {
            acc0 = mfma16(a[u].x, bf[u].x, acc0);
            acc1 = mfma16(a[u + 1].x, bf[u + 1].x, acc1);
            acc0 = mfma16(a[u].y, bf[u].y, acc0);
            acc1 = mfma16(a[u + 1].y, bf[u + 1].y, acc1);
            acc0 = mfma16(a[u].z, bf[u].z, acc0);
            acc1 = mfma16(a[u + 1].z, bf[u + 1].z, acc1);
            acc0 = mfma16(a[u].w, bf[u].w, acc0);
            acc1 = mfma16(a[u + 1].w, bf[u + 1].w, acc1);
          }
        }
        dx_pending1 = acc0 + acc1;
        if (p.dx_add_chunk == 1) dx_pending1 += gv;
      }
    }
    // (no barrier here: every buffer the next tile rewrites was last read in front of a barrier all waves have passed by then)
  }
  if constexpr (N0 > 0) store_dx();
  // ---- d beta / d gamma: this workgroup's partial row
  f32x4 sb, sg;
  sb.x = sum_rows(sum_b.x); sb.y = sum_rows(sum_b.y); sb.z = sum_rows(sum_b.z); sb.w = sum_rows(sum_b.w);
  sg.x = sum_rows(sum_g.x); sg.y = sum_rows(sum_g.y); sg.z = sum_rows(sum_g.z); sg.w = sum_rows(sum_g.w);
  if (j == 0) {
    float* lp = p.ln_partial + (int64_t)blockIdx.x * 2 * D;
    *reinterpret_cast<f32x4*>(lp + f0) = sb;
    *reinterpret_cast<f32x4*>(lp + D + f0) = sg;
  }
}

int persist_grid(int64_t rows) {
  const int64_t tiles = gnc::ceil_div(rows, (int64_t)R16);
  return (int)(tiles < gnc::num_cu() ? tiles : gnc::num_cu());
}

template <int NT>
int launch_b(const BPlan& p, hipStream_t stream) {
  const size_t smem = ((size_t)3 * 16 * (NT * 16 + 4) + 4 * NT * 16) * sizeof(float);
  const int cap = 2 * gnc::num_cu();
  const int grid = p.num_tiles < cap ? p.num_tiles : cap;
  mlp_bwd_col16_kernel<NT><<<dim3((unsigned)grid), dim3(NT * 64), smem, stream>>>(p);
  return gnc::check_launch("mlp_bwd_col16_kernel");
}

bool al16b(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; }

int last_matmul(const gnc_mlp_desc_t& d) {
  int last = -1;
  for (int s = 0; s < d.num_segments; ++s)
    if (d.seg[s].mode == GNC_SEG_MATMUL) last = s;
  return last;
}

}  // namespace

// Shape question (pointers are looked at for alignment only): would the small-batch data kernel serve this backward, given
// that the forward's post-activations are at hand?
bool gnc_mlp::bwd_col16_supported(const gnc_mlp_desc_t& d) {
  static const bool off = getenv("GNC_NO_BWD_COL16") != nullptr;  // A/B switch
  if (off || d.rows < 1 || d.rows > bwd_col16_max_rows() || d.rows >= INT32_MAX) return false;
  const int L = d.num_linear;
  if (L < 2 || L > NTR) return false;
  const int H = d.out_dim[0], od = d.out_dim[L - 1];
  static const bool narrow = getenv("GNC_COL16_D64") != nullptr;
  if (H > 128 || od > 128 || H % 4 != 0 || (!narrow && H <= 64 && od <= 64)) return false;
  const int n0 = (d.in_dim[0] + CK - 1) / CK;
  if ((L - 1) + n0 > NTR) return false;
  for (int l = 0; l < L; ++l) {
    const int ldw = ldw_of(d, l);
    if ((int64_t)d.out_dim[l] * ldw * 4 > 0x7fffffffll) return false;
  }
  if (d.ln_gamma && (ldw_of(d, L - 1) % 4 != 0 || !al16b(d.weight[L - 1]))) return false;  // the forward chunk: 16-B pieces
  const int lm = last_matmul(d);
  if (lm < 0) return false;
  // a residual's gradient lands on whole chunks of dx only
  if (!d.seg[lm].index && d.seg[lm].width == od && d.seg[lm].wcol % CK != 0) return false;
  return true;
}

int gnc_mlp::bwd_col16_ln_partial_rows(int64_t rows) { return (int)gnc::ceil_div(rows, (int64_t)R16); }

// Large batches at exactly 128 features with saved activations (the c2 training configuration's edge processors): the
// register-resident variant.  Shape fields and alignment only.
bool gnc_mlp::bwd_col16_persist_supported(const gnc_mlp_desc_t& d) {
  static const bool off = getenv("GNC_NO_BWD_PERSIST") != nullptr;  // A/B switch: the 32-row streamed kernel
  if (off || d.rows <= bwd_col16_max_rows() || d.rows >= INT32_MAX / 2) return false;
  if (d.num_linear != 3 || !d.ln_gamma || !d.ln_beta) return false;
  for (int l = 0; l < 3; ++l)
    if (d.out_dim[l] != 128 || (l > 0 && d.in_dim[l] != 128) || (l == 2 && (ldw_of(d, l) % 4 != 0 || !al16b(d.weight[l]))) ||
        (int64_t)128 * ldw_of(d, l) * 4 > 0x7fffffffll)
      return false;
  if (d.in_dim[0] > 256 || !al16b(d.ln_gamma) || (d.bias[2] && !al16b(d.bias[2]))) return false;
  const int lm = last_matmul(d);
  if (lm < 0) return false;
  // a residual's gradient is folded into dx only as the whole (one) chunk
  if (!d.seg[lm].index && d.seg[lm].width == 128 && d.seg[lm].wcol % CK != 0) return false;
  return true;
}
int gnc_mlp::bwd_col16_persist_ln_partial_rows(int64_t rows) { return persist_grid(rows); }

int gnc_mlp::launch_bwd_col16_persist(const gnc_mlp_bwd_desc_t& bd, hipStream_t stream) {
  const gnc_mlp_desc_t& d = bd.fwd;
  constexpr int D = 128;
  GNC_REQUIRE(bd.grad_out || bd.grad_gather, "gnc_mlp_backward_f32: grad_out is null");
  GNC_REQUIRE(!bd.grad_out || (bd.ld_grad_out % 4 == 0 && al16b(bd.grad_out) && bd.ld_grad_out >= D),
              "gnc_mlp_backward_f32: grad_out must be 16-B aligned with ld %% 4 == 0");
  GNC_REQUIRE(!bd.grad_gather || (bd.ld_grad_gather % 4 == 0 && al16b(bd.grad_gather) && bd.grad_gather_index && bd.ld_grad_gather >= D),
              "gnc_mlp_backward_f32: grad_gather must be 16-B aligned with ld %% 4 == 0");
  GNC_REQUIRE(bd.ln_partial && al16b(bd.ln_partial), "gnc_mlp_backward_f32: ln_partial is required (gnc_mlp_backward_ln_partial_rows rows)");
  GNC_REQUIRE(!bd.dx || bd.ld_dx >= d.in_dim[0], "gnc_mlp_backward_f32: ld_dx < in_dim[0]");
  BPlan p = {};
  p.rows = (int)d.rows;
  p.num_tiles = (int)gnc::ceil_div(d.rows, (int64_t)R16);
  p.L = 3; p.H = D; p.od = D;
  p.n0 = bd.dx ? (d.in_dim[0] > CK ? 2 : 1) : 0;
  p.k_in = d.in_dim[0];
  p.vec_dx = (bd.dx && bd.ld_dx % 4 == 0 && d.in_dim[0] % 4 == 0 && al16b(bd.dx)) ? 1 : 0;
  p.has_ln = 1;
  p.has_g = bd.grad_out ? 1 : 0;
  p.has_gg = bd.grad_gather ? 1 : 0;
  for (int l = 0; l < 3; ++l) {
    GNC_REQUIRE(bd.dz[l] && al16b(bd.dz[l]), "gnc_mlp_backward_f32: dz[%d] must be given, 16-B aligned", l);
    p.dz[l] = bd.dz[l];
    if (l < 2) {
      GNC_REQUIRE(bd.act[l] && al16b(bd.act[l]), "gnc_mlp_backward_f32: act[%d] must be given, 16-B aligned", l);
      p.act[l] = bd.act[l];
    }
  }
  p.fw = {d.weight[2], ldw_of(d, 2), (127 * ldw_of(d, 2) + D) * 4, D};
  p.tr[0] = {d.weight[2], ldw_of(d, 2), (127 * ldw_of(d, 2) + D) * 4, D};
  p.tr[1] = {d.weight[1], ldw_of(d, 1), (127 * ldw_of(d, 1) + D) * 4, D};
  p.tr[2] = {d.weight[0], ldw_of(d, 0), (127 * ldw_of(d, 0) + d.in_dim[0]) * 4, D};
  if (d.in_dim[0] > CK) p.tr[3] = {d.weight[0] + CK, ldw_of(d, 0), (127 * ldw_of(d, 0) + d.in_dim[0] - CK) * 4, D};
  p.bias_last = d.bias[2];
  p.gamma = d.ln_gamma;
  p.eps = d.ln_eps;
  p.grad_out = bd.grad_out;
  p.ld_g = bd.ld_grad_out;
  p.gg = bd.grad_gather;
  p.ld_gg = bd.ld_grad_gather;
  p.gg_index = bd.grad_gather_index;
  p.gg_rows = bd.grad_gather_rows > INT32_MAX ? (uint32_t)INT32_MAX : (uint32_t)bd.grad_gather_rows;
  p.dx = bd.dx;
  p.ld_dx = bd.ld_dx;
  p.dx_add_chunk = -1;
  if (bd.dx && bd.dx_add_grad_out) {
    const int lm = last_matmul(d);
    GNC_REQUIRE(lm >= 0 && d.seg[lm].width == D && !d.seg[lm].index && d.seg[lm].wcol % CK == 0 && d.seg[lm].wcol + D <= d.in_dim[0],
                "gnc_mlp_backward_f32: dx_add_grad_out needs a row-ordered MATMUL segment as wide as the output on a chunk boundary");
    p.dx_add_chunk = d.seg[lm].wcol / CK;
  }
  p.ln_partial = bd.ln_partial;
  const size_t smem = ((size_t)3 * 16 * (8 * 16 + 4) + 4 * 8 * 16 + (p.n0 > 1 ? (size_t)8 * NKB * 64 * 4 : 0)) * sizeof(float);
  const dim3 grid((unsigned)persist_grid(d.rows)), block(512);
  if (p.n0 > 1) {
    static bool attr_set = false;
    if (!attr_set) {
      int rc = gnc::check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_bwd_col16_persist_kernel<2, false>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                              "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
      if (rc) return rc;
      attr_set = true;
    }
    GNC_REQUIRE(!bd.grad_gather, "gnc_mlp_backward_f32: grad_gather with a two-chunk dx is not instantiated");
    mlp_bwd_col16_persist_kernel<2, false><<<grid, block, smem, stream>>>(p);
    return gnc::check_launch("mlp_bwd_col16_persist_kernel");
  }
  if (bd.dx) {
    if (bd.grad_gather) mlp_bwd_col16_persist_kernel<1, true><<<grid, block, smem, stream>>>(p);
    else mlp_bwd_col16_persist_kernel<1, false><<<grid, block, smem, stream>>>(p);
  } else {
    if (bd.grad_gather) mlp_bwd_col16_persist_kernel<0, true><<<grid, block, smem, stream>>>(p);
    else mlp_bwd_col16_persist_kernel<0, false><<<grid, block, smem, stream>>>(p);
  }
  return gnc::check_launch("mlp_bwd_col16_persist_kernel");
}

int gnc_mlp::launch_bwd_col16(const gnc_mlp_bwd_desc_t& bd, hipStream_t stream) {
  const gnc_mlp_desc_t& d = bd.fwd;
  const int L = d.num_linear, H = d.out_dim[0], od = d.out_dim[L - 1];
  GNC_REQUIRE(bd.grad_out || bd.grad_gather, "gnc_mlp_backward_f32: grad_out is null");
  GNC_REQUIRE(!bd.grad_out || (bd.ld_grad_out >= od && (int64_t)d.rows * bd.ld_grad_out * 4 <= 0x7fffffffll),
              "gnc_mlp_backward_f32: grad_out: ld < out_dim (or a table beyond 2 GiB)");
  GNC_REQUIRE(!bd.grad_gather || (bd.ld_grad_gather % 4 == 0 && al16b(bd.grad_gather) && bd.grad_gather_index && bd.ld_grad_gather >= od),
              "gnc_mlp_backward_f32: grad_gather must be 16-B aligned with ld %% 4 == 0");
  GNC_REQUIRE(!d.ln_gamma || bd.ln_partial, "gnc_mlp_backward_f32: the small-batch kernel forms the LayerNorm sums itself (ln_partial)");
  GNC_REQUIRE(!bd.dx || bd.ld_dx >= d.in_dim[0], "gnc_mlp_backward_f32: ld_dx < in_dim[0]");
  BPlan p = {};
  p.rows = (int)d.rows;
  p.num_tiles = bwd_col16_ln_partial_rows(d.rows);
  p.L = L; p.H = H; p.od = od;
  p.n0 = bd.dx ? (d.in_dim[0] + CK - 1) / CK : 0;
  p.k_in = d.in_dim[0];
  p.has_ln = d.ln_gamma ? 1 : 0;
  p.has_g = bd.grad_out ? 1 : 0;
  p.has_gg = bd.grad_gather ? 1 : 0;
  for (int l = 0; l < L; ++l) {
    p.out_dim[l] = d.out_dim[l];
    p.in_dim[l] = d.in_dim[l];
    GNC_REQUIRE(bd.dz[l], "gnc_mlp_backward_f32: dz[%d] is null", l);
    GNC_REQUIRE(al16b(bd.dz[l]), "gnc_mlp_backward_f32: dz[%d] must be 16-B aligned", l);
    p.dz[l] = bd.dz[l];
    if (l < L - 1) {
      GNC_REQUIRE(bd.act[l] && al16b(bd.act[l]), "gnc_mlp_backward_f32: act[%d] must be given, 16-B aligned", l);
      p.act[l] = bd.act[l];
    }
  }
  if (d.ln_gamma) {
    const int ldw = ldw_of(d, L - 1);
    p.fw = {d.weight[L - 1], ldw, ((od - 1) * ldw + d.in_dim[L - 1]) * 4, d.in_dim[L - 1]};
    p.bias_last = d.bias[L - 1];
    p.gamma = d.ln_gamma;
    p.eps = d.ln_eps;
  }
  int c = 0;
  for (int l = L - 1; l >= 1; --l, ++c) {
    const int ldw = ldw_of(d, l);
    p.tr[c] = {d.weight[l], ldw, ((d.out_dim[l] - 1) * ldw + d.in_dim[l]) * 4, d.out_dim[l]};
  }
  for (int q = 0; q < p.n0; ++q, ++c) {
    const int ldw = ldw_of(d, 0);
    p.tr[c] = {d.weight[0] + q * CK, ldw, ((H - 1) * ldw + d.in_dim[0] - q * CK) * 4, H};
  }
  p.grad_out = bd.grad_out;
  p.ld_g = bd.ld_grad_out;
  p.gg = bd.grad_gather;
  p.ld_gg = bd.ld_grad_gather;
  p.gg_index = bd.grad_gather_index;
  p.gg_rows = bd.grad_gather_rows > INT32_MAX ? (uint32_t)INT32_MAX : (uint32_t)bd.grad_gather_rows;
  p.dx = bd.dx;
  p.ld_dx = bd.ld_dx;
  p.vec_dx = (bd.dx && bd.ld_dx % 4 == 0 && d.in_dim[0] % 4 == 0 && al16b(bd.dx)) ? 1 : 0;
  p.dx_add_chunk = -1;
  if (bd.dx && bd.dx_add_grad_out) {
    const int lm = last_matmul(d);
    GNC_REQUIRE(lm >= 0 && d.seg[lm].width == od && !d.seg[lm].index && d.seg[lm].wcol % CK == 0,
                "gnc_mlp_backward_f32: dx_add_grad_out needs a row-ordered last MATMUL segment as wide as the output");
    p.dx_add_chunk = d.seg[lm].wcol / CK;
  }
  p.ln_partial = d.ln_gamma ? bd.ln_partial : nullptr;
  const int wmax = H > od ? H : od;
  return wmax <= 64 ? launch_b<4>(p, stream) : launch_b<8>(p, stream);
}
