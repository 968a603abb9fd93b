// K4, streaming FAST variant for hidden widths 129..256, built on v_mfma_f32_16x16x4_f32.
//
// At 256 features the 32-row formulation needs 2 x 128 accumulator registers per wave (the current and
// the next layer), which forces one wave per SIMD and still spills (47 % of the fp32 MFMA peak at config
// c5).  With the 16x16x4 shape a wave owns 16 data rows: the same 256 features are 16 tiles x 4
// registers = 64 registers, two layers fit in 128, and the kernel runs 8 waves per CU (2 per SIMD); the 256-wide
// instance still spills 68 registers (272 B of scratch), the 128-wide ones none.  The formulation is the same as mlp_fused.hip, transposed: accumulator register r of tile t on
// lane (j = lane & 15, g = lane >> 4) holds feature 16t + 4g + r of data row j, which is exactly the B
// operand (k = 4g + r within the 16-wide k block t) of the next Linear; the matching A operand is
// W[n][16t + 4g .. +3], one ds_read_b128.  Weight chunks ([features][64 k], padded rows of 68 floats) are
// prefetched into registers under the MFMAs of the previous chunk and written between two barriers (one
// 70 KB buffer: two do not fit next to the row tiles); rows are staged 16 per wave.
#include <stdlib.h>

#include <type_traits>

#include "mlp_device16.h"

using namespace gnc_mlp;

namespace {

constexpr int MAX_STEPS16 = 16;
constexpr int MAX_WCHUNKS16 = 40;

struct Plan16 {
  int num_steps;
  int num_wchunks;
  struct { short seg, c0, add, pad; } step[MAX_STEPS16];
  struct { short layer, kbase, klimit, pad; } wc[MAX_WCHUNKS16];
};

// NTH / NTO: 16-feature tiles of the hidden / output width (4, 8 or 16; NTO may be 1 for the decoder)
// AGG: fused aggregation epilogue (gnc_mlp_desc_t.agg_out, see mlp_resident.hip / mlp_stream.hip): every wave owns a
// CONTIGUOUS range of 16-row wave tiles and carries the running sum of the destination in progress, one register per 64
// output columns; the workgroup still steps through the weight chunks in lockstep, so waves whose range is one tile
// shorter run a last iteration on a tile past the table's end (loads return zeros / clamped rows, stores are dropped).
// SAVE: training forward (gnc_mlp_desc_t.save_act): the hidden layers' post-activations are also written
// WV: waves per workgroup (8; 2 for small batches: one wave per SIMD on a quarter of the rows per workgroup, four times as many
// CUs busy - the reference's one-graph-per-call regime is MFMA-LATENCY bound: a wave's 768 dependent 16x16x4 steps per row tile)
template <int NTH, int NTO, bool DBUF, bool AGG = false, bool SAVE = false, int WV = W16>
__global__ __launch_bounds__(WV * 64) void mlp_stream16_kernel(const gnc_mlp_desc_t d, const Plan16 pl, const int num_tiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NTW = NTH > NTO ? NTH : NTO;
  constexpr int WROWS = NTW * 16;             // weight rows per chunk buffer
  constexpr int CH = WROWS * LDSW;
  constexpr int PSTRIDE = WROWS;
  constexpr int NTV = WV * 64;
  constexpr int RPP = NTV / 16;              // weight rows staged per pass (32)
  constexpr int NW = WROWS / RPP;
  constexpr int NCHI = (NTH + 3) / 4;         // 64-column chunks of a hidden-width input
  constexpr int NCHO = (NTO + 3) / 4;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15;   // feature row of the A operand == data row of the B operand / accumulator column
  const int g = lane >> 4;
  const int L = d.num_linear;
  const int out_dim = d.out_dim[L - 1];
  const int rows = (int)d.rows;
  float* wbuf = lds;  // DBUF: two chunk buffers (<= 128 features); at 256 features only one 70 KB buffer fits
  float* pbuf = lds + (DBUF ? 2 : 1) * CH;
  int gq = 0;
  float* abuf = pbuf + (L + 2) * PSTRIDE + wave * R16 * LDSW;

  stage_params<NTV>(pbuf, d, PSTRIDE, tid);

  auto wload = [&](f32x4 (&wr)[NW], int q) {
    const int ftid = wave * 64 + fresh_lane();  // roles recomputed per use: nothing lane-derived is carried across the loop
    const int wc4 = ftid & 15, wr0 = ftid >> 4;
    const int layer = pl.wc[q].layer;
    const float* W = d.weight[layer];
    const int ldw = ldw_of(d, layer);
    const int nrows = d.out_dim[layer];
    const int klimit = pl.wc[q].klimit;
    const int kbase = pl.wc[q].kbase;
    if (kbase + KC <= klimit && ldw % 4 == 0 && (reinterpret_cast<uintptr_t>(W) & 15u) == 0 &&
        (int64_t)nrows * ldw * 4 <= 0xffffffffll) {
      // full 64-column chunk of an aligned matrix: window over the matrix (rows >= nrows read as 0), no masks
      const __amdgpu_buffer_rsrc_t w =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, nrows * ldw * 4, 0x00020000);
      const uint32_t off0 = (uint32_t)(wr0 * ldw + kbase + wc4 * 4) * 4u;
#pragma unroll
      for (int p = 0; p < NW; ++p) wr[p] = window_load(w, off0 + (uint32_t)(p * RPP * 4) * (uint32_t)ldw);
      return;
    }
    const int col = kbase + wc4 * 4;
    const int colc = col < klimit ? col : kbase;
#pragma unroll
    for (int p = 0; p < NW; ++p) {
      const int n = p * RPP + wr0;
      const int nc = n < nrows ? n : nrows - 1;
      f32x4 v = *reinterpret_cast<const f32x4*>(W + (int64_t)nc * ldw + colc);
      const bool rowok = n < nrows;
      v.x = (rowok && col + 0 < klimit) ? v.x : 0.f; v.y = (rowok && col + 1 < klimit) ? v.y : 0.f;
      v.z = (rowok && col + 2 < klimit) ? v.z : 0.f; v.w = (rowok && col + 3 < klimit) ? v.w : 0.f;
      wr[p] = v;
    }
  };
  auto wstore = [&](const f32x4 (&wr)[NW], float* buf) {
    const int ftid = wave * 64 + fresh_lane();
    const int wc4 = ftid & 15, wr0 = ftid >> 4;
#pragma unroll
    for (int p = 0; p < NW; ++p) *reinterpret_cast<f32x4*>(buf + (p * RPP + wr0) * LDSW + wc4 * 4) = wr[p];
  };
  // Tile schedule in WAVE tiles (16 rows each).  Default: workgroup tile t = blockIdx.x + k * gridDim.x, the wave takes
  // wave tile t * 8 + wave.  AGG: one contiguous range per wave.  `iters` is workgroup-uniform.
  const int num_wtiles = (rows + R16 - 1) / R16;
  const int last_wt = num_wtiles - 1;
  const int gwave = (int)blockIdx.x * WV + wave;
  int agg_t0 = 0, agg_cnt = 0, iters;
  if constexpr (AGG) {
    const int tw = (int)gridDim.x * WV, qq = num_wtiles / tw, rem = num_wtiles - qq * tw;
    agg_t0 = gwave * qq + (gwave < rem ? gwave : rem);
    agg_cnt = qq + (gwave < rem ? 1 : 0);
    iters = qq + ((int)blockIdx.x * WV < rem ? 1 : 0);
  } else {
    iters = (num_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  }
  auto wtile_of = [&](int k) -> int {  // wave tile of iteration k (>= num_wtiles: past the end)
    if constexpr (AGG) return k < agg_cnt ? agg_t0 + k : num_wtiles;
    return ((int)blockIdx.x + k * (int)gridDim.x) * WV + wave;
  };
  auto load_idx = [&](int wt_, int s) -> int {
    const int tc = wt_ < last_wt ? wt_ : last_wt;
    int r = tc * R16 + (lane & 15);
    r = r < rows ? r : rows - 1;
    const int32_t* ip = d.seg[s].index;
    return ip ? ip[r] : r;
  };
  auto load_rows = [&](f32x4 (&pre)[NP16], int s, int c0, int idxv, int tile_of) {
    const int fl = fresh_lane();
    const int c4 = fl & 15, rs = fl >> 4;
    const float* base = d.seg[s].ptr;
    const int ld = d.seg[s].ld;
    const int col = c0 + c4 * 4 < ld ? c0 + c4 * 4 : 0;
    if (d.seg[s].index == nullptr) {  // row-ordered: window at the tile's first row
      load_tile_rows<false, NP16>(pre, base, ld, (int64_t)tile_of * R16, rows, (uint32_t)(rs * ld + col) * 4u);
    } else {
      const int64_t tbytes = d.seg[s].table_rows * (int64_t)ld * 4;
      if (tbytes > 0 && tbytes <= 0xffffffffll) {
        // stated table below 4 GiB: bounds-checked window over the table, the row's byte offset travels by
        // ds_bpermute, one 32-bit add per access (an id outside the table reads zeros)
        const __amdgpu_buffer_rsrc_t w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)(uint32_t)tbytes, 0x00020000);
        const int row_bytes = idxv * (ld * 4);
#pragma unroll
        for (int p = 0; p < NP16; ++p) pre[p] = window_load(w, (uint32_t)__shfl(row_bytes, p * 4 + rs, 64) + (uint32_t)(col * 4));
      } else {
#pragma unroll
        for (int p = 0; p < NP16; ++p) {  // table of 4 GiB or more: flat addresses, the id checked by hand (outside -> zeros)
          const int tr = __shfl(idxv, p * 4 + rs, 64);
          const bool ok = (uint64_t)(int64_t)tr < (uint64_t)d.seg[s].table_rows;
          const f32x4 v = *reinterpret_cast<const f32x4*>(base + (int64_t)(ok ? tr : 0) * ld + col);
          pre[p] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
    }
  };
  auto stage = [&](const f32x4 (&pre)[NP16], int c0, int width) {
    compiler_lds_barrier();
    const int fl = fresh_lane();
    const int c4 = fl & 15, rs = fl >> 4;
    const int c = c0 + c4 * 4;
    if (c0 + KC <= width) {  // full chunk: nothing to mask
#pragma unroll
      for (int p = 0; p < NP16; ++p) *reinterpret_cast<f32x4*>(abuf + (p * 4 + rs) * LDSW + c4 * 4) = pre[p];
    } else {
#pragma unroll
      for (int p = 0; p < NP16; ++p) {
        f32x4 v = pre[p];
        v.x = c + 0 < width ? v.x : 0.f; v.y = c + 1 < width ? v.y : 0.f;
        v.z = c + 2 < width ? v.z : 0.f; v.w = c + 3 < width ? v.w : 0.f;
        *reinterpret_cast<f32x4*>(abuf + (p * 4 + rs) * LDSW + c4 * 4) = v;
      }
    }
    compiler_lds_barrier();
  };

  int ids[GNC_MAX_SEGMENTS], ids_next[GNC_MAX_SEGMENTS];
  int kit = 0;
  int tile = wtile_of(0);  // the wave tile in progress
#pragma unroll
  for (int s = 0; s < GNC_MAX_SEGMENTS; ++s) {
    ids[s] = s < d.num_segments ? load_idx(tile, s) : 0;
    ids_next[s] = s < d.num_segments ? load_idx(wtile_of(1), s) : 0;
  }
  auto id_of = [&](int s) {
    int v = 0;
#pragma unroll
    for (int k = 0; k < GNC_MAX_SEGMENTS; ++k) v = k == s ? ids[k] : v;
    return v;
  };
  f32x4 cur[NP16];
  load_rows(cur, pl.step[0].seg, pl.step[0].c0, id_of(pl.step[0].seg), tile);
  f32x4 wreg[NW];
  wload(wreg, 0);
  wstore(wreg, wbuf);
  __syncthreads();
  auto prefetch_next_chunk = [&](int q) { wload(wreg, q + 1 < pl.num_wchunks ? q + 1 : 0); };
  auto publish_next_chunk = [&]() {
    if constexpr (DBUF) {
      wstore(wreg, wbuf + ((gq + 1) & 1) * CH);
      __syncthreads();
      ++gq;
    } else {
      __syncthreads();  // everyone is done reading the buffer
      wstore(wreg, wbuf);
      __syncthreads();
    }
  };
  auto cur_w = [&]() -> const float* { return wbuf + (DBUF ? (gq & 1) : 0) * CH; };

  // fused aggregation state (AGG): destination ids of the tile's rows, running sums (lane = feature, per 64 columns)
  auto agg_ids = [&](int wt_) -> int {
    int r = wt_ * R16 + (lane & 15);
    r = r < rows ? r : rows - 1;
    return d.agg_index[r];
  };
  int aid = 0, aid_next = 0;
  float agg_acc[AGG ? NCHO : 1];
#pragma unroll
  for (int c = 0; c < (AGG ? NCHO : 1); ++c) agg_acc[c] = 0.f;
  int agg_cur = -1, agg_first_dst = -1;  // wave-uniform
  bool agg_first = true;
  if constexpr (AGG) aid = agg_ids(tile);

  while (kit < iters) {
    const int row0 = tile * R16;
    const int ntile = wtile_of(kit + 1);
    if constexpr (AGG) aid_next = agg_ids(ntile);
    int q = 0;

    // ------------------------------------------------------------------ first Linear
    f32x4 hid[NTH];
    init_bias16<NTH>(hid, pbuf, g);
    // stage the rows of step `st` and request those of the step after it (the next tile's first one after the last)
    auto stage_and_advance = [&](int st) {
      stage(cur, pl.step[st].c0, d.seg[pl.step[st].seg].width);
      const bool wrap = st + 1 >= pl.num_steps;
      const int nst = wrap ? 0 : st + 1;
      if (wrap) {
#pragma unroll
        for (int k = 0; k < GNC_MAX_SEGMENTS; ++k) {
          ids[k] = ids_next[k];
          if (k < d.num_segments) ids_next[k] = load_idx(wtile_of(kit + 2), k);
        }
      }
      load_rows(cur, pl.step[nst].seg, pl.step[nst].c0, id_of(pl.step[nst].seg), wrap ? ntile : tile);
    };
    int st = 0;
    for (; st < pl.num_steps && !pl.step[st].add; ++st) {  // MATMUL steps: one staged 64-column chunk of weight[0] each
      const int c0 = pl.step[st].c0, width = d.seg[pl.step[st].seg].width;
      stage_and_advance(st);
      prefetch_next_chunk(q);
      const int kc = width - c0 < KC ? width - c0 : KC;
      mma16_chunk_from_lds<NTH>(hid, abuf, cur_w(), (kc + 15) >> 4, i, g);
      publish_next_chunk();
      ++q;
    }
    // additive steps (rows already in the hidden width), listed slab-major by the plan: the accumulator tiles a slab
    // goes to are a compile-time constant here (a run-time tile index costs a v_cndmask per accumulator component)
#define GNC_ADD_SLAB(CC_)                                                        \
  if constexpr (CC_ < NCHI) {                                                    \
    for (; st < pl.num_steps && pl.step[st].c0 == CC_ * KC; ++st) {               \
      stage_and_advance(st);                                                     \
      add_slab16<NTH, 4 * CC_>(hid, abuf, i, g);                                  \
    }                                                                            \
  }
    GNC_ADD_SLAB(0) GNC_ADD_SLAB(1) GNC_ADD_SLAB(2) GNC_ADD_SLAB(3)
#undef GNC_ADD_SLAB

    f32x4 o[NTO];
    if (L == 1) {
      if (d.ln_gamma) layer_norm16<NTH>(hid, pbuf + L * PSTRIDE, pbuf + (L + 1) * PSTRIDE, out_dim, d.ln_eps, g);
      if constexpr (NTO == NTH) {
#pragma unroll
        for (int t = 0; t < NTO; ++t) o[t] = hid[t];
      }
    } else {
      relu16<NTH>(hid);
      // training forward (gnc_mlp_desc_t.save_act): the INPUT of the layer being computed leaves as whole rows, one 64-column
      // slab next to each of that layer's weight chunks (through the wave's LDS tile, idle between the first Linear's
      // staging and the epilogue).  All slabs of a tensor in one go right after its ReLU measured +1.7 ms per c5 edge
      // launch, +0.15 ms of it the LDS transposes: the 8 waves of all 256 CUs run in step, the burst (33 MB chip-wide)
      // takes longer to drain than the next chunk's MFMAs last, and the next weight chunk's vmcnt wait sits it out.
      auto save_slab = [&](const f32x4 (&acc)[NTH], int l, auto cc_) {
        constexpr int cc = decltype(cc_)::value;
        const int width = d.out_dim[l];
        compiler_lds_barrier();
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
          if (4 * cc + cb < NTH) *reinterpret_cast<f32x4*>(abuf + i * LDSW + 16 * cb + 4 * g) = acc[4 * cc + cb < NTH ? 4 * cc + cb : 0];
        compiler_lds_barrier();
        const int fl = fresh_lane();
        store_staged_rows<NP16, true>(abuf, d.save_act[l] + cc * KC, width, width - cc * KC < KC ? width - cc * KC : KC, row0, rows,
                                      fl & 15, fl >> 4);
        compiler_lds_barrier();
      };
      for (int l = 1; l < L - 1; ++l) {
        f32x4 nxt[NTH];
        init_bias16<NTH>(nxt, pbuf + l * PSTRIDE, g);
#define GNC_HID_CHUNK(C_)                                                                        \
  if constexpr (C_ < NCHI) {                                                                     \
    if (C_ * KC < d.in_dim[l]) {                                                                 \
      prefetch_next_chunk(q);                                                                    \
      if constexpr (SAVE) save_slab(hid, l - 1, std::integral_constant<int, C_>{});               \
      mma16_chunk_from_regs<NTH, NTH, C_>(nxt, hid, cur_w(), d.in_dim[l], i, g);   \
      publish_next_chunk();                                                                      \
      ++q;                                                                                       \
    }                                                                                            \
  }
        GNC_HID_CHUNK(0) GNC_HID_CHUNK(1) GNC_HID_CHUNK(2) GNC_HID_CHUNK(3)
#undef GNC_HID_CHUNK
        relu16<NTH>(nxt);
#pragma unroll
        for (int t = 0; t < NTH; ++t) hid[t] = nxt[t];
      }
      init_bias16<NTO>(o, pbuf + (L - 1) * PSTRIDE, g);
#define GNC_OUT_CHUNK(C_)                                                                          \
  if constexpr (C_ < NCHI) {                                                                       \
    if (C_ * KC < d.in_dim[L - 1]) {                                                               \
      prefetch_next_chunk(q);                                                                      \
      if constexpr (SAVE) save_slab(hid, L - 2, std::integral_constant<int, C_>{});                 \
      mma16_chunk_from_regs<NTH, NTO, C_>(o, hid, cur_w(), d.in_dim[L - 1], i, g);   \
      publish_next_chunk();                                                                        \
      ++q;                                                                                         \
    }                                                                                              \
  }
      GNC_OUT_CHUNK(0) GNC_OUT_CHUNK(1) GNC_OUT_CHUNK(2) GNC_OUT_CHUNK(3)
#undef GNC_OUT_CHUNK
      if (d.ln_gamma) layer_norm16<NTO>(o, pbuf + L * PSTRIDE, pbuf + (L + 1) * PSTRIDE, out_dim, d.ln_eps, g);
    }

    // collect the next tile's prefetched rows before the asm stores join the memory queue
#pragma unroll
    for (int p = 0; p < NP16; ++p) asm volatile("" ::"v"(cur[p]));

    // ------------------------------------------------------------------ epilogue: 64 output columns at a time
    const int fle = fresh_lane();
    const int c4 = fle & 15, rs = fle >> 4;
#pragma unroll
    for (int cc = 0; cc < NCHO; ++cc) {
      if (cc * KC < out_dim) {
        compiler_lds_barrier();
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
          if (4 * cc + cb < NTO) *reinterpret_cast<f32x4*>(abuf + i * LDSW + 16 * cb + 4 * g) = o[4 * cc + cb < NTO ? 4 * cc + cb : 0];
        compiler_lds_barrier();
        if constexpr (AGG) {
          // the residual joins the staged tile first (whole rows), so that the LDS tile holds the FINAL rows that are
          // both stored and summed per destination
          const int wcc = out_dim - cc * KC < KC ? out_dim - cc * KC : KC;
          if (d.residual) {
            f32x4 rv[NP16];
            load_tile_rows<false, NP16>(rv, d.residual + cc * KC, d.ld_residual, row0, rows,
                                        (uint32_t)(rs * d.ld_residual + (c4 * 4 < wcc ? c4 * 4 : 0)) * 4u);
#pragma unroll
            for (int p = 0; p < NP16; ++p) {
              float* qp = abuf + (p * 4 + rs) * LDSW + c4 * 4;
              *reinterpret_cast<f32x4*>(qp) = *reinterpret_cast<const f32x4*>(qp) + rv[p];
            }
            compiler_lds_barrier();
          }
          store_staged_rows<NP16, true>(abuf, d.out + cc * KC, d.ld_out, wcc, row0, rows, c4, rs);
          // walk of this slab's 16 row values (lane = column of the slab): wave-uniform control (a ballot of "row starts
          // a new destination"), one add per row.  Every slab replays the walk from the tile's starting state; the state
          // is committed after the last slab.
          float rowv[R16];
#pragma unroll
          for (int r = 0; r < R16; ++r) rowv[r] = abuf[r * LDSW + (fle & 63)];
          const int valid = rows - row0 < R16 ? (rows - row0 > 0 ? rows - row0 : 0) : R16;
          int prv = __shfl_up(aid, 1, 64);
          prv = (fle & 63) == 0 ? agg_cur : prv;
          const unsigned vmask = valid >= R16 ? 0xffffu : ((1u << valid) - 1u);
          const unsigned bnd = (unsigned)(__ballot(aid != prv) & 0xffffull) & vmask;
          int w_cur = agg_cur, w_first_dst = agg_first_dst;
          bool w_first = agg_first;
          float acc = agg_acc[cc];
#pragma unroll
          for (int r = 0; r < R16; ++r) {
            if (r < valid) {
              if ((bnd >> r) & 1u) {
                if (w_cur >= 0) {
                  if (w_first) {
                    w_first = false;
                    w_first_dst = w_cur;
                  } else if (cc * KC + (fle & 63) < out_dim) {
                    float* dstp = d.agg_out + (int64_t)w_cur * d.ld_agg;
                    asm volatile("s_nop 4\n\tglobal_store_dword %0, %1, %2\n\ts_nop 0" ::"v"((cc * KC + (fle & 63)) * 4), "v"(acc), "s"(dstp) : "memory");
                  }
                }
                w_cur = __builtin_amdgcn_readlane(aid, r);
                acc = 0.f;
              }
              acc += rowv[r];
            }
          }
          agg_acc[cc] = acc;
          if (cc == NCHO - 1 || (cc + 1) * KC >= out_dim) {  // last slab of the tile: commit the walk's state
            agg_cur = w_cur;
            agg_first = w_first;
            agg_first_dst = w_first_dst;
          }
        } else {
          store_staged_rows<NP16, true>(abuf, d.out + cc * KC, d.ld_out, out_dim - cc * KC < KC ? out_dim - cc * KC : KC, row0, rows, c4,
                                        rs, d.residual ? d.residual + cc * KC : nullptr, d.ld_residual);
        }
      }
    }
    if constexpr (AGG) aid = aid_next;
    compiler_lds_barrier();
    tile = ntile;
    ++kit;
  }
  if constexpr (AGG) {  // the range's first and last destination (possibly the same, possibly none) go to the fix-up
    if (lane == 0) {
      d.agg_fix[2 * gwave] = agg_first ? agg_cur : agg_first_dst;
      d.agg_fix[2 * gwave + 1] = agg_cur;
    }
  }
}

template <int NTH, int NTO, bool AGG = false, bool SAVE = false, int WV = W16>
int launch16(const gnc_mlp_desc_t& d, const Plan16& pl, hipStream_t stream) {
  constexpr int NTW = NTH > NTO ? NTH : NTO;
  constexpr bool DBUF = NTW <= 8;
  const size_t smem =
      ((size_t)(DBUF ? 2 : 1) * NTW * 16 * LDSW + (size_t)(d.num_linear + 2) * NTW * 16 + (size_t)WV * R16 * LDSW) * sizeof(float);
  if (smem > 160 * 1024) {
    gnc::set_error("mlp_stream16: LDS budget exceeded (%zu bytes)", smem);
    return GNC_ERR_UNSUPPORTED;
  }
  static bool attr_set = false;
  if (!attr_set) {
    int rc = gnc::check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_stream16_kernel<NTH, NTO, DBUF, AGG, SAVE, WV>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                            "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    if (rc) return rc;
    attr_set = true;
  }
  const int64_t num_tiles = gnc::ceil_div(d.rows, (int64_t)WV * R16);
  int64_t grid = num_tiles < gnc::num_cu() ? num_tiles : gnc::num_cu();
  if constexpr (AGG) grid = gnc::num_cu();  // agg_fix has two entries for every wave of the full grid (8 waves per workgroup)
  mlp_stream16_kernel<NTH, NTO, DBUF, AGG, SAVE, WV><<<dim3((unsigned)grid), dim3(WV * 64), smem, stream>>>(d, pl, (int)num_tiles);
  return gnc::check_launch("mlp_stream16_kernel");
}

bool al16p(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

int gnc_mlp::launch_stream16(const gnc_mlp_desc_t& d, hipStream_t stream, bool* launched, bool probe_only) {
  *launched = false;
  static const bool disabled = getenv("GNC_MLP_NO_STREAM16") != nullptr;  // A/B switch for benchmarking
  if (disabled || d.rows >= INT32_MAX) return GNC_OK;
  const int L = d.num_linear;
  if (L > 1 && d.activation != GNC_ACT_RELU) return GNC_OK;
  const int H = d.out_dim[0], od = d.out_dim[L - 1];
  if (H > 256 || od > 256) return GNC_OK;
  if (d.residual && (d.ld_residual % 4 != 0 || !al16p(d.residual))) return GNC_OK;
  if (d.save_act[0]) {  // saved post-activations: whole rows of 16-B pieces
    if (L < 2) return GNC_OK;
    for (int l = 0; l < L - 1; ++l)
      if (!d.save_act[l] || d.out_dim[l] % 4 != 0 || !al16p(d.save_act[l])) return GNC_OK;
  }
  for (int l = 0; l < L; ++l)
    if (ldw_of(d, l) % 4 != 0 || !al16p(d.weight[l])) return GNC_OK;
  Plan16 pl = {};
  for (int s = 0; s < d.num_segments; ++s) {  // MATMUL segments first, chunk by chunk
    const gnc_mlp_segment_t& sg = d.seg[s];
    if (sg.ld % 4 != 0 || !al16p(sg.ptr)) return GNC_OK;
    if (sg.mode == GNC_SEG_ADD) continue;
    if (sg.wcol % 4 != 0) return GNC_OK;
    for (int c0 = 0; c0 < sg.width; c0 += KC) {
      if (pl.num_steps >= MAX_STEPS16 || pl.num_wchunks >= MAX_WCHUNKS16) return GNC_OK;
      pl.step[pl.num_steps++] = {(short)s, (short)c0, (short)0, 0};
      pl.wc[pl.num_wchunks++] = {(short)0, (short)(sg.wcol + c0), (short)(sg.wcol + sg.width), 0};
    }
  }
  for (int c0 = 0; c0 < H; c0 += KC)  // then the additive ones, slab-major (the kernel walks the slabs with a static index)
    for (int s = 0; s < d.num_segments; ++s) {
      if (d.seg[s].mode != GNC_SEG_ADD) continue;
      if (pl.num_steps >= MAX_STEPS16) return GNC_OK;
      pl.step[pl.num_steps++] = {(short)s, (short)c0, (short)1, 0};
    }
  if (pl.num_wchunks == 0) return GNC_OK;
  for (int l = 1; l < L; ++l)
    for (int c = 0; c * KC < d.in_dim[l]; ++c) {
      if (pl.num_wchunks >= MAX_WCHUNKS16) return GNC_OK;
      pl.wc[pl.num_wchunks++] = {(short)l, (short)(c * KC), (short)d.in_dim[l], 0};
    }
  const bool save = d.save_act[0] != nullptr;  // training forward: SAVE instances (template flag) of the 256-wide kernel and of the small-batch one
  const bool small_rows = d.rows <= (int64_t)2 * R16 * gnc::num_cu();
  if (save && !((H > 128 || (small_rows && H > 64)) && od > 16)) return GNC_OK;
  // small batches at up to 128 features (the reference's one-graph-per-call regime): 2-wave workgroups of 16-row tiles
  const bool small128 = H <= 128 && od <= 128 && (!save || od > 16) && d.rows <= (int64_t)2 * R16 * gnc::num_cu();
  if (d.agg_out) {  // aggregation epilogue: the 256-wide instance (129..256 output features) and the small-batch 128-wide one
    if (L <= 1 || !d.agg_index || !d.agg_fix || d.ld_agg < od) return GNC_OK;
    if (small128 && od > 64) {
      *launched = true;
      if (probe_only) return GNC_OK;
      return save ? launch16<8, 8, true, true, 2>(d, pl, stream) : launch16<8, 8, true, false, 2>(d, pl, stream);
    }
    if (!(od > 128)) return GNC_OK;
    *launched = true;
    if (probe_only) return GNC_OK;
    return save ? launch16<16, 16, true, true>(d, pl, stream) : launch16<16, 16, true>(d, pl, stream);
  }
  *launched = true;
  if (probe_only) return GNC_OK;
  if (save && small128) return launch16<8, 8, false, true, 2>(d, pl, stream);
  if (save) return launch16<16, 16, false, true>(d, pl, stream);
  if (small128) {
    if (od <= 16 && L > 1) return launch16<8, 1, false, false, 2>(d, pl, stream);
    return launch16<8, 8, false, false, 2>(d, pl, stream);
  }
  if (H <= 128 && od <= 128) {
    if (od <= 16 && L > 1) return launch16<8, 1>(d, pl, stream);
    return launch16<8, 8>(d, pl, stream);
  }
  if (od <= 16 && L > 1) return launch16<16, 1>(d, pl, stream);
  return launch16<16, 16>(d, pl, stream);
}
