// K4, streaming FAST variant of the fused MLP for hidden widths 65..256 (formulation: mlp_fused.hip).
//
// The weights of a 128- or 256-wide MLP do not fit in LDS (edge processor at D=128: 192 KB), so
// they stream L2 -> registers -> LDS in [features][64 k] chunks through a DOUBLE buffer: while the
// waves run the MFMAs of chunk q out of buffer q&1, every thread already holds its share of chunk
// q+1 in registers (loads issued before the MFMAs) and drops it into buffer (q+1)&1 afterwards;
// one barrier per chunk then publishes it.  The chunk sequence of a tile (first Linear's MATMUL
// chunks, then every later Linear) is the same for every tile, so the stream never stops: the
// last chunk of a tile prefetches chunk 0 of the next.
//
// The per-wave side is the resident kernel's pipeline (mlp_resident.hip): unconditional row loads one
// step ahead in registers (buffer windows for row-ordered tables, weight chunks and stated gather tables;
// clamped ids), gather ids one tile ahead, additive (W-split) segments summed in registers, hidden
// activations in accumulators across layers, stores issued from inline asm so hipcc's vmcnt waits stay
// counted.  The 128-wide instance can also carry the aggregation epilogue (template flag AGG).
#include <stdlib.h>

#include "mlp_device.h"

using namespace gnc_mlp;

namespace {

constexpr int MAX_STEPS = 16;
constexpr int MAX_WCHUNKS = 40;

struct StreamPlan {
  int num_steps;    // first-Linear staging steps per tile
  int num_wchunks;  // weight chunks per tile
  struct {
    short seg;   // segment staged by this step
    short seg2;  // second segment of a combined ADD step, or -1
    short c0;    // first column of the 64-column chunk
    short add;   // 1 = additive step (no weights)
  } step[MAX_STEPS];
  struct {
    short layer;
    short kbase;   // first weight column of the chunk
    short klimit;  // columns at or beyond this are zero
    short pad;
  } wc[MAX_WCHUNKS];
};

// ADD2: additive segments are staged in pairs summed in registers (needs a second row set; off for the
// 256-wide instance, which has no registers to spare).
// DBUF: two weight buffers and one barrier per chunk; false (256-wide layers, whose 70 KB chunks do not
// fit twice): one buffer, the prefetched registers are written between two barriers.
// AGG: fused aggregation epilogue (gnc_mlp_desc_t.agg_out, see mlp_resident.hip): every wave owns a CONTIGUOUS range
// of 32-row tiles and carries the running sum of the destination in progress, one register per 64 output columns;
// a workgroup still steps through the weight chunks in lockstep, so waves whose range is one tile shorter run a
// last iteration on a tile past the table's end (loads return zeros / clamped rows, stores are dropped).
// SAVE: training forward (gnc_mlp_desc_t.save_act): the hidden layers' post-activations are also written
template <int HT, int OT, int WAVES, bool DBUF, bool ADD2, bool AGG = false, bool SAVE = false>
__global__ __launch_bounds__(WAVES * 64) void mlp_stream_kernel(const gnc_mlp_desc_t d, const StreamPlan pl,
                                                                const int num_tiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NT = WAVES * 64;
  constexpr int WT = HT > OT ? HT : OT;
  constexpr int CH = WT * 32 * LDSW;          // floats per weight chunk buffer
  constexpr int PSTRIDE = WT * 32;
  constexpr int RPP = NT / 16;                // weight rows staged per pass
  constexpr int NW = (WT * 32) / RPP;         // float4 registers per thread for one weight chunk
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31;
  const int h = lane >> 5;
  const int c4 = lane & 15;
  const int rs = lane >> 4;
  const int wc4 = tid & 15;  // weight staging role
  const int wr0 = tid >> 4;
  const int L = d.num_linear;
  const int out_dim = d.out_dim[L - 1];
  const int rows = (int)d.rows;
  float* wbuf = lds;                              // [DBUF ? 2 : 1][CH]
  float* pbuf = lds + (DBUF ? 2 : 1) * CH;
  float* abuf = pbuf + (L + 2) * PSTRIDE + wave * RPW * LDSW;

  stage_params<NT>(pbuf, d, PSTRIDE, tid);

  // this thread's share of weight chunk q -> registers (unconditional loads, zero outside the matrix)
  auto wload = [&](f32x4 (&wr)[NW], int q) {
    const int layer = pl.wc[q].layer;
    const float* W = d.weight[layer];
    const int ldw = ldw_of(d, layer);
    const int nrows = d.out_dim[layer];
    const int klimit = pl.wc[q].klimit;
    const int kbase = pl.wc[q].kbase;
    if (kbase + KC <= klimit && ldw % 4 == 0 && (reinterpret_cast<uintptr_t>(W) & 15u) == 0 &&
        (int64_t)nrows * ldw * 4 <= 0xffffffffll) {
      // full 64-column chunk of an aligned matrix: a window over the matrix, rows >= nrows read as 0 through
      // the bounds check, no masks, one 32-bit add per load
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
#pragma unroll
    for (int p = 0; p < NW; ++p) *reinterpret_cast<f32x4*>(buf + (p * RPP + wr0) * LDSW + wc4 * 4) = wr[p];
  };

  // Tile schedule in WAVE tiles (32 rows each).  Default: workgroup-tile t = blockIdx.x + k * gridDim.x, the wave takes
  // rows (t * WAVES + wave) * 32.  AGG: contiguous ranges, see above.  `iters` is workgroup-uniform.
  const int num_wtiles = (rows + RPW - 1) / RPW;
  const int gwave = (int)blockIdx.x * WAVES + wave;
  int agg_t0 = 0, agg_cnt = 0, iters;
  if constexpr (AGG) {
    const int tw = (int)gridDim.x * WAVES, q = num_wtiles / tw, rem = num_wtiles - q * tw;
    agg_t0 = gwave * q + (gwave < rem ? gwave : rem);
    agg_cnt = q + (gwave < rem ? 1 : 0);
    iters = q + ((int)blockIdx.x * WAVES < rem ? 1 : 0);
    if (iters == 0) {  // workgroup-uniform: nothing to do (small batches on the full grid) but to report "no destination"
      if (lane == 0) {
        d.agg_fix[2 * gwave] = -1;
        d.agg_fix[2 * gwave + 1] = -1;
      }
      return;
    }
  } else {
    iters = (num_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  }
  auto wtile_of = [&](int k) -> int {  // wave tile of iteration k (>= num_wtiles: past the end)
    if constexpr (AGG) return k < agg_cnt ? agg_t0 + k : num_wtiles;
    return ((int)blockIdx.x + k * (int)gridDim.x) * WAVES + wave;
  };
  // table row of tile row (lane & 31), clamped so the load is always legal
  auto load_idx = [&](int wt, int s) -> int {
    int r = wt * RPW + (lane & 31);
    r = r < rows ? r : rows - 1;
    const int32_t* ip = d.seg[s].index;
    return ip ? ip[r] : r;
  };
  // `tile_of` = the WAVE tile these rows belong to (row-ordered segments are read through a window at its first row)
  auto load_rows = [&](f32x4 (&pre)[NP], int s, int c0, int idxv, int tile_of) {
    const float* base = d.seg[s].ptr;
    const int ld = d.seg[s].ld;
    const int col = c0 + c4 * 4 < ld ? c0 + c4 * 4 : 0;
    if (d.seg[s].index == nullptr) {
      load_tile_rows(pre, base, ld, (int64_t)tile_of * RPW, rows, (uint32_t)(rs * ld + col) * 4u);
    } else {
      const int64_t tbytes = d.seg[s].table_rows * (int64_t)ld * 4;
      if (tbytes > 0 && tbytes <= 0xffffffffll) {
        // stated table below 4 GiB: bounds-checked window over the table, the row's byte offset travels by
        // ds_bpermute, one 32-bit add per access (an id outside the table reads zeros)
        const __amdgpu_buffer_rsrc_t w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)(uint32_t)tbytes, 0x00020000);
        const int row_bytes = idxv * (ld * 4);
#pragma unroll
        for (int p = 0; p < NP; ++p) pre[p] = window_load(w, (uint32_t)__shfl(row_bytes, p * 4 + rs, 64) + (uint32_t)(col * 4));
      } else {
#pragma unroll
        for (int p = 0; p < NP; ++p) {  // table of 4 GiB or more: flat addresses, the id checked by hand (outside -> zeros)
          const int tr = __shfl(idxv, p * 4 + rs, 64);
          const bool ok = (uint64_t)(int64_t)tr < (uint64_t)d.seg[s].table_rows;
          const f32x4 v = *reinterpret_cast<const f32x4*>(base + (int64_t)(ok ? tr : 0) * ld + col);
          pre[p] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
    }
  };
  auto stage = [&](const f32x4 (&pre)[NP], int c0, int width) {
    compiler_lds_barrier();
    const int c = c0 + c4 * 4;
    if (c0 + KC <= width) {  // full chunk: nothing to mask
#pragma unroll
      for (int p = 0; p < NP; ++p) *reinterpret_cast<f32x4*>(abuf + (p * 4 + rs) * LDSW + c4 * 4) = pre[p];
    } else {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        f32x4 v = pre[p];
        v.x = c + 0 < width ? v.x : 0.f; v.y = c + 1 < width ? v.y : 0.f;
        v.z = c + 2 < width ? v.z : 0.f; v.w = c + 3 < width ? v.w : 0.f;
        *reinterpret_cast<f32x4*>(abuf + (p * 4 + rs) * LDSW + c4 * 4) = v;
      }
    }
    compiler_lds_barrier();
  };

  // gather ids per segment: ids[s] = ids of the tile whose rows are requested next, ids_next[s] = the
  // tile after that (fetched a whole tile before they are needed)
  int ids[GNC_MAX_SEGMENTS], ids_next[GNC_MAX_SEGMENTS];
  int kit = 0;               // iteration counter
  int tile = wtile_of(0);    // wave tile of this iteration
#pragma unroll
  for (int s = 0; s < GNC_MAX_SEGMENTS; ++s) {
    ids[s] = s < d.num_segments ? load_idx(tile, s) : 0;
    ids_next[s] = s < d.num_segments ? load_idx(wtile_of(1), s) : 0;
  }

  f32x4 cur[NP], cur2[NP];  // rows of the next step to stage (cur2: second segment of a combined ADD)
  {
    const int s0 = pl.step[0].seg;
    int idv = 0;
#pragma unroll
    for (int s = 0; s < GNC_MAX_SEGMENTS; ++s) idv = s == s0 ? ids[s] : idv;
    load_rows(cur, s0, pl.step[0].c0, idv, tile);
    if (ADD2 && pl.step[0].seg2 >= 0) {
      const int s2 = pl.step[0].seg2;
#pragma unroll
      for (int s = 0; s < GNC_MAX_SEGMENTS; ++s) idv = s == s2 ? ids[s] : idv;
      load_rows(cur2, s2, pl.step[0].c0, idv, tile);
    }
  }
  // weight chunk 0 -> buffer 0
  f32x4 wreg[NW];
  wload(wreg, 0);
  wstore(wreg, wbuf);
  __syncthreads();
  int gq = 0;  // running chunk counter: chunk q of any tile sits in buffer gq & 1

  // next-chunk bookkeeping shared by every weight-consuming stage
  auto prefetch_next_chunk = [&](int q) {  // q = chunk being computed (index within the tile)
    const int nq = q + 1 < pl.num_wchunks ? q + 1 : 0;
    wload(wreg, nq);
  };
  auto publish_next_chunk = [&]() {
    if constexpr (DBUF) {
      wstore(wreg, wbuf + ((gq + 1) & 1) * CH);
      __syncthreads();
      ++gq;
    } else {
      __syncthreads();  // everyone is done reading the single buffer
      wstore(wreg, wbuf);
      __syncthreads();
    }
  };

  // fused aggregation state (AGG): destination ids of the tile's rows, running sums (lane = feature, per 64 columns)
  constexpr int OCHA = (OT + 1) / 2;
  auto agg_ids = [&](int wt) -> int {
    int r = wt * RPW + (lane & 31);
    r = r < rows ? r : rows - 1;
    return d.agg_index[r];
  };
  int aid = 0, aid_next = 0;
  float agg_acc[OCHA];
#pragma unroll
  for (int c = 0; c < OCHA; ++c) agg_acc[c] = 0.f;
  int agg_cur = -1, agg_first_dst = -1;  // wave-uniform
  bool agg_first = true;
  if constexpr (AGG) aid = agg_ids(tile);

  while (kit < iters) {
    const int row0 = tile * RPW;
    const int ntile = wtile_of(kit + 1);
    if constexpr (AGG) aid_next = agg_ids(ntile);
    int q = 0;

    // ------------------------------------------------------------------ first Linear
    f32x16 hid[HT];
    f32x16 acc2[WT];  // second accumulator set: next hidden layer, then the output layer
    init_bias<HT>(hid, pbuf, h);
    for (int st = 0; st < pl.num_steps; ++st) {
      const int s = pl.step[st].seg, s2 = pl.step[st].seg2, c0 = pl.step[st].c0;
      const bool additive = pl.step[st].add != 0;
      const int width = d.seg[s].width;
      if (ADD2 && s2 >= 0) {
#pragma unroll
        for (int p = 0; p < NP; ++p) cur[p] += cur2[p];
      }
      stage(cur, c0, width);
      // request the next step's rows (this tile, or the first step of the next tile)
      {
        const bool wrap = st + 1 >= pl.num_steps;
        const int nst = wrap ? 0 : st + 1;
        const int ns = pl.step[nst].seg, ns2 = pl.step[nst].seg2, nc0 = pl.step[nst].c0;
        if (wrap) {  // all of this tile's gathers are out: move on to the next tile's ids
#pragma unroll
          for (int k = 0; k < GNC_MAX_SEGMENTS; ++k) {
            ids[k] = ids_next[k];
            if (k < d.num_segments) ids_next[k] = load_idx(wtile_of(kit + 2), k);
          }
        }
        int idv = 0;
#pragma unroll
        for (int k = 0; k < GNC_MAX_SEGMENTS; ++k) idv = k == ns ? ids[k] : idv;
        load_rows(cur, ns, nc0, idv, wrap ? ntile : tile);
        if (ADD2 && ns2 >= 0) {
#pragma unroll
          for (int k = 0; k < GNC_MAX_SEGMENTS; ++k) idv = k == ns2 ? ids[k] : idv;
          load_rows(cur2, ns2, nc0, idv, wrap ? ntile : tile);
        }
      }
      if (additive) {
        switch (c0 >> 6) {  // wave-uniform: keeps the accumulator index static
          case 0: add_chunk_from_lds<HT, 0>(hid, abuf, i, h); break;
          case 1: add_chunk_from_lds<HT, 2>(hid, abuf, i, h); break;
          case 2: add_chunk_from_lds<HT, 4>(hid, abuf, i, h); break;
          default: add_chunk_from_lds<HT, 6>(hid, abuf, i, h); break;
        }
      } else {
        prefetch_next_chunk(q);
        const int kc = width - c0 < KC ? width - c0 : KC;
        mma_chunk_from_lds<HT>(hid, abuf, wbuf + (DBUF ? (gq & 1) : 0) * CH, (kc + 7) >> 3, i, h);
        publish_next_chunk();
        ++q;
      }
    }

    if (L == 1) {
      if (d.ln_gamma) layer_norm_tiles<HT>(hid, pbuf + L * PSTRIDE, pbuf + (L + 1) * PSTRIDE, out_dim, d.ln_eps, h);
    } else {
      relu_tiles<HT>(hid);
      // training forward (gnc_mlp_desc_t.save_act): the post-activations leave as whole rows, 64 columns at a time
      // through the wave's LDS tile (idle between the first Linear's staging and the epilogue)
      auto save_rows = [&](const f32x16 (&acc)[HT], int l) {
        const int width = d.out_dim[l];
#pragma unroll
        for (int cc = 0; cc < (HT + 1) / 2; ++cc) {
          if (cc * KC < width) {
            compiler_lds_barrier();
            switch (cc) {
              case 0: chunk_to_lds<HT, 0>(acc, abuf, i, h); break;
              case 1: chunk_to_lds<HT, 2>(acc, abuf, i, h); break;
              case 2: chunk_to_lds<HT, 4>(acc, abuf, i, h); break;
              default: chunk_to_lds<HT, 6>(acc, abuf, i, h); break;
            }
            compiler_lds_barrier();
            const int fl = fresh_lane();
            store_staged_rows<NP, true>(abuf, d.save_act[l] + cc * KC, width, width - cc * KC < KC ? width - cc * KC : KC, row0, rows,
                                        fl & 15, fl >> 4);
            compiler_lds_barrier();
          }
        }
      };
      if constexpr (SAVE) save_rows(hid, 0);
      // ---------------------------------------------------------------- hidden layers 1 .. L-2
      for (int l = 1; l < L - 1; ++l) {
        f32x16 (&nxt)[HT] = reinterpret_cast<f32x16 (&)[HT]>(acc2);
        init_bias<HT>(nxt, pbuf + l * PSTRIDE, h);
#pragma unroll
        for (int c = 0; c < (HT + 1) / 2; ++c) {
          if (c * KC < d.in_dim[l]) {
            prefetch_next_chunk(q);
            mma_chunk_from_regs<HT, HT>(nxt, hid, wbuf + (DBUF ? (gq & 1) : 0) * CH, c, d.in_dim[l], i, h);
            publish_next_chunk();
            ++q;
          }
        }
        relu_tiles<HT>(nxt);
#pragma unroll
        for (int t = 0; t < HT; ++t) hid[t] = nxt[t];
        if constexpr (SAVE) save_rows(hid, l);
      }
    }

    // ------------------------------------------------------------------ last Linear, LayerNorm, epilogue
    f32x16 (&o)[OT] = reinterpret_cast<f32x16 (&)[OT]>(acc2);
    if (L > 1) {
      init_bias<OT>(o, pbuf + (L - 1) * PSTRIDE, h);
#pragma unroll
      for (int c = 0; c < (HT + 1) / 2; ++c) {
        if (c * KC < d.in_dim[L - 1]) {
          prefetch_next_chunk(q);
          mma_chunk_from_regs<HT, OT>(o, hid, wbuf + (DBUF ? (gq & 1) : 0) * CH, c, d.in_dim[L - 1], i, h);
          publish_next_chunk();
          ++q;
        }
      }
      if (d.ln_gamma) layer_norm_tiles<OT>(o, pbuf + L * PSTRIDE, pbuf + (L + 1) * PSTRIDE, out_dim, d.ln_eps, h);
    } else {
      if constexpr (OT == HT) {
#pragma unroll
        for (int t = 0; t < OT; ++t) o[t] = hid[t];
      }
    }

    // collect the next tile's prefetched rows before the asm stores join the memory queue
#pragma unroll
    for (int p = 0; p < NP; ++p) asm volatile("" ::"v"(cur[p]));

    constexpr int OCH = (OT + 1) / 2;
    float rowv[AGG ? OCH : 1][RPW];  // AGG: final row values per 64-column chunk, lane = feature
#pragma unroll
    for (int cc = 0; cc < OCH; ++cc) {
      if (cc * KC < out_dim) {
        compiler_lds_barrier();
        switch (cc) {  // compile-time after unrolling; keeps the accumulator index static
          case 0: chunk_to_lds<OT, 0>(o, abuf, i, h); break;
          case 1: chunk_to_lds<OT, 2>(o, abuf, i, h); break;
          case 2: chunk_to_lds<OT, 4>(o, abuf, i, h); break;
          default: chunk_to_lds<OT, 6>(o, abuf, i, h); break;
        }
        compiler_lds_barrier();
        if constexpr (AGG) {
          // the residual joins the staged tile first (whole rows), so that the LDS tile holds the FINAL rows
          // that are both stored and summed per destination
          const int wcc = out_dim - cc * KC < KC ? out_dim - cc * KC : KC;
          if (d.residual) {
            f32x4 rv[NP];
            load_tile_rows(rv, d.residual + cc * KC, d.ld_residual, row0, rows, (uint32_t)(rs * d.ld_residual + (c4 * 4 < wcc ? c4 * 4 : 0)) * 4u);
#pragma unroll
            for (int p = 0; p < NP; ++p) {
              float* q = abuf + (p * 4 + rs) * LDSW + c4 * 4;
              *reinterpret_cast<f32x4*>(q) = *reinterpret_cast<const f32x4*>(q) + rv[p];
            }
            compiler_lds_barrier();
          }
          store_staged_rows<NP, true>(abuf, d.out + cc * KC, d.ld_out, wcc, row0, rows, c4, rs);
          // this chunk's 32 row values (lane = feature) are kept; ONE scalar walk over the rows follows the last chunk
#pragma unroll
          for (int r = 0; r < RPW; ++r) rowv[cc][r] = abuf[r * LDSW + lane];
        } else {
          store_staged_rows<NP, true>(abuf, d.out + cc * KC, d.ld_out, out_dim - cc * KC < KC ? out_dim - cc * KC : KC, row0, rows, c4, rs,
                                      d.residual ? d.residual + cc * KC : nullptr, d.ld_residual);
        }
      } else if constexpr (AGG) {
#pragma unroll
        for (int r = 0; r < RPW; ++r) rowv[cc][r] = 0.f;
      }
    }
    if constexpr (AGG) {
      // walk of the tile's rows (see mlp_resident.hip): wave-uniform control, one add per row and column chunk
      const int valid = rows - row0 < RPW ? (rows - row0 > 0 ? rows - row0 : 0) : RPW;
      int prv = __shfl_up(aid, 1, 64);
      prv = lane == 0 ? agg_cur : prv;
      const unsigned long long vmask = valid >= 32 ? 0xffffffffull : ((1ull << valid) - 1ull);
      const unsigned long long bnd = __ballot(aid != prv) & vmask;
#pragma unroll
      for (int r = 0; r < RPW; ++r) {
        if (r < valid) {
          if ((bnd >> r) & 1ull) {
            if (agg_cur >= 0) {
              if (agg_first) {
                agg_first = false;
                agg_first_dst = agg_cur;
              } else {
                float* dstp = d.agg_out + (int64_t)agg_cur * d.ld_agg;
#pragma unroll
                for (int c = 0; c < OCH; ++c)
                  if (c * KC + lane < out_dim)
                    asm volatile("s_nop 4\n\tglobal_store_dword %0, %1, %2\n\ts_nop 0" ::"v"((c * KC + lane) * 4), "v"(agg_acc[c]), "s"(dstp) : "memory");
              }
            }
            agg_cur = __builtin_amdgcn_readlane(aid, r);
#pragma unroll
            for (int c = 0; c < OCH; ++c) agg_acc[c] = 0.f;
          }
#pragma unroll
          for (int c = 0; c < OCH; ++c) agg_acc[c] += rowv[c][r];
        }
      }
      aid = aid_next;
    }
    compiler_lds_barrier();
    tile = ntile;
    ++kit;
  }
  if constexpr (AGG) {
    if (lane == 0) {
      d.agg_fix[2 * gwave] = agg_first ? agg_cur : agg_first_dst;
      d.agg_fix[2 * gwave + 1] = agg_cur;
    }
  }
}

template <int HT, int OT, int WAVES, bool DBUF, bool ADD2, bool AGG = false, bool SAVE = false>
int launch(const gnc_mlp_desc_t& d, const StreamPlan& pl, hipStream_t stream) {
  constexpr int WT = HT > OT ? HT : OT;
  const size_t smem =
      ((size_t)(DBUF ? 2 : 1) * WT * 32 * LDSW + (size_t)(d.num_linear + 2) * WT * 32 + (size_t)WAVES * RPW * LDSW) * sizeof(float);
  if (smem > 160 * 1024) {
    gnc::set_error("mlp_stream: LDS budget exceeded (%zu bytes)", smem);
    return GNC_ERR_UNSUPPORTED;
  }
  static bool attr_set = false;
  if (!attr_set) {
    int rc = gnc::check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_stream_kernel<HT, OT, WAVES, DBUF, ADD2, AGG, SAVE>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                            "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    if (rc) return rc;
    attr_set = true;
  }
  const int64_t num_tiles = gnc::ceil_div(d.rows, (int64_t)WAVES * RPW);
  int64_t grid = num_tiles < gnc::num_cu() ? num_tiles : gnc::num_cu();  // one persistent workgroup per CU
  if constexpr (AGG) grid = gnc::num_cu() * 8 / WAVES;  // agg_fix has two entries for each of the 8 x num_cu waves of the full grid
  mlp_stream_kernel<HT, OT, WAVES, DBUF, ADD2, AGG, SAVE><<<dim3((unsigned)grid), dim3(WAVES * 64), smem, stream>>>(d, pl, (int)num_tiles);
  return gnc::check_launch("mlp_stream_kernel");
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

int gnc_mlp::launch_stream(const gnc_mlp_desc_t& d, int T, bool narrow_out, hipStream_t stream, bool* launched,
                           bool probe_only) {
  *launched = false;
  static const bool disabled = getenv("GNC_MLP_NO_STREAM2") != nullptr;  // A/B switch for benchmarking
  if (disabled || d.rows >= INT32_MAX - (1 << 22)) return GNC_OK;  // row numbers of prefetched tiles stay below 2^31
  const int L = d.num_linear;
  if (L > 1 && d.activation != GNC_ACT_RELU) return GNC_OK;
  if (d.residual && (d.ld_residual % 4 != 0 || !al16(d.residual))) return GNC_OK;
  if (d.save_act[0]) {  // saved post-activations: whole rows of 16-B pieces
    if (L < 2) return GNC_OK;
    for (int l = 0; l < L - 1; ++l)
      if (!d.save_act[l] || d.out_dim[l] % 4 != 0 || !al16(d.save_act[l])) return GNC_OK;
  }
  for (int l = 0; l < L; ++l)
    if (ldw_of(d, l) % 4 != 0 || !al16(d.weight[l])) return GNC_OK;

  StreamPlan pl = {};
  // steps: MATMUL segments in listed order, then the additive ones (pairs merged)
  int add_seg[GNC_MAX_SEGMENTS], nadd = 0;
  for (int s = 0; s < d.num_segments; ++s) {
    const gnc_mlp_segment_t& g = d.seg[s];
    if (g.ld % 4 != 0 || !al16(g.ptr)) return GNC_OK;
    if (g.mode == GNC_SEG_ADD) { add_seg[nadd++] = s; continue; }
    if (g.wcol % 4 != 0) return GNC_OK;
    for (int c0 = 0; c0 < g.width; c0 += KC) {
      if (pl.num_steps >= MAX_STEPS || pl.num_wchunks >= MAX_WCHUNKS) return GNC_OK;
      pl.step[pl.num_steps++] = {(short)s, (short)-1, (short)c0, (short)0};
      const int kl = g.wcol + g.width;
      pl.wc[pl.num_wchunks++] = {(short)0, (short)(g.wcol + c0), (short)kl, 0};
    }
  }
  if (pl.num_wchunks == 0) return GNC_OK;
  const int pair = T < 4 ? 2 : 1;  // the 128/256-wide instances stage additive segments one by one
  for (int a = 0; a < nadd; a += pair) {
    const int s = add_seg[a], s2 = (pair == 2 && a + 1 < nadd) ? add_seg[a + 1] : -1;
    for (int c0 = 0; c0 < d.seg[s].width; c0 += KC) {
      if (pl.num_steps >= MAX_STEPS) return GNC_OK;
      pl.step[pl.num_steps++] = {(short)s, (short)s2, (short)c0, (short)1};
    }
  }
  for (int l = 1; l < L; ++l)
    for (int c = 0; c * KC < d.in_dim[l]; ++c) {
      if (pl.num_wchunks >= MAX_WCHUNKS) return GNC_OK;
      pl.wc[pl.num_wchunks++] = {(short)l, (short)(c * KC), (short)d.in_dim[l], 0};
    }

  const bool save = d.save_act[0] != nullptr;  // training forward: SAVE instances of the 128-wide kernel (a template flag:
  if (save && (T != 4 || narrow_out)) return GNC_OK;  // the run-time branch alone cost the c2 inference forward 6 %)
  if (d.agg_out) {  // fused aggregation epilogue: the 128-wide instance, whole 16-B output pieces
    const int od = d.out_dim[L - 1];
    if (!(T == 4 && !narrow_out && od % 4 == 0 && d.ld_out % 4 == 0 && al16(d.out) && d.agg_index && d.agg_fix && d.ld_agg >= od))
      return GNC_OK;
    *launched = true;
    if (probe_only) return GNC_OK;
    // small batches (the reference's one-graph-per-call regime): 2-wave workgroups as below, four times as many of them
    if (d.rows <= (int64_t)2 * RPW * gnc::num_cu())
      return save ? launch<4, 4, 2, true, false, true, true>(d, pl, stream) : launch<4, 4, 2, true, false, true>(d, pl, stream);
    return save ? launch<4, 4, 8, true, false, true, true>(d, pl, stream) : launch<4, 4, 8, true, false, true>(d, pl, stream);
  }
  *launched = true;
  if (probe_only) return GNC_OK;
  if (save) {
    const bool small_rows = d.rows <= (int64_t)2 * RPW * gnc::num_cu();
    return small_rows ? launch<4, 4, 2, true, false, false, true>(d, pl, stream) : launch<4, 4, 8, true, false, false, true>(d, pl, stream);
  }
  // Small batches (the reference's own regime: ONE ~1000-node graph per call, main.py:60): with 8-wave workgroups a
  // 2,000-row launch is 8 workgroups, each streaming the whole weight sequence with nothing to overlap it (38 us per
  // forward launch, 145 us per backward launch in the captured training step's trace).  2-wave workgroups spread the
  // same rows over 4x as many CUs; the weights come out of L2 either way.
  const bool small = T == 4 && d.rows <= (int64_t)2 * RPW * gnc::num_cu();
  if (small) return narrow_out ? launch<4, 1, 2, true, true>(d, pl, stream) : launch<4, 4, 2, true, false>(d, pl, stream);
  if (narrow_out) {  // out width <= 32 (the decoder): one output tile
    switch (T) {
      case 1: return launch<1, 1, 8, true, true>(d, pl, stream);
      case 2: return launch<2, 1, 8, true, true>(d, pl, stream);
      case 4: return launch<4, 1, 4, true, true>(d, pl, stream);
      default: return launch<8, 1, 4, false, false>(d, pl, stream);
    }
  }
  switch (T) {
    case 1: return launch<1, 1, 8, true, true>(d, pl, stream);
    case 2: return launch<2, 2, 8, true, true>(d, pl, stream);
    case 4: {
      static const bool w4 = getenv("GNC_STREAM_W4") != nullptr;  // A/B: 4 waves x 512 registers instead of 8 x 256
      return w4 ? launch<4, 4, 4, true, false>(d, pl, stream) : launch<4, 4, 8, true, false>(d, pl, stream);
    }
    default: return launch<8, 8, 4, false, false>(d, pl, stream);
  }
}
