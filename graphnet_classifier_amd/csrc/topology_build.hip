// Topology build for graph-ordered batches: edge_index -> (rowptr, perm, src_sorted, dst_sorted) in ONE pass family,
// without a global radix sort.
//
// The reference's batches are disjoint unions of small graphs whose edges are contiguous in edge_index
// (utils/dataloader.py:33-53 yields one self-contained graph per item; every BASELINE config is such a batch).  The
// stable destination sort of such a list decomposes: position i of the edge list is a CUT when every destination before
// it is smaller than every destination from it on; the sorted order of the whole list is then the sorted order of the
// part before the cut followed by the sorted order of the part after it, and the parts can be sorted independently, in
// place.  Cuts are found from per-tile min / max (two tiny kernels), then every workgroup owns the range from the first
// cut inside its 2048-edge tile to the first cut at or behind the tile's end and sorts it in LDS: counting sort over the
// range's destination span (integer LDS atomics; the order inside a destination is then fixed by ranking every edge
// among the edge ids of its destination = ascending id = the reference's edge order, models/GNN.py:18-20), and writes
// the row pointers, the permutation and both permuted endpoint vectors itself.  Deterministic.
//
// Anything the LDS path cannot take (a range longer than 4096 edges, i.e. a graph with more than 2048 edges or an
// edge list that is not graph-ordered; a destination span above 4096 nodes) raises a DEVICE flag
// (status[2]); the general path - a stable LSD radix sort written for this file, ONE kernel that returns at once
// unless the flag is set - is enqueued behind it, so the call never needs the host to look at the flag (capturable, no
// sync).  Callers that synchronise anyway may pass gated_fallback = 0 and run gnc_csr_build (rocPRIM) when they see the flag.
#include <stdlib.h>

#include "gnc_common.h"

namespace {

constexpr int TT = 2048;     // edges per tile
constexpr int CAP = 2 * TT;  // edges a workgroup can hold: its tile and the next one
constexpr int KMAX = 4096;   // destination span of one owned range (with the other LDS arrays: 3 workgroups per CU)
constexpr int NET = 32;      // in-degrees up to this are sorted by a register network, larger ones by the whole workgroup
constexpr int BIGQ = CAP / (NET + 1) + 1;
constexpr int NT = 256;
constexpr int PER = CAP / NT;  // consecutive window positions per thread (16)
constexpr int IMAX = 0x7fffffff;

template <typename IdxT>
__device__ __forceinline__ int checked_id(IdxT v, int64_t n, bool* bad) {
  const bool ok = (v >= 0) & ((int64_t)v < n);
  *bad |= !ok;
  return ok ? (int)v : 0;  // memory-safe stand-in; the flag makes the host raise (or the forward poison its output)
}

// ---- pass A: per-tile min / max of the (sanitised) destinations.  Also clears the three status words for the kernels behind it
// (pass C re-reads and validates every destination, so nothing is flagged here).  NOT a hipMemsetAsync: inside a captured
// hipGraph the runtime's memset node was seen to write a foreign byte pattern (0x04...) into the flags on replays that followed
// an unrelated reduction launched outside the graph (round 3, any-topology training step) - the flags of a capturable build are
// written by this library's own kernels only.
template <typename IdxT>
__global__ __launch_bounds__(NT) void topo_tile_minmax(const IdxT* __restrict__ dst, int64_t E, int64_t N, int* __restrict__ tmin,
                                                       int* __restrict__ tmax, int* __restrict__ status) {
  __shared__ int smin[NT / 64], smax[NT / 64];
  if (blockIdx.x == 0 && threadIdx.x < 3) status[threadIdx.x] = 0;
  const int64_t base = (int64_t)blockIdx.x * TT;
  int mn = IMAX, mx = -1;
  bool bad = false;
#pragma unroll
  for (int k = 0; k < TT / NT; ++k) {
    const int64_t e = base + k * NT + threadIdx.x;
    if (e < E) {
      const int v = checked_id(dst[e], N, &bad);
      mn = v < mn ? v : mn;
      mx = v > mx ? v : mx;
    }
  }
  (void)bad;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int a = __shfl_xor(mn, o, 64), b = __shfl_xor(mx, o, 64);
    mn = a < mn ? a : mn;
    mx = b > mx ? b : mx;
  }
  if ((threadIdx.x & 63) == 0) {
    smin[threadIdx.x >> 6] = mn;
    smax[threadIdx.x >> 6] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < NT / 64; ++w) {
      mn = smin[w] < mn ? smin[w] : mn;
      mx = smax[w] > mx ? smax[w] : mx;
    }
    tmin[blockIdx.x] = mn;
    tmax[blockIdx.x] = mx;
  }
}

// an empty edge list: row pointers and flags cleared by a kernel (see pass A for why not a memset)
__global__ void topo_clear(int* __restrict__ rowptr, int64_t count, int* __restrict__ status) {
  if (blockIdx.x == 0 && threadIdx.x < 3) status[threadIdx.x] = 0;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < count; k += (int64_t)gridDim.x * blockDim.x) rowptr[k] = 0;
}

// ---- pass B (one workgroup): P[t] = max over tiles before t (-1), S[t] = min over tiles after t (IMAX)
__global__ __launch_bounds__(1024) void topo_tile_scan(const int* __restrict__ tmin, const int* __restrict__ tmax, int num_tiles,
                                                       int* __restrict__ P, int* __restrict__ S, unsigned* __restrict__ fb_bar) {
  __shared__ int cmax[1024], cmin[1024];
  if (fb_bar && threadIdx.x == 0) *fb_bar = 0;  // arrival counter of the general path's grid barrier
  const int per = (num_tiles + 1023) / 1024;
  const int t0 = threadIdx.x * per, t1 = t0 + per < num_tiles ? t0 + per : num_tiles;
  int mx = -1, mn = IMAX;
  for (int t = t0; t < t1; ++t) {
    mx = tmax[t] > mx ? tmax[t] : mx;
    mn = tmin[t] < mn ? tmin[t] : mn;
  }
  // exclusive prefix max / exclusive suffix min over the 1024 per-thread values: wave scans + the 16 wave totals
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int pmx = mx, smn = mn;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int a = __shfl_up(pmx, o, 64);
    if (lane >= o) pmx = a > pmx ? a : pmx;
    const int b = __shfl_down(smn, o, 64);
    if (lane + o < 64) smn = b < smn ? b : smn;
  }
  if (lane == 63) cmax[wave] = pmx;
  if (lane == 0) cmin[wave] = smn;
  __syncthreads();
  int ex = __shfl_up(pmx, 1, 64);
  ex = lane == 0 ? -1 : ex;
  for (int w = 0; w < wave; ++w) ex = cmax[w] > ex ? cmax[w] : ex;
  int sx = __shfl_down(smn, 1, 64);
  sx = lane == 63 ? IMAX : sx;
  for (int w = 15; w > wave; --w) sx = cmin[w] < sx ? cmin[w] : sx;
  __syncthreads();
  cmax[threadIdx.x] = ex;
  cmin[threadIdx.x] = sx;
  __syncthreads();
  int run = cmax[threadIdx.x];
  for (int t = t0; t < t1; ++t) {
    P[t] = run;
    run = tmax[t] > run ? tmax[t] : run;
  }
  run = cmin[threadIdx.x];
  for (int t = t1 - 1; t >= t0; --t) {
    S[t] = run;
    run = tmin[t] < run ? tmin[t] : run;
  }
}

// ---- pass C: every workgroup sorts the range that starts at the first cut inside its tile
template <typename IdxT, bool WITH_ENDPOINTS>
__global__ __launch_bounds__(NT) void topo_tile_sort(const IdxT* __restrict__ dst, const IdxT* __restrict__ src, int64_t E, int64_t N,
                                                     const int* __restrict__ P, const int* __restrict__ S, int num_tiles,
                                                     int* __restrict__ rowptr, int* __restrict__ perm, int* __restrict__ src_sorted,
                                                     int* __restrict__ dst_sorted, int* __restrict__ status, int phase_limit,
                                                     unsigned* __restrict__ single_tile_bar) {
  __shared__ int skey[CAP];
  __shared__ int hist[KMAX];
  __shared__ unsigned short sidx[CAP];
  __shared__ int wmax[NT / 64], wmin[NT / 64];
  __shared__ int bigq[BIGQ];
  __shared__ int sh_c0, sh_c1, sh_lo, sh_hi, sh_mprev, sh_nbig;
  const int t = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t wbase = (int64_t)t * TT;
  const int n_win = (int)((E - wbase) < CAP ? (E - wbase) : CAP);
  const int last_tile = (int)((wbase + n_win - 1) / TT);  // last tile the window touches
  bool bad = false;
  // ONE tile (a single small graph: the reference's one-graph-per-call regime): passes A and B are not launched - there is
  // nothing in front of or behind the window - and this workgroup clears the flags and the general path's counter itself
  const bool single = num_tiles == 1 && single_tile_bar != nullptr;
  if (single) {
    if (tid < 3) status[tid] = 0;
    if (tid == 3) *single_tile_bar = 0;
  }
  for (int k = tid; k < CAP; k += NT) skey[k] = k < n_win ? checked_id(dst[wbase + k], N, &bad) : IMAX;
  if (tid == 0) {
    sh_c0 = IMAX; sh_c1 = IMAX; sh_lo = IMAX; sh_hi = -1; sh_mprev = -1; sh_nbig = 0;
  }
  __syncthreads();
  // exclusive prefix max / inclusive suffix min at this thread's PER consecutive positions
  int kv[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) kv[k] = skey[tid * PER + k];
  int lmax = -1, lmin = IMAX;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    if (tid * PER + k < n_win) lmax = kv[k] > lmax ? kv[k] : lmax;
    lmin = kv[k] < lmin ? kv[k] : lmin;  // positions past the window hold IMAX
  }
  // across threads: wave scans by shuffles, the four wave totals through LDS
  int pmax = lmax, smin = lmin;  // inclusive scans first
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int a = __shfl_up(pmax, o, 64);
    if (lane >= o) pmax = a > pmax ? a : pmax;
    const int b = __shfl_down(smin, o, 64);
    if (lane + o < 64) smin = b < smin ? b : smin;
  }
  if (lane == 63) wmax[wave] = pmax;
  if (lane == 0) wmin[wave] = smin;
  __syncthreads();
  int before = single ? -1 : P[t];  // everything in front of the window
  for (int w = 0; w < wave; ++w) before = wmax[w] > before ? wmax[w] : before;
  int after = single ? IMAX : S[last_tile];  // everything behind the window
  for (int w = NT / 64 - 1; w > wave; --w) after = wmin[w] < after ? wmin[w] : after;
  int ex = __shfl_up(pmax, 1, 64);   // exclusive prefix max of the threads before this one
  ex = lane == 0 ? -1 : ex;
  ex = ex > before ? ex : before;
  int sx = __shfl_down(smin, 1, 64);  // suffix min of the threads behind this one
  sx = lane == 63 ? IMAX : sx;
  sx = sx < after ? sx : after;
  // per position: pm[k] = max of everything before it, sm[k] = min of everything from it on
  int pm[PER + 1], sm[PER + 1];
  pm[0] = ex;
#pragma unroll
  for (int k = 0; k < PER; ++k) pm[k + 1] = (tid * PER + k < n_win && kv[k] > pm[k]) ? kv[k] : pm[k];
  sm[PER] = sx;
#pragma unroll
  for (int k = PER - 1; k >= 0; --k) sm[k] = kv[k] < sm[k + 1] ? kv[k] : sm[k + 1];
  int c0 = IMAX, c1 = IMAX;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int pos = tid * PER + k;
    const bool cut = pos <= n_win && pm[k] < sm[k];  // pos == n_win < CAP: the end of the list (sm = IMAX)
    if (cut && pos < TT && pos < n_win) c0 = pos < c0 ? pos : c0;
    if (cut && pos >= TT) c1 = pos < c1 ? pos : c1;
  }
  if (tid == NT - 1 && n_win == CAP && pm[PER] < sm[PER]) c1 = CAP < c1 ? CAP : c1;  // the boundary behind the window
  if (n_win < TT && tid == 0) c1 = n_win;  // a last tile shorter than TT: the list ends inside it
  if (c0 != IMAX) atomicMin(&sh_c0, c0);
  if (c1 != IMAX) atomicMin(&sh_c1, c1);
  __syncthreads();
  c0 = sh_c0;
  c1 = sh_c1;
  if (bad) status[0] = 1;
  if (phase_limit == 1) return;
  if (c0 == IMAX) return;  // no range starts in this tile
  if (c1 == IMAX || c1 > n_win) {
    if (tid == 0) status[2] = 1;  // the range does not end inside the window: general path
    return;
  }
  // ---- the owned range [c0, c1): span of its destinations
  int rmin = IMAX, rmax = -1;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int pos = tid * PER + k;
    if (pos >= c0 && pos < c1) {
      rmin = kv[k] < rmin ? kv[k] : rmin;
      rmax = kv[k] > rmax ? kv[k] : rmax;
    }
    if (pos == c0) sh_mprev = pm[k];
  }
  // wave reduction first: 256 same-address LDS atomics serialise (measured: 30 us of the kernel at 10 M edges)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int a = __shfl_xor(rmin, o, 64), b = __shfl_xor(rmax, o, 64);
    rmin = a < rmin ? a : rmin;
    rmax = b > rmax ? b : rmax;
  }
  if (lane == 0 && rmin != IMAX) {
    atomicMin(&sh_lo, rmin);
    atomicMax(&sh_hi, rmax);
  }
  __syncthreads();
  const int lo = sh_lo, hi = sh_hi, mprev = sh_mprev;
  const int span = hi - lo + 1;
  if (span > KMAX) {
    if (tid == 0) status[2] = 1;
    return;
  }
  for (int k = tid; k < span; k += NT) hist[k] = 0;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int pos = tid * PER + k;
    if (pos >= c0 && pos < c1) atomicAdd(&hist[kv[k] - lo], 1);
  }
  __syncthreads();
  if (phase_limit == 2) return;
  // exclusive scan of the counts (each thread a contiguous slice of bins), row pointers on the way
  const int bper = (span + NT - 1) / NT;
  const int b0 = tid * bper, b1 = b0 + bper < span ? b0 + bper : span;
  int tot = 0;
  for (int k = b0; k < b1; ++k) tot += hist[k];
  int inc = tot;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int a = __shfl_up(inc, o, 64);
    if (lane >= o) inc += a;
  }
  if (lane == 63) wmax[wave] = inc;
  __syncthreads();
  int off = inc - tot;
  for (int w = 0; w < wave; ++w) off += wmax[w];
  const int64_t gbase = wbase + c0;  // sorted position of the range's first edge = its position in the list
  for (int k = b0; k < b1; ++k) {
    const int c = hist[k];
    hist[k] = off;
    rowptr[lo + k] = (int)(gbase + off);
    off += c;
  }
  // destinations without edges between the previous range and this one, and behind the last range
  for (int v = mprev + 1 + tid; v < lo; v += NT) rowptr[v] = (int)gbase;
  if (wbase + c1 == E)
    for (int64_t v = (int64_t)hi + 1 + tid; v <= N; v += NT) rowptr[v] = (int)E;
  __syncthreads();
  if (phase_limit == 3) return;
  // placement (order inside a destination arbitrary for now)
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int pos = tid * PER + k;
    if (pos >= c0 && pos < c1) sidx[atomicAdd(&hist[kv[k] - lo], 1)] = (unsigned short)pos;
  }
  __syncthreads();
  if (phase_limit == 4) return;
  // Stable order inside a destination = ascending edge id (the reference's summation order): every destination's list
  // is sorted in place.  hist[k] is now the END of bin k.  Lists of up to NET ids: one thread per destination, the list
  // in registers, a bitonic network (static indices, min / max only; per-step LDS round trips - an insertion sort by one
  // thread, or every edge counting the smaller ids of its list - measured 10x slower).  Longer lists (hubs) are queued
  // and ranked by the whole workgroup afterwards.
  for (int k = tid; k < span; k += NT) {
    const int s = k == 0 ? 0 : hist[k - 1], d = hist[k] - s;
    if (d > NET) {
      bigq[atomicAdd(&sh_nbig, 1)] = k;
    } else if (d > 1) {
      int v[NET];
#pragma unroll
      for (int i = 0; i < NET; ++i) v[i] = i < d ? (int)sidx[s + i] : IMAX;
#pragma unroll
      for (int kk = 2; kk <= NET; kk <<= 1)
#pragma unroll
        for (int j = kk >> 1; j > 0; j >>= 1)
#pragma unroll
          for (int i = 0; i < NET; ++i) {
            const int l = i ^ j;
            if (l > i) {
              const int a = v[i], b = v[l];
              const bool up = (i & kk) == 0;
              v[i] = up ? (a < b ? a : b) : (a > b ? a : b);
              v[l] = up ? (a > b ? a : b) : (a < b ? a : b);
            }
          }
#pragma unroll
      for (int i = 0; i < NET; ++i)
        if (i < d) sidx[s + i] = (unsigned short)v[i];
    }
  }
  __syncthreads();
  const int nbig = sh_nbig;
  for (int b = 0; b < nbig; ++b) {  // workgroup-uniform
    const int k = bigq[b];
    const int s = k == 0 ? 0 : hist[k - 1], d = hist[k] - s;
    int mine[PER], rk[PER];  // d <= CAP: at most PER ids per thread
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int i = u * NT + tid;
      mine[u] = i < d ? (int)sidx[s + i] : IMAX;
      rk[u] = 0;
    }
    for (int q = 0; q < d; ++q) {
      const int o = sidx[s + q];  // broadcast read
#pragma unroll
      for (int u = 0; u < PER; ++u) rk[u] += o < mine[u] ? 1 : 0;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PER; ++u)
      if (u * NT + tid < d) sidx[s + rk[u]] = (unsigned short)mine[u];
    __syncthreads();
  }
  if (phase_limit == 5) return;
  const int n = c1 - c0;
  bool bad_src = false;
  for (int q = tid; q < n; q += NT) {
    const int li = sidx[q];
    perm[gbase + q] = (int)(wbase + li);
    if constexpr (WITH_ENDPOINTS) {
      dst_sorted[gbase + q] = skey[li];
      src_sorted[gbase + q] = checked_id(src[wbase + li], N, &bad_src);
    }
  }
  if (bad_src) status[1] = 1;
}

// ---- general path: stable LSD radix sort, 8-bit digits, ONE kernel gated by status[2] ---------------------------------
// The LDS path enqueues this kernel behind itself on every call (no host look at the flag: capturable), so what it costs
// when the flag is NOT set matters: one launch that returns at once (the first version was 11 gated launches per call, 50 us
// of dispatch at c3).  When the flag IS set the passes of the sort are separated by a grid barrier: the grid is at most two
// one-wave workgroups per CU (64 KB of LDS each), i.e. every workgroup is resident, and the barrier gives up after
// FB_TIMEOUT ticks of the 100 MHz clock (status[2] = 2, status[0] = 1: every caller's validity check fails loudly) rather
// than spin for ever if that assumption is ever broken.
constexpr int FB_NT = 64;       // one wave per block: cnt[256][64] counters in LDS
constexpr int FB_MAXB = 512;    // blocks (= chunks of the list)
constexpr unsigned long long FB_TIMEOUT = 500000000ull;  // 5 s

struct FbGeom {
  int nblocks;
  int chunk;  // elements per block
  int run;    // consecutive elements per thread
};

FbGeom fb_geom(int64_t E) {
  FbGeom g;
  int64_t nb = (E + 4095) / 4096;
  int64_t cap = (int64_t)gnc::num_cu() * 2;  // co-resident: the grid barrier needs every block on the chip
  cap = cap < FB_MAXB ? cap : FB_MAXB;
  g.nblocks = (int)(nb < 1 ? 1 : (nb > cap ? cap : nb));
  g.chunk = (int)((E + g.nblocks - 1) / g.nblocks);
  g.run = (g.chunk + FB_NT - 1) / FB_NT;
  return g;
}

// every block arrives once per barrier; the counter only grows (zeroed by topo_tile_scan in front of the sort kernels)
__device__ __forceinline__ bool fb_grid_barrier(unsigned* bar, unsigned target) {
  __shared__ int ok_s;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = wall_clock64();
    int ok = 1;
    while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(16);
      if (wall_clock64() - t0 > FB_TIMEOUT) {
        ok = 0;
        break;
      }
    }
    ok_s = ok;
  }
  __syncthreads();
  __threadfence();
  return ok_s != 0;
}

template <typename IdxT, bool WITH_ENDPOINTS>
__global__ __launch_bounds__(FB_NT) void fb_sort_all(const int* gate, const IdxT* dst, const IdxT* src, int64_t E, int64_t N, FbGeom g,
                                                     int bits, unsigned* keys_a, unsigned* keys_b, int* vals_a, int* vals_b,
                                                     unsigned* tab, unsigned* part, unsigned* bar, int* rowptr, int* perm,
                                                     int* src_sorted, int* dst_sorted, int* status) {
  if (*gate == 0) return;
  __shared__ unsigned cnt[256][FB_NT];
  const int tid = threadIdx.x, blk = blockIdx.x, nb = g.nblocks;
  const int64_t gtid = (int64_t)blk * FB_NT + tid, gsz = (int64_t)nb * FB_NT;
  unsigned arrivals = 0;
#define FB_BARRIER()                                         \
  do {                                                       \
    arrivals += (unsigned)nb;                                \
    if (!fb_grid_barrier(bar, arrivals)) {                   \
      if (tid == 0) {                                        \
        status[2] = 2;                                       \
        status[0] = 1;                                       \
      }                                                      \
      return;                                                \
    }                                                        \
  } while (0)
  {  // keys = validated destinations, values = edge ids
    bool bad = false;
    for (int64_t e = gtid; e < E; e += gsz) {
      keys_a[e] = (unsigned)checked_id(dst[e], N, &bad);
      vals_a[e] = (int)e;
    }
    if (bad) status[0] = 1;
  }
  FB_BARRIER();
  unsigned *ka = keys_a, *kb = keys_b;
  int *va = vals_a, *vb = vals_b;
  const int64_t e0 = (int64_t)blk * g.chunk + (int64_t)tid * g.run;
  int64_t e1 = e0 + g.run;
  {
    const int64_t cend = ((int64_t)blk + 1) * g.chunk;
    e1 = e1 < cend ? e1 : cend;
    e1 = e1 < E ? e1 : E;
  }
  for (int shift = 0; shift < bits; shift += 8) {
    // (1) per-thread digit counts of this block's chunk (kept in LDS for the placement), per-block totals digit-major
    for (int d = 0; d < 256; ++d) cnt[d][tid] = 0;
    for (int64_t e = e0; e < e1; ++e) ++cnt[(ka[e] >> shift) & 255u][tid];
    __syncthreads();
    for (int d = tid; d < 256; d += FB_NT) {
      unsigned s = 0;
      for (int k = 0; k < FB_NT; ++k) s += cnt[d][k];
      tab[(int64_t)d * nb + blk] = s;  // digit-major: the scan order of an LSD pass
    }
    FB_BARRIER();
    // (2) exclusive scan of the 256 * nb totals: block b owns entries [256 b, 256 b + 256); slice totals first ...
    unsigned v[4], incl[4];
    {
      unsigned run = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = tab[(int64_t)blk * 256 + j * 64 + tid];
        unsigned x = v[j];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const unsigned a = __shfl_up(x, o, 64);
          if (tid >= o) x += a;
        }
        incl[j] = x + run;
        run += __shfl(x, 63, 64);
      }
      if (tid == 0) part[blk] = run;
    }
    FB_BARRIER();
    // ... then every block adds the totals of the slices in front of its own and writes its slice back
    {
      unsigned off = 0;
      for (int b = tid; b < blk; b += FB_NT) off += part[b];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) off += __shfl_xor(off, o, 64);
#pragma unroll
      for (int j = 0; j < 4; ++j) tab[(int64_t)blk * 256 + j * 64 + tid] = off + incl[j] - v[j];
    }
    FB_BARRIER();
    // (3) placement: first output slot of (digit d, thread k) - stable in list order
    for (int d = tid; d < 256; d += FB_NT) {
      unsigned run = tab[(int64_t)d * nb + blk];
      for (int k = 0; k < FB_NT; ++k) {
        const unsigned c = cnt[d][k];
        cnt[d][k] = run;
        run += c;
      }
    }
    __syncthreads();
    for (int64_t e = e0; e < e1; ++e) {
      const unsigned k = ka[e];
      const unsigned pos = cnt[(k >> shift) & 255u][tid]++;
      kb[pos] = k;
      vb[pos] = va[e];
    }
    FB_BARRIER();
    unsigned* tk = ka; ka = kb; kb = tk;
    int* tv = va; va = vb; vb = tv;
  }
#undef FB_BARRIER
  // row pointers from the boundaries of the sorted keys, permutation, permuted endpoints
  bool bad = false;
  for (int64_t i = gtid; i <= E; i += gsz) {
    const int64_t prev = i == 0 ? -1 : (int64_t)ka[i - 1];
    const int64_t cur = i == E ? N : (int64_t)ka[i];
    for (int64_t q = prev + 1; q <= cur; ++q) rowptr[q] = (int)i;
    if (i < E) {
      perm[i] = va[i];
      if constexpr (WITH_ENDPOINTS) {
        dst_sorted[i] = (int)ka[i];
        src_sorted[i] = checked_id(src[va[i]], N, &bad);
      }
    }
  }
  if (bad) status[1] = 1;
}

constexpr size_t kAlign = 256;
inline size_t align_up(size_t x) { return (x + kAlign - 1) / kAlign * kAlign; }

unsigned key_bits(int64_t n) {  // bits of the largest key, n - 1
  unsigned b = 1;
  while (b < 32 && ((uint64_t)(n > 0 ? n - 1 : 0) >> b) != 0) ++b;
  return b;
}

struct Workspace {
  int *tmin, *tmax, *P, *S;
  unsigned *keys_a, *keys_b, *tab, *part, *bar;
  int *vals_a, *vals_b;
  size_t bytes;
};

Workspace carve(void* base_, int64_t E, bool fallback) {
  const int64_t tiles = (E + TT - 1) / TT;
  uintptr_t p = (reinterpret_cast<uintptr_t>(base_) + kAlign - 1) / kAlign * kAlign;
  const uintptr_t p0 = p;
  auto take = [&](size_t n) { uintptr_t q = p; p += align_up(n); return q; };
  Workspace w;
  w.tmin = reinterpret_cast<int*>(take((size_t)tiles * 4));
  w.tmax = reinterpret_cast<int*>(take((size_t)tiles * 4));
  w.P = reinterpret_cast<int*>(take((size_t)tiles * 4));
  w.S = reinterpret_cast<int*>(take((size_t)tiles * 4));
  w.keys_a = w.keys_b = w.tab = w.part = w.bar = nullptr;
  w.vals_a = w.vals_b = nullptr;
  if (fallback) {
    w.keys_a = reinterpret_cast<unsigned*>(take((size_t)E * 4));
    w.keys_b = reinterpret_cast<unsigned*>(take((size_t)E * 4));
    w.vals_a = reinterpret_cast<int*>(take((size_t)E * 4));
    w.vals_b = reinterpret_cast<int*>(take((size_t)E * 4));
    w.tab = reinterpret_cast<unsigned*>(take((size_t)256 * FB_MAXB * 4));
    w.part = reinterpret_cast<unsigned*>(take((size_t)FB_MAXB * 4));
    w.bar = reinterpret_cast<unsigned*>(take(4));
  }
  w.bytes = (size_t)(p - p0) + kAlign;
  return w;
}

template <typename IdxT, bool WITH_ENDPOINTS>
int build(const IdxT* src, const IdxT* dst, int64_t E, int64_t N, int* rowptr, int* perm, int* src_sorted, int* dst_sorted, int* status,
          void* workspace, size_t workspace_bytes, int gated_fallback, hipStream_t stream) {
  int rc = GNC_OK;
  if (E == 0) {
    int64_t cb = gnc::ceil_div(N + 1, 256);
    cb = cb < 1024 ? cb : 1024;
    topo_clear<<<(unsigned)cb, 256, 0, stream>>>(rowptr, N + 1, status);
    return gnc::check_launch("topo_clear");
  }
  const Workspace w = carve(workspace, E, gated_fallback != 0);
  if (workspace_bytes < w.bytes) {
    gnc::set_error("gnc_topology_build: workspace too small (%zu < %zu)", workspace_bytes, w.bytes);
    return GNC_ERR_WORKSPACE;
  }
  const int tiles = (int)((E + TT - 1) / TT);
  // the general path's barrier counter; without a gated general path a dummy word of the workspace (tmin) takes the store
  unsigned* single_bar = tiles == 1 ? (w.bar ? w.bar : reinterpret_cast<unsigned*>(w.tmin)) : nullptr;
  if (tiles > 1) {
    topo_tile_minmax<IdxT><<<tiles, NT, 0, stream>>>(dst, E, N, w.tmin, w.tmax, status);
    rc = gnc::check_launch("topo_tile_minmax");
    if (rc) return rc;
    topo_tile_scan<<<1, 1024, 0, stream>>>(w.tmin, w.tmax, tiles, w.P, w.S, w.bar);
    rc = gnc::check_launch("topo_tile_scan");
    if (rc) return rc;
  }
  const char* pl = getenv("GNC_TOPO_PHASE_LIMIT");  // developer probe: stop the sort kernel after phase n (timing only)
  topo_tile_sort<IdxT, WITH_ENDPOINTS><<<tiles, NT, 0, stream>>>(dst, src, E, N, w.P, w.S, tiles, rowptr, perm, src_sorted, dst_sorted,
                                                               status, pl ? atoi(pl) : 0, single_bar);
  rc = gnc::check_launch("topo_tile_sort");
  if (rc || !gated_fallback) return rc;
  // general path, gated on the device by status[2]: one launch
  const FbGeom g = fb_geom(E);
  fb_sort_all<IdxT, WITH_ENDPOINTS><<<g.nblocks, FB_NT, 0, stream>>>(status + 2, dst, src, E, N, g, (int)key_bits(N), w.keys_a, w.keys_b,
                                                                  w.vals_a, w.vals_b, w.tab, w.part, w.bar, rowptr, perm, src_sorted,
                                                                  dst_sorted, status);
  return gnc::check_launch("gnc_topology_build(general path)");
}

}  // namespace

extern "C" size_t gnc_topology_workspace_bytes(int64_t num_nodes, int64_t num_edges, int32_t gated_fallback) {
  if (num_nodes < 0 || num_edges < 0 || num_nodes >= INT32_MAX || num_edges >= INT32_MAX - CAP) {
    gnc::set_error("gnc_topology_workspace_bytes: sizes out of int32 range (N=%lld, E=%lld)", (long long)num_nodes, (long long)num_edges);
    return 0;
  }
  return carve(nullptr, num_edges > 0 ? num_edges : 1, gated_fallback != 0).bytes;
}

extern "C" int gnc_topology_build(const void* src, const void* dst, int32_t index_bytes, int64_t num_edges, int64_t num_nodes,
                                  int32_t* rowptr, int32_t* perm, int32_t* src_sorted, int32_t* dst_sorted, int32_t* status,
                                  void* workspace, size_t workspace_bytes, int32_t gated_fallback, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GNC_REQUIRE(num_nodes >= 0 && num_edges >= 0 && num_nodes < INT32_MAX && num_edges < INT32_MAX - CAP,
              "gnc_topology_build: sizes out of int32 range (N=%lld, E=%lld)", (long long)num_nodes, (long long)num_edges);
  GNC_REQUIRE(index_bytes == 8 || index_bytes == 4, "gnc_topology_build: index_bytes must be 8 (int64) or 4 (int32)");
  GNC_REQUIRE(rowptr && status, "gnc_topology_build: rowptr/status must not be null");
  GNC_REQUIRE(num_edges == 0 || (dst && perm && workspace), "gnc_topology_build: null dst/perm/workspace");
  const bool ends = src != nullptr;
  GNC_REQUIRE(!ends || num_edges == 0 || (src_sorted && dst_sorted), "gnc_topology_build: src given without src_sorted/dst_sorted");
  if (index_bytes == 8) {
    const int64_t *s = (const int64_t*)src, *d = (const int64_t*)dst;
    return ends ? build<int64_t, true>(s, d, num_edges, num_nodes, rowptr, perm, src_sorted, dst_sorted, status, workspace, workspace_bytes, gated_fallback, stream)
                : build<int64_t, false>(s, d, num_edges, num_nodes, rowptr, perm, nullptr, nullptr, status, workspace, workspace_bytes, gated_fallback, stream);
  }
  const int32_t *s = (const int32_t*)src, *d = (const int32_t*)dst;
  return ends ? build<int32_t, true>(s, d, num_edges, num_nodes, rowptr, perm, src_sorted, dst_sorted, status, workspace, workspace_bytes, gated_fallback, stream)
              : build<int32_t, false>(s, d, num_edges, num_nodes, rowptr, perm, nullptr, nullptr, status, workspace, workspace_bytes, gated_fallback, stream);
}
