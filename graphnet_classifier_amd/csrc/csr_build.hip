// Topology build: int64 edge destinations -> destination-sorted CSR (rowptr, perm).
//
// One-off per topology (the edge_index of a batch is constant across the L GN blocks and,
// for pixel/patch graphs, across the whole dataset: reference
// utils/image_to_graph/image_to_graph_optimized.py:42-47 caches it as well).
//
// The sort must be STABLE so that the summation order inside every destination segment is
// the original edge order, which is the order the reference's sequential index_add_
// (models/GNN.py:20) adds in.  A stable LSD radix sort over only the significant key bits is
// used (rocPRIM device radix sort); the key narrowing/validation and the row-pointer
// extraction are hand-written.  No atomics anywhere, so the result is deterministic.
#include <rocprim/device/device_radix_sort.hpp>

#include "gnc_common.h"

namespace {

constexpr size_t kAlign = 256;
inline size_t align_up(size_t x) { return (x + kAlign - 1) / kAlign * kAlign; }

// keys[e] = destination id narrowed to u32, or the sentinel `num_nodes` for an out-of-range
// id (sorts behind every valid key and is excluded by rowptr[num_nodes]).
__global__ void csr_prepare_keys(const int64_t* __restrict__ index, int64_t num_edges, int64_t num_nodes,
                                 uint32_t* __restrict__ keys, int32_t* __restrict__ ids,
                                 int32_t* __restrict__ status) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  bool bad = false;
  for (; e < num_edges; e += stride) {
    const int64_t v = index[e];
    const bool ok = (v >= 0) & (v < num_nodes);
    bad |= !ok;
    keys[e] = ok ? (uint32_t)v : (uint32_t)num_nodes;
    ids[e] = (int32_t)e;
  }
  if (bad) *status = 1;  // benign race: every writer stores the same value
}

// rowptr[v] = first sorted position whose key is >= v, for v in [0, num_nodes].
// Thread i looks at the boundary between sorted positions i-1 and i and fills the row
// pointers of every node id in (key[i-1], key[i]].
__global__ void csr_rowptr_from_sorted(const uint32_t* __restrict__ sorted_keys, int64_t num_edges,
                                       int64_t num_nodes, int32_t* __restrict__ rowptr) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i <= num_edges; i += stride) {
    const int64_t prev = (i == 0) ? -1 : (int64_t)sorted_keys[i - 1];
    const int64_t cur = (i == num_edges) ? num_nodes : (int64_t)sorted_keys[i];
    for (int64_t v = prev + 1; v <= cur; ++v) rowptr[v] = (int32_t)i;
  }
}

__global__ void permute_index_i64_i32(const int64_t* __restrict__ src, const int32_t* __restrict__ perm,
                                      int64_t n, int32_t* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = (int32_t)src[perm ? perm[i] : i];
}

__global__ void permute_index_checked_i64_i32(const int64_t* __restrict__ src, const int32_t* __restrict__ perm, int64_t n,
                                              int64_t num_nodes, int32_t* __restrict__ out, int32_t* __restrict__ status) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  bool bad = false;
  for (; i < n; i += stride) {
    const int64_t v = src[perm ? perm[i] : i];
    const bool ok = v >= 0 && v < num_nodes;
    bad |= !ok;
    out[i] = ok ? (int32_t)v : 0;
  }
  if (bad) *status = 1;  // every writer stores the same value
}

unsigned key_bits(int64_t num_nodes) {  // bits needed to represent the sentinel value num_nodes
  unsigned b = 1;
  while (b < 32 && ((uint64_t)num_nodes >> b) != 0) ++b;
  return b;
}

int sort_temp_bytes(int64_t num_edges, unsigned bits, size_t* bytes) {
  size_t tmp = 0;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                           (const int32_t*)nullptr, (int32_t*)nullptr, (size_t)num_edges, 0u, bits);
  if (e != hipSuccess) return gnc::check_hip(e, "rocprim::radix_sort_pairs(size query)");
  *bytes = tmp;
  return GNC_OK;
}

int grid_for(int64_t n) {
  int64_t g = gnc::ceil_div(n > 0 ? n : 1, gnc::kBlock);
  const int64_t cap = gnc::num_cu() * 8;
  return (int)(g < cap ? g : cap);
}

}  // namespace

extern "C" size_t gnc_csr_workspace_bytes(int64_t num_nodes, int64_t num_edges) {
  if (num_nodes < 0 || num_edges < 0 || num_nodes >= INT32_MAX || num_edges >= INT32_MAX) {
    gnc::set_error("gnc_csr_workspace_bytes: sizes out of int32 range (N=%lld, E=%lld)", (long long)num_nodes,
                   (long long)num_edges);
    return 0;
  }
  size_t sort_bytes = 0;
  if (num_edges > 0 && sort_temp_bytes(num_edges, key_bits(num_nodes), &sort_bytes) != GNC_OK) return 0;
  const size_t e = (size_t)num_edges;
  // keys_in | keys_out | ids_in | rocPRIM temporaries
  return 3 * align_up(e * 4) + align_up(sort_bytes) + kAlign;
}

extern "C" int gnc_csr_build(const int64_t* index, int64_t num_edges, int64_t num_nodes, int32_t* rowptr,
                             int32_t* perm, int32_t* status, void* workspace, size_t workspace_bytes,
                             void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GNC_REQUIRE(num_nodes >= 0 && num_edges >= 0 && num_nodes < INT32_MAX && num_edges < INT32_MAX,
              "gnc_csr_build: sizes out of int32 range (N=%lld, E=%lld)", (long long)num_nodes, (long long)num_edges);
  GNC_REQUIRE(rowptr && status, "gnc_csr_build: rowptr/status must not be null");
  GNC_REQUIRE(num_edges == 0 || (index && perm && workspace), "gnc_csr_build: null index/perm/workspace");

  int rc = gnc::check_hip(hipMemsetAsync(status, 0, sizeof(int32_t), stream), "hipMemsetAsync(status)");
  if (rc) return rc;
  if (num_edges == 0) {
    rc = gnc::check_hip(hipMemsetAsync(rowptr, 0, (size_t)(num_nodes + 1) * sizeof(int32_t), stream),
                        "hipMemsetAsync(rowptr)");
    return rc;
  }

  const unsigned bits = key_bits(num_nodes);
  size_t sort_bytes = 0;
  rc = sort_temp_bytes(num_edges, bits, &sort_bytes);
  if (rc) return rc;
  const size_t e4 = align_up((size_t)num_edges * 4);
  const size_t need = 3 * e4 + align_up(sort_bytes);
  uintptr_t base = (reinterpret_cast<uintptr_t>(workspace) + kAlign - 1) / kAlign * kAlign;
  const size_t slack = base - reinterpret_cast<uintptr_t>(workspace);
  if (workspace_bytes < need + slack) {
    gnc::set_error("gnc_csr_build: workspace too small (%zu < %zu)", workspace_bytes, need + slack);
    return GNC_ERR_WORKSPACE;
  }
  uint32_t* keys_in = reinterpret_cast<uint32_t*>(base);
  uint32_t* keys_out = reinterpret_cast<uint32_t*>(base + e4);
  int32_t* ids_in = reinterpret_cast<int32_t*>(base + 2 * e4);
  void* sort_tmp = reinterpret_cast<void*>(base + 3 * e4);

  csr_prepare_keys<<<grid_for(num_edges), gnc::kBlock, 0, stream>>>(index, num_edges, num_nodes, keys_in, ids_in,
                                                                     status);
  rc = gnc::check_launch("csr_prepare_keys");
  if (rc) return rc;
  rc = gnc::check_hip(rocprim::radix_sort_pairs(sort_tmp, sort_bytes, (const uint32_t*)keys_in, keys_out,
                                                (const int32_t*)ids_in, perm, (size_t)num_edges, 0u, bits, stream),
                      "rocprim::radix_sort_pairs");
  if (rc) return rc;
  csr_rowptr_from_sorted<<<grid_for(num_edges + 1), gnc::kBlock, 0, stream>>>(keys_out, num_edges, num_nodes, rowptr);
  return gnc::check_launch("csr_rowptr_from_sorted");
}

extern "C" int gnc_permute_index_checked_i64_i32(const int64_t* src, const int32_t* perm, int64_t n, int64_t num_nodes,
                                                 int32_t* out, int32_t* status, void* stream_) {
  GNC_REQUIRE(n >= 0 && num_nodes >= 0, "gnc_permute_index_checked_i64_i32: negative size");
  if (n == 0) return GNC_OK;
  GNC_REQUIRE(src && out && status, "gnc_permute_index_checked_i64_i32: null pointer");
  permute_index_checked_i64_i32<<<grid_for(n), gnc::kBlock, 0, (hipStream_t)stream_>>>(src, perm, n, num_nodes, out, status);
  return gnc::check_launch("permute_index_checked_i64_i32");
}

extern "C" int gnc_permute_index_i64_i32(const int64_t* src, const int32_t* perm, int64_t n, int32_t* out,
                                         void* stream_) {
  GNC_REQUIRE(n >= 0, "gnc_permute_index_i64_i32: negative size");
  if (n == 0) return GNC_OK;
  GNC_REQUIRE(src && out, "gnc_permute_index_i64_i32: null pointer");
  permute_index_i64_i32<<<grid_for(n), gnc::kBlock, 0, (hipStream_t)stream_>>>(src, perm, n, out);
  return gnc::check_launch("permute_index_i64_i32");
}
