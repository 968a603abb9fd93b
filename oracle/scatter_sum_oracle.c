/*
 * Scalar C restatement of the reference's scatter-add aggregation and of the stable
 * destination sort.  TEST INFRASTRUCTURE ONLY (checker + scalar CPU baseline); never linked
 * into or called from the product path.  Parity status: pinned by tests/golden/g1_scatter.npz
 * (captured from the reference's own scatter_sum) in tests/test_oracle_golden.py.
 *
 * Follows reference models/GNN.py:18-20:
 *     out = src.new_zeros((dim_size, src.size(1)));  out.index_add_(0, index, src)
 * i.e. rows are added in edge order, one fp32 add per element.
 *
 * Build: make -C oracle   (gcc -O2 -shared -fPIC; -ffp-contract=off keeps plain fp32 adds)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* out[N, D] (zero-initialised here) += src[e, :] at row index[e], e ascending.
 * returns 0, or -1 if an index is out of range (nothing is written for that edge). */
int oracle_scatter_sum_f32(const float* src, const int64_t* index, int64_t num_edges, int64_t feat_dim,
                           int64_t num_nodes, float* out) {
  int bad = 0;
  memset(out, 0, (size_t)(num_nodes * feat_dim) * sizeof(float));
  for (int64_t e = 0; e < num_edges; ++e) {
    const int64_t v = index[e];
    if (v < 0 || v >= num_nodes) { bad = -1; continue; }
    float* o = out + v * feat_dim;
    const float* s = src + e * feat_dim;
    for (int64_t c = 0; c < feat_dim; ++c) o[c] += s[c];
  }
  return bad;
}

/* Stable counting sort of edges by destination: rowptr[N+1], perm[E] (perm[k] = original edge
 * id at sorted position k).  What gnc_csr_build must reproduce exactly. */
int oracle_csr_build(const int64_t* index, int64_t num_edges, int64_t num_nodes, int32_t* rowptr, int32_t* perm) {
  memset(rowptr, 0, (size_t)(num_nodes + 1) * sizeof(int32_t));
  for (int64_t e = 0; e < num_edges; ++e) {
    if (index[e] < 0 || index[e] >= num_nodes) return -1;
    rowptr[index[e] + 1]++;
  }
  for (int64_t v = 0; v < num_nodes; ++v) rowptr[v + 1] += rowptr[v];
  /* second pass places edges in edge order behind a per-destination cursor -> stable */
  {
    int32_t* cur = (int32_t*)malloc((size_t)(num_nodes > 0 ? num_nodes : 1) * sizeof(int32_t));
    if (!cur) return -2;
    memcpy(cur, rowptr, (size_t)num_nodes * sizeof(int32_t));
    for (int64_t e = 0; e < num_edges; ++e) perm[cur[index[e]]++] = (int32_t)e;
    free(cur);
  }
  return 0;
}
