/*
 * gnc_hip.h -- C ABI of the MI355X (gfx950) message-passing engine that stands behind the
 * GraphNet forward hot path of alexisvannson/GraphNet_Classifier.
 *
 * This is the drop-in boundary.  The reference has no FFI of its own:
 * its narrowest seam is the module-level Python name `scatter_sum` bound at import
 * (reference models/GNN.py:4-21, where an optional C++ extension or an `index_add_` fallback
 * is chosen) plus the `torch.nn.Module` forward methods of `MLP`, `EdgeProcessor`,
 * `NodeProcessor`, `GraphProcessor` and `GraphNet`.  Each entry point below names the
 * reference line(s) whose work it replaces.  A maintainer of the reference binds these with
 * `ctypes` (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless marked host;
 *   - `stream` is a `hipStream_t` passed as `void*`; nothing here synchronises the stream,
 *     allocates, or frees: outputs and workspaces are caller-allocated (the Python side uses
 *     torch's caching allocator) so calls can be captured into a hipGraph;
 *   - return value: 0 = GNC_OK, negative = error (see enum); `gnc_last_error_string()` gives
 *     a thread-local description of the last failure on the calling thread;
 *   - matrices are row-major fp32; `ld*` are leading dimensions in ELEMENTS;
 *   - indices are int32 inside the engine; the int64 `edge_index` of the reference
 *     (utils/dataloader.py:51) is narrowed once, by `gnc_csr_build`.
 */
#ifndef GNC_HIP_H
#define GNC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GNC_ABI_VERSION 19

enum {
  GNC_OK = 0,
  GNC_ERR_INVALID_ARGUMENT = -1, /* null pointer, negative size, misaligned buffer ...      */
  GNC_ERR_UNSUPPORTED = -2,      /* shape outside what the kernels are instantiated for      */
  GNC_ERR_WORKSPACE = -3,        /* workspace smaller than gnc_*_workspace_bytes() reports   */
  GNC_ERR_HIP = -4               /* a HIP runtime call or a kernel launch failed             */
};

#define GNC_MAX_SEGMENTS 4 /* concat segments feeding the first Linear of an MLP */
#define GNC_MAX_LINEAR 8   /* Linear layers in one MLP (hidden_layers + 1)       */

enum { GNC_ACT_RELU = 0, GNC_ACT_IDENTITY = 1, GNC_ACT_TANH = 2, GNC_ACT_SIGMOID = 3,
       GNC_ACT_SILU = 4, GNC_ACT_GELU = 5, GNC_ACT_LEAKY_RELU = 6, GNC_ACT_ELU = 7 };

/* ---- diagnostics --------------------------------------------------------------------- */
int gnc_abi_version(void);
const char* gnc_last_error_string(void);
/* Name of the device code object target this library was built for ("gfx950"). */
const char* gnc_target_arch(void);

/* ---- topology: edge list -> destination-sorted CSR ------------------------------------
 * Replaces nothing the reference does explicitly: the reference re-derives "who sends to
 * node v" inside every `index_add_` (models/GNN.py:18-20).  Here the destination column of
 * `edge_index` (models/GNN.py:98 `_, col = edge_index`) is sorted once per topology.
 *
 *   index   [E] int64   destination (or source) node id per edge, any order
 *   rowptr  [N+1] int32 out: edges of node v are sorted positions rowptr[v] .. rowptr[v+1]-1
 *   perm    [E] int32   out: perm[k] = ORIGINAL edge id at sorted position k.  The sort is
 *                       stable, so inside a segment edges keep their original order and the
 *                       per-destination summation order equals the reference's edge order.
 *   status  [1] int32   out: 0, or 1 if any index was outside [0, N) (such edges are dropped
 *                       instead of faulting; the reference raises IndexError there)
 */
size_t gnc_csr_workspace_bytes(int64_t num_nodes, int64_t num_edges);
int gnc_csr_build(const int64_t* index, int64_t num_edges, int64_t num_nodes,
                  int32_t* rowptr, int32_t* perm, int32_t* status,
                  void* workspace, size_t workspace_bytes, void* stream);

/* out[k] = (int32) src[perm[k]]   -- narrows and reorders an int64 edge_index row. */
int gnc_permute_index_i64_i32(const int64_t* src, const int32_t* perm, int64_t n, int32_t* out, void* stream);
/* Same, and validates on the way: a value outside [0, num_nodes) sets *status (device int32, caller zero-fills)
 * to 1 and is stored as 0, so that nothing downstream can index out of range before the host has looked at
 * the flag (models/GNN.py:18-20: `index_add_` raises IndexError for such an id). */
int gnc_permute_index_checked_i64_i32(const int64_t* src, const int32_t* perm, int64_t n, int64_t num_nodes, int32_t* out,
                                      int32_t* status, void* stream);

/* ---- topology of a graph-ordered batch in one call (ABI 17) --------------------------------
 * Everything `GraphNet.forward` needs of `edge_index` (models/GNN.py:98 `_, col = edge_index`, the x[row] / x[col]
 * gathers of MetaLayer at :215) from ONE call: the destination CSR (rowptr, perm: the stable sort of gnc_csr_build) and
 * both endpoint vectors narrowed to int32 and reordered by it (src_sorted[k] = src[perm[k]], dst_sorted[k] = dst[perm[k]];
 * what gnc_permute_index_checked_i64_i32 / gnc_permute_index_i64_i32 give).  Batches of small graphs whose edges are
 * contiguous in edge_index (the reference's loader yields one self-contained graph per item, utils/dataloader.py:33-53;
 * graphs of at most 2048 edges) are sorted range by range in LDS without a global radix sort;
 * any other edge list raises status[2] ON THE DEVICE and is sorted by a general stable LSD radix sort: ONE kernel that
 * is enqueued behind it and returns at once unless the flag is set - so the call never synchronises and can be
 * captured (gated_fallback = 1).  A caller that reads the flags back anyway may pass gated_fallback = 0 (smaller
 * workspace) and run gnc_csr_build + the permutes itself when status[2] is set.
 *
 *   src, dst      [E] int64 (index_bytes = 8) or int32 (index_bytes = 4) node ids; src may be NULL: then only rowptr and
 *                 perm are produced (src_sorted / dst_sorted ignored)
 *   status [3]    out: [0] = a destination outside [0, N), [1] = a source outside [0, N) (such ids are replaced by 0:
 *                 nothing downstream can index out of range; models/GNN.py:18-20 raises IndexError there),
 *                 [2] = 1: the LDS path could not take this edge list (informational when gated_fallback = 1: the general
 *                 path - one gated launch whose passes meet at a grid barrier - has sorted it); 2: that barrier timed out
 *                 (5 s; the outputs are invalid, status[0] is raised as well so that every validity check fails)
 */
size_t gnc_topology_workspace_bytes(int64_t num_nodes, int64_t num_edges, int32_t gated_fallback);
int gnc_topology_build(const void* src, const void* dst, int32_t index_bytes, int64_t num_edges, int64_t num_nodes,
                       int32_t* rowptr, int32_t* perm, int32_t* src_sorted, int32_t* dst_sorted, int32_t* status,
                       void* workspace, size_t workspace_bytes, int32_t gated_fallback, void* stream);

/* ---- K1: scatter-sum neighbourhood aggregation ------------------------------------------
 * Replaces `scatter_sum(edge_attr, col, dim=0)` (models/GNN.py:99 -> :11-21):
 *   out[v, :] = sum over sorted positions k in [rowptr[v], rowptr[v+1]) of src[row(k), :]
 * with row(k) = perm[k], or k itself when `perm` is NULL (messages already stored in
 * destination-sorted order).  No atomics: each destination is summed by one lane group in
 * ascending k, so the result is bitwise reproducible and every out row is written
 * (zero for a node without in-edges), no memset needed.
 */
int gnc_scatter_sum_csr_f32(const float* src, int64_t ld_src, const int32_t* rowptr, const int32_t* perm,
                            int64_t num_nodes, int64_t num_edges, int32_t feat_dim,
                            float* out, int64_t ld_out, void* stream);

/* ---- K2: row gather -----------------------------------------------------------------------
 * Replaces `x[row]`, `x[col]` of PyG MetaLayer.forward (called at models/GNN.py:215) and is the
 * backward of K1:  out[r, :] = table[index[r], :].
 */
int gnc_gather_rows_f32(const float* table, int64_t ld_table, const int32_t* index, int64_t num_rows,
                        int32_t feat_dim, float* out, int64_t ld_out, void* stream);
/* out[r, :] = table[index[r], :] + addend[r, :]: the gradient of an edge latent that feeds both the aggregation
 * (backward of K1 = this gather) and the next block (addend), formed in one pass. */
int gnc_gather_rows_add_f32(const float* table, int64_t ld_table, const int32_t* index, const float* addend,
                            int64_t ld_addend, int64_t num_rows, int32_t feat_dim, float* out, int64_t ld_out, void* stream);

/* ---- K6: edge features ------------------------------------------------------------------
 * Replaces models/GNN.py:299-302:  rel = pos[dst[e]] - pos[src[e]];  out[e] = [rel, sum|rel|].
 * pos is [N, space_dim]; out is [E, ld_out] with ld_out >= space_dim + 1; columns past the
 * features are written as zero (ld_out = 4 for space_dim = 2 gives 16-B rows the MLP kernels can
 * load with one vector instruction).
 */
int gnc_edge_features_f32(const float* pos, int32_t space_dim, const int32_t* src, const int32_t* dst,
                          int64_t num_edges, float* out, int64_t ld_out, void* stream);

/* ABI 19.  Deferred validation of a topology (status flags still on the device when the forward is enqueued): overwrite the
 * `count` contiguous floats of `out` with NaN when any of flags[0 .. nflags) is non-zero; one launch that returns at once
 * otherwise.  Keeps a forward on an edge_index with out-of-range ids (models/GNN.py:18-20 raises IndexError there) from
 * yielding plausible numbers before the host has read the flags.  nflags <= 16. */
int gnc_poison_if_flagged_f32(float* out, int64_t count, const int32_t* flags, int32_t nflags, void* stream);

/* ---- K4: fused MLP ----------------------------------------------------------------------
 * Replaces `MLP.forward` (models/MLP.py:45-47 over the Sequential built at :24-37) together
 * with the `torch.cat` in front of it and the residual add behind it in the processors
 * (models/GNN.py:58-62 and :100-102):
 *
 *   in[r]  = concat over segments s of  seg[s].ptr[ row_s(r), 0:width_s ]      (never materialised)
 *            row_s(r) = seg[s].index[r] if index != NULL else r                 (fused gather, K2+K3)
 *   h      = act(in  W0^T + b0);  h = act(h Wl^T + bl) ...;  y = h W_last^T + b_last
 *   y      = LayerNorm(y) * gamma + beta        if ln_gamma != NULL  (eps = ln_eps)
 *   out[r] = y + residual[r]                    if residual != NULL
 *
 * Weights are nn.Linear layout [out_dim, in_dim] row-major (state-dict tensors are passed
 * as they are).  fp32 in, fp32 MFMA (v_mfma_f32_32x32x2_f32), fp32 out.
 * Limits: hidden widths and out width <= 256, 1 <= num_linear <= GNC_MAX_LINEAR, all
 * hidden layers the same width (true by construction in models/MLP.py:24-27).
 * in_dim[0] is the summed width of the GNC_SEG_MATMUL segments.  When `residual` is the table of
 * the LAST listed segment (no index, same ld, width == out width <= 64) the kernel takes the
 * residual from the rows it has already staged instead of reading them again.  num_linear == 1 is a plain
 * (optionally LayerNorm-ed) projection.
 *
 * W-split.  W0 [x_src | x_dst | e]^T = Ws x[src] + Wd x[dst] + We e: the node-side products can
 * be formed once per NODE (two num_linear == 1 launches over [N, Dn]) and merely gathered and
 * added per EDGE (GNC_SEG_ADD segments), which removes 2/3 of the first Linear's per-edge MFMA
 * work of models/GNN.py:58-61.  ld_weight lets We be passed as a column slice of W0.
 */
enum { GNC_SEG_MATMUL = 0, /* columns of the virtual concat: multiplied by their slice of weight[0]      */
       GNC_SEG_ADD = 1     /* rows already projected to the hidden width: added to the first Linear's
                              pre-activation (width must equal out_dim[0]); see "W-split" below       */ };

typedef struct gnc_mlp_segment {
  const float* ptr;     /* [*, ld] table                                   */
  const int32_t* index; /* NULL, or [rows] row ids into the table          */
  int32_t width;        /* columns taken from the table                    */
  int32_t ld;           /* leading dimension of the table (elements)       */
  int32_t mode;         /* GNC_SEG_MATMUL | GNC_SEG_ADD                    */
  int32_t wcol;         /* MATMUL: first column of weight[0] this segment multiplies (segments may be
                           listed, i.e. staged, in any order; the concat order lives here)          */
  int64_t table_rows;   /* rows of the table behind ptr = bound of the ids in `index`; REQUIRED (> 0) whenever
                           index != NULL.  Every kernel variant treats an id outside [0, table_rows) the same
                           way: the row reads as zeros, never a fault (tables below 4 GiB through the hardware
                           bounds check of a buffer window, larger ones by an explicit compare).
                           Ignored when index == NULL */
} gnc_mlp_segment_t;

typedef struct gnc_mlp_desc {
  int32_t num_segments;
  int32_t num_linear;
  int32_t activation;   /* GNC_ACT_*                                       */
  float act_param;      /* negative slope / alpha for LEAKY_RELU / ELU     */
  gnc_mlp_segment_t seg[GNC_MAX_SEGMENTS];
  const float* weight[GNC_MAX_LINEAR]; /* [out_dim[l], in_dim[l]]          */
  int32_t ld_weight[GNC_MAX_LINEAR];   /* row stride of weight[l]; 0 = in_dim[l] (contiguous) */
  const float* bias[GNC_MAX_LINEAR];   /* [out_dim[l]] or NULL             */
  int32_t in_dim[GNC_MAX_LINEAR];
  int32_t out_dim[GNC_MAX_LINEAR];
  const float* ln_gamma; /* NULL = no LayerNorm                            */
  const float* ln_beta;
  float ln_eps;
  const float* residual; /* NULL, or [rows, ld_residual]                   */
  int32_t ld_residual;
  float* out;            /* [rows, ld_out]                                 */
  int32_t ld_out;
  int64_t rows;
  /* Fused aggregation epilogue ("CSR-ordered segmented-sum epilogue"); all NULL / 0 = off.
   * With agg_out set, the launch ALSO forms agg_out[v, :] = sum of the output rows r with agg_index[r] == v,
   * added in ascending r starting from 0.0 - bit for bit what gnc_scatter_sum_csr_f32 gives on `out` -
   * for every destination whose rows lie inside ONE wave's contiguous row range.  Contract:
   *   - agg_index [rows] int32 is non-decreasing (rows are in destination-CSR order);
   *   - destinations without rows are not written by the kernel: gnc_agg_fixup_f32 zero-fills them;
   *   - destinations cut by a range boundary are not written either: the kernel lists them (and -1 for
   *     idle waves) in agg_fix [gnc_mlp_agg_fix_len()] int32, and gnc_agg_fixup_f32 recomputes exactly
   *     those rows from `out` through the row pointers afterwards (same stream).
   * Served by the weights-resident kernel (widths <= 64), the 32-row streaming kernel (65..128) and the 16-row
   * streaming kernel (129..256, num_linear > 1): ask gnc_mlp_agg_supported(); gnc_mlp_forward_f32 returns
   * GNC_ERR_UNSUPPORTED otherwise.  Not offered for small batches at 65..128 features (gnc_mlp_small_batch_supported):
   * there gnc_scatter_sum_csr_f32 on the output rows is the cheaper route. */
  float* agg_out;           /* [*, ld_agg] or NULL */
  int32_t ld_agg;
  const int32_t* agg_index; /* [rows] destination of each output row */
  int32_t* agg_fix;         /* [gnc_mlp_agg_fix_len()] destinations left to gnc_agg_fixup_f32 */
  /* ABI 16, training forward: save_act[l] (l < num_linear - 1) = NULL, or [rows, out_dim[l]] contiguous: the launch
   * also writes the post-activation output of Linear l (what torch.autograd keeps for the backward of
   * models/MLP.py:45-47), so that gnc_mlp_backward_f32 can read it (act_given) instead of recomputing the forward of
   * every tile.  All of them or none; only where gnc_mlp_save_act_supported() says so. */
  float* save_act[GNC_MAX_LINEAR];
  /* ABI 19, K6 as the prologue of the edge encoder (models/GNN.py:299-302 feeding :306).  ef_pos = NULL: off.
   * With ef_pos set, the rows of segment 0 - the only segment: MATMUL, index NULL, width = ef_space_dim + 1, its ptr / ld
   * ignored - are COMPUTED by the launch instead of read:
   *   row r = [ pos[ef_dst[r]] - pos[ef_src[r]] , sum_d |pos[ef_dst[r]][d] - pos[ef_src[r]][d]| ]
   * (the differences summed in dimension order starting from the first: bit for bit what gnc_edge_features_f32 stores), so
   * the [rows, space_dim + 1] table never exists in HBM.  pos [ef_nodes, ef_space_dim] contiguous; an id outside
   * [0, ef_nodes) reads as a zero position, never a fault.  Inference only (no save_act, no agg_out); served where
   * gnc_mlp_edge_features_supported() says so (the weights-resident kernel, ef_space_dim = 2, hidden widths <= 64, a batch
   * above the small-batch limit); gnc_mlp_forward_f32 returns GNC_ERR_UNSUPPORTED otherwise and the caller runs
   * gnc_edge_features_f32 + a plain segment. */
  const float* ef_pos;
  const int32_t* ef_src;
  const int32_t* ef_dst;
  int64_t ef_nodes;
  int32_t ef_space_dim;
} gnc_mlp_desc_t;

/* 0 if gnc_mlp_forward_f32 can run this description, GNC_ERR_UNSUPPORTED otherwise
 * (pointers are not inspected, only the shape fields). */
/* sizeof(gnc_mlp_desc_t) as compiled into the library, for binding cross-checks. */
size_t gnc_sizeof_mlp_desc(void);
int gnc_mlp_supported(const gnc_mlp_desc_t* desc /* host */);
int gnc_mlp_forward_f32(const gnc_mlp_desc_t* desc /* host */, void* stream);
/* fused aggregation epilogue: 0 if this description can run with agg_out set (shape fields only) */
int gnc_mlp_agg_supported(const gnc_mlp_desc_t* desc /* host */);
int gnc_mlp_agg_fix_len(void); /* entries of agg_fix (2 per wave of the persistent grid) */
/* ABI 19: 0 if the kernel that gnc_mlp_forward_f32 picks for this description reads its operands where they lie: weight
 * matrices of any row stride / alignment (the [H, 3] matrix of nn.Linear(3, H), models/GNN.py:251-256) and a contiguous
 * [rows, 3] table as the only segment (the node features of models/GNN.py:305) - no zero-padded copies needed.  Every other
 * kernel reads rows of 16-B pieces (tables and weights 16-B aligned, leading dimensions % 4 == 0). */
int gnc_mlp_operands_in_place_supported(const gnc_mlp_desc_t* desc /* host */);
/* ABI 19: 0 if gnc_mlp_forward_f32 serves this description with ef_pos set (pointer fields: alignment tests only) */
int gnc_mlp_edge_features_supported(const gnc_mlp_desc_t* desc /* host */);
/* ABI 18.  0 if the small-batch kernel (one 16-row tile per workgroup, the waves split the output features; rows up to
 * 128 x the number of CUs, widths 65..128, ReLU) serves this description EXACTLY AS GIVEN: it is the only kernel that reads
 * tables and weights whose rows are not 16-B pieces (the reference's [N, 3] inputs and nn.Linear(3, H) weights,
 * models/GNN.py:251-253) in place; for every other launch the caller hands over zero-padded copies.  The reference's own
 * regime (one graph per call: main.py:60, utils/train_model.py:35-45, utils/inference.py:59) runs on it. */
int gnc_mlp_small_batch_supported(const gnc_mlp_desc_t* desc /* host */);
/* ABI 18: out = a W[:, 0:dn] + b W[:, dn:2 dn] for W [hidden, ld_w] as nn.Linear holds it (read transposed, no copy): the
 * node-side gradient of the W-split first Linear (dx = d(ps) Ws + d(pd) Wd) in one launch for a small batch (hidden = dn = 128,
 * 16-B aligned rows); GNC_ERR_UNSUPPORTED otherwise, nothing launched. */
int gnc_mlp_projection_t2_f32(const float* a, int64_t ld_a, const float* b, int64_t ld_b, int64_t rows, const float* w, int64_t ld_w,
                              int32_t hidden, int32_t dn, float* out, int64_t ld_out, void* stream);
int64_t gnc_mlp_small_batch_max_rows(void); /* the row limit of the small-batch forward kernel (128 x CUs unless overridden) */
/* ABI 18: out_a = x wa^T and out_b = x wb^T (no bias) over the same row-ordered x [rows, in_dim] in ONE launch: the two
 * node-side products of the W-split first Linear (models/GNN.py:58-61 formed once per node).  Two shapes: the small-batch
 * projection (in_dim = out_dim = 128, rows within the small-batch limit, 16-B aligned rows, ld_wa == ld_wb) and - since ABI 19 -
 * a batch above that limit at in_dim <= 64, out_dim 36..64 (a multiple of 4) on the weights-resident kernel (rows read and
 * staged once, both matrices resident in LDS, read where they lie whatever their row stride).  GNC_ERR_UNSUPPORTED otherwise,
 * nothing launched - issue two gnc_mlp_forward_f32 launches then. */
int gnc_mlp_dual_projection_f32(const float* x, int64_t ld_x, int64_t rows, const float* wa, int64_t ld_wa, const float* wb,
                                int64_t ld_wb, int32_t in_dim, int32_t out_dim, float* out_a, float* out_b, int64_t ld_out,
                                void* stream);
/* 0 if gnc_mlp_forward_f32 can run this description with save_act set (every kernel but the generic fallback: ReLU, hidden
 * widths that are multiples of 4 and at most 256, aligned tables), GNC_ERR_UNSUPPORTED otherwise (shape fields only) */
int gnc_mlp_save_act_supported(const gnc_mlp_desc_t* desc /* host */);
/* out[v, :] = sum over k in [rowptr[v], rowptr[v+1]) of src[k, :] (ascending k) for the n_fix destinations
 * v = fix[j] (entries < 0 or >= num_nodes are skipped; duplicates are harmless), and out[v, :] = 0 for every
 * destination v < num_nodes without rows: together with the epilogue every row of `out` is then defined. */
int gnc_agg_fixup_f32(const float* src, int64_t ld_src, const int32_t* rowptr, const int32_t* fix, int32_t n_fix,
                      int64_t num_nodes, int32_t feat_dim, float* out, int64_t ld_out, void* stream);

/* ---- K8: backward of the fused MLP ---------------------------------------------------------
 * Replaces what PyTorch's autograd does for `loss.backward()` (utils/train_model.py:41) through
 * models/MLP.py:45-47 and the concat/residual around it.  Two kernels:
 *
 * gnc_mlp_backward_f32: given the forward description `fwd` (its `out` field is ignored) and
 *   grad_out = dL/d(out) [rows, out_dim], recomputes the forward per 32-row tile - or, with act_given (ABI 16), reads the
 *   post-activations the training forward saved (gnc_mlp_desc_t.save_act) and recomputes only the last Linear for the
 *   LayerNorm statistics - and writes
 *     act[l]  [rows, out_dim[l]]  post-activation output of Linear l, l < num_linear-1 (input of Linear l+1); an INPUT
 *                                 with act_given
 *     dz[l]   [rows, out_dim[l]]  dL/d(pre-activation of Linear l); dz[num_linear-1] is dL/d(pre-LayerNorm y)
 *     dx      [rows, in_dim[0]]   dL/d(MATMUL part of the input concat), column order = wcol; NULL to skip
 *     yhat    [rows, out_dim]     normalised pre-affine output of the LayerNorm (required with LayerNorm):
 *                                 d beta = colsum(grad_out), d gamma = colsum(grad_out * yhat) -> gnc_colsum_pair_f32
 *   The gradient of an ADD segment's table rows is dz[0] itself (scatter it by the segment's index);
 *   the gradient of the residual is grad_out.  Weight gradients: dW_l = dz[l]^T (input of Linear l),
 *   db_l = column sums of dz[l] -> gnc_xty_f32.
 *   Supported: ReLU, 2 <= num_linear <= 7, tables 16-B aligned with ld % 4 == 0; widths <= 64 with every
 *   segment <= 64 columns and MATMUL segments listed before 0 or 2 ADD segments run with LDS-resident
 *   weights, other shapes up to width 128 stream the weights on 32-row tiles, widths 129..256 on 16-row tiles
 *   (v_mfma_f32_16x16x4_f32; both streaming forms need 16-B aligned weights, the 16-row one stated gather tables);
 *   gnc_mlp_backward_supported() answers from the shape fields alone.
 *
 * gnc_xty_f32: partial[w] = [ A^T B (M x K, row-major) | column sums of A (M) ] over the rows worker w
 *   streamed; sum the gnc_xty_partials_for(rows, M, K) rows it writes (fixed order => reproducible).  Both operands
 *   above 64 columns: M, K <= 256, one partial per WORKGROUP (the row tile is shared by the workgroup's waves, each
 *   row of A and B is read once per launch); otherwise M, K <= 128 and one partial per wave
 *   (= gnc_xty_partials(rows), which is also an upper bound of every count).
 * gnc_colsum_pair_f32: partial[w] = [ colsum(G) (width) | colsum(G * Y) (width) ], same partial count.
 */
typedef struct gnc_mlp_bwd_desc {
  gnc_mlp_desc_t fwd;
  const float* grad_out;
  int32_t ld_grad_out;
  float* act[GNC_MAX_LINEAR];
  float* dz[GNC_MAX_LINEAR];
  float* dx;
  int32_t ld_dx;
  float* yhat;
  int32_t dx_add_grad_out; /* 1: the LAST MATMUL segment is also the residual and out_dim equals its width: its dx
                              columns get grad_out added in the kernel (honoured by the weights-resident variant;
                              gnc_mlp_backward_dx_add_honoured() tells) */
  float* ln_partial;       /* NULL, or [gnc_mlp_backward_ln_partial_rows(fwd), 2 * out_dim]: row w receives
                              [ colsum(grad_out) | colsum(grad_out * yhat) ] over the rows wave w processed (sum the
                              rows in order: d beta, d gamma).  With it `yhat` is neither needed nor written and
                              gnc_colsum_pair_f32 is not needed: 3 x [rows, out_dim] of HBM traffic less per launch. */
  float* dw_partial[GNC_MAX_LINEAR]; /* all NULL, or for l < num_linear: [gnc_mlp_backward_fused_rows(fwd), M_l*K_l + M_l]
                              with M_l = out_dim[l], K_l = in_dim[l] (l = 0: the width of the MATMUL segment): row w
                              receives [ dW_l row-major | db_l ] over the rows wave w processed - the layout of
                              gnc_xty_f32, formed INSIDE the data kernel.  With it act[], dz[1..], yhat are neither
                              needed nor written (dz[0] still is when given); ln_partial is required with LayerNorm. */
  /* ABI 16: a gathered part of the output gradient.  When grad_gather is given, the gradient of output row r is
   *   (grad_out ? grad_out[r] : 0) + grad_gather[grad_gather_index[r]]
   * (table [grad_gather_rows, ld_grad_gather], first out_dim columns; an id outside the table contributes zeros).
   * This is the backward of the scatter-sum that consumed the rows (models/GNN.py:99: d agg[col[e]] / d e'[e] = I)
   * folded into the K8 launch: no [rows, out_dim] gather pass in front of it.  grad_out may then be NULL (the rows
   * had no other consumer).  Only kernels for which gnc_mlp_backward_grad_gather_honoured() returns 1 accept it.
   * grad_sum: device scratch [rows, ld_grad_sum >= out_dim] (16-B aligned, ld % 4 == 0), required when grad_out,
   * grad_gather and dx_add_grad_out are all given: the kernel parks the summed rows there for the residual path of
   * dx (contents undefined afterwards). */
  const float* grad_gather;
  int32_t ld_grad_gather;
  const int32_t* grad_gather_index;
  int64_t grad_gather_rows;
  float* grad_sum;
  int32_t ld_grad_sum;
  /* ABI 16: 1 = act[l] (l < num_linear - 1) are INPUTS: the post-activations the forward saved (gnc_mlp_desc_t.save_act,
   * contiguous [rows, out_dim[l]]); the kernel reads them instead of recomputing the forward of every tile.  Only where
   * gnc_mlp_backward_saved_act_honoured() returns 1 (the fused data + weight-gradient kernel: dw_partial given). */
  int32_t act_given;
  /* ABI 18: with act_given, fwd.save_act[l] carries the same pointers as act[l]: the shape queries that only take the forward
   * description (gnc_mlp_backward_ln_partial_rows, gnc_mlp_backward_dx_add_honoured, gnc_mlp_backward_supported) then answer
   * for the kernel that will run - small batches (rows <= 32 x CUs, widths 65..128) with saved activations run the column-split
   * (above that, at exactly 128 features, its register-resident variant)
   * data kernel (mlp_bwd_col16.hip): one ln_partial row per 16-row tile, grad_gather and dx_add_grad_out folded in. */
} gnc_mlp_bwd_desc_t;

size_t gnc_sizeof_mlp_bwd_desc(void);
int gnc_mlp_backward_supported(const gnc_mlp_desc_t* fwd /* host */);
int gnc_mlp_backward_dx_add_honoured(const gnc_mlp_desc_t* fwd /* host */); /* 1 / 0 */
/* ABI 18: 1 if the small-batch data kernel serves this backward with the operands exactly as they lie (fwd.save_act = the saved
 * post-activations): no 16-B-row copies of [N, 3] inputs / nn.Linear(3, H) weights are needed in front of it */
int gnc_mlp_backward_small_batch_supported(const gnc_mlp_desc_t* fwd /* host */);
/* 1 if gnc_mlp_backward_f32 accepts desc->grad_gather for this description (grad_gather* and dw_partial filled in as
 * for the call), else 0: the caller then gathers the rows itself (gnc_gather_rows_f32 / gnc_gather_rows_add_f32) */
int gnc_mlp_backward_grad_gather_honoured(const gnc_mlp_bwd_desc_t* desc /* host */);
/* 1 if gnc_mlp_backward_f32 accepts desc->act_given for this description (act[] and dw_partial filled in as for the call) */
int gnc_mlp_backward_saved_act_honoured(const gnc_mlp_bwd_desc_t* desc /* host */);
/* rows of `ln_partial` this description needs, 0 if its backward kernel cannot form the LayerNorm sums in flight */
int gnc_mlp_backward_ln_partial_rows(const gnc_mlp_desc_t* fwd /* host */);
/* rows of `dw_partial[l]` (and then also of `ln_partial`) if this description can run the fused data + weight-
 * gradient kernel (three Linear layers, widths <= 64, one row-ordered MATMUL segment and 0 or 2 gathered ADD
 * segments with stated tables), else 0 */
int gnc_mlp_backward_fused_rows(const gnc_mlp_desc_t* fwd /* host */);
int gnc_mlp_backward_f32(const gnc_mlp_bwd_desc_t* desc /* host */, void* stream);
int gnc_xty_partials(int64_t rows);
int gnc_xty_partials_for(int64_t rows, int32_t M, int32_t K);
int gnc_xty_f32(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t rows, int32_t M, int32_t K,
                float* partial, int32_t num_partials, void* stream);
int gnc_colsum_pair_f32(const float* G, int64_t ldg, const float* Y, int64_t ldy, int64_t rows, int32_t width,
                        float* partial, int32_t num_partials, void* stream);
/* Fixed-order sum of per-wave partial rows written by the kernels above (and by dw_partial / ln_partial):
 * for j < M*K: dW[(j / K) * ld_dw + j % K] = sum_p partial[p * stride + j]; for j < M: db[j] = sum_p partial[p * stride + M*K + j]
 * (db may be NULL).  Lets a weight-gradient block land in its place inside a larger matrix without a PyTorch sum + copies. */
int gnc_reduce_partials_f32(const float* partial, int32_t num_partials, int32_t stride, int32_t M, int32_t K,
                            float* dW, int64_t ld_dw, float* db, void* stream);

/* ---- weight gradients of a small batch (ABI 18) ------------------------------------------------------------
 * The reference trains on one graph per step (utils/train_model.py:35-45): ~2000 rows per product.  gnc_xty_small_f32
 * forms SEVERAL products in ONE launch, without partial buffers or a reduction launch:
 *   kind 0:  dw[m, k] = sum over r of a[r, m] * b[r, k]   (dw [m, ld_dw]; any row stride, so a block of a wider matrix)
 *            db[m]    = sum over r of a[r, m]             (db may be NULL)
 *   kind 1:  dw[c]    = sum over r of a[r, c], c < m      (rows of per-tile partial sums -> one row; b, k, db unused)
 * Sums run in a fixed order (bitwise reproducible).  rows <= gnc_xty_small_max_rows() for every job. */
#define GNC_XTY_MAX_JOBS 8
typedef struct gnc_xty_job {
  const float* a;
  int64_t lda;
  const float* b;
  int64_t ldb;
  int64_t rows;
  int32_t m, k;
  float* dw;
  int64_t ld_dw;
  float* db;
  int32_t kind;
} gnc_xty_job_t;
int gnc_xty_small_max_rows(void);
int gnc_xty_small_f32(const gnc_xty_job_t* jobs /* host */, int32_t num_jobs, void* stream);

/* ---- backward of an MLP with an activation other than ReLU (ABI 17) ---------------------------------------
 * models/MLP.py:21 accepts any nn.<Name>; the reference's entry points only use ReLU, which the fused K8 kernels
 * above serve.  For the other activations of GNC_ACT_* the backward runs LAYER BY LAYER on this library: the
 * pre-activations are recomputed with single-Linear gnc_mlp_forward_f32 launches, these three row-wise / elementwise
 * kernels provide act, act' and the LayerNorm backward, the products da = dz W are single-Linear launches on the
 * transposed weights and the weight gradients come from gnc_xty_f32 / gnc_colsum_pair_f32.
 *   gnc_activation_f32           out = act(z)
 *   gnc_activation_backward_f32  grad_z = grad_act * act'(z)      (the derivative torch.autograd uses for nn.<Name>)
 *   gnc_layer_norm_backward_f32  for out = LayerNorm(y) * gamma + beta: grad_y and the normalised y_hat of every row
 *                                (d gamma = colsum(grad_out * y_hat), d beta = colsum(grad_out))
 */
int gnc_activation_f32(const float* z, int64_t ld_z, int64_t rows, int32_t width, int32_t activation, float act_param,
                       float* out, int64_t ld_out, void* stream);
int gnc_activation_backward_f32(const float* z, int64_t ld_z, const float* grad_act, int64_t ld_grad, int64_t rows,
                                int32_t width, int32_t activation, float act_param, float* grad_z, int64_t ld_out, void* stream);
int gnc_layer_norm_backward_f32(const float* y, int64_t ld_y, const float* gamma, const float* grad_out, int64_t ld_grad,
                                int64_t rows, int32_t width, float eps, float* grad_y, int64_t ld_gy, float* yhat,
                                int64_t ld_yhat, void* stream);

/* ---- read-out classifier of ONE graph (ABI 18) -------------------------------------------------------------------
 * Replaces `LinearClassifier.forward` (models/GNN.py:312-325, called at :340) for the single-graph call of the reference's loops:
 * logits = fc3(relu(fc2(relu(fc1(y))))) over the flattened node outputs y [F]; weights as nn.Linear holds them.  One launch
 * forward (h1 [H1], h2 [H2]: the post-ReLU hidden vectors, kept for the backward; `ticket`: a device uint32 that is 0 before the
 * first launch and is left at 0 by every launch), one launch backward (all six parameter gradients, contiguous, and dy [F];
 * dy may be NULL).  H1, H2 <= 1024, C <= 64.  Fixed summation order (bitwise reproducible). */
int gnc_readout_forward_f32(const float* y, int64_t F, const float* w1, int64_t ld1, const float* b1, int32_t H1, const float* w2,
                            int64_t ld2, const float* b2, int32_t H2, const float* w3, int64_t ld3, const float* b3, int32_t C, float* h1,
                            float* h2, float* logits, uint32_t* ticket, void* stream);
int gnc_readout_backward_f32(const float* grad_logits, const float* y, int64_t F, const float* w1, int64_t ld1, int32_t H1, const float* w2,
                             int64_t ld2, int32_t H2, const float* w3, int64_t ld3, int32_t C, const float* h1, const float* h2, float* dw1,
                             float* db1, float* dw2, float* db2, float* dw3, float* db3, float* dy, void* stream);

/* ---- graph construction on the device ---------------------------
 * Inputs: an already resized uint8 RGB image [H, W, C] in HBM.  Outputs: the tensors
 * utils/dataloader.py:49-51 builds (x, pos float32; edge_index int64 [2, E] row-major), in the
 * reference's node and edge order.
 *
 * gnc_grid_edges_i64   utils/image_to_graph/image_to_graph_optimized.py:7-39 (one-directional 4-neighbour
 *                      grid, optional two diagonal families); E = gnc_grid_num_edges().
 * gnc_pixel_nodes_f32  optimized.py:73-79: x = raw pixel values, pos = (row, col).
 * gnc_patch_nodes_f32  image_to_graph_patch.py:30-47: mean colour and integer centre per patch.
 * gnc_rag_build        image_to_graph_superpixel.py:36-71 after SLIC: `labels` [H, W] int32 in [0, H*W);
 *                      x [<=H*W, 3] mean of img/255 per label, pos = centroid (y, x), labels compacted in
 *                      ascending order (np.unique); edge_index rows have leading dimension ld_edges >=
 *                      4*H*W; counts[0] = segments, counts[1] = directed edges, counts[2] = 1 if a label
 *                      was out of range.
 */
int64_t gnc_grid_num_edges(int32_t H, int32_t W, int32_t diagonals);
int gnc_grid_edges_i64(int32_t H, int32_t W, int32_t diagonals, int64_t* edge_index, void* stream);
int gnc_pixel_nodes_f32(const uint8_t* img, int32_t H, int32_t W, int32_t C, float* x, float* pos, void* stream);
int gnc_patch_nodes_f32(const uint8_t* img, int32_t H, int32_t W, int32_t C, int32_t patch, float* x, float* pos,
                        void* stream);
size_t gnc_rag_workspace_bytes(int32_t H, int32_t W);
int gnc_rag_build(const int32_t* labels, const uint8_t* img, int32_t H, int32_t W, float* x, float* pos,
                  int64_t* edge_index, int64_t ld_edges, int32_t* counts, void* workspace, size_t workspace_bytes,
                  void* stream);

/* ---- fused Adam over flat buffers -------------------------------
 * Replaces `optimizer.step()` of utils/train_model.py:42 for `optim.Adam(model.parameters(), lr=1e-3)` (:9):
 * every parameter is a view of `param` [n], every gradient a view of `grad` [n]; one launch updates all of them
 * with torch.optim.Adam's arithmetic (amsgrad off).  `step` is a DEVICE int64 holding the number of updates made
 * so far (the call advances it first; bias corrections are derived from it on the device in double precision), so
 * the call sequence is the same on every step and can be replayed from a hipGraph.  `scratch2` is 2 floats of
 * device scratch.  All four buffers 16-B aligned.
 */
int gnc_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                      float beta1, float beta2, float eps, float weight_decay, int64_t* step, float* scratch2,
                      void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GNC_HIP_H */
