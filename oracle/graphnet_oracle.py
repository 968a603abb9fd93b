"""CPU oracle for the GraphNet forward hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-torch (CPU tensors, explicit arithmetic) restatement of the
reference's forward path.  It exists to CHECK the HIP path; it is never the
thing shipped or measured as the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  Nothing under ``graphnet_classifier_amd/`` imports it.

Parity status: PINNED.  Every function below is checked in
``tests/test_oracle_golden.py`` against golden vectors captured by running the
reference's own classes (``/root/reference/models/GNN.py``, ``models/MLP.py``)
in the build container; the capture script is ``tests/golden/make_golden.py``
and the vectors are the ``tests/golden/*.npz`` files.  The reference ships no
tests of its own, so those captured vectors are the pin.

Each function cites the reference lines (relative to /root/reference) it
restates.  The model weights are passed as a flat ``dict[str, Tensor]`` keyed
exactly like the reference's ``state_dict``.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import Tensor

LN_EPS = 1e-5  # torch.nn.LayerNorm default, used by models/MLP.py:35


# --------------------------------------------------------------------------
# a1: scatter_sum  (models/GNN.py:11-21, the index_add_ fallback)
# --------------------------------------------------------------------------
def scatter_sum(src: Tensor, index: Tensor, dim: int = 0, dim_size: int | None = None) -> Tensor:
    """out[index[e], :] += src[e, :] with out zero-initialised.

    models/GNN.py:12-13  dim != 0 -> NotImplementedError
    models/GNN.py:14-15  1-D src is treated as [E, 1]
    models/GNN.py:16-17  dim_size defaults to max(index)+1 (0 when empty)
    models/GNN.py:18-20  zero-init + index_add_

    The accumulation is written as an explicit edge-ordered loop over
    destinations (numpy ``add.at`` processes repeated indices sequentially in
    edge order) so it does not lean on the ATen kernel it is checking.
    """
    if dim != 0:
        raise NotImplementedError("fallback scatter_sum currently supports dim=0 only")
    if src.ndim == 1:
        src = src.unsqueeze(-1)
    if dim_size is None:
        dim_size = int(index.max().item()) + 1 if index.numel() > 0 else 0
    out = np.zeros((dim_size, src.size(1)), dtype=src.detach().numpy().dtype)
    if index.numel() > 0:
        np.add.at(out, index.detach().numpy().astype(np.int64), src.detach().numpy())
    return torch.from_numpy(out)


def scatter_sum_fast(src: Tensor, index: Tensor, dim_size: int) -> Tensor:
    """Same result as :func:`scatter_sum` for large inputs (``np.add.at`` is slow).

    Sorts edges stably by destination and sums each segment in edge order with a
    float32 running sum, i.e. the same summation order as the sequential loop.
    Used by the parity tests at sizes where the python-level loop would take
    minutes, and by bench.py's cpu_baseline as the scalar 'port'.
    """
    src_np = src.detach().numpy()
    idx = index.detach().numpy().astype(np.int64)
    order = np.argsort(idx, kind="stable")
    sorted_idx = idx[order]
    counts = np.bincount(sorted_idx, minlength=dim_size)
    out = np.zeros((dim_size, src_np.shape[1]), dtype=src_np.dtype)
    maxdeg = int(counts.max()) if counts.size else 0
    starts = np.concatenate([[0], np.cumsum(counts)[:-1]])
    # vectorised over destinations, sequential over the position inside a segment:
    for k in range(maxdeg):
        live = counts > k
        rows = order[starts[live] + k]
        out[live] += src_np[rows]
    return torch.from_numpy(out)


# --------------------------------------------------------------------------
# a5: MLP  (models/MLP.py:24-37 construction, :45-47 forward)
# --------------------------------------------------------------------------
def linear(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    """nn.Linear: y = x @ W^T + b, W is [out, in] (models/MLP.py:24-27)."""
    return x @ w.t() + b


def layer_norm(x: Tensor, gamma: Tensor, beta: Tensor, eps: float = LN_EPS) -> Tensor:
    """nn.LayerNorm over the last dim, biased variance (models/MLP.py:29-35)."""
    mean = x.mean(dim=-1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps) * gamma + beta


def mlp_forward(sd: dict, prefix: str, x: Tensor) -> Tensor:
    """MLP.forward (models/MLP.py:45-47): flatten to 2-D, cast to float32, run
    Linear/ReLU ... Linear [LayerNorm].  The Sequential indices are
    0,2,4,.. for the Linear layers and the last index for the LayerNorm
    (models/MLP.py:24-37); they are discovered from the state-dict keys.
    """
    x = x.reshape(x.size(0), -1)
    dtype = sd[f"{prefix}.model.0.weight"].dtype
    x = x.to(dtype)  # reference: x.float(); fp64 weight dicts are used for cross-checks
    idx = sorted({int(k[len(prefix) + 7:].split(".")[0]) for k in sd if k.startswith(prefix + ".model.")})
    lin = [i for i in idx if sd[f"{prefix}.model.{i}.weight"].ndim == 2]
    norm = [i for i in idx if sd[f"{prefix}.model.{i}.weight"].ndim == 1]
    for n, i in enumerate(lin):
        x = linear(x, sd[f"{prefix}.model.{i}.weight"], sd[f"{prefix}.model.{i}.bias"])
        if n + 1 < len(lin):
            x = x.clamp_min(0)  # nn.ReLU
    for i in norm:
        x = layer_norm(x, sd[f"{prefix}.model.{i}.weight"], sd[f"{prefix}.model.{i}.bias"])
    return x


# --------------------------------------------------------------------------
# a3 / a2 / a4: EdgeProcessor, NodeProcessor, MetaLayer glue
# --------------------------------------------------------------------------
def edge_processor(sd: dict, prefix: str, src: Tensor, dest: Tensor, edge_attr: Tensor) -> Tensor:
    """models/GNN.py:57-64: MLP(cat[src, dest, e]) + e."""
    out = torch.cat([src, dest, edge_attr], -1)
    out = mlp_forward(sd, prefix + ".edge_processor", out)
    return out + edge_attr


def scatter_sum_index_add(src: Tensor, index: Tensor, dim_size: int) -> Tensor:
    """Literally the reference's fallback body (models/GNN.py:18-20): zero-init + ATen
    ``index_add_``.  Used when the oracle is TIMED as the CPU baseline (bench.py), so that the
    baseline runs the same multi-threaded ATen kernel the reference runs; checked equal to the
    edge-ordered loop in tests/test_oracle_golden.py."""
    out = src.new_zeros((dim_size, src.size(1)))
    out.index_add_(0, index.long(), src)
    return out


SCATTER_IMPL = {"sorted_loop": scatter_sum_fast, "index_add": scatter_sum_index_add}
_scatter = scatter_sum_fast


def set_scatter_impl(name: str) -> None:
    global _scatter
    _scatter = SCATTER_IMPL[name]


def node_processor(sd: dict, prefix: str, x: Tensor, edge_index: Tensor, edge_attr: Tensor) -> Tensor:
    """models/GNN.py:95-104: MLP(cat[x, scatter_sum(e, col)]) + x."""
    col = edge_index[1]
    agg = _scatter(edge_attr, col, x.size(0))
    out = torch.cat([x, agg], dim=-1)
    out = mlp_forward(sd, prefix + ".node_processor", out)
    return out + x


def gn_block(sd: dict, prefix: str, x: Tensor, edge_index: Tensor, edge_attr: Tensor):
    """PyG MetaLayer contract as used at models/GNN.py:146-165,:215: gather
    x[row], x[col]; edge model; node model (no global model, u = batch = None)."""
    row, col = edge_index[0], edge_index[1]
    edge_attr = edge_processor(sd, prefix + ".edge_model", x[row], x[col], edge_attr)
    x = node_processor(sd, prefix + ".node_model", x, edge_index, edge_attr)
    return x, edge_attr


# --------------------------------------------------------------------------
# a7: GraphNet.forward (models/GNN.py:297-309), a6: GraphProcessor (:213-216)
# --------------------------------------------------------------------------
def edge_features(pos: Tensor, edge_index: Tensor) -> Tensor:
    """models/GNN.py:299-302: [pos[col]-pos[row], sum|pos[col]-pos[row]|]."""
    rel = pos[edge_index[1]] - pos[edge_index[0]]
    dist = rel.abs().sum(dim=1)
    return torch.cat([rel, dist.unsqueeze(1)], dim=1)


def n_blocks_of(sd: dict, prefix: str = "") -> int:
    pre = prefix + "graph_processor.blocks."
    ids = {int(k[len(pre):].split(".")[0]) for k in sd if k.startswith(pre)}
    return max(ids) + 1 if ids else 0


def graphnet_forward(sd: dict, x: Tensor, pos: Tensor, edge_index: Tensor, prefix: str = "",
                     return_latents: bool = False):
    """GraphNet.forward.  ``prefix`` is '' for a bare GraphNet state dict and
    'graph_net.' for a CombinedModel state dict."""
    e = edge_features(pos, edge_index)
    h = mlp_forward(sd, prefix + "node_encoder", x)
    e = mlp_forward(sd, prefix + "edge_encoder", e)
    for b in range(n_blocks_of(sd, prefix)):
        h, e = gn_block(sd, f"{prefix}graph_processor.blocks.{b}", h, edge_index, e)
    y = mlp_forward(sd, prefix + "node_decoder", h)
    if return_latents:
        return y, h, e
    return y


# --------------------------------------------------------------------------
# a8: CombinedModel / LinearClassifier (models/GNN.py:312-341)
# --------------------------------------------------------------------------
def classifier_forward(sd: dict, v: Tensor, prefix: str = "classifier.") -> Tensor:
    """models/GNN.py:320-325: fc1 ReLU fc2 ReLU fc3 on a 1-D vector."""
    v = linear(v, sd[prefix + "fc1.weight"], sd[prefix + "fc1.bias"]).clamp_min(0)
    v = linear(v, sd[prefix + "fc2.weight"], sd[prefix + "fc2.bias"]).clamp_min(0)
    return linear(v, sd[prefix + "fc3.weight"], sd[prefix + "fc3.bias"])


def combined_forward(sd: dict, x: Tensor, pos: Tensor, edge_index: Tensor) -> Tensor:
    """models/GNN.py:334-341: graph_net -> flatten -> classifier; logits are 1-D."""
    y = graphnet_forward(sd, x, pos, edge_index, prefix="graph_net.")
    return classifier_forward(sd, y.flatten())


# --------------------------------------------------------------------------
# input format of the pixel/patch builders (utils/image_to_graph/image_to_graph_optimized.py:7-39)
# --------------------------------------------------------------------------
def grid_edge_index(H: int, W: int, diagonals: bool = False) -> np.ndarray:
    """One-directional 4-neighbour grid: all left->right edges (row-major), then
    all top->bottom edges, then optionally the two diagonal families
    (optimized.py:19-37).  Returns int64 [2, E]."""
    src, dst = [], []
    for r in range(H):
        for c in range(W - 1):
            src.append(r * W + c); dst.append(r * W + c + 1)
    for r in range(H - 1):
        for c in range(W):
            src.append(r * W + c); dst.append((r + 1) * W + c)
    if diagonals:
        for r in range(H - 1):
            for c in range(W - 1):
                src.append(r * W + c); dst.append((r + 1) * W + c + 1)
        for r in range(H - 1):
            for c in range(W - 1):
                src.append(r * W + c + 1); dst.append((r + 1) * W + c)
    return np.array([src, dst], dtype=np.int64).reshape(2, -1)


def to_dtype(sd: dict, dtype) -> dict:
    return {k: v.to(dtype) for k, v in sd.items()}
