"""Deterministic synthetic inputs of the hot path (BASELINE.json configs).

The dataset of the reference is not shipped and real SLIC segmentations cannot be produced
offline, so benchmarks and full-size parity tests use these generators.  All use
``numpy.random.default_rng(PCG64)`` with seed = 1000 + config index and return host tensors in
the format ``utils/dataloader.py:49-51`` produces: ``x`` float32 [N, F], ``pos`` float32 [N, 2],
``edge_index`` int64 [2, E] (row 0 = source, row 1 = destination), plus ``graph_ptr`` int64
[G+1] (node offsets of the block-diagonal batch).

Edge order mimics the superpixel builder (reference
utils/image_to_graph/image_to_graph_superpixel.py:54-66): each undirected pair {i<j}, in
lexicographic order, is emitted as [i,j] then [j,i]; destinations are therefore NOT sorted at
the boundary.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch


@dataclass
class GraphBatch:
    x: torch.Tensor           # [N, F] float32
    pos: torch.Tensor         # [N, 2] float32
    edge_index: torch.Tensor  # [2, E] int64
    graph_ptr: torch.Tensor   # [G+1] int64 node offsets
    edge_ptr: torch.Tensor    # [G+1] int64 edge offsets

    @property
    def num_graphs(self) -> int:
        return self.graph_ptr.numel() - 1

    @property
    def num_nodes(self) -> int:
        return self.x.size(0)

    @property
    def num_edges(self) -> int:
        return self.edge_index.size(1)

    def to(self, device):
        return GraphBatch(self.x.to(device), self.pos.to(device), self.edge_index.to(device), self.graph_ptr, self.edge_ptr)

    def slice_graphs(self, g0: int, g1: int) -> "GraphBatch":
        """Graphs [g0, g1) as their own batch with local node ids (used for graph-id sharding)."""
        n0, n1 = int(self.graph_ptr[g0]), int(self.graph_ptr[g1])
        e0, e1 = int(self.edge_ptr[g0]), int(self.edge_ptr[g1])
        return GraphBatch(self.x[n0:n1], self.pos[n0:n1], self.edge_index[:, e0:e1] - n0,
                          self.graph_ptr[g0:g1 + 1] - n0, self.edge_ptr[g0:g1 + 1] - e0)


def _pairs_to_edges(i: np.ndarray, j: np.ndarray) -> np.ndarray:
    """undirected pairs (i<j, already in lexicographic order) -> [2, 2P] interleaved [i,j],[j,i]."""
    ei = np.empty((2, 2 * i.size), dtype=np.int64)
    ei[0, 0::2], ei[1, 0::2] = i, j
    ei[0, 1::2], ei[1, 1::2] = j, i
    return ei


def random_pair_graphs(num_graphs: int, nodes_per_graph: int, pairs_per_graph: int, feat_dim: int, seed: int) -> GraphBatch:
    """C3/C5 family: per graph ``pairs_per_graph`` distinct undirected pairs sampled uniformly
    without replacement -> 2*pairs directed edges; in-degree ~ Binomial."""
    rng = np.random.default_rng(seed)
    n = nodes_per_graph
    iu, ju = np.triu_indices(n, k=1)  # lexicographic (i, j), i < j
    total_pairs = iu.size
    if pairs_per_graph > total_pairs:
        raise ValueError("more pairs than the graph has")
    # sample without replacement per graph: the pairs_per_graph smallest of total_pairs random keys
    chunks = []
    step = max(1, min(num_graphs, (1 << 24) // total_pairs))
    for g0 in range(0, num_graphs, step):
        g1 = min(num_graphs, g0 + step)
        keys = rng.random((g1 - g0, total_pairs), dtype=np.float32)
        sel = np.argpartition(keys, pairs_per_graph - 1, axis=1)[:, :pairs_per_graph]
        sel.sort(axis=1)
        off = (np.arange(g0, g1, dtype=np.int64) * n)[:, None]
        ei = _pairs_to_edges((iu[sel] + off).ravel(), (ju[sel] + off).ravel())
        chunks.append(ei)
    edge_index = np.concatenate(chunks, axis=1)
    num_nodes = num_graphs * n
    x = rng.random((num_nodes, feat_dim), dtype=np.float32)
    pos = (rng.random((num_nodes, 2), dtype=np.float32) * 32.0).astype(np.float32)
    graph_ptr = np.arange(num_graphs + 1, dtype=np.int64) * n
    edge_ptr = np.arange(num_graphs + 1, dtype=np.int64) * (2 * pairs_per_graph)
    return GraphBatch(torch.from_numpy(x), torch.from_numpy(pos), torch.from_numpy(edge_index),
                      torch.from_numpy(graph_ptr), torch.from_numpy(edge_ptr))


def _triangulated_grid_pairs(a: int, b: int, diag_flip: np.ndarray):
    """a x b jittered grid, every cell split by one diagonal (diag_flip[cell] picks which):
    planar triangulation like a superpixel region-adjacency graph.  Returns pairs (i<j) sorted."""
    idx = np.arange(a * b).reshape(a, b)
    pi = [idx[:, :-1].ravel(), idx[:-1, :].ravel()]
    pj = [idx[:, 1:].ravel(), idx[1:, :].ravel()]
    tl, tr, bl, br = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel()
    pi.append(np.where(diag_flip, tr, tl))
    pj.append(np.where(diag_flip, bl, br))
    i, j = np.concatenate(pi), np.concatenate(pj)
    lo, hi = np.minimum(i, j), np.maximum(i, j)
    order = np.lexsort((hi, lo))
    return lo[order], hi[order]


def superpixel_like_graphs(num_graphs: int, seed: int, shapes=((12, 12), (12, 13), (13, 13))) -> GraphBatch:
    """C1/C2 family: ~150-node planar triangulations (144/156/169 nodes, ~770-912 directed edges,
    in-degree 2..8), x ~ U[0,1)^3 (mean-RGB range of superpixel.py:28,43), pos = jittered cell
    centres in [0,32)^2."""
    rng = np.random.default_rng(seed)
    which = rng.integers(0, len(shapes), size=num_graphs)
    xs, ps, eis, gptr, eptr = [], [], [], [0], [0]
    for g in range(num_graphs):
        a, b = shapes[which[g]]
        flip = rng.random((a - 1) * (b - 1)) < 0.5
        lo, hi = _triangulated_grid_pairs(a, b, flip)
        eis.append(_pairs_to_edges(lo, hi) + gptr[-1])
        rr, cc = np.meshgrid(np.arange(a), np.arange(b), indexing="ij")
        centre = np.stack([(rr.ravel() + 0.5) * (32.0 / a), (cc.ravel() + 0.5) * (32.0 / b)], axis=1)
        jitter = (rng.random((a * b, 2)) - 0.5) * np.array([32.0 / a, 32.0 / b]) * 0.6
        ps.append((centre + jitter).astype(np.float32))
        xs.append(rng.random((a * b, 3), dtype=np.float32))
        gptr.append(gptr[-1] + a * b)
        eptr.append(eptr[-1] + eis[-1].shape[1])
    return GraphBatch(torch.from_numpy(np.concatenate(xs)), torch.from_numpy(np.concatenate(ps)),
                      torch.from_numpy(np.concatenate(eis, axis=1)), torch.tensor(gptr, dtype=torch.int64),
                      torch.tensor(eptr, dtype=torch.int64))


# ---- BASELINE.json configs ------------------------------------------------------------------
WORKLOADS = {
    # name: (generator kwargs, GraphNet kwargs, description)
    "c2": dict(kind="superpixel", num_graphs=10_000, seed=1001, width=128, n_blocks=3,
               desc="10k superpixel-like graphs x ~150 nodes x ~850 edges, feat_dim=3, D=128, 3 GN blocks"),
    "c3": dict(kind="pairs", num_graphs=6_250, nodes=160, pairs=800, seed=1002, width=64, n_blocks=2,
               desc="1M nodes / 10M edges (6250 graphs x 160 nodes x 1600 edges), D=64, 2 GN blocks"),
    "c5": dict(kind="pairs", num_graphs=3_125, nodes=160, pairs=800, seed=1004, width=256, n_blocks=3,
               desc="500k nodes / 5M edges (3125 graphs x 160 nodes x 1600 edges), D=256, 3 GN blocks"),
}


def graphnet_kwargs(width: int, n_blocks: int, out_channels: int = 1) -> dict:
    """All latent/hidden widths equal to ``width``."""
    return dict(num_local_features=3, space_dim=2, out_channels=out_channels, n_blocks=n_blocks,
                out_dim_node=width, out_dim_edge=width, hidden_dim_node=width, hidden_dim_edge=width,
                hidden_dim_decoder=width, hidden_dim_processor_node=width, hidden_dim_processor_edge=width)


def make_workload(name: str, scale: float = 1.0) -> tuple[GraphBatch, dict]:
    """Returns (batch, GraphNet kwargs).  ``scale`` < 1 shrinks the number of graphs (tests,
    CPU-baseline samples); the per-graph statistics are unchanged."""
    w = WORKLOADS[name]
    g = max(1, int(round(w["num_graphs"] * scale)))
    if w["kind"] == "superpixel":
        batch = superpixel_like_graphs(g, w["seed"])
    else:
        batch = random_pair_graphs(g, w["nodes"], w["pairs"], 3, w["seed"])
    return batch, graphnet_kwargs(w["width"], w["n_blocks"])
