"""Graph topology for the HIP message-passing path: destination-sorted CSR, built once per
``edge_index`` and cached.

The reference re-derives the aggregation pattern inside every ``index_add_``
(models/GNN.py:18-20) and gathers ``x[row]``, ``x[col]`` through int64 indices on every GN
block (PyG MetaLayer, called at models/GNN.py:215).  Here ``edge_index`` is converted once:

* ``rowptr`` int32 [N+1], ``perm`` int32 [E]  -- stable sort of the edges by destination
  (``perm[k]`` = original edge id at sorted position ``k``);
* ``src_sorted`` / ``dst_sorted`` int32 [E]    -- endpoints of the edge at sorted position k;
* ``row32`` / ``col32`` int32 [E]               -- endpoints in ORIGINAL edge order (for the
  operator-level API and its backward).

The pixel/patch builders reuse one topology per image size (reference
utils/image_to_graph/image_to_graph_optimized.py:42-47 caches it with lru_cache); the cache
below does the same by content hash for host tensors and by identity for device tensors.
"""
from __future__ import annotations

import hashlib
from collections import OrderedDict

import torch

from . import native


# How an out-of-range node id in ``edge_index`` is reported (the reference raises IndexError inside its scatter's
# ``index_add_``, models/GNN.py:18-20):
#   "sync"     (default) the topology build reads its two device flags back - one host sync per build - and raises;
#   "deferred" the build does NOT synchronise: the flags stay on the device, the kernels run on sanitised ids (never a
#              fault), ``GraphNet`` poisons its output with NaN on the device when a flag is set, and
#              ``check_deferred()`` - called by whoever next synchronises anyway (bench.py: after the timed region) -
#              reads every pending flag with one sync and raises the IndexError then.  This is what lets the host run
#              ahead of the GPU across steps (small per-rank batches of a strong-scaling run are launch-bound otherwise).
_VALIDATION = "sync"
_pending: list = []


def set_validation(mode: str) -> None:
    global _VALIDATION
    if mode not in ("sync", "deferred"):
        raise ValueError("validation mode must be 'sync' or 'deferred'")
    _VALIDATION = mode


def check_deferred() -> None:
    """Read every pending out-of-range flag (one host sync) and raise the IndexError a synchronous build would have.
    A topology whose flag is set is evicted from the caches first, so a later call with the same ``edge_index`` builds
    (and reports) it again instead of hitting an entry whose flag nobody will read a second time."""
    global _pending
    flags, _pending = _pending, []
    if flags:
        bad = torch.stack([f.any() for f, _ in flags]).tolist()
        first = None
        for is_bad, (status, n) in zip(bad, flags):
            if is_bad:
                _default_cache.evict_status(status)
                first = n if first is None else first
        if first is not None:
            raise IndexError(f"edge_index has node ids outside [0, {first})")


class GraphTopology:
    """Device-resident CSR view of one ``edge_index`` [2, E] (int64) over ``num_nodes`` nodes."""

    def __init__(self, edge_index: torch.Tensor, num_nodes: int, device=None, validate: bool | str = True):
        if edge_index.dim() != 2 or edge_index.size(0) != 2:
            raise ValueError(f"edge_index must be [2, E], got {tuple(edge_index.shape)}")
        device = torch.device(device) if device is not None else edge_index.device
        if device.type != "cuda":
            raise RuntimeError("GraphTopology lives on the GPU: pass device='cuda' (no CPU fallback exists)")
        ei = edge_index.to(device=device, dtype=torch.int64, non_blocking=True)
        self.num_nodes = int(num_nodes)
        self.num_edges = int(ei.size(1))
        self.device = device
        row, col = ei[0].contiguous(), ei[1].contiguous()
        mode = _VALIDATION if validate is True else validate
        # ONE call (gnc_topology_build): graph-ordered batches are sorted range by range in LDS, anything else raises a
        # device flag and takes a general path.  Deferred / unvalidated builds enqueue that general path gated on the
        # device (no host sync, capturable); a synchronous build reads the flags back anyway and reruns through rocPRIM.
        sync = bool(mode) and mode != "deferred"
        self.rowptr, self.perm, self.src_sorted, self.dst_sorted, flags = native.topology_build(row, col, self.num_nodes,
                                                                                                gated_fallback=not sync)
        self._row, self._col = row, col  # int32 copies in ORIGINAL edge order are built on first use
        self._row32 = None
        self._col32 = None
        self._inv_perm = None
        self._csc = None
        self.status = flags[:2]  # device int32 [2]: out-of-range destination / source flags
        if mode == "deferred":
            _pending.append((self.status, self.num_nodes))
            # bound the list; never from inside a hipGraph capture (the read-back is a host sync)
            if len(_pending) > 4096 and not torch.cuda.is_current_stream_capturing():
                check_deferred()
        elif mode:
            # one host sync per topology build; the reference syncs on every scatter
            # (models/GNN.py:16-17 `index.max().item()`) and raises IndexError for a bad index
            bad_dst, bad_src, general = flags.tolist()  # all flags, one sync
            if bad_dst or bad_src:
                raise IndexError(f"edge_index has node ids outside [0, {self.num_nodes})")
            if general:  # not a graph-ordered batch of small graphs: global radix sort (rocPRIM) + the two permute passes
                self.rowptr, self.perm, status = native.csr_build(col, self.num_nodes)
                self.src_sorted = native.permute_index_checked(row, self.perm, self.num_nodes, status[1:])
                self.dst_sorted = native.permute_index(col, self.perm)
                self.status = status
                if any(status.tolist()):  # the LDS path had not looked at every source before it gave up
                    raise IndexError(f"edge_index has node ids outside [0, {self.num_nodes})")
        self.deferred = mode == "deferred"

    @property
    def row32(self) -> torch.Tensor:
        """int32 [E] source ids in original edge order (operator-level calls only)."""
        if self._row32 is None:
            self._row32 = native.permute_index(self._row, None)
        return self._row32

    @property
    def col32(self) -> torch.Tensor:
        """int32 [E] destination ids in original edge order (backward of the operator-level scatter_sum)."""
        if self._col32 is None:
            self._col32 = native.permute_index(self._col, None)
        return self._col32

    @property
    def inv_perm(self) -> torch.Tensor:
        """int32 [E]: sorted position of original edge e (inverse of ``perm``)."""
        if self._inv_perm is None:
            inv = torch.empty_like(self.perm)
            inv[self.perm.long()] = torch.arange(self.num_edges, dtype=torch.int32, device=self.device)
            self._inv_perm = inv
        return self._inv_perm

    @property
    def csc(self):
        """(rowptr, perm) of the SOURCE-sorted order, in sorted-edge numbering: used by the
        backward of the fused gather (grad wrt x[src]) -- built on first use."""
        if self._csc is None:
            # sources of a graph-ordered batch are as local as its destinations: same LDS path (int32 ids, no endpoints)
            rp, pm, _, _, _ = native.topology_build(None, self.src_sorted, self.num_nodes, gated_fallback=True)
            self._csc = (rp, pm)
        return self._csc


class TopologyCache:
    """Small LRU of topologies.  Host tensors are keyed by content (shape + blake2 digest of the
    bytes: a 32x32 pixel grid is 32 KB), device tensors by identity and version; a cached
    device entry keeps its ``edge_index`` alive so the address cannot be recycled under it."""

    def __init__(self, capacity: int = 16):
        self.capacity = capacity
        self._entries: OrderedDict = OrderedDict()
        self.hits = 0
        self.misses = 0

    @staticmethod
    def _key(edge_index: torch.Tensor, num_nodes: int, device):
        if edge_index.is_cuda:
            # shape alone does not pin the content: pairs.t() and pairs.view(2, E) share address, shape and version
            return ("dev", edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape), tuple(edge_index.stride()),
                    str(edge_index.dtype), num_nodes, str(device))
        buf = edge_index.contiguous().numpy().tobytes()
        return ("host", hashlib.blake2b(buf, digest_size=16).digest(), tuple(edge_index.shape), str(edge_index.dtype),
                num_nodes, str(device))

    def get(self, edge_index: torch.Tensor, num_nodes: int, device) -> GraphTopology:
        key = self._key(edge_index, num_nodes, device)
        hit = self._entries.get(key)
        if hit is not None:
            self._entries.move_to_end(key)
            self.hits += 1
            topo = hit[0]
            if topo.deferred and _VALIDATION == "sync":
                # built without a read-back, asked for again in synchronous mode: read its flags now (one sync, once)
                topo.deferred = False
                if any(topo.status.tolist()):
                    del self._entries[key]
                    raise IndexError(f"edge_index has node ids outside [0, {topo.num_nodes})")
            return topo
        self.misses += 1
        topo = GraphTopology(edge_index, num_nodes, device=device)
        self._entries[key] = (topo, edge_index if edge_index.is_cuda else None)
        while len(self._entries) > self.capacity:
            self._entries.popitem(last=False)
        return topo

    def clear(self):
        self._entries.clear()

    def evict_status(self, status: torch.Tensor) -> None:
        """Drop the entry whose topology owns ``status`` (check_deferred: its flag was set)."""
        for key, (topo, _) in list(self._entries.items()):
            if getattr(topo, "status", None) is status:
                del self._entries[key]


_default_cache = TopologyCache()


class DestinationCSR:
    """What the operator-level ``scatter_sum(src, index)`` needs and nothing more: the stable destination sort of ONE
    index vector (``rowptr``, ``perm``) and, for its backward, the int32 copy of the index.  One host sync (the
    out-of-range flag -> IndexError, as models/GNN.py:18-20's ``index_add_`` raises)."""

    def __init__(self, index: torch.Tensor, num_nodes: int, device):
        idx = index.to(device=device, dtype=torch.int64, non_blocking=True).contiguous()
        self.num_nodes = int(num_nodes)
        self.rowptr, self.perm, status = native.csr_build(idx, self.num_nodes)
        self._index = idx
        self._col32 = None
        if int(status[0].item()):
            raise IndexError(f"scatter_sum: index has entries outside [0, {self.num_nodes})")

    @property
    def col32(self) -> torch.Tensor:
        if self._col32 is None:
            self._col32 = native.permute_index(self._index, None)
        return self._col32


class _DestinationCache(TopologyCache):
    """Keyed on the index vector itself (the public scatter_sum used to stack a fresh [2, E] tensor per call, which
    could never hit the identity-keyed cache and pinned dead copies in the LRU)."""

    def get(self, index: torch.Tensor, num_nodes: int, device) -> DestinationCSR:
        key = self._key(index, num_nodes, device)
        hit = self._entries.get(key)
        if hit is not None:
            self._entries.move_to_end(key)
            self.hits += 1
            return hit[0]
        self.misses += 1
        csr = DestinationCSR(index, num_nodes, device)
        self._entries[key] = (csr, index if index.is_cuda else None)
        while len(self._entries) > self.capacity:
            self._entries.popitem(last=False)
        return csr


_destination_cache = _DestinationCache(capacity=4)


def get_destination_csr(index: torch.Tensor, num_nodes: int, device) -> DestinationCSR:
    return _destination_cache.get(index, num_nodes, device)


def get_topology(edge_index: torch.Tensor, num_nodes: int, device) -> GraphTopology:
    return _default_cache.get(edge_index, num_nodes, device)


def clear_topology_cache():
    _default_cache.clear()
    _destination_cache.clear()
