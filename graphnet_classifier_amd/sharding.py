"""Multi-GPU layout of the hot path: one process per GPU, graphs sharded by graph id.

A batch is a disjoint union of small graphs and no edge crosses graphs
(reference utils/dataloader.py:33-53 yields one self-contained graph per item), so forward and
backward need no exchange at all.  The single collective of a training step is one
``all_reduce(SUM)`` over ONE flat fp32 buffer holding every parameter gradient, inserted
between ``loss.backward()`` and ``optimizer.step()`` (reference utils/train_model.py:41-42).
The buffer is 682,339 floats = 2.73 MB at the reference defaults: latency-bound on xGMI, so it
is sent as a single bucket (``backend="nccl"`` is RCCL on ROCm; ``gloo`` is used by the CPU tests).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def shard_ranges(edge_ptr, world_size: int):
    """Contiguous graph-id ranges [(g0, g1)] * world_size, balanced by edge count (the forward's
    cost is ~5 D^2 MACs per edge vs 4 D^2 per node and E ~ 5-10 N).  ``edge_ptr`` is the [G+1]
    prefix sum of edges per graph.  Every graph belongs to exactly one rank; ranks may be empty
    only when there are fewer graphs than ranks."""
    ep = np.asarray(edge_ptr, dtype=np.int64)
    g = ep.size - 1
    total = int(ep[-1])
    cuts = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        c = int(np.searchsorted(ep, target, side="left"))
        # nearest boundary, monotone, and leave at least one graph per remaining rank if possible
        if c > 0 and abs(ep[c - 1] - target) <= abs(ep[min(c, g)] - target):
            c -= 1
        c = max(c, cuts[-1] + (1 if g >= world_size else 0))
        c = min(c, g - (world_size - r) if g >= world_size else g)
        cuts.append(max(c, cuts[-1]))
    cuts.append(g)
    return [(cuts[r], cuts[r + 1]) for r in range(world_size)]


def _world(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


class FlatGradAllReduce:
    """Every parameter gradient as a VIEW of one flat fp32 buffer, and one collective over that buffer.

    Protocol of a step (utils/train_model.py:40-42 with the collective inserted)::

        reducer.zero_grad()      # p.grad = None: backward hands its freshly computed tensors over, no zero-fill,
                                 # no per-parameter accumulate launch
        loss.backward()
        reducer()                # pack (ONE multi-tensor launch) + ONE all-reduce; afterwards p.grad IS the view
        optimizer.step()         # reads p.grad = views of the flat buffer (a fused optimizer reads `flat` itself)

    * zero copy out: after the call ``p.grad`` aliases ``flat``; nothing is copied back per parameter;
    * the copy in is one ``torch._foreach_copy_`` over all parameters (one or two multi-tensor kernels, never a
      ``copyBuffer`` per parameter); a gradient that autograd accumulated in place into last step's view (a caller
      that skipped ``zero_grad``) is already in the buffer and is not touched;
    * at world size 1 with ``pack_always=False`` the call returns at once (a true no-op);
    * ``average=True`` divides by the world size after the SUM (gradient of the mean of per-rank mean losses).
      Ranks with unequal graph counts (``shard_ranges`` balances by edges) should instead scale their LOCAL loss by
      ``1 / global_graph_count`` (sum-reduced loss) and pass ``average=False``: the SUM of those gradients is exactly
      the gradient of the global-batch mean loss.
    """

    def __init__(self, params, group=None, average: bool = True, pack_always: bool = False, flat: torch.Tensor | None = None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.average = average
        self.pack_always = pack_always
        self.numel = sum(p.numel() for p in self.params)
        p0 = self.params[0]
        self.flat = flat if flat is not None else torch.zeros(self.numel, dtype=torch.float32, device=p0.device)
        assert self.flat.numel() == self.numel and self.flat.dtype == torch.float32
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.collectives = 0  # number of all-reduce calls issued (tests assert one per step)

    def zero_grad(self) -> None:
        for p in self.params:
            p.grad = None

    def pack(self, grads=None) -> torch.Tensor:
        """Bring every gradient into the flat buffer and make ``p.grad`` the view of it.  ``grads``: the gradients
        (one per parameter, None = unused) when they did not land in ``p.grad`` - a step that differentiated private
        leaf aliases of the parameters (train.CapturedTrainStep) hands them over here."""
        srcs, dsts = [], []
        for k, (p, v) in enumerate(zip(self.params, self.views)):
            g = p.grad if grads is None else grads[k]
            if g is None:
                v.zero_()  # parameter unused by this step's graph
            elif g.data_ptr() != v.data_ptr():
                srcs.append(g if g.dtype == torch.float32 else g.float())
                dsts.append(v)
            p.grad = v
        if srcs:
            torch._foreach_copy_(dsts, srcs)
        return self.flat

    def __call__(self, grads=None) -> torch.Tensor | None:
        """Call between backward() and optimizer.step(); returns the flat (reduced) buffer."""
        world = _world(self.group)
        if world == 1 and not self.pack_always:
            return None
        self.pack(grads)
        return self.allreduce()

    def allreduce(self) -> torch.Tensor:
        """The collective alone, over an already packed buffer (a step whose forward + backward + pack replay from a
        hipGraph issues it directly after the replay)."""
        world = _world(self.group)
        if world > 1:
            if self.flat.is_cuda and dist.get_backend(self.group) == "gloo":  # CPU rehearsal backend: stage through the host
                host = self.flat.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                self.flat.copy_(host)
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)  # ONE collective per step (RCCL)
            self.collectives += 1
            if self.average:
                self.flat.div_(world)
        return self.flat
