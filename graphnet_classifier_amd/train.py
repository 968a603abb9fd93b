"""Training loop of the reference (``utils/train_model.py:8-81``) on the HIP path - SURVEY.md section 8, row f3.

``train(model, dataset, epochs, patience=5, output_path='weights', start_weights=None)`` has the reference's
signature and observable behaviour: Adam(lr=1e-3) (:9), CrossEntropyLoss (:10), one optimizer step per sample in
dataset order (:35-45), average training loss per epoch, early stopping on it (:57-69), ``best_model_epoch{k}.pth``
/ ``final_model.pth`` (:61-62, :72-73) and the timestamped log file with the reference's line formats (:22-30,
:52-54, :76-80).  What differs is how a step runs and how the run is organised:

* parameters and gradients are views of two flat fp32 buffers (``FlatParameters``), the optimizer is ONE fused Adam
  launch over them (``FusedAdam`` -> ``gnc_adam_step_f32``) instead of a walk over 76 tensors;
* the running loss is accumulated on the device (float64 scalar) and read once per epoch, where the reference
  synchronises on ``loss.item()`` after every sample (:44);
* when consecutive samples share one topology (pixel / patch graphs: the edge list depends on the image size only,
  utils/image_to_graph/image_to_graph_optimized.py:42-47) the whole step - forward, loss, backward, gradient pack,
  Adam - is captured into a hipGraph once and replayed per sample (``CapturedTrainStep``): one launch per step
  instead of ~150;
* the run's side effects live in three small objects: ``RunJournal`` (the log file and the console lines),
  ``PlateauStopper`` (best loss so far, epochs without improvement) and ``CheckpointShelf`` (the ``.pth`` files).

Saved ``.pth`` files hold CPU tensors under the reference's state-dict keys, so they load in the reference as they
are (``utils/inference.py:40-45``) and vice versa.
"""
from __future__ import annotations

import os
import time
from datetime import datetime

import torch
import torch.distributed as dist
import torch.nn as nn

from . import native
from .MLP import require_gpu_param
from .sharding import FlatGradAllReduce

ALIGN = 64  # floats: every parameter starts on a 256-B boundary of the flat buffer (the MLP kernels want 16-B aligned weights)


def _layout(params, align: int = ALIGN):
    offs, off = [], 0
    for p in params:
        offs.append(off)
        off += (p.numel() + align - 1) // align * align
    return offs, off


class FlatParameters:
    """Re-homes every trainable parameter of ``module`` as a view of ONE flat fp32 buffer (``flat``) and provides the
    matching flat gradient buffer (``grad``; ``p.grad`` become views of it through ``reducer``).  Offsets are padded
    to 256 B; the gaps hold zeros in both buffers and stay zero under Adam (0 gradient -> 0 update).

    ``state_dict`` keys, shapes and values are unchanged; ``module.to(...)`` afterwards would detach the views, so
    construct this last."""

    def __init__(self, module: nn.Module, group=None, average: bool = True):
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        self.names = [n for n, _ in named]
        self.params = [p for _, p in named]
        if not self.params:
            raise ValueError("module has no trainable parameters")
        dev = require_gpu_param(self.params[0], "FlatParameters")
        if any(p.dtype != torch.float32 or p.device != dev for p in self.params):
            raise TypeError("FlatParameters: float32 parameters on one GPU expected")
        self.offsets, self.numel = _layout(self.params)
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):
                view = self.flat[off:off + p.numel()].view_as(p)
                view.copy_(p)
                p.data = view
        self.reducer = _AlignedReducer(self.params, self.grad, self.offsets, group, average)


class _AlignedReducer(FlatGradAllReduce):
    """FlatGradAllReduce over a caller-provided flat buffer with padded offsets (always packs: the fused optimizer reads
    the flat buffer)."""

    def __init__(self, params, flat, offsets, group, average):
        self.params, self.group, self.average, self.pack_always = list(params), group, average, True
        self.flat, self.numel = flat, flat.numel()
        self.views = [flat[off:off + p.numel()].view_as(p) for p, off in zip(self.params, offsets)]
        self.collectives = 0


class FusedAdam:
    """torch.optim.Adam(params, lr, betas, eps, weight_decay) semantics (amsgrad off) as one launch over the flat
    buffers of a ``FlatParameters`` (``gnc_adam_step_f32``).  The step counter is a device tensor, so ``step()`` is
    the same launch sequence every time (hipGraph-capturable)."""

    def __init__(self, flat_params: FlatParameters, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0):
        self.fp = flat_params
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        dev = flat_params.flat.device
        self.exp_avg = torch.zeros_like(flat_params.flat)
        self.exp_avg_sq = torch.zeros_like(flat_params.flat)
        self.step_count = torch.zeros(1, dtype=torch.int64, device=dev)
        self._scratch = torch.zeros(2, dtype=torch.float32, device=dev)

    def zero_grad(self, set_to_none: bool = True) -> None:
        self.fp.reducer.zero_grad()

    def step(self, reduce: bool = True, grads=None) -> None:
        """Gradient pack (+ the one all-reduce when a process group with more than one rank is up) and the update.
        ``grads``: see FlatGradAllReduce.pack."""
        if reduce:
            self.fp.reducer(grads)
        native.adam_step(self.fp.flat, self.fp.grad, self.exp_avg, self.exp_avg_sq, self.step_count, self._scratch, self.lr,
                         self.betas[0], self.betas[1], self.eps, self.weight_decay)

    def state_snapshot(self):
        return [t.clone() for t in (self.fp.flat, self.exp_avg, self.exp_avg_sq, self.step_count)]

    def state_restore(self, snap) -> None:
        for dst, src in zip((self.fp.flat, self.exp_avg, self.exp_avg_sq, self.step_count), snap):
            dst.copy_(src)


def _same_topology(a: torch.Tensor, b: torch.Tensor) -> bool:
    return a is b or (a.shape == b.shape and a.dtype == b.dtype and a.device == b.device and bool(torch.equal(a, b)))


def _world(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


class _Through(nn.Module):
    """``functional_call`` target: runs ``fn(model, *args)`` with the model's parameters swapped for the given ones."""

    def __init__(self, model: nn.Module, fn):
        super().__init__()
        self.model, self.fn = model, fn

    def forward(self, *args):
        return self.fn(self.model, *args)


class CapturedTrainStep:
    """One training step (utils/train_model.py:37-42: forward, CE loss, zero_grad, backward, Adam) for ONE fixed
    topology, captured into a hipGraph and replayed per sample.  The loss of every replay is added to ``loss_sum``
    (device float64) inside the graph.  Warm-up steps run before the capture; parameters, optimizer state, module
    buffers and the loss accumulator are restored afterwards, so constructing this object does not train.

    **Safe whatever the caller ran before.**  A parameter's ``AccumulateGrad`` node remembers the stream that was
    current when the node was created, and the node lives as long as ANY autograd graph that reached the parameter
    (a ``loss`` tensor the caller still holds is enough).  Had the caller trained eagerly on the legacy default
    stream, a backward inside the capture would find those nodes, and the autograd engine would make the legacy
    stream wait on an event of the capturing stream and accumulate there: a default-stream operation in the middle
    of a stream capture, which takes the HIP runtime down at ``hipStreamEndCapture`` (tools/repro_capture3.py
    reproduces it in plain PyTorch: MODE=stale_default crashes, fresh_default / stale_side / alias_default do not).
    The captured step therefore never differentiates the module's own parameter tensors: it runs the module through
    ``torch.func.functional_call`` on PRIVATE leaf aliases over the same storage (views of the flat buffer), whose
    autograd nodes are born on the warm-up / capture stream and die with each step's graph, and hands their gradients
    to the reducer's pack.  Nothing of the caller's autograd history is reachable from the capture.

    ``forward(model, x, pos, edge_index) -> logits`` replaces the default ``model((x, pos, edge_index))`` (a batched
    read-out, for instance); ``loss_scale`` multiplies the loss (``1 / global graph count`` of a sharded batch).
    With a process group of more than one rank the graph holds forward + backward + gradient pack, and every call
    issues the ONE all-reduce and the fused Adam launch directly after the replay.

    ``edge_capacity=C``: the step is captured for ANY edge list of at most C edges over the sample's node count (a new
    topology per sample: superpixel graphs) - see the comment in ``__init__``.  A custom ``forward`` then receives the PADDED
    buffers (``x`` / ``pos`` with the dummy rows behind the first ``num_nodes``, ``edge_index`` [2, C]); it must build its
    topology from them in every call (``GraphTopology(..., validate="deferred")``, never the cache) and read out the first
    ``num_nodes`` rows only.  ``check()`` raises the IndexError of a replayed edge list with ids outside the graph."""

    def __init__(self, model: nn.Module, optimizer: FusedAdam, criterion, sample, label, loss_sum: torch.Tensor, *,
                 forward=None, loss_scale: float = 1.0, capture_error_mode: str = "global", edge_capacity: int | None = None):
        x, pos, edge_index = sample
        dev = require_gpu_param(next(model.parameters()), "CapturedTrainStep")
        fp = optimizer.fp
        self.optimizer = optimizer
        self.edge_index_host = edge_index
        self.edge_capacity = edge_capacity
        self.num_nodes = int(x.size(0))
        self.label = torch.as_tensor(label).to(dev).clone()
        self.loss_sum = loss_sum
        self.collective_outside = _world(fp.reducer.group) > 1
        from .topology import GraphTopology, get_topology
        if edge_capacity is None:
            self.x = x.to(device=dev, dtype=torch.float32).clone()
            self.pos = pos.to(device=dev, dtype=torch.float32).clone()
            self.edge_index = edge_index.to(dev)
            # the build's host sync happens here, outside the capture; the reference keeps the CSR arrays alive for the graph
            self.topo = get_topology(self.edge_index, self.x.size(0), dev)
            fwd = forward if forward is not None else (lambda mod, xx, pp, ee: mod((xx, pp, ee)))
        else:
            # ANY topology over the same node count (the reference's superpixel graphs: a new region adjacency per image,
            # utils/image_to_graph/image_to_graph_superpixel.py:31-66, one graph per optimizer step, main.py:60): the captured
            # buffers hold extra DUMMY nodes with zero features and ``edge_capacity`` edges, the unused tail of which are
            # self-loops of the dummies (spread round-robin, at most 8 each: ONE dummy would be a hub of hundreds of rows
            # that the per-destination kernels - K1, the topology build's in-destination ranking - walk serially: 0.95 ms
            # per step against 0.6).  Those rows only ever talk to dummies, whose outputs the read-out never sees, so
            # they contribute exact zeros to the logits and to every gradient; the topology build (device flags, no host
            # sync) is PART of the captured step, so a replay sorts whatever edge list the buffer holds.
            n, e = self.num_nodes, int(edge_index.size(1))
            if e > edge_capacity:
                raise ValueError(f"CapturedTrainStep: {e} edges exceed edge_capacity {edge_capacity}")
            if any(isinstance(m, nn.modules.batchnorm._BatchNorm) for m in model.modules()):
                # batch statistics span ALL rows: the dummy rows would enter them (LayerNorm is per row and unaffected)
                raise NotImplementedError("CapturedTrainStep(edge_capacity=...): not with BatchNorm layers (norm_type='BatchNorm1d')")
            if forward is None and not (hasattr(model, "graph_net") and hasattr(model, "classifier")):
                raise TypeError("CapturedTrainStep(edge_capacity=...): pass forward= for a module that is not a CombinedModel")
            dummies = max(1, (edge_capacity + 7) // 8)
            self.x = torch.zeros(n + dummies, *x.shape[1:], dtype=torch.float32, device=dev)
            self.pos = torch.zeros(n + dummies, *pos.shape[1:], dtype=torch.float32, device=dev)
            self._tail = n + torch.arange(edge_capacity, dtype=torch.int64, device=dev) % dummies  # slot k's dummy self-loop
            self.edge_index = self._tail.repeat(2, 1)
            self.x[:n].copy_(x)
            self.pos[:n].copy_(pos)
            self.edge_index[:, :e].copy_(edge_index)
            self.topo, self._status = None, None
            self._range_flag = torch.zeros((), dtype=torch.bool, device=dev)

            def padded_forward(mod, xx, pp, ee):
                topo = GraphTopology(ee, xx.size(0), device=dev, validate="deferred")  # never the cache: built in every step
                self._status = topo.status  # the capture's own flags: every replay rewrites them
                y = mod.graph_net.forward_device(xx, pp, topo)
                return mod.classifier(y[:n].flatten())
            fwd = forward if forward is not None else padded_forward
        through = _Through(model, fwd)
        # private leaves over the parameters' storage (the flat buffer): see the class docstring
        wanted = dict(zip(fp.names, range(len(fp.names))))
        alias, self._leaves = {}, [None] * len(fp.names)
        for name, p in model.named_parameters():
            leaf = p.detach()
            if name in wanted:
                leaf.requires_grad_(True)
                self._leaves[wanted[name]] = leaf
            alias["model." + name] = leaf
        if any(l is None for l in self._leaves):
            raise RuntimeError("CapturedTrainStep: the optimizer's FlatParameters were built over another module")
        leaves = self._leaves

        def one_step():
            logits = torch.func.functional_call(through, alias, (self.x, self.pos, self.edge_index))
            loss = criterion(logits, self.label)
            if loss_scale != 1.0:
                loss = loss * loss_scale
            for leaf in leaves:                                                        # utils/train_model.py:40
                leaf.grad = None
            loss.backward()                                                            # :41
            grads = [leaf.grad for leaf in leaves]
            if self.collective_outside:
                fp.reducer.pack(grads)
            else:
                optimizer.step(grads=grads)                                            # :42 (pack + fused Adam)
            self.loss_sum.add_(loss.detach().double())

        snap = optimizer.state_snapshot()
        buffers = [(b, b.clone()) for b in model.buffers()]  # BatchNorm statistics must not absorb the warm-up either
        keep = self.loss_sum.clone()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(3):  # kernel attributes, allocator pools, lazily built source-sorted CSR
                one_step()
                if self.collective_outside:
                    self._finish()
        torch.cuda.current_stream(dev).wait_stream(side)
        for leaf in leaves:
            leaf.grad = None
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode=capture_error_mode):
            one_step()
        optimizer.state_restore(snap)
        with torch.no_grad():
            for b, saved in buffers:
                b.copy_(saved)
        self.loss_sum.copy_(keep)

    def _finish(self) -> None:
        """Multi-rank tail of a step: the ONE collective over the packed flat buffer, then the fused Adam launch."""
        self.optimizer.fp.reducer.allreduce()
        self.optimizer.step(reduce=False)

    def matches(self, sample) -> bool:
        x, pos, edge_index = sample
        if self.edge_capacity is not None:
            return (x.size(0) == self.num_nodes and x.shape[1:] == self.x.shape[1:] and pos.shape[1:] == self.pos.shape[1:]
                    and edge_index.dim() == 2 and edge_index.size(1) <= self.edge_capacity)
        return x.shape == self.x.shape and pos.shape == self.pos.shape and _same_topology(edge_index, self.edge_index_host)

    def replay(self) -> None:
        """The step on whatever the captured input buffers hold (``self.x`` / ``self.pos`` / ``self.label``)."""
        self.graph.replay()
        if self.collective_outside:
            self._finish()

    def __call__(self, sample, label) -> None:
        x, pos, edge_index = sample
        if self.edge_capacity is None:
            self.x.copy_(x, non_blocking=True)
            self.pos.copy_(pos, non_blocking=True)
        else:
            n, e = self.num_nodes, int(edge_index.size(1))
            if x.size(0) != n or e > self.edge_capacity:
                raise ValueError(f"CapturedTrainStep: sample with {x.size(0)} nodes / {e} edges does not fit the captured "
                                 f"{n} nodes / {self.edge_capacity} edges")
            # ids in [n, n + dummies) would pass the topology build's range check but are not nodes of THIS graph
            if e and not edge_index.is_cuda:
                if int(edge_index.max()) >= n or int(edge_index.min()) < 0:  # host tensor (the loader's): checked here, at once
                    raise IndexError(f"edge_index has node ids outside [0, {n})")
            elif e:
                self._range_flag |= (edge_index >= n).any()  # device tensor: no sync, read by check()
            self.x[:n].copy_(x, non_blocking=True)
            self.pos[:n].copy_(pos, non_blocking=True)
            self.edge_index[:, :e].copy_(edge_index, non_blocking=True)
            self.edge_index[:, e:].copy_(self._tail[e:])  # the tail: self-loops of the dummy nodes
        self.label.copy_(torch.as_tensor(label), non_blocking=True)
        self.replay()

    def check(self) -> None:
        """Padded form: read the device flags of the LAST replayed topology build (one host sync) and raise the IndexError the
        reference's scatter raises for node ids outside the graph (models/GNN.py:18-20)."""
        status = getattr(self, "_status", None)
        if status is not None and bool((status.any() | self._range_flag).item()):
            self._range_flag.zero_()
            raise IndexError(f"edge_index has node ids outside [0, {self.num_nodes})")


# --------------------------------------------------------------------------- the run's side effects
class RunJournal:
    """Everything the reference's train() writes or prints besides checkpoints: the timestamped
    ``training_logs_<stamp>.txt`` (header :26-30, two lines per epoch :52-54, footer :76-80) and the console lines
    (:19, :48, :50, :63, :68, :74).  The line formats are contract (golden G8 compares them)."""

    RULE = "-" * 50

    def __init__(self, directory: str, epochs: int, patience: int):
        os.makedirs(directory, exist_ok=True)
        print(f"Training model in {directory}")
        opened = datetime.now()
        self.epochs = epochs
        self.path = os.path.join(directory, f"training_logs_{opened.strftime('%Y%m%d_%H%M%S')}.txt")
        self._append([f"Training started at: {opened.strftime('%Y-%m-%d %H:%M:%S')}", f"Epochs: {epochs}, Patience: {patience}",
                      f"Output path: {directory}", self.RULE], mode="w")

    def _append(self, lines, mode: str = "a") -> None:
        with open(self.path, mode) as f:
            f.write("".join(line + "\n" for line in lines))

    def epoch(self, number: int, avg_loss: float, seconds: float) -> None:
        tag = f"Epoch {number}/{self.epochs}"
        print(f"{tag}, avg_loss={avg_loss:.4f}")
        print(f"epoch: {number} needed {seconds} time")
        self._append([f"{tag}, avg_loss={avg_loss:.4f}", f"{tag}, needed {seconds / 60:.2f} minutes"])

    def saved(self, kind: str, path: str) -> None:
        print(f"Saved {kind} model: {path}")

    def stopped_early(self, number: int) -> None:
        print(f"Early stopping at epoch {number}")

    def close(self, best_loss: float, final_path: str) -> None:
        self._append([self.RULE, f"Training completed at: {datetime.now().strftime('%Y-%m-%d %H:%M:%S')}",
                      f"Best loss achieved: {best_loss:.4f}", f"Final model saved: {final_path}"])


class PlateauStopper:
    """Early stopping on the average TRAINING loss (utils/train_model.py:57-69): an epoch either sets a new best (strictly
    lower) or counts as stale; ``patience`` stale epochs in a row end the run."""

    def __init__(self, patience: int):
        self.patience, self.best, self.stale = patience, float("inf"), 0

    def observe(self, value: float) -> bool:
        """True when ``value`` is a new best."""
        if value < self.best:
            self.best, self.stale = value, 0
            return True
        self.stale += 1
        return False

    @property
    def exhausted(self) -> bool:
        return self.stale >= self.patience


class CheckpointShelf:
    """The ``.pth`` files of a run (:61-62, :72-73): state dicts under the reference's keys, CPU tensors."""

    def __init__(self, model: nn.Module, directory: str):
        self.model, self.directory = model, directory

    def _write(self, filename: str) -> str:
        path = os.path.join(self.directory, filename)
        torch.save({k: v.detach().cpu().clone() for k, v in self.model.state_dict().items()}, path)
        return path

    def best(self, epoch_number: int) -> str:
        return self._write(f"best_model_epoch{epoch_number}.pth")

    def final(self) -> str:
        return self._write("final_model.pth")


class _SampleStepper:
    """Runs one optimizer step per sample: eagerly, or - once two consecutive samples have shared a topology - as a
    replay of the captured step."""

    def __init__(self, model, optimizer, criterion, loss_sum, device, capture: bool):
        self.model, self.optimizer, self.criterion, self.loss_sum, self.device = model, optimizer, criterion, loss_sum, device
        self.capture = capture
        self.captured: CapturedTrainStep | None = None
        self.padded: CapturedTrainStep | None = None
        self._padded_captures = 0
        self._previous = None

    MAX_PADDED_CAPTURES = 4  # a dataset whose edge counts keep outgrowing the capacity goes back to eager steps

    def _try_replay(self, sample, label) -> bool:
        prev = self._previous
        same_nodes = prev is not None and prev[0].shape == sample[0].shape and prev[1].shape == sample[1].shape
        if self.captured is None and same_nodes and _same_topology(prev[2], sample[2]):
            self.captured = CapturedTrainStep(self.model, self.optimizer, self.criterion, sample, label, self.loss_sum)
        if self.captured is not None and self.captured.matches(sample):
            self.captured(sample, label)
            return True
        # a NEW topology over the same node count (superpixel graphs): the padded form, topology build inside the graph
        if self.padded is not None and not self.padded.matches(sample) and sample[0].size(0) == self.padded.num_nodes:
            self.padded = None  # more edges than the capacity: capture again with room to spare
        if (self.padded is None and same_nodes and self._padded_captures < self.MAX_PADDED_CAPTURES
                and hasattr(self.model, "graph_net") and hasattr(self.model, "classifier")
                and not any(isinstance(m, nn.modules.batchnorm._BatchNorm) for m in self.model.modules())):
            most = max(int(prev[2].size(1)), int(sample[2].size(1)))
            capacity = (most * 3 // 2 + 255) // 256 * 256
            self.padded = CapturedTrainStep(self.model, self.optimizer, self.criterion, sample, label, self.loss_sum,
                                            edge_capacity=capacity)
            self._padded_captures += 1
        if self.padded is not None and self.padded.matches(sample):
            self.padded(sample, label)
            return True
        self._previous = sample
        return False

    def check(self) -> None:
        if self.padded is not None:
            self.padded.check()

    def __call__(self, sample, label) -> None:
        dev = self.device
        is_graph = isinstance(sample, (tuple, list)) and len(sample) == 3
        if self.capture and is_graph and self._try_replay(sample, label):
            return
        if is_graph:
            # edge_index stays where it is: a host tensor is looked up in the topology cache by content, so equal
            # topologies are sorted once, not once per sample
            sample = (sample[0].to(dev, non_blocking=True), sample[1].to(dev, non_blocking=True), sample[2])
        else:
            sample = sample.to(dev, non_blocking=True)
        logits = self.model(sample)                                                            # utils/train_model.py:37
        loss = self.criterion(logits, torch.as_tensor(label).to(dev, non_blocking=True))      # :38
        self.optimizer.zero_grad()                                                             # :40
        loss.backward()                                                                        # :41
        self.optimizer.step()                                                                  # :42
        self.loss_sum += loss.detach().double()                                                # :44, without the per-sample sync


def train(model, dataset, epochs, patience=5, output_path='weights', start_weights=None, *, capture: bool = True, lr: float = 1e-3):
    """utils/train_model.py:8-81 (same positional arguments, files and log lines).  Returns a dict with the per-epoch
    average losses (the reference returns None; nothing in it reads the return value)."""
    if start_weights:
        model.load_state_dict(torch.load(start_weights, map_location="cpu"))           # :14-15
    dev = require_gpu_param(next(model.parameters()), "train")
    optimizer = FusedAdam(FlatParameters(model), lr=lr)                                # :9
    criterion = nn.CrossEntropyLoss()                                                  # :10
    journal = RunJournal(output_path, epochs, patience)
    shelf = CheckpointShelf(model, output_path)
    stopper = PlateauStopper(patience)
    loss_sum = torch.zeros((), dtype=torch.float64, device=dev)
    # this loop steps one rank's model on its own samples: with a multi-rank process group up, a captured step would
    # record (or, under gloo, host-stage) a collective per sample that nothing here asked for
    stepper = _SampleStepper(model, optimizer, criterion, loss_sum, dev, capture and _world() == 1)
    history = []
    # Belt and braces: the loop runs on a side stream, so that the eager steps in front of a capture never touch the
    # legacy default stream.  The capture itself no longer depends on it (CapturedTrainStep differentiates private
    # aliases of the parameters, see its docstring for the hipStreamEndCapture crash this used to be the only guard of).
    run_stream = torch.cuda.Stream(device=dev)
    run_stream.wait_stream(torch.cuda.current_stream(dev))
    try:
        with torch.cuda.stream(run_stream):
            for epoch in range(1, epochs + 1):
                started = time.time()
                loss_sum.zero_()
                steps = 0
                for sample, label in dataset:                                          # :35
                    stepper(sample, label)
                    steps += 1
                avg_loss = float(loss_sum.item()) / max(1, steps)                      # :47 (the epoch's one host sync)
                stepper.check()  # replayed topology builds: node ids outside the graph raise here (models/GNN.py:18-20)
                history.append(avg_loss)
                journal.epoch(epoch, avg_loss, time.time() - started)
                if stopper.observe(avg_loss):                                          # :57-66
                    journal.saved("best", shelf.best(epoch))
                if stopper.exhausted:                                                  # :67-69
                    journal.stopped_early(epoch)
                    break
            final_path = shelf.final()                                                 # :72-74
            journal.saved("final", final_path)
            journal.close(stopper.best, final_path)
    finally:
        torch.cuda.current_stream(dev).wait_stream(run_stream)
    return {"avg_loss": history, "best_loss": stopper.best, "log_path": journal.path, "captured": stepper.captured is not None, "captured_any_topology": stepper.padded is not None,
            "optimizer": optimizer}
