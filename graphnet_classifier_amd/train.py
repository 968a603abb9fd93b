"""Training loop of the reference (``utils/train_model.py:8-81``) on the HIP path - SURVEY.md section 8, row f3.

``train(model, dataset, epochs, patience=5, output_path='weights', start_weights=None)`` has the reference's
signature and observable behaviour: Adam(lr=1e-3) (:9), CrossEntropyLoss (:10), one optimizer step per sample in
dataset order (:35-45), average training loss per epoch, early stopping on it (:57-69), ``best_model_epoch{k}.pth``
/ ``final_model.pth`` (:61-62, :72-73) and the timestamped log file with the reference's line formats (:22-30,
:52-54, :76-80).  What differs is how a step runs:

* parameters and gradients are views of two flat fp32 buffers (``FlatParameters``), the optimizer is ONE fused Adam
  launch over them (``FusedAdam`` -> ``gnc_adam_step_f32``) instead of a walk over 76 tensors;
* the running loss is accumulated on the device (float64 scalar) and read once per epoch, where the reference
  synchronises on ``loss.item()`` after every sample (:44);
* when consecutive samples share one topology (pixel / patch graphs: the edge list depends on the image size only,
  utils/image_to_graph/image_to_graph_optimized.py:42-47) the whole step - forward, loss, backward, gradient pack,
  Adam - is captured into a hipGraph once and replayed per sample (``CapturedTrainStep``): one launch per step
  instead of ~150.

Saved ``.pth`` files hold CPU tensors under the reference's state-dict keys, so they load in the reference as they
are (``utils/inference.py:40-45``) and vice versa.
"""
from __future__ import annotations

import os
import time
from datetime import datetime

import torch
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
        self.params = [p for p in module.parameters() if p.requires_grad]
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

    def step(self, reduce: bool = True) -> None:
        """Gradient pack (+ the one all-reduce when a process group with more than one rank is up) and the update."""
        if reduce:
            self.fp.reducer()
        native.adam_step(self.fp.flat, self.fp.grad, self.exp_avg, self.exp_avg_sq, self.step_count, self._scratch, self.lr,
                         self.betas[0], self.betas[1], self.eps, self.weight_decay)

    def state_snapshot(self):
        return [t.clone() for t in (self.fp.flat, self.exp_avg, self.exp_avg_sq, self.step_count)]

    def state_restore(self, snap) -> None:
        for dst, src in zip((self.fp.flat, self.exp_avg, self.exp_avg_sq, self.step_count), snap):
            dst.copy_(src)


def _same_topology(a: torch.Tensor, b: torch.Tensor) -> bool:
    return a is b or (a.shape == b.shape and a.dtype == b.dtype and a.device == b.device and bool(torch.equal(a, b)))


class CapturedTrainStep:
    """One training step (utils/train_model.py:37-42: forward, CE loss, zero_grad, backward, Adam) for ONE fixed
    topology, captured into a hipGraph and replayed per sample.  The loss of every replay is added to ``loss_sum``
    (device float64) inside the graph.  Warm-up steps run before the capture; parameters and optimizer state are
    restored afterwards, so constructing this object does not train."""

    def __init__(self, model: nn.Module, optimizer: FusedAdam, criterion, sample, label, loss_sum: torch.Tensor):
        x, pos, edge_index = sample
        dev = require_gpu_param(next(model.parameters()), "CapturedTrainStep")
        self.edge_index_host = edge_index
        self.x = x.to(device=dev, dtype=torch.float32).clone()
        self.pos = pos.to(device=dev, dtype=torch.float32).clone()
        self.edge_index = edge_index.to(dev)
        self.label = torch.as_tensor(label).to(dev).clone()
        self.loss_sum = loss_sum
        from .topology import get_topology
        # the build's host sync happens here, outside the capture; the reference keeps the CSR arrays alive for the graph
        self.topo = get_topology(self.edge_index, self.x.size(0), dev)

        def one_step():
            logits = model((self.x, self.pos, self.edge_index))
            loss = criterion(logits, self.label)
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
            self.loss_sum.add_(loss.detach().double())

        snap = optimizer.state_snapshot()
        keep = self.loss_sum.clone()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(3):  # kernel attributes, allocator pools, lazily built source-sorted CSR
                one_step()
        torch.cuda.current_stream(dev).wait_stream(side)
        optimizer.zero_grad()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            one_step()
        optimizer.state_restore(snap)
        self.loss_sum.copy_(keep)

    def matches(self, sample) -> bool:
        x, pos, edge_index = sample
        return x.shape == self.x.shape and pos.shape == self.pos.shape and _same_topology(edge_index, self.edge_index_host)

    def __call__(self, sample, label) -> None:
        x, pos, _ = sample
        self.x.copy_(x, non_blocking=True)
        self.pos.copy_(pos, non_blocking=True)
        self.label.copy_(torch.as_tensor(label), non_blocking=True)
        self.graph.replay()


def train(model, dataset, epochs, patience=5, output_path='weights', start_weights=None, *, capture: bool = True, lr: float = 1e-3):
    """utils/train_model.py:8-81 (same positional arguments, files and log lines).  Returns a dict with the per-epoch
    average losses (the reference returns None; nothing in it reads the return value)."""
    if start_weights:
        model.load_state_dict(torch.load(start_weights, map_location="cpu"))           # :14-15
    dev = require_gpu_param(next(model.parameters()), "train")
    flat = FlatParameters(model)
    optimizer = FusedAdam(flat, lr=lr)                                                 # :9
    criterion = nn.CrossEntropyLoss()                                                  # :10

    os.makedirs(output_path, exist_ok=True)                                            # :18
    print(f"Training model in {output_path}")
    timestamp = datetime.now().strftime("%Y%m%d_%H%M%S")
    log_path = os.path.join(output_path, f'training_logs_{timestamp}.txt')
    with open(log_path, "w") as the_file:                                              # :26-30
        the_file.write(f"Training started at: {datetime.now().strftime('%Y-%m-%d %H:%M:%S')}\n")
        the_file.write(f"Epochs: {epochs}, Patience: {patience}\n")
        the_file.write(f"Output path: {output_path}\n")
        the_file.write("-" * 50 + "\n")

    def save(path):
        torch.save({k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, path)

    loss_sum = torch.zeros((), dtype=torch.float64, device=dev)
    history = []
    # The whole loop runs on a side stream.  A hipGraph capture that follows eager autograd steps issued on the
    # legacy default stream takes the HIP runtime down at hipStreamEndCapture (reproduced: tools/repro_capture2.py,
    # MODE=eager_first segfaults, MODE=eager_side does not); PyTorch's whole-network capture recipe asks for a
    # side-stream warm-up for the same reason.  Nothing here depends on the default stream.
    run_stream = torch.cuda.Stream(device=dev)
    run_stream.wait_stream(torch.cuda.current_stream(dev))
    try:
        with torch.cuda.stream(run_stream):
            return _train_loop(model, dataset, epochs, patience, output_path, dev, optimizer, criterion, capture, log_path, save,
                               loss_sum, history)
    finally:
        torch.cuda.current_stream(dev).wait_stream(run_stream)


def _train_loop(model, dataset, epochs, patience, output_path, dev, optimizer, criterion, capture, log_path, save, loss_sum, history):
    best_loss = float('inf')
    patience_counter = 0
    captured: CapturedTrainStep | None = None
    prev_sample = None
    for epoch in range(epochs):
        checkpoint1 = time.time()
        loss_sum.zero_()
        num_batches = 0
        for sample, label in dataset:                                                  # :35
            graph_sample = isinstance(sample, (tuple, list)) and len(sample) == 3
            if capture and graph_sample:
                if captured is None and prev_sample is not None and prev_sample[0].shape == sample[0].shape \
                        and _same_topology(prev_sample[2], sample[2]):
                    captured = CapturedTrainStep(model, optimizer, criterion, sample, label, loss_sum)
                if captured is not None and captured.matches(sample):
                    captured(sample, label)
                    num_batches += 1
                    continue
                prev_sample = sample
            if graph_sample:
                # edge_index stays where it is: a host tensor is looked up in the topology cache by content, so equal
                # topologies are sorted once, not once per sample
                sample = (sample[0].to(dev, non_blocking=True), sample[1].to(dev, non_blocking=True), sample[2])
            else:
                sample = sample.to(dev, non_blocking=True)
            logits = model(sample)                                                     # :37
            loss = criterion(logits, torch.as_tensor(label).to(dev, non_blocking=True))  # :38
            optimizer.zero_grad()                                                      # :40
            loss.backward()                                                            # :41
            optimizer.step()                                                           # :42
            loss_sum += loss.detach().double()                                         # :44, without the per-sample sync
            num_batches += 1

        avg_loss = float(loss_sum.item()) / max(1, num_batches)                        # :47 (the epoch's one host sync)
        history.append(avg_loss)
        print(f"Epoch {epoch+1}/{epochs}, avg_loss={avg_loss:.4f}")
        checkpoint2 = time.time()
        print(f"epoch: {epoch + 1} needed {checkpoint2 - checkpoint1} time")
        with open(log_path, "a") as the_file:                                          # :52-54
            the_file.write(f"Epoch {epoch+1}/{epochs}, avg_loss={avg_loss:.4f}\n")
            the_file.write(f"Epoch {epoch+1}/{epochs}, needed {(checkpoint2 - checkpoint1) / 60:.2f} minutes\n")

        if avg_loss < best_loss:                                                       # :57-66
            best_loss = avg_loss
            patience_counter = 0
            best_model_path = os.path.join(output_path, f'best_model_epoch{epoch+1}.pth')
            save(best_model_path)
            print(f"Saved best model: {best_model_path}")
        else:
            patience_counter += 1
        if patience_counter >= patience:                                               # :67-69
            print(f"Early stopping at epoch {epoch+1}")
            break

    final_model_path = os.path.join(output_path, 'final_model.pth')                    # :72-74
    save(final_model_path)
    print(f"Saved final model: {final_model_path}")
    with open(log_path, "a") as the_file:                                              # :76-80
        the_file.write("-" * 50 + "\n")
        the_file.write(f"Training completed at: {datetime.now().strftime('%Y-%m-%d %H:%M:%S')}\n")
        the_file.write(f"Best loss achieved: {best_loss:.4f}\n")
        the_file.write(f"Final model saved: {final_model_path}\n")
    return {"avg_loss": history, "best_loss": best_loss, "log_path": log_path, "captured": captured is not None,
            "optimizer": optimizer}
