"""Drop-in for the reference's ``models/MLP.py``: same constructor, same ``state_dict``
keys (``model.<i>.weight`` ...), forward executed by the fused HIP kernel K4.

Reference: models/MLP.py:5-47.
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch import Tensor

from . import functional as Fn
from . import native


def default_device() -> torch.device:
    """Device new modules are created on: the current ROCm device when one is visible (the
    reference never calls ``.to(device)``, so a drop-in has to place itself), else CPU -- on
    which only construction and state-dict handling work, never ``forward``."""
    if torch.cuda.is_available():
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def require_gpu_param(p: Tensor, what: str) -> torch.device:
    if not p.is_cuda:
        raise RuntimeError(
            f"{what}: parameters are on {p.device}; the forward path exists only as HIP kernels for the MI355X "
            f"(no CPU fallback).  Construct the module on a machine with a visible GPU or call .to('cuda').")
    return p.device


class MLP(nn.Module):
    def __init__(
        self,
        in_dim: int,
        out_dim: int,
        hidden_dim: int = 128,
        hidden_layers: int = 2,
        activation: str = "ReLU",
        initializer: None | str = None,
        norm_type: None | str = "LayerNorm",
    ):
        """Linear/act x hidden_layers, Linear, optional norm (models/MLP.py:24-37).

        The Sequential built here has exactly the reference's module order (Linear at even positions,
        the shared activation module between them, the norm last) so ``state_dict`` keys and the random
        initialisation under a given seed are the same."""
        super().__init__()
        self.activation_name = activation
        self.activation = getattr(nn, activation)()  # AttributeError for an unknown name, as in the reference
        if norm_type is not None:
            assert norm_type in ["LayerNorm", "BatchNorm1d"]  # models/MLP.py:30-33
        self.norm_type = norm_type
        widths = [in_dim] + [hidden_dim] * max(int(hidden_layers), 1) + [out_dim]
        chain = []
        for k in range(len(widths) - 1):
            chain.append(nn.Linear(widths[k], widths[k + 1]))
            if k + 2 < len(widths):
                chain.append(self.activation)
        if norm_type is not None:
            chain.append(getattr(nn, norm_type)(out_dim))
        self.model = nn.Sequential(*chain)
        if initializer is not None:
            self.initializer = getattr(nn.init, initializer)
            for tensor in self.model.parameters():
                if tensor.requires_grad and tensor.dim() > 1:
                    self.initializer(tensor)
        self.to(default_device())

    # -- pieces of the Sequential the kernel consumes ------------------------------------
    def _linears(self):
        return [m for m in self.model if isinstance(m, nn.Linear)]

    def _act_param(self) -> float:
        a = self.activation
        return float(getattr(a, "negative_slope", getattr(a, "alpha", 0.0)))

    def forward_segments(self, segments, residual: Tensor | None = None, rows: int | None = None) -> Tensor:
        """Run the MLP on the virtual concat of ``segments`` = [(table, int32 index | None)]
        (device tensors), optionally adding ``residual``: concat, gathers, Linear chain,
        LayerNorm and residual are one kernel launch."""
        lin = self._linears()
        require_gpu_param(lin[0].weight, "MLP")
        if self.activation_name not in native.ACTIVATIONS:
            raise NotImplementedError(f"activation nn.{self.activation_name} has no HIP kernel "
                                      f"(available: {sorted(native.ACTIVATIONS)})")
        norm = self.model[-1] if self.norm_type is not None else None
        ln = (norm.weight, norm.bias, norm.eps) if isinstance(norm, nn.LayerNorm) else None
        fuse_res = residual if not isinstance(norm, nn.BatchNorm1d) else None
        y = Fn.fused_mlp(segments, [m.weight for m in lin], [m.bias for m in lin], ln=ln,
                         activation=self.activation_name, act_param=self._act_param(), residual=fuse_res, rows=rows)
        if isinstance(norm, nn.BatchNorm1d):  # batch statistics span all rows: PyTorch-ROCm op on the GPU
            y = norm(y)
            if residual is not None:
                y = y + residual
        return y

    def forward_edge_features(self, pos: Tensor, src: Tensor, dst: Tensor) -> Tensor | None:
        """This MLP on the edge features of models/GNN.py:299-302 WITHOUT storing them (K6 as the launch's prologue); None
        when that form is not available (training, BatchNorm, a shape / batch size no kernel serves this way)."""
        lin = self._linears()
        if self.activation_name != "ReLU" or isinstance(self.model[-1], nn.BatchNorm1d):
            return None
        if torch.is_grad_enabled() and (pos.requires_grad or any(p.requires_grad for p in self.parameters())):
            return None  # the backward needs the feature rows (dW_0)
        norm = self.model[-1] if self.norm_type is not None else None
        ln = (norm.weight, norm.bias, norm.eps) if isinstance(norm, nn.LayerNorm) else None
        return native.mlp_forward_edge_features(pos, src, dst, [m.weight for m in lin], [m.bias for m in lin], ln=ln)

    def forward(self, x: Tensor):
        """models/MLP.py:45-47: flatten to (rows, -1), cast to float32, run the Sequential."""
        dev = require_gpu_param(self.model[0].weight, "MLP")
        src_device = x.device
        x = x.view(x.size(0), -1).to(device=dev, dtype=torch.float32)
        y = self.forward_segments([(x, None)])
        return y if src_device == dev else y.to(src_device)
