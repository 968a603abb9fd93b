#!/usr/bin/env python3
"""Runs only the c3 edge-processor MLP launch a few times (for rocprofv3 --pmc passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphnet_classifier_amd import native  # noqa: E402

dev = "cuda:0"
n, e, d = 1_000_000, 10_000_000, 64
mode = sys.argv[1] if len(sys.argv) > 1 else "wsplit"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
g = torch.Generator().manual_seed(0)
gsz = 160
goe = (torch.arange(e) // (e // (n // gsz))).clamp_(max=n // gsz - 1)
src = (goe * gsz + torch.randint(0, gsz, (e,), generator=g)).int().to(dev)
dst = torch.sort(goe * gsz + torch.randint(0, gsz, (e,), generator=g))[0].int().to(dev)
x = torch.randn(n, d, device=dev)
ea = torch.randn(e, d, device=dev)


def lin(o, i):
    return torch.randn(o, i, device=dev) / i ** 0.5, torch.randn(o, device=dev) * 0.1


(w0, b0), (w1, b1), (w2, b2) = lin(d, 3 * d), lin(d, d), lin(d, d)
ln = (torch.ones(d, device=dev), torch.zeros(d, device=dev), 1e-5)
for _ in range(iters):
    if mode == "wsplit":
        ps = native.mlp_forward([(x, None)], [w0[:, :d]], [None])
        pd = native.mlp_forward([(x, None)], [w0[:, d:2 * d]], [None])
        y = native.mlp_forward([(ps, src), (pd, dst), (ea, None)], [w0[:, 2 * d:], w1, w2], [b0, b1, b2], ln=ln,
                               residual=ea, modes=[1, 1, 0])
    elif mode == "concat":
        y = native.mlp_forward([(x, src), (x, dst), (ea, None)], [w0, w1, w2], [b0, b1, b2], ln=ln, residual=ea)
    elif mode == "enc":
        e0 = torch.randn(e, 3, device=dev)
        (v0, c0) = lin(d, 3)
        y = native.mlp_forward([(e0, None)], [v0, w1, w2], [c0, b1, b2], ln=ln)
torch.cuda.synchronize()
print("done", mode, float(y[0, 0]))
