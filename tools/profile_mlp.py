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
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(iters):
    if it == 1:
        ev0.record()
    if mode == "wsplit":
        ps = native.mlp_forward([(x, None)], [w0[:, :d]], [None])
        pd = native.mlp_forward([(x, None)], [w0[:, d:2 * d]], [None])
        y = native.mlp_forward([(ps, src), (pd, dst), (ea, None)], [w0[:, 2 * d:], w1, w2], [b0, b1, b2], ln=ln,
                               residual=ea, modes=[1, 1, 0])
    elif mode == "wsplit_agg":  # the same launch with the fused aggregation epilogue (dst is sorted above)
        ps = native.mlp_forward([(x, None)], [w0[:, :d]], [None])
        pd = native.mlp_forward([(x, None)], [w0[:, d:2 * d]], [None])
        if it == 0:
            rowptr = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev),
                                torch.cumsum(torch.bincount(dst.long(), minlength=n), 0)]).int()
        y, agg = native.mlp_forward([(ps, src), (pd, dst), (ea, None)], [w0[:, 2 * d:], w1, w2], [b0, b1, b2], ln=ln,
                                    residual=ea, modes=[1, 1, 0], aggregate=(dst, rowptr, n))
    elif mode == "concat":
        y = native.mlp_forward([(x, src), (x, dst), (ea, None)], [w0, w1, w2], [b0, b1, b2], ln=ln, residual=ea)
    elif mode == "enc":
        e0 = torch.randn(e, 3, device=dev)
        (v0, c0) = lin(d, 3)
        y = native.mlp_forward([(e0, None)], [v0, w1, w2], [c0, b1, b2], ln=ln)
ev1.record()
torch.cuda.synchronize()
if iters > 1:
    print("ms_per_iter", mode, ev0.elapsed_time(ev1) / (iters - 1))
print("done", mode, float(y[0, 0]))

# phase probe of the last resident-kernel launch (library built with `make PROBE=1` only)
import ctypes  # noqa: E402

import numpy as np  # noqa: E402

lib = native.load_library()
if hasattr(lib, "gnc_phase_probe_read"):
    buf = np.zeros(2048 * 12, dtype=np.uint64)
    lib.gnc_phase_probe_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    lib.gnc_phase_probe_read(buf.ctypes.data, buf.nbytes)
    b = buf.reshape(2048, 12).astype(np.float64)
    tiles = (e + 31) // 32 / 2048
    names = ["wait+stage", "L0 mfma", "add step", "hidden", "last", "LN+transpose", "epilogue", "load issue"]
    print("cycles per tile and wave:", {n: round(v / tiles) for n, v in zip(names, b[:, :8].mean(0))},
          "total", round(b[:, 8].mean() / tiles))
    print("shader clock GHz during the kernel:", round(float((b[:, 8] / b[:, 9]).mean()) * 0.1, 3))
