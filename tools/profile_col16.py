#!/usr/bin/env python3
"""Phase times of the small-batch forward kernel (mlp_col16.hip) on the default model's launch shapes at one ~1000-node graph.
Needs the probe build: `make -C graphnet_classifier_amd/csrc probe_col16`, GNC_LIB_PATH=build/libgnc_probe_col16.so.
Prints, per shape, the per-launch time (HIP events) and the median over waves of the cycles between the probe stamps."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphnet_classifier_amd import native  # noqa: E402

dev = "cuda:0"
N, E, D = 1024, 1984, 128
g = torch.Generator().manual_seed(0)


def lin(o, i):
    return torch.randn(o, i, device=dev) / i ** 0.5, torch.randn(o, device=dev) * 0.1


def mlp3(i, o=D):
    (w0, b0), (w1, b1), (w2, b2) = lin(D, i), lin(D, D), lin(o, D)
    return [w0, w1, w2], [b0, b1, b2]


ln = (torch.ones(D, device=dev), torch.zeros(D, device=dev), 1e-5)
x3 = torch.rand(N, 3, device=dev)
x = torch.randn(N, D, device=dev)
agg = torch.randn(N, D, device=dev)
e = torch.randn(E, D, device=dev)
src = torch.randint(0, N, (E,), generator=g).int().to(dev)
dst = torch.sort(torch.randint(0, N, (E,), generator=g))[0].int().to(dev)
ps, pd = torch.randn(N, D, device=dev), torch.randn(N, D, device=dev)
rowptr = torch.zeros(N + 1, dtype=torch.int32, device=dev)
rowptr[1:] = torch.cumsum(torch.bincount(dst.long(), minlength=N), 0).int()

wE, bE = mlp3(D)
wN, bN = mlp3(2 * D)
wX, bX = mlp3(3)
wD, bD = mlp3(D, 1)
wP, _ = lin(D, D)
shapes = {
    "node encoder  [N,3] -> 128 x3, LN": lambda: native.mlp_forward([(x3, None)], wX, bX, ln=ln, rows=N),
    "projection    [N,128] -> 128": lambda: native.mlp_forward([(x, None)], [wP], [None], rows=N),
    "edge W-split  e + ps[src] + pd[dst], LN, res": lambda: native.mlp_forward(
        [(ps, src), (pd, dst), (e, None)], wE, bE, ln=ln, residual=e, rows=E, modes=[native.SEG_ADD, native.SEG_ADD, native.SEG_MATMUL]),
    "edge W-split + aggregation": lambda: native.mlp_forward(
        [(ps, src), (pd, dst), (e, None)], wE, bE, ln=ln, residual=e, rows=E, modes=[native.SEG_ADD, native.SEG_ADD, native.SEG_MATMUL],
        aggregate=(dst, rowptr, N)),
    "node [x|agg] -> 128 x3, LN, res": lambda: native.mlp_forward([(x, None), (agg, None)], wN, bN, ln=ln, residual=x, rows=N),
    "decoder 128 -> 1": lambda: native.mlp_forward([(x, None)], wD, bD, rows=N),
}
lib = native.load_library()
have_probe = hasattr(lib, "gnc_col_probe_read")
names = ["prologue", "ids+rows", "barrier", "Linear 0", "Linears 1..", "LayerNorm", "store", "aggregate"]
for name, fn in shapes.items():
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(50):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    t = sorted(p.elapsed_time(q) for p, q in evs)
    line = f"{name:48s} launch {t[len(t) // 2] * 1e3:6.1f} us (min {t[0] * 1e3:5.1f})"
    if have_probe:
        rows = E if "edge" in name else N
        nwg = (rows + 15) // 16
        buf = np.zeros(1024 * 8 * 12, dtype=np.uint64)
        lib.gnc_col_probe_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        lib.gnc_col_probe_read(buf.ctypes.data, buf.nbytes)
        p = buf.reshape(1024, 8, 12)[:nwg].astype(np.int64)
        dl = np.diff(p[:, :, :9], axis=2).reshape(-1, 8)
        med = np.median(dl, axis=0)
        tot = np.median(p[:, :, 8] - p[:, :, 0])
        line += "  cycles: " + " ".join(f"{n}={int(m)}" for n, m in zip(names, med)) + f"  total={int(tot)}"
    print(line, flush=True)
