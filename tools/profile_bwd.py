#!/usr/bin/env python3
"""Runs the c3 edge-processor BACKWARD launch (fused K8 kernel) a few times and, with the probe build
(`make -C graphnet_classifier_amd/csrc probe_bwd`, GNC_LIB_PATH=build/libgnc_probe_bwd.so), prints the cycles per
phase of its tile loop."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphnet_classifier_amd import native  # noqa: E402

dev = "cuda:0"
n, e, d = 1_000_000, 10_000_000, 64
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4
# second argument: 0 = row-ordered grad_out (default), 1 = grad_out + gathered part (ABI 16 grad_gather), 2 = gathered part alone
gg_mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
# third argument: 1 = the forward's saved post-activations are read (ABI 16 act_given) instead of being recomputed
saved = int(sys.argv[3]) if len(sys.argv) > 3 else 0
g = torch.Generator().manual_seed(0)
gsz = 160
goe = (torch.arange(e) // (e // (n // gsz))).clamp_(max=n // gsz - 1)
src = (goe * gsz + torch.randint(0, gsz, (e,), generator=g)).int().to(dev)
dst = torch.sort(goe * gsz + torch.randint(0, gsz, (e,), generator=g))[0].int().to(dev)
ps, pd = torch.randn(n, d, device=dev), torch.randn(n, d, device=dev)
ea = torch.randn(e, d, device=dev)
gout = torch.randn(e, d, device=dev)
gagg = torch.randn(n, d, device=dev)


def lin(o, i):
    return torch.randn(o, i, device=dev) / i ** 0.5, torch.randn(o, device=dev) * 0.1


(w0, b0), (w1, b1), (w2, b2) = lin(d, d), lin(d, d), lin(d, d)
ln = (torch.ones(d, device=dev), torch.zeros(d, device=dev), 1e-5)
segs = [(ps, src), (pd, dst), (ea, None)]
modes = [native.SEG_ADD, native.SEG_ADD, native.SEG_MATMUL]
acts = []
if saved:
    native.mlp_forward(segs, [w0, w1, w2], [b0, b1, b2], ln=ln, residual=ea, rows=e, modes=modes, save_act=acts)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(iters):
    if it == 1:
        ev0.record()
    r = native.mlp_backward(segs, [w0, w1, w2], [b0, b1, b2], ln, None if gg_mode == 2 else gout, rows=e, modes=modes, need_dx=True,
                            residual=ea, grad_gather=(gagg, dst) if gg_mode else None, saved_act=acts if saved else None)
ev1.record()
torch.cuda.synchronize()
print("saved_act", r["saved_act_used"], "gg_mode", gg_mode, "ms_per_launch", ev0.elapsed_time(ev1) / (iters - 1), "fused" if "dw" in r else "split",
      "gathered in the launch" if (gg_mode and r["grad_out"] is None) else "")
lib = native.load_library()
reader = "gnc_phase_probe_bwd_sv_read" if saved else "gnc_phase_probe_bwd_read"
if hasattr(lib, reader):
    waves = 1024
    buf = np.zeros(4096 * 12, dtype=np.uint64)
    getattr(lib, reader).argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    getattr(lib, reader)(buf.ctypes.data, buf.nbytes)
    b = buf.reshape(4096, 12)[:waves].astype(np.float64)
    tiles = (e + 31) // 32 / waves
    names = ["L0+add+relu", "L1+relu", "L2+LN bwd+sums", "transposes+dW2", "da1+mask", "loads+layer1", "gathers+layer0", "collect+stores+park"]
    print("cycles per tile and wave:", {n: round(v / tiles) for n, v in zip(names, b[:, :8].mean(0))}, "total", round(b[:, 8].mean() / tiles))
    print("shader clock GHz during the kernel:", round(float((b[:, 8] / b[:, 9]).mean()) * 0.1, 3))
