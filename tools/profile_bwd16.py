#!/usr/bin/env python3
"""Runs the c5 edge-processor BACKWARD launch (16-row streamed K8 kernel, widths 256, saved post-activations) a few times
and, with the probe build (`make -C graphnet_classifier_amd/csrc probe_b16`, GNC_LIB_PATH=build/libgnc_probe_b16.so), prints
the cycles per phase of its tile loop.  argv: [iters] [width] [rows]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphnet_classifier_amd import native  # noqa: E402

dev = "cuda:0"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6
d = int(sys.argv[2]) if len(sys.argv) > 2 else 256
e = int(sys.argv[3]) if len(sys.argv) > 3 else 5_000_000
n = e // 10
g = torch.Generator().manual_seed(0)
gsz = 160
goe = (torch.arange(e) // (e // (n // gsz))).clamp_(max=n // gsz - 1)
src = (goe * gsz + torch.randint(0, gsz, (e,), generator=g)).int().to(dev)
dst = torch.sort(goe * gsz + torch.randint(0, gsz, (e,), generator=g))[0].int().to(dev)
ps, pd = torch.randn(n, d, device=dev), torch.randn(n, d, device=dev)
ea = torch.randn(e, d, device=dev)
gout = torch.randn(e, d, device=dev)


def lin(o, i):
    return torch.randn(o, i, device=dev) / i ** 0.5, torch.randn(o, device=dev) * 0.1


(w0, b0), (w1, b1), (w2, b2) = lin(d, d), lin(d, d), lin(d, d)
ln = (torch.ones(d, device=dev), torch.zeros(d, device=dev), 1e-5)
segs = [(ps, src), (pd, dst), (ea, None)]
modes = [native.SEG_ADD, native.SEG_ADD, native.SEG_MATMUL]
acts = []
native.mlp_forward(segs, [w0, w1, w2], [b0, b1, b2], ln=ln, residual=ea, rows=e, modes=modes, save_act=acts)
def launch():
    return native.mlp_backward(segs, [w0, w1, w2], [b0, b1, b2], ln, gout, rows=e, modes=modes, need_dx=True, residual=ea, saved_act=acts)


# warm-up until the launch time settles (a fresh box can take a second to reach its clocks), then per-launch HIP events
import time
t_end = time.time() + 1.5
while time.time() < t_end:
    r = launch()
    torch.cuda.synchronize()
evs = []
for it in range(iters):
    a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    r = launch()
    b_.record()
    evs.append((a, b_))
torch.cuda.synchronize()
times = sorted(x.elapsed_time(y) for x, y in evs)
ms = times[len(times) // 2]
print("per-launch ms:", [round(t, 2) for t in times])
flops = 2.0 * e * 4 * d * d  # LayerNorm recompute + two transposed products + dx
print(f"width {d} rows {e}: saved_act {r['saved_act_used']} ms_per_launch {ms:.3f} -> {flops / ms / 1e9:.1f} TFLOP/s executed "
      f"({flops / ms / 1e9 / 157.3:.3f} of the fp32 MFMA peak)")
lib = native.load_library()
reader = "gnc_phase_probe_b16_read" if d > 128 else "gnc_phase_probe_b32_read"
if hasattr(lib, reader):
    waves = 256 * 8
    buf = np.zeros(4096 * 12, dtype=np.uint64)
    getattr(lib, reader).argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    getattr(lib, reader)(buf.ctypes.data, buf.nbytes)
    b = buf.reshape(4096, 12)[:waves].astype(np.float64)
    tiles = (e + (15 if d > 128 else 31)) // (16 if d > 128 else 32) / waves
    names = (["saved tiles -> masks", "grad_out slabs", "last Linear (LN stats)", "LN backward + sums", "emit dz_last", "W^T products + masks + emits", "dx", "-"]
             if d > 128 else ["saved tiles -> masks", "last Linear (LN stats)", "grad_out slabs", "LN backward + sums", "emit dz_last", "W^T products + masks + emits", "dx", "-"])
    print("cycles per tile and wave:", {k: round(v / tiles) for k, v in zip(names, b[:, :8].mean(0))}, "total", round(b[:, 8].mean() / tiles),
          "(MFMA alone: 4096 x 32 = 131072 per tile and wave pair -> 2 waves per SIMD)")
    print("shader clock GHz during the kernel:", round(float((b[:, 8] / b[:, 9]).mean()) * 0.1, 3))
