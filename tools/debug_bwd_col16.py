#!/usr/bin/env python3
"""Small-batch K8 data kernel (mlp_bwd_col16.hip) against float64 formulas on the SAME saved activations (no ReLU-side
ambiguity): argv = d rows nadd(0|2)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphnet_classifier_amd import native  # noqa: E402

dev = "cuda:0"
d = int(sys.argv[1]) if len(sys.argv) > 1 else 100
e = int(sys.argv[2]) if len(sys.argv) > 2 else 999
nadd = int(sys.argv[3]) if len(sys.argv) > 3 else 0
n = 301
g = torch.Generator().manual_seed(1)
r = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
ws = [(r(d, d) / d ** 0.5).to(dev) for _ in range(3)]
bs = [(r(d) * 0.1).to(dev) for _ in range(3)]
ln = ((1 + 0.1 * r(d)).to(dev), (0.1 * r(d)).to(dev), 1e-5)
ea = r(e, d).to(dev)
src = torch.randint(0, n, (e,), generator=g).int().to(dev)
dst = torch.sort(torch.randint(0, n, (e,), generator=g))[0].int().to(dev)
ps, pd_ = r(n, d).to(dev), r(n, d).to(dev)
if nadd:
    segs, modes = [(ps, src), (pd_, dst), (ea, None)], [native.SEG_ADD, native.SEG_ADD, native.SEG_MATMUL]
else:
    segs, modes = [(ea, None)], None
acts = []
out = native.mlp_forward(segs, ws, bs, ln=ln, residual=ea, rows=e, modes=modes, save_act=acts)
gout = r(e, d).to(dev)
res = native.mlp_backward(segs, ws, bs, ln, gout, rows=e, modes=modes, need_dx=True, residual=ea, saved_act=acts)
torch.cuda.synchronize()
print("saved_act_used", res["saved_act_used"], "residual_folded", res["residual_folded"])
D = lambda t: t.double().cpu()  # noqa: E731
a0, a1 = D(acts[0]), D(acts[1])
z2 = a1 @ D(ws[2]).t() + D(bs[2])
mean = z2.mean(1, keepdim=True)
var = ((z2 - mean) ** 2).mean(1, keepdim=True)
rstd = 1 / torch.sqrt(var + 1e-5)
yh = (z2 - mean) * rstd
gg = D(gout)
dy = gg * D(ln[0])
dz2 = rstd * (dy - dy.mean(1, keepdim=True) - yh * (dy * yh).mean(1, keepdim=True))
dz1 = (dz2 @ D(ws[2])) * (a1 > 0)
dz0 = (dz1 @ D(ws[1])) * (a0 > 0)
dx = dz0 @ D(ws[0]) + (gg if res["residual_folded"] else 0)
for name, got, want in (("dz2", res["dz"][2], dz2), ("dz1", res["dz"][1], dz1), ("dz0", res["dz"][0], dz0), ("dx", res["dx"], dx)):
    err = (D(got) - want).abs()
    bad = (err.amax(1) > 1e-4).nonzero().flatten()
    print(name, "max err", float(err.max()), "bad rows", bad[:12].tolist(), "of", len(bad))
sb, sg = res["ln_sums"]
print("d beta err", float((D(sb) - gg.sum(0)).abs().max()), "d gamma err", float((D(sg) - (gg * yh).sum(0)).abs().max()))
