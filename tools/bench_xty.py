#!/usr/bin/env python3
"""Times native.xty (dW = dz^T a) on one shape: `python tools/bench_xty.py ROWS M K [iters]`; prints ms per call (all
block launches + partial sums), TFLOP/s and the HBM rate of reading both operands once."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from graphnet_classifier_amd import native

rows, m, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
native.load_library()
a = torch.randn(rows, m, device="cuda:0")
b = torch.randn(rows, k, device="cuda:0")
for _ in range(3):
    native.xty(a, b)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    native.xty(a, b)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f"xty rows={rows} M={m} K={k}: {ms:.3f} ms  {2.0 * rows * m * k / ms / 1e9:.1f} TFLOP/s  "
      f"{4.0 * rows * (m + k) / ms / 1e6:.0f} GB/s (operands once)")
