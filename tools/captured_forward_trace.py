#!/usr/bin/env python3
"""Replays the captured per-sample forward (pixel R = 32, default D = 128 model) N times: run it under
`rocprofv3 --kernel-trace --stats` (tools/prof_kernels.sh) to see what a replay is made of."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphnet_classifier_amd.GNN import CapturedForward, CombinedModel, GraphNet  # noqa: E402
from graphnet_classifier_amd.image_to_graph import create_grid_edges_optimized  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
torch.manual_seed(0)
ei = create_grid_edges_optimized(32, 32).cpu()
rr, cc = np.meshgrid(np.arange(32), np.arange(32), indexing="ij")
x = torch.rand(1024, 3) * 255
pos = torch.from_numpy(np.stack([rr.ravel(), cc.ravel()], 1).astype(np.float32))
model = CombinedModel(GraphNet(num_local_features=3, space_dim=2, out_channels=1, n_blocks=3), num_nodes=1024, classes=2).eval()
cap = CapturedForward(model, x, pos, ei)
xd = x.to("cuda:0")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    cap(xd)
torch.cuda.synchronize()
print("ms per replayed forward", (time.perf_counter() - t0) / n * 1e3)
