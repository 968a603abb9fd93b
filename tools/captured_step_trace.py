#!/usr/bin/env python3
"""Replays the captured per-sample training step (pixel R = 32, default D = 128 model) N times: run it under
`rocprofv3 --kernel-trace --stats` to see what the ~3.9 ms of a replay are made of."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphnet_classifier_amd.GNN import CombinedModel, GraphNet  # noqa: E402
from graphnet_classifier_amd.image_to_graph import create_grid_edges_optimized  # noqa: E402
from graphnet_classifier_amd.train import CapturedTrainStep, FlatParameters, FusedAdam  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
R = int(sys.argv[2]) if len(sys.argv) > 2 else 32  # pixel graph R x R (128 = the reference's default image size, main.py)
torch.manual_seed(0)
ei = create_grid_edges_optimized(R, R).cpu()
rr, cc = np.meshgrid(np.arange(R), np.arange(R), indexing="ij")
x = torch.rand(R * R, 3) * 255
pos = torch.from_numpy(np.stack([rr.ravel(), cc.ravel()], 1).astype(np.float32))
label = torch.tensor(1)
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    model = CombinedModel(GraphNet(num_local_features=3, space_dim=2, out_channels=1, n_blocks=3), num_nodes=R * R, classes=2)
    model.train()
    opt = FusedAdam(FlatParameters(model), lr=1e-3)
    loss_sum = torch.zeros((), dtype=torch.float64, device="cuda:0")
    step = CapturedTrainStep(model, opt, torch.nn.CrossEntropyLoss(), (x, pos, ei), label, loss_sum)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step((x, pos, ei), label)
    torch.cuda.synchronize()
    print("ms per replayed step", (time.perf_counter() - t0) / n * 1e3)
