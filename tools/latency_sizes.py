#!/usr/bin/env python3
"""Per-sample latency against graph size (pixel graphs R x R, default D = 128 model): replayed forward and captured training
step.  GNC_COL16_MAX_ROWS=<rows> moves the small-batch limit (0 = the throughput kernels everywhere)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphnet_classifier_amd.GNN import CapturedForward, CombinedModel, GraphNet  # noqa: E402
from graphnet_classifier_amd.image_to_graph import create_grid_edges_optimized  # noqa: E402
from graphnet_classifier_amd.train import CapturedTrainStep, FlatParameters, FusedAdam  # noqa: E402

sizes = [int(v) for v in sys.argv[1:]] or [16, 32, 64, 90, 96, 128, 192]
for R in sizes:
    torch.manual_seed(0)
    ei = create_grid_edges_optimized(R, R).cpu()
    rr, cc = np.meshgrid(np.arange(R), np.arange(R), indexing="ij")
    x = torch.rand(R * R, 3) * 255
    pos = torch.from_numpy(np.stack([rr.ravel(), cc.ravel()], 1).astype(np.float32))
    model = CombinedModel(GraphNet(num_local_features=3, space_dim=2, out_channels=1, n_blocks=3), num_nodes=R * R, classes=2).eval()
    cap = CapturedForward(model, x, pos, ei)
    xd = x.to("cuda:0")
    for _ in range(20):
        cap(xd)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        cap(xd)
    torch.cuda.synchronize()
    fwd = (time.perf_counter() - t0) / 200 * 1e3
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        model.train()
        opt = FusedAdam(FlatParameters(model), lr=1e-3)
        loss_sum = torch.zeros((), dtype=torch.float64, device="cuda:0")
        step = CapturedTrainStep(model, opt, torch.nn.CrossEntropyLoss(), (x, pos, ei), torch.tensor(1), loss_sum)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            step((x, pos, ei), torch.tensor(1))
        torch.cuda.synchronize()
        trn = (time.perf_counter() - t0) / 50 * 1e3
    print(f"R={R} N={R * R} E={ei.size(1)}: replayed forward {fwd:.3f} ms, captured training step {trn:.3f} ms", flush=True)
