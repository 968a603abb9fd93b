#!/usr/bin/env python3
"""Replays the any-topology captured training step (CapturedTrainStep(edge_capacity=...)) on changing superpixel-like graphs;
run under `rocprofv3 --kernel-trace --stats` for the kernel nodes of one replay (python3 directly after `--`)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphnet_classifier_amd import synthetic  # noqa: E402
from graphnet_classifier_amd.GNN import CombinedModel, GraphNet  # noqa: E402
from graphnet_classifier_amd.train import CapturedTrainStep, FlatParameters, FusedAdam  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
fixed = len(sys.argv) > 2 and sys.argv[2] == "fixed"
graphs = [synthetic.superpixel_like_graphs(1, seed=2000 + k, shapes=((12, 12),)) for k in range(8)]
graphs = [(g.x.cuda(), g.pos.cuda(), g.edge_index.cuda()) for g in graphs]
model = CombinedModel(GraphNet(num_local_features=3, space_dim=2, out_channels=1, n_blocks=3), num_nodes=144, classes=2)
model.train()
opt = FusedAdam(FlatParameters(model), lr=1e-3)
loss_sum = torch.zeros((), dtype=torch.float64, device="cuda:0")
label = torch.tensor(1)
cap = max(int(e.size(1)) for _, _, e in graphs)
step = CapturedTrainStep(model, opt, torch.nn.CrossEntropyLoss(), graphs[0], label, loss_sum,
                         edge_capacity=None if fixed else (cap * 3 // 2 + 255) // 256 * 256)
torch.cuda.synchronize()
for k in range(reps):
    step(graphs[0 if fixed else k % 8], label)
torch.cuda.synchronize()
print("replays", reps, "fixed" if fixed else f"edge capacity {step.edge_capacity}", "loss_sum", float(loss_sum.item()))
