#!/usr/bin/env python3
"""Host-side cost of the reference's own regime - ONE small graph per optimizer step, a NEW topology every step
(superpixel graphs), default D = 128 model: wall time per eager training step and a cProfile of the loop."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphnet_classifier_amd import native, synthetic  # noqa: E402
from graphnet_classifier_amd.GNN import CombinedModel, GraphNet  # noqa: E402
from graphnet_classifier_amd.train import FlatParameters, FusedAdam  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
native.load_library()
graphs = [synthetic.superpixel_like_graphs(1, seed=100 + k, shapes=((12, 12),)) for k in range(8)]
graphs = [(g.x.cuda(), g.pos.cuda(), g.edge_index.cuda()) for g in graphs]  # device-resident samples (the graph builders' output)
model = CombinedModel(GraphNet(num_local_features=3, space_dim=2, out_channels=1, n_blocks=3), num_nodes=144, classes=2)
model.train()
opt = FusedAdam(FlatParameters(model), lr=1e-3)
crit = torch.nn.CrossEntropyLoss()
label = torch.tensor(1, device="cuda:0")


def step(k):
    g = graphs[k % len(graphs)]
    opt.zero_grad()
    loss = crit(model(g), label)
    loss.backward()
    opt.step()


for k in range(10):
    step(k)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(steps):
    step(k)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"eager training step, new topology every step: host {t_host / steps * 1e3:.3f} ms, with the final sync {(time.perf_counter() - t0) / steps * 1e3:.3f} ms")
pr = cProfile.Profile()
pr.enable()
for k in range(steps):
    step(k)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
