#!/usr/bin/env python3
"""Host-side cost of one forward step (CSR build + GraphNet.forward) at a given scale of c3: wall time of the enqueue
loop without a device sync inside it, and a cProfile of the same loop (which Python frames the launches spend their
time in).  `python tools/host_profile.py [scale] [steps]`"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphnet_classifier_amd import native, synthetic, topology  # noqa: E402
from graphnet_classifier_amd.GNN import GraphNet  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.125
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
native.load_library()
topology.set_validation("deferred")
batch, kw = synthetic.make_workload("c3", scale)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = GraphNet(**kw).to(dev).eval()
x, pos, ei = batch.x.to(dev), batch.pos.to(dev), batch.edge_index.to(dev)


def step():
    topology.clear_topology_cache()
    with torch.no_grad():
        return model(x, pos, ei)


for _ in range(10):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"scale {scale}: host enqueue {t_host / steps * 1e3:.3f} ms per step, with the final sync {t_all / steps * 1e3:.3f} ms per step")
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
