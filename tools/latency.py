#!/usr/bin/env python3
"""Per-graph latency of the default model (main.py:72: D=128, 3 blocks) the way the reference drives it:
one graph per call, host tensors in, logits out (utils/train_model.py:35-45)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphnet_classifier_amd import synthetic  # noqa: E402
from graphnet_classifier_amd.GNN import CombinedModel, GraphNet  # noqa: E402
from graphnet_classifier_amd.image_to_graph import create_grid_edges_optimized  # noqa: E402

torch.manual_seed(0)
cases = {}
ei = create_grid_edges_optimized(32, 32).cpu()
rr, cc = np.meshgrid(np.arange(32), np.arange(32), indexing="ij")
cases["pixel R=32 (N=1024, E=1984)"] = (torch.rand(1024, 3) * 255, torch.from_numpy(np.stack([rr.ravel(), cc.ravel()], 1).astype(np.float32)), ei)
b = synthetic.superpixel_like_graphs(1, seed=1000, shapes=((12, 12),))
cases["superpixel-like (N=144, E~790)"] = (b.x, b.pos, b.edge_index)
for name, (x, pos, ei) in cases.items():
    model = CombinedModel(GraphNet(num_local_features=3, space_dim=2, out_channels=1, n_blocks=3), num_nodes=x.size(0), classes=2)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    crit = torch.nn.CrossEntropyLoss()
    label = torch.tensor(1)
    from graphnet_classifier_amd.GNN import CapturedForward
    cap = CapturedForward(model.eval(), x, pos, ei)
    xd = x.to("cuda:0")
    ts = []
    for it in range(50):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = cap(xd)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"{name:36s} hipGraph replay (device input) median {1e3*np.median(ts[5:]):7.3f} ms", flush=True)
    # the reference's regime: ONE graph per optimizer step (main.py:60).  eager = torch.optim.Adam over 76 tensors and a
    # loss.item() per step (utils/train_model.py:37-44); fused = flat parameters + one fused Adam launch, loss kept on
    # the device; captured = the whole step (forward + CE + backward + gradient pack + Adam) replayed from one hipGraph
    from graphnet_classifier_amd.train import CapturedTrainStep, FlatParameters, FusedAdam
    tmodel = CombinedModel(GraphNet(num_local_features=3, space_dim=2, out_channels=1, n_blocks=3), num_nodes=x.size(0), classes=2)
    tmodel.train()
    fopt = FusedAdam(FlatParameters(tmodel), lr=1e-3)
    loss_sum = torch.zeros((), dtype=torch.float64, device="cuda:0")
    cap_step = CapturedTrainStep(tmodel, fopt, crit, (x, pos, ei), label, loss_sum)
    xdev, posdev, labdev = x.to("cuda:0"), pos.to("cuda:0"), label.to("cuda:0")
    for mode in ("fused", "captured"):
        ts = []
        for it in range(60):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            if mode == "fused":
                logits = tmodel((xdev, posdev, ei))
                loss = crit(logits, labdev)
                fopt.zero_grad(); loss.backward(); fopt.step()
                loss_sum += loss.detach().double()
            else:
                cap_step((x, pos, ei), label)  # host tensors in, as the reference's loader yields them
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(f"{name:36s} train/{mode:8s} median {1e3*np.median(ts[5:]):7.3f} ms  min {1e3*np.min(ts[5:]):7.3f} ms", flush=True)
    model.train()
    for mode in ("forward", "train"):
        ts = []
        for it in range(30):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            if mode == "forward":
                with torch.no_grad():
                    logits = model((x, pos, ei))
            else:
                logits = model((x, pos, ei))
                loss = crit(logits, label)
                opt.zero_grad(); loss.backward(); opt.step()
                _ = loss.item()
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(f"{name:36s} {mode:8s} median {1e3*np.median(ts[5:]):7.3f} ms  min {1e3*np.min(ts[5:]):7.3f} ms", flush=True)

# The reference's superpixel regime: a NEW topology every optimizer step over the same node count
# (utils/image_to_graph/image_to_graph_superpixel.py:31-66).  eager = what a changing topology cost before (host-bound);
# captured / any topology = ONE hipGraph whose buffers hold a dummy node and spare edge slots, topology build inside the graph.
from graphnet_classifier_amd.train import CapturedTrainStep, FlatParameters, FusedAdam  # noqa: E402
graphs = [synthetic.superpixel_like_graphs(1, seed=2000 + k, shapes=((12, 12),)) for k in range(8)]
graphs = [(g.x, g.pos, g.edge_index) for g in graphs]
tmodel = CombinedModel(GraphNet(num_local_features=3, space_dim=2, out_channels=1, n_blocks=3), num_nodes=144, classes=2)
tmodel.train()
fopt = FusedAdam(FlatParameters(tmodel), lr=1e-3)
crit = torch.nn.CrossEntropyLoss()
label = torch.tensor(1)
loss_sum = torch.zeros((), dtype=torch.float64, device="cuda:0")
dev_graphs = [(x.cuda(), p.cuda(), e.cuda()) for x, p, e in graphs]
labdev = label.cuda()
ts = []
for it in range(60):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    logits = tmodel(dev_graphs[it % 8])
    loss = crit(logits, labdev)
    fopt.zero_grad(); loss.backward(); fopt.step()
    loss_sum += loss.detach().double()
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(f"{'new topology every step (N=144)':36s} train/fused (eager)            median {1e3*np.median(ts[5:]):7.3f} ms", flush=True)
cap = max(int(e.size(1)) for _, _, e in graphs)
step = CapturedTrainStep(tmodel, fopt, crit, graphs[0], label, loss_sum, edge_capacity=(cap * 3 // 2 + 255) // 256 * 256)
for src, tag in ((graphs, "host tensors in"), (dev_graphs, "device tensors in")):
    ts = []
    for it in range(60):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        step(src[it % 8], label)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"{'new topology every step (N=144)':36s} train/captured, any topology ({tag}) median {1e3*np.median(ts[5:]):7.3f} ms  "
          f"(edge capacity {step.edge_capacity})", flush=True)
from graphnet_classifier_amd.GNN import CapturedForward  # noqa: E402
imodel = CombinedModel(GraphNet(num_local_features=3, space_dim=2, out_channels=1, n_blocks=3), num_nodes=144, classes=2).eval()
for tag, runner in (("forward (eager)", None), ("forward/captured, any topology", CapturedForward(imodel, *graphs[0], edge_capacity=step.edge_capacity))):
    ts = []
    for it in range(60):
        g = dev_graphs[it % 8]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if runner is None:
            with torch.no_grad():
                out = imodel(g)
        else:
            out = runner(*g)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"{'new topology every call (N=144)':36s} {tag:46s} median {1e3*np.median(ts[5:]):7.3f} ms", flush=True)
try:
    step.check()
except IndexError:
    print("flags:", step._status.tolist(), bool(step._range_flag), "loss_sum", float(loss_sum))
    raise
