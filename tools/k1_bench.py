#!/usr/bin/env python3
"""K1 (gnc_scatter_sum_csr_f32) alone on the BASELINE workloads' own row pointers: CSR-ordered messages [E, D] and the permuted
form, HIP-event time per launch and GB/s of algorithmic bytes (E*D*4 read + N*D*4 written + row pointers [+ perm])."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphnet_classifier_amd import native, synthetic  # noqa: E402
from graphnet_classifier_amd.topology import get_topology  # noqa: E402

for name, scale in ((n, 1.0) for n in (sys.argv[1:] or ["c3"])):
    batch, kw = synthetic.make_workload(name, scale)
    d = kw["out_dim_edge"]
    topo = get_topology(batch.edge_index.cuda(), batch.num_nodes, torch.device("cuda:0"))
    e, n = batch.num_edges, batch.num_nodes
    msg = torch.randn(e, d, device="cuda:0")
    for tag, perm in (("sorted", None), ("perm", topo.perm)):
        fn = lambda: native.scatter_sum_csr(msg, topo.rowptr, perm, n)
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
        ev[0].record()
        for i in range(20):
            fn()
            ev[i + 1].record()
        torch.cuda.synchronize()
        ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(20)]
        byts = e * d * 4 + n * d * 4 + (n + 1) * 4 + (e * 4 if perm is not None else 0)
        print(f"{name} D={d} {tag:6s}: median {np.median(ts):.4f} ms  min {np.min(ts):.4f} ms  "
              f"{byts / np.median(ts) / 1e6:.0f} GB/s = {byts / np.median(ts) / 1e6 / 8000:.3f} of 8 TB/s", flush=True)
