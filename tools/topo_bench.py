#!/usr/bin/env python3
"""Times the topology build of a workload's batch: gnc_topology_build (LDS path; with / without the gated general path
enqueued behind it) against the rocPRIM path (gnc_csr_build + two permutes) it replaces."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphnet_classifier_amd import native, synthetic  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
batch, kw = synthetic.make_workload(name, 1.0)
ei = batch.edge_index.to("cuda:0")
row, col = ei[0].contiguous(), ei[1].contiguous()
n = batch.num_nodes


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


def old():
    rowptr, perm, status = native.csr_build(col, n)
    return rowptr, perm, native.permute_index_checked(row, perm, n, status[1:]), native.permute_index(col, perm)


t_old, o = timed(old)
t_new, a = timed(lambda: native.topology_build(row, col, n, gated_fallback=True))
t_new0, b = timed(lambda: native.topology_build(row, col, n, gated_fallback=False))
print(f"{name}: E={batch.num_edges} rocPRIM path {t_old:.3f} ms | LDS path + gated general path {t_new:.3f} ms (flags {a[4].tolist()}) | "
      f"LDS path alone {t_new0:.3f} ms; equal: {all(torch.equal(x, y) for x, y in zip(o, a[:4]))}")

for ph in (1, 2, 3, 4, 5):
    os.environ["GNC_TOPO_PHASE_LIMIT"] = str(ph)
    t, _ = timed(lambda: native.topology_build(row, col, n, gated_fallback=False))
    print(f"  LDS path stopped after phase {ph}: {t:.3f} ms")
os.environ.pop("GNC_TOPO_PHASE_LIMIT")
