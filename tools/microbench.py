#!/usr/bin/env python3
"""Developer micro-benchmark of the individual HIP kernels (not the contract bench; see bench.py).

    python tools/microbench.py [--nodes 1000000] [--edges 10000000] [--dim 64] [--iters 20]
"""
import argparse
import json
import time

import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphnet_classifier_amd import native  # noqa: E402


def timeit(fn, iters, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(iters)]
    return float(np.median(ts)), float(np.min(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=1_000_000)
    ap.add_argument("--edges", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--skip-mlp", action="store_true")
    a = ap.parse_args()
    dev = "cuda:0"
    n, e, d = a.nodes, a.edges, a.dim
    g = torch.Generator(device="cpu").manual_seed(0)
    # block-diagonal batch of 160-node graphs like config C3: edges stay inside a graph
    gsz = 160
    graph_of_edge = torch.arange(e) // (e // (n // gsz))
    graph_of_edge.clamp_(max=n // gsz - 1)
    row = graph_of_edge * gsz + torch.randint(0, gsz, (e,), generator=g)
    col = graph_of_edge * gsz + torch.randint(0, gsz, (e,), generator=g)
    perm_e = torch.randperm(e, generator=g)  # unsorted at the boundary
    row, col = row[perm_e].to(dev), col[perm_e].to(dev)
    res = {"nodes": n, "edges": e, "dim": d}

    t0 = time.perf_counter()
    rowptr, perm, status = native.csr_build(col, n)
    torch.cuda.synchronize()
    res["csr_build_first_ms"] = (time.perf_counter() - t0) * 1e3
    med, mn = timeit(lambda: native.csr_build(col, n), 5, 1)
    res["csr_build_ms"] = med

    src = torch.randn(e, d, device=dev)
    out = torch.empty(n, d, device=dev)
    alg_bytes = 4 * d * e + 4 * d * n + 4 * e + 4 * (n + 1)
    for name, p in (("scatter_perm", perm), ("scatter_sorted", None)):
        med, mn = timeit(lambda: native.scatter_sum_csr(src, rowptr, p, n, out=out), a.iters)
        res[name + "_ms"] = med
        res[name + "_GBps"] = alg_bytes / med / 1e6
        res[name + "_Gedges_s"] = e / med / 1e6
    med, mn = timeit(lambda: out.index_add_(0, col, src), 3, 1)
    res["torch_index_add_ms"] = med

    src32, dst32 = native.permute_index(row, perm), native.permute_index(col, perm)
    x = torch.randn(n, d, device=dev)
    med, mn = timeit(lambda: native.gather_rows(x, src32), a.iters)
    res["gather_ms"] = med
    res["gather_GBps"] = (8 * d * e + 4 * e) / med / 1e6

    # PCIe-inclusive hand-over of one c3-sized batch (host tensors as the reference's loader yields them)
    hx, hp, hei = torch.rand(n, 3), torch.rand(n, 2), torch.stack([row, col]).cpu()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _ = hx.to(dev), hp.to(dev), hei.to(dev)
    torch.cuda.synchronize()
    res["h2d_pageable_ms"] = (time.perf_counter() - t0) * 1e3
    hx, hp, hei = hx.pin_memory(), hp.pin_memory(), hei.pin_memory()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _ = hx.to(dev, non_blocking=True), hp.to(dev, non_blocking=True), hei.to(dev, non_blocking=True)
    torch.cuda.synchronize()
    res["h2d_pinned_ms"] = (time.perf_counter() - t0) * 1e3
    res["h2d_bytes"] = hx.numel() * 4 + hp.numel() * 4 + hei.numel() * 8

    if not a.skip_mlp:
        def lin(o, i):
            return torch.randn(o, i, device=dev) / i ** 0.5, torch.randn(o, device=dev) * 0.1
        ws, bs = zip(lin(d, 3 * d), lin(d, d), lin(d, d))
        ln = (torch.ones(d, device=dev), torch.zeros(d, device=dev), 1e-5)
        ea = src
        fn = lambda: native.mlp_forward([(x, src32), (x, dst32), (ea, None)], ws, bs, ln=ln, residual=ea)
        med, mn = timeit(fn, max(3, a.iters // 4), 1)
        flops = 2.0 * e * (3 * d * d + d * d + d * d)
        res["edge_mlp_ms"] = med
        res["edge_mlp_TFLOPs"] = flops / med / 1e9
        ws, bs = zip(lin(d, 2 * d), lin(d, d), lin(d, d))
        fn = lambda: native.mlp_forward([(x, None), (out, None)], ws, bs, ln=ln, residual=x)
        med, mn = timeit(fn, max(3, a.iters // 4), 1)
        res["node_mlp_ms"] = med
        res["node_mlp_TFLOPs"] = 2.0 * n * 4 * d * d / med / 1e9
    print(json.dumps(res))


if __name__ == "__main__":
    main()
