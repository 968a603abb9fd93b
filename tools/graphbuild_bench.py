#!/usr/bin/env python3
"""Timing of the device-side superpixel graph build (post-SLIC part) on synthetic Voronoi label images.
(The CPU comparison quoted in DESIGN.md was taken with the test-suite oracle; tools never import oracle/.)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphnet_classifier_amd import image_to_graph as I2G  # noqa: E402

rng = np.random.default_rng(0)
for size, nseg in ((32, 100), (128, 100), (256, 400)):
    img = rng.integers(0, 256, size=(size, size, 3), dtype=np.uint8)
    pts = rng.random((nseg, 2)) * size
    yy, xx = np.meshgrid(np.arange(size), np.arange(size), indexing="ij")
    seg = ((yy[..., None] - pts[:, 0]) ** 2 + (xx[..., None] - pts[:, 1]) ** 2).argmin(-1).astype(np.int32)
    I2G.superpixel_graph_from_labels(img, seg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        x, pos, ei = I2G.superpixel_graph_from_labels(img, seg)
    torch.cuda.synchronize()
    gpu = (time.perf_counter() - t0) / 10
    print(f"R={size} segments={x.size(0)} edges={ei.size(1)}: device {gpu*1e3:.2f} ms (incl. H2D + 1 sync)", flush=True)
