"""Shared helpers for the tests: golden-fixture loading."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def sub_state_dict(arrays, tag, device="cpu"):
    """Arrays saved as '<tag>key' -> {key: tensor}."""
    return {k[len(tag):]: torch.from_numpy(np.array(v)).to(device) for k, v in arrays.items() if k.startswith(tag)}


def t(a, device="cpu"):
    return torch.from_numpy(np.array(a)).to(device)


def max_abs(a, b):
    return float((a.double() - b.double()).abs().max()) if a.numel() else 0.0
