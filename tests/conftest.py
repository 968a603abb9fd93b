"""pytest configuration: registers the ``gpu`` marker and puts the repo root on sys.path.

``-m "not gpu"`` runs everywhere (oracle vs golden vectors, host logic, C-ABI symbol
checks, gloo multi-process tests); ``-m gpu`` needs a real MI355X and calls the HIP path
through the C-ABI library.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (runs the HIP kernels through the C-ABI)")


def pytest_sessionstart(session):
    """The shared library is a build artefact (git-ignored).  In a fresh checkout build it once, exactly as
    __graft_entry__.build() does (hipcc cross-compiles gfx950 without a GPU), so that the C-ABI tests have
    something to load; on the GPU box the prebuilt file travels with the snapshot and nothing happens."""
    lib = os.path.join(ROOT, "graphnet_classifier_amd", "libgnc_hip.so")
    if not os.path.exists(lib):
        import subprocess
        subprocess.run(["make", "-j", "4", "-C", os.path.join(ROOT, "graphnet_classifier_amd", "csrc")], check=True,
                       stdout=subprocess.DEVNULL)


def pytest_collection_modifyitems(config, items):
    """GPU tests fail loudly on a GPU-less host only when explicitly selected; when the
    whole suite is run without -m they are skipped there."""
    import torch
    if torch.cuda.is_available():
        return
    markexpr = config.getoption("-m") or ""
    if "gpu" in markexpr and "not gpu" not in markexpr:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
