"""ctypes access to the C oracle (oracle/scatter_sum_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libgnc_oracle.so")
_lib = None


def build():
    subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_scatter_sum_f32.restype = ctypes.c_int
        _lib.oracle_csr_build.restype = ctypes.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def scatter_sum(src: np.ndarray, index: np.ndarray, num_nodes: int) -> np.ndarray:
    src = np.ascontiguousarray(src, dtype=np.float32)
    index = np.ascontiguousarray(index, dtype=np.int64)
    out = np.empty((num_nodes, src.shape[1]), dtype=np.float32)
    rc = load().oracle_scatter_sum_f32(_p(src), _p(index), ctypes.c_int64(src.shape[0]), ctypes.c_int64(src.shape[1]),
                                       ctypes.c_int64(num_nodes), _p(out))
    if rc != 0:
        raise IndexError("index out of range")
    return out


def csr_build(index: np.ndarray, num_nodes: int):
    index = np.ascontiguousarray(index, dtype=np.int64)
    rowptr = np.empty(num_nodes + 1, dtype=np.int32)
    perm = np.empty(index.size, dtype=np.int32)
    rc = load().oracle_csr_build(_p(index), ctypes.c_int64(index.size), ctypes.c_int64(num_nodes), _p(rowptr), _p(perm))
    if rc != 0:
        raise IndexError("index out of range")
    return rowptr, perm
