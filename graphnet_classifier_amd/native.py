"""ctypes binding of ``libgnc_hip.so`` (C ABI: ``include/gnc_hip.h``).

PyTorch is used for device memory and streams only: every call below passes raw
``data_ptr()`` values and the current HIP stream to the library.  There is NO
fallback: if the shared library is missing or a call fails, a ``RuntimeError``
is raised -- the product path never silently routes around the HIP kernels.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int32, c_int64, c_size_t, c_void_p

import torch

LIB_NAME = "libgnc_hip.so"
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)

GNC_MAX_SEGMENTS = 4
GNC_MAX_LINEAR = 8
ABI_VERSION = 19

ACTIVATIONS = {  # nn.<Name> accepted by the reference's MLP(activation=...) (models/MLP.py:21)
    "ReLU": 0, "Identity": 1, "Tanh": 2, "Sigmoid": 3, "SiLU": 4, "GELU": 5, "LeakyReLU": 6, "ELU": 7,
}

# every symbol include/gnc_hip.h declares: (restype, argtypes)
_SIGNATURES = {
    "gnc_abi_version": (c_int32, []),
    "gnc_last_error_string": (c_char_p, []),
    "gnc_target_arch": (c_char_p, []),
    "gnc_mlp_agg_supported": (c_int32, [c_void_p]),
    "gnc_mlp_edge_features_supported": (c_int32, [c_void_p]),
    "gnc_mlp_operands_in_place_supported": (c_int32, [c_void_p]),
    "gnc_mlp_save_act_supported": (c_int32, [c_void_p]),
    "gnc_mlp_agg_fix_len": (c_int32, []),
    "gnc_mlp_small_batch_supported": (c_int32, [c_void_p]),
    "gnc_mlp_small_batch_max_rows": (c_int64, []),
    "gnc_mlp_projection_t2_f32": (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int32, c_int32, c_void_p,
                                            c_int64, c_void_p]),
    "gnc_mlp_dual_projection_f32": (c_int32, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32,
                                              c_void_p, c_void_p, c_int64, c_void_p]),
    "gnc_mlp_backward_small_batch_supported": (c_int32, [c_void_p]),
    "gnc_readout_forward_f32": (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int32, c_void_p, c_int64, c_void_p, c_int32,
                                          c_void_p, c_int64, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gnc_readout_backward_f32": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int32, c_void_p, c_int64, c_int32, c_void_p,
                                           c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_void_p]),
    "gnc_xty_small_max_rows": (c_int32, []),
    "gnc_xty_small_f32": (c_int32, [c_void_p, c_int32, c_void_p]),
    "gnc_agg_fixup_f32": (c_int32, [c_void_p, c_int64, c_void_p, c_void_p, c_int32, c_int64, c_int32, c_void_p, c_int64,
                                    c_void_p]),
    "gnc_csr_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "gnc_csr_build": (c_int32, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "gnc_topology_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int32]),
    "gnc_topology_build": (c_int32, [c_void_p, c_void_p, c_int32, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_size_t, c_int32, c_void_p]),
    "gnc_permute_index_i64_i32": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "gnc_permute_index_checked_i64_i32": (c_int32, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "gnc_scatter_sum_csr_f32": (c_int32, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int32, c_void_p,
                                          c_int64, c_void_p]),
    "gnc_gather_rows_f32": (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_int32, c_void_p, c_int64, c_void_p]),
    "gnc_gather_rows_add_f32": (c_int32, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int32, c_void_p, c_int64,
                                          c_void_p]),
    "gnc_poison_if_flagged_f32": (c_int32, [c_void_p, c_int64, c_void_p, c_int32, c_void_p]),
    "gnc_edge_features_f32": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    "gnc_sizeof_mlp_desc": (c_size_t, []),
    "gnc_mlp_supported": (c_int32, [c_void_p]),
    "gnc_mlp_forward_f32": (c_int32, [c_void_p, c_void_p]),
    "gnc_sizeof_mlp_bwd_desc": (c_size_t, []),
    "gnc_mlp_backward_supported": (c_int32, [c_void_p]),
    "gnc_mlp_backward_dx_add_honoured": (c_int32, [c_void_p]),
    "gnc_mlp_backward_grad_gather_honoured": (c_int32, [c_void_p]),
    "gnc_mlp_backward_saved_act_honoured": (c_int32, [c_void_p]),
    "gnc_mlp_backward_ln_partial_rows": (c_int32, [c_void_p]),
    "gnc_mlp_backward_fused_rows": (c_int32, [c_void_p]),
    "gnc_mlp_backward_f32": (c_int32, [c_void_p, c_void_p]),
    "gnc_xty_partials": (c_int32, [c_int64]),
    "gnc_xty_partials_for": (c_int32, [c_int64, c_int32, c_int32]),
    "gnc_xty_f32": (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int32, c_int32, c_void_p, c_int32, c_void_p]),
    "gnc_grid_num_edges": (c_int64, [c_int32, c_int32, c_int32]),
    "gnc_grid_edges_i64": (c_int32, [c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "gnc_pixel_nodes_f32": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "gnc_patch_nodes_f32": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "gnc_rag_workspace_bytes": (c_size_t, [c_int32, c_int32]),
    "gnc_rag_build": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                                c_void_p, c_size_t, c_void_p]),
    "gnc_colsum_pair_f32": (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int32, c_void_p, c_int32, c_void_p]),
    "gnc_reduce_partials_f32": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int64, c_void_p, c_void_p]),
    "gnc_activation_f32": (c_int32, [c_void_p, c_int64, c_int64, c_int32, c_int32, c_float, c_void_p, c_int64, c_void_p]),
    "gnc_activation_backward_f32": (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int32, c_int32, c_float, c_void_p, c_int64,
                                              c_void_p]),
    "gnc_layer_norm_backward_f32": (c_int32, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int32, c_float, c_void_p, c_int64,
                                              c_void_p, c_int64, c_void_p]),
    "gnc_adam_step_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_float,
                                    c_void_p, c_void_p, c_void_p]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


class MlpSegment(Structure):
    _fields_ = [("ptr", c_void_p), ("index", c_void_p), ("width", c_int32), ("ld", c_int32), ("mode", c_int32),
                ("wcol", c_int32), ("table_rows", c_int64)]


class MlpDesc(Structure):
    _fields_ = [
        ("num_segments", c_int32), ("num_linear", c_int32), ("activation", c_int32), ("act_param", c_float),
        ("seg", MlpSegment * GNC_MAX_SEGMENTS),
        ("weight", c_void_p * GNC_MAX_LINEAR), ("ld_weight", c_int32 * GNC_MAX_LINEAR),
        ("bias", c_void_p * GNC_MAX_LINEAR),
        ("in_dim", c_int32 * GNC_MAX_LINEAR), ("out_dim", c_int32 * GNC_MAX_LINEAR),
        ("ln_gamma", c_void_p), ("ln_beta", c_void_p), ("ln_eps", c_float),
        ("residual", c_void_p), ("ld_residual", c_int32),
        ("out", c_void_p), ("ld_out", c_int32),
        ("rows", c_int64),
        ("agg_out", c_void_p), ("ld_agg", c_int32), ("agg_index", c_void_p), ("agg_fix", c_void_p),
        ("save_act", c_void_p * GNC_MAX_LINEAR),
        ("ef_pos", c_void_p), ("ef_src", c_void_p), ("ef_dst", c_void_p), ("ef_nodes", c_int64), ("ef_space_dim", c_int32),
    ]


class MlpBwdDesc(Structure):
    _fields_ = [
        ("fwd", MlpDesc), ("grad_out", c_void_p), ("ld_grad_out", c_int32),
        ("act", c_void_p * GNC_MAX_LINEAR), ("dz", c_void_p * GNC_MAX_LINEAR),
        ("dx", c_void_p), ("ld_dx", c_int32), ("yhat", c_void_p), ("dx_add_grad_out", c_int32), ("ln_partial", c_void_p),
        ("dw_partial", c_void_p * GNC_MAX_LINEAR),
        ("grad_gather", c_void_p), ("ld_grad_gather", c_int32), ("grad_gather_index", c_void_p), ("grad_gather_rows", c_int64),
        ("grad_sum", c_void_p), ("ld_grad_sum", c_int32), ("act_given", c_int32),
    ]


class XtyJob(Structure):
    _fields_ = [("a", c_void_p), ("lda", c_int64), ("b", c_void_p), ("ldb", c_int64), ("rows", c_int64), ("m", c_int32),
                ("k", c_int32), ("dw", c_void_p), ("ld_dw", c_int64), ("db", c_void_p), ("kind", c_int32)]


GNC_XTY_MAX_JOBS = 8

_lib = None


def load_library() -> ctypes.CDLL:
    """Load the in-tree shared library (torch is imported first so that its HIP runtime is
    the one already mapped).  Raises RuntimeError when it is missing or has the wrong ABI."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("GNC_LIB_PATH", LIB_PATH)  # developer override: A/B builds and the phase-probe build
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C graphnet_classifier_amd/csrc`.  There is no CPU/PyTorch fallback for the hot path.")
    lib = ctypes.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here means the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    if lib.gnc_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{LIB_NAME}: ABI version {lib.gnc_abi_version()} != expected {ABI_VERSION}")
    if lib.gnc_sizeof_mlp_bwd_desc() != ctypes.sizeof(MlpBwdDesc):
        raise RuntimeError(f"{LIB_NAME}: gnc_mlp_bwd_desc_t is {lib.gnc_sizeof_mlp_bwd_desc()} bytes in C but "
                           f"{ctypes.sizeof(MlpBwdDesc)} in the ctypes binding")
    if lib.gnc_sizeof_mlp_desc() != ctypes.sizeof(MlpDesc):
        raise RuntimeError(f"{LIB_NAME}: gnc_mlp_desc_t is {lib.gnc_sizeof_mlp_desc()} bytes in C but "
                           f"{ctypes.sizeof(MlpDesc)} in the ctypes binding")
    _lib = lib
    return lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load_library().gnc_last_error_string()
        raise RuntimeError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _require_cuda(*tensors: torch.Tensor) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("graphnet_classifier_amd: the hot path runs on the GPU only; got a CPU tensor "
                               "(there is deliberately no CPU fallback)")


def _rowmajor(t: torch.Tensor) -> torch.Tensor:
    """2-D fp32 tensor with unit column stride (row stride may exceed the width)."""
    if t.dtype != torch.float32:
        raise TypeError(f"expected float32, got {t.dtype}")
    if t.dim() != 2:
        raise ValueError(f"expected a 2-D tensor, got shape {tuple(t.shape)}")
    if t.size(1) > 0 and t.stride(1) != 1 or (t.size(0) > 1 and t.stride(0) < t.size(1)):
        t = t.contiguous()
    return t


def _ld(t: torch.Tensor) -> int:
    if t.size(0) > 1 or t.stride(0) >= t.size(1):
        return max(t.stride(0), 1)
    return max(t.size(1), 1)  # a single row whose stride says nothing


# --------------------------------------------------------------------------- kernel timers
class KernelTimers:
    """Optional HIP-event timing of individual launches (bench.py uses it for the roofline
    object).  Events are recorded on the stream the kernel is launched on (torch's current
    stream); nothing is synchronised until ``summary()`` is called after the timed region."""

    def __init__(self, only: str | None = None):
        self.events = {}
        self.work = {}
        self._pool = []
        self.only = only  # time launches of this name alone (every event record is a few microseconds of GPU time)

    def reserve(self, launches: int) -> None:
        """Create (and record once, which is what makes the runtime allocate them) the events of ``launches``
        launches ahead of a timed region: growing the runtime's event pool costs tens of milliseconds each time
        it happens (seen as one 50-70 ms step in every fresh process around the 900th event)."""
        stream = torch.cuda.current_stream()
        for _ in range(2 * launches):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(stream)
            self._pool.append(ev)

    def num_launches(self) -> int:
        return sum(len(v) for v in self.events.values())

    def launch(self, name: str, stream_tensor: torch.Tensor, fn, work: float = 0.0):
        if self.only is not None and name != self.only:
            return fn()
        start = self._pool.pop() if self._pool else torch.cuda.Event(enable_timing=True)
        end = self._pool.pop() if self._pool else torch.cuda.Event(enable_timing=True)
        start.record(torch.cuda.current_stream(stream_tensor.device))
        rc = fn()
        end.record(torch.cuda.current_stream(stream_tensor.device))
        self.events.setdefault(name, []).append((start, end))
        self.work.setdefault(name, []).append(work)
        return rc

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, evs in self.events.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            out[name] = {"launches": len(ms), "avg_ms": sum(ms) / len(ms), "min_ms": min(ms),
                         "avg_work": sum(self.work[name]) / len(ms)}
        return out


_timers: KernelTimers | None = None


def set_kernel_timers(timers: KernelTimers | None) -> None:
    global _timers
    _timers = timers


def _launch(name: str, t: torch.Tensor, fn, work: float = 0.0):
    return _timers.launch(name, t, fn, work) if _timers is not None else fn()


# --------------------------------------------------------------------------- topology
def csr_build(index: torch.Tensor, num_nodes: int):
    """index [E] int64 (device) -> (rowptr int32 [N+1], perm int32 [E], status int32 [2]: [0] = out-of-range flag of the build, [1] = scratch flag for a checked permute)."""
    lib = load_library()
    _require_cuda(index)
    if index.dtype != torch.int64:
        index = index.long()
    index = index.contiguous()
    e = index.numel()
    dev = index.device
    rowptr = torch.empty(num_nodes + 1, dtype=torch.int32, device=dev)
    perm = torch.empty(e, dtype=torch.int32, device=dev)
    status = torch.zeros(2, dtype=torch.int32, device=dev)  # [0]: written by the build, [1]: free for a checked permute
    with torch.cuda.device(dev):
        nbytes = lib.gnc_csr_workspace_bytes(num_nodes, e)
        if nbytes == 0:
            _check(-1, "gnc_csr_workspace_bytes")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _check(lib.gnc_csr_build(index.data_ptr(), e, num_nodes, rowptr.data_ptr(), perm.data_ptr(), status.data_ptr(),
                                 ws.data_ptr(), nbytes, _stream(index)), "gnc_csr_build")
    return rowptr, perm, status


def topology_build(src: torch.Tensor | None, dst: torch.Tensor, num_nodes: int, gated_fallback: bool = True):
    """(rowptr int32 [N+1], perm int32 [E], src_sorted int32 [E] | None, dst_sorted int32 [E] | None, status int32 [3]) of
    one edge list in ONE call (include/gnc_hip.h, gnc_topology_build): graph-ordered batches are sorted range by range in
    LDS; anything else raises status[2] on the device and - with ``gated_fallback`` - is sorted by the general kernels
    enqueued behind it.  ``src`` / ``dst``: int64 or int32 [E] on the device; ``src=None`` gives rowptr and perm only."""
    lib = load_library()
    _require_cuda(dst, src)
    if dst.dtype not in (torch.int64, torch.int32):
        dst = dst.long()
    dst = dst.contiguous()
    if src is not None:
        src = src.to(dst.dtype).contiguous()
    e, dev = dst.numel(), dst.device
    rowptr = torch.empty(num_nodes + 1, dtype=torch.int32, device=dev)
    perm = torch.empty(e, dtype=torch.int32, device=dev)
    ends = torch.empty(2, e, dtype=torch.int32, device=dev) if src is not None else None
    status = torch.empty(3, dtype=torch.int32, device=dev)  # zeroed by the call
    with torch.cuda.device(dev):
        nbytes = lib.gnc_topology_workspace_bytes(num_nodes, e, 1 if gated_fallback else 0)
        if nbytes == 0:
            _check(-1, "gnc_topology_workspace_bytes")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _check(lib.gnc_topology_build(src.data_ptr() if src is not None else None, dst.data_ptr(), dst.element_size(), e, num_nodes,
                                      rowptr.data_ptr(), perm.data_ptr(), ends[0].data_ptr() if ends is not None else None,
                                      ends[1].data_ptr() if ends is not None else None, status.data_ptr(), ws.data_ptr(), nbytes,
                                      1 if gated_fallback else 0, _stream(dst)), "gnc_topology_build")
    return rowptr, perm, (ends[0] if ends is not None else None), (ends[1] if ends is not None else None), status


def permute_index_checked(src: torch.Tensor, perm: torch.Tensor | None, num_nodes: int, status: torch.Tensor) -> torch.Tensor:
    """``permute_index`` that also flags ids outside [0, num_nodes) in ``status`` (int32 [1] on the device, zero
    before the call) and stores them as 0."""
    lib = load_library()
    _require_cuda(src, status)
    src = src.contiguous()
    out = torch.empty(src.numel(), dtype=torch.int32, device=src.device)
    with torch.cuda.device(src.device):
        _check(lib.gnc_permute_index_checked_i64_i32(src.data_ptr(), perm.data_ptr() if perm is not None else None, src.numel(),
                                                     num_nodes, out.data_ptr(), status.data_ptr(), _stream(src)),
               "gnc_permute_index_checked_i64_i32")
    return out


def permute_index(src: torch.Tensor, perm: torch.Tensor | None) -> torch.Tensor:
    """int64 [E] -> int32 [E], reordered by perm (sorted position -> original edge)."""
    lib = load_library()
    _require_cuda(src)
    src = src.contiguous()
    out = torch.empty(src.numel(), dtype=torch.int32, device=src.device)
    with torch.cuda.device(src.device):
        _check(lib.gnc_permute_index_i64_i32(src.data_ptr(), perm.data_ptr() if perm is not None else None, src.numel(),
                                             out.data_ptr(), _stream(src)), "gnc_permute_index_i64_i32")
    return out


# --------------------------------------------------------------------------- K1 / K2 / K6
def scatter_sum_csr(src: torch.Tensor, rowptr: torch.Tensor, perm: torch.Tensor | None, num_nodes: int,
                    out: torch.Tensor | None = None) -> torch.Tensor:
    lib = load_library()
    _require_cuda(src, rowptr)
    src = _rowmajor(src)
    e, d = src.shape
    if out is None:
        out = torch.empty(num_nodes, d, dtype=torch.float32, device=src.device)
    # algorithmic bytes of this launch: messages once, output once, row pointers,
    # and one int32 per edge only when the kernel has to go through the permutation
    work = 4.0 * d * e + 4.0 * d * num_nodes + 4.0 * (num_nodes + 1) + (4.0 * e if perm is not None else 0.0)
    with torch.cuda.device(src.device):
        _check(_launch("scatter_sum_csr" + ("_perm" if perm is not None else "_sorted"), src,
                       lambda: lib.gnc_scatter_sum_csr_f32(src.data_ptr(), _ld(src), rowptr.data_ptr(),
                                                           perm.data_ptr() if perm is not None else None, num_nodes,
                                                           e, d, out.data_ptr(), _ld(out), _stream(src)), work),
               "gnc_scatter_sum_csr_f32")
    return out


def gather_rows(table: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
    lib = load_library()
    _require_cuda(table, index)
    table = _rowmajor(table)
    if index.dtype != torch.int32:
        raise TypeError("gather_rows expects an int32 index")
    n, d = index.numel(), table.size(1)
    out = torch.empty(n, d, dtype=torch.float32, device=table.device)
    with torch.cuda.device(table.device):
        _check(lib.gnc_gather_rows_f32(table.data_ptr(), _ld(table), index.data_ptr(), n, d, out.data_ptr(), _ld(out),
                                       _stream(table)), "gnc_gather_rows_f32")
    return out


def gather_rows_add(table: torch.Tensor, index: torch.Tensor, addend: torch.Tensor) -> torch.Tensor:
    """out[r] = table[index[r]] + addend[r] in one pass."""
    lib = load_library()
    _require_cuda(table, index, addend)
    table, addend = _rowmajor(table), _rowmajor(addend)
    if index.dtype != torch.int32:
        raise TypeError("gather_rows_add expects an int32 index")
    n, d = index.numel(), table.size(1)
    if addend.shape != (n, d):
        raise ValueError(f"addend must be [{n}, {d}], got {tuple(addend.shape)}")
    out = torch.empty(n, d, dtype=torch.float32, device=table.device)
    with torch.cuda.device(table.device):
        _check(lib.gnc_gather_rows_add_f32(table.data_ptr(), _ld(table), index.data_ptr(), addend.data_ptr(), _ld(addend), n, d,
                                           out.data_ptr(), _ld(out), _stream(table)), "gnc_gather_rows_add_f32")
    return out


def edge_features(pos: torch.Tensor, src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    lib = load_library()
    _require_cuda(pos, src, dst)
    pos = pos.float().contiguous()
    e, sd = src.numel(), pos.size(1)
    ld = (sd + 1 + 3) // 4 * 4  # rows padded to a multiple of 16 B (zeros) for the MLP kernels' vector loads
    buf = torch.empty(e, ld, dtype=torch.float32, device=pos.device)
    with torch.cuda.device(pos.device):
        _check(lib.gnc_edge_features_f32(pos.data_ptr(), sd, src.data_ptr(), dst.data_ptr(), e, buf.data_ptr(), ld,
                                         _stream(pos)), "gnc_edge_features_f32")
    return buf[:, :sd + 1]


def poison_if_flagged_(out: torch.Tensor, flags: torch.Tensor) -> torch.Tensor:
    """In place: ``out`` (contiguous fp32) becomes NaN when any of the int32 device ``flags`` is set; one launch that returns
    at once otherwise (deferred topology validation)."""
    lib = load_library()
    _require_cuda(out, flags)
    if not out.is_contiguous() or out.dtype != torch.float32 or flags.dtype != torch.int32 or not flags.is_contiguous():
        raise TypeError("poison_if_flagged_: contiguous fp32 output and contiguous int32 flags expected")
    with torch.cuda.device(out.device):
        _check(lib.gnc_poison_if_flagged_f32(out.data_ptr(), out.numel(), flags.data_ptr(), flags.numel(), _stream(out)),
               "gnc_poison_if_flagged_f32")
    return out


# --------------------------------------------------------------------------- K4
def make_mlp_desc(segments, weights, biases, ln, activation: str, act_param: float, residual, out, rows: int):
    """Fill a gnc_mlp_desc_t.  ``segments`` = [(table, index_or_None, width, mode, wcol)], tensors must stay
    alive until the call returns (they are enqueued on the current stream)."""
    if len(segments) > GNC_MAX_SEGMENTS or not (1 <= len(weights) <= GNC_MAX_LINEAR):
        raise NotImplementedError(f"MLP with {len(segments)} segments / {len(weights)} Linear layers is outside the HIP kernel")
    if activation not in ACTIVATIONS:
        raise NotImplementedError(f"activation nn.{activation} has no HIP kernel (supported: {sorted(ACTIVATIONS)})")
    d = MlpDesc()
    d.num_segments = len(segments)
    d.num_linear = len(weights)
    d.activation = ACTIVATIONS[activation]
    d.act_param = act_param
    for s, (table, index, width, mode, wcol) in enumerate(segments):
        d.seg[s].wcol = wcol
        d.seg[s].ptr = table.data_ptr()
        d.seg[s].index = index.data_ptr() if index is not None else None
        d.seg[s].width = width
        d.seg[s].ld = _ld(table)
        d.seg[s].mode = mode
        d.seg[s].table_rows = table.size(0) if index is not None else 0
    for l, (w, b) in enumerate(zip(weights, biases)):
        d.weight[l] = w.data_ptr()
        d.ld_weight[l] = _ld(w)
        d.bias[l] = b.data_ptr() if b is not None else None
        d.out_dim[l], d.in_dim[l] = w.shape
    if ln is not None:
        d.ln_gamma, d.ln_beta, d.ln_eps = ln[0].data_ptr(), ln[1].data_ptr(), ln[2]
    if residual is not None:
        d.residual, d.ld_residual = residual.data_ptr(), _ld(residual)
    d.out, d.ld_out = out.data_ptr(), _ld(out)
    d.rows = rows
    return d


SEG_MATMUL, SEG_ADD = 0, 1

def _vector_rows(t: torch.Tensor) -> torch.Tensor:
    """Rows the kernels can read with 16-B vector loads: row stride a multiple of 4 floats and a 16-B
    aligned base.  Anything else (the reference's 3-column inputs and [H, 3] first-layer weights) is
    copied into a zero-padded buffer (one pad launch) and handed over as a column slice of it.  Nothing is
    cached: the copy is made from the live tensor on every call, so in-place edits through ``.data``
    (which do not bump ``_version``), ``load_state_dict`` and optimizer steps are always seen, and a
    hipGraph capture records the pad itself instead of a pointer to a stale copy."""
    if t.data_ptr() % 16 == 0 and _ld(t) % 4 == 0:
        return t
    cols = (t.size(1) + 3) // 4 * 4
    if t.stride(1) != 1 or t.stride(0) != t.size(1):
        t = t.contiguous()
    if cols == t.size(1):  # only the base address was off (a column slice of a wider tensor): fresh, aligned copy
        return t.clone(memory_format=torch.contiguous_format)
    return torch.nn.functional.pad(t, (0, cols - t.size(1)))[:, :t.size(1)]


_SMALL_ROWS = None


def _small_batch_rows(lib) -> int:
    """Row limit of the small-batch forward kernel (asked once)."""
    global _SMALL_ROWS
    if _SMALL_ROWS is None:
        _SMALL_ROWS = int(lib.gnc_mlp_small_batch_max_rows())
    return _SMALL_ROWS


def _rows_of(segments, rows) -> int:
    if rows is not None:
        return int(rows)
    t0, i0 = segments[0]
    return int(i0.numel() if i0 is not None else t0.size(0))


def _prepare_mlp(segments, weights, biases, residual, rows, modes, vector_rows: bool = True):
    """Shared argument preparation of the forward and backward launches (see mlp_forward).  ``vector_rows=False`` hands
    tables and weights over as they are (row-major, any alignment): only for a launch whose kernel reads them in place
    (gnc_mlp_small_batch_supported, gnc_mlp_operands_in_place_supported)."""
    _vr = _vector_rows if vector_rows else (lambda t: t)
    _vw = _vr
    modes = list(modes) if modes is not None else [SEG_MATMUL] * len(segments)
    segs, wcol = [], 0
    for (table, index), mode in zip(segments, modes):
        _require_cuda(table, index)
        table = _vr(_rowmajor(table))
        if index is not None and index.dtype != torch.int32:
            raise TypeError("segment index must be int32")
        segs.append((table, index, table.size(1), mode, wcol if mode == SEG_MATMUL else 0))
        if mode == SEG_MATMUL:
            wcol += table.size(1)
    # kernel-side order: MATMUL segments first (the one that is also the residual last among them, so
    # its rows can stay in registers for the residual add), then the additive segments
    def _rank(sg):
        is_res = (residual is not None and sg[1] is None and sg[0].data_ptr() == residual.data_ptr()
                  and sg[0].shape == residual.shape)
        return (sg[3] == SEG_ADD, is_res)
    segs.sort(key=_rank)
    if rows is None:
        t0, i0 = segments[0]
        rows = i0.numel() if i0 is not None else t0.size(0)
    weights = [_vw(_rowmajor(w.detach())) for w in weights]
    biases = [b.contiguous() if b is not None else None for b in biases]
    if residual is not None:
        residual = _rowmajor(residual)
    return segs, weights, biases, residual, rows, modes


SAVE_ACT = os.environ.get("GNC_NO_SAVED_ACT") is None  # A/B switch: training forwards keep nothing, backwards recompute


def _backward_reads_saved_act(lib, desc, any_tensor, need_dx: bool = True) -> bool:
    """Would gnc_mlp_backward_f32 read saved post-activations for this description?  (Shape question only: the K8 kernel
    of some shapes - the weights-resident data kernel of the node processors - recomputes regardless, and a forward that
    saved for it would write tensors nobody reads.)  ``need_dx``: whether that backward will be asked for the input
    gradient (the kernel choice, and with it the answer, depends on it)."""
    bd = MlpBwdDesc()
    ctypes.memmove(ctypes.byref(bd.fwd), ctypes.byref(desc), ctypes.sizeof(MlpDesc))
    ptr = any_tensor.data_ptr()  # any 16-B aligned device address: the query looks at alignment only
    for l in range(desc.num_linear - 1):
        bd.act[l] = ptr
    bd.act_given, bd.dx = 1, (ptr if need_dx else None)
    if lib.gnc_mlp_backward_fused_rows(ctypes.byref(bd.fwd)) > 0:
        bd.dw_partial[0] = ptr
    return lib.gnc_mlp_backward_saved_act_honoured(ctypes.byref(bd)) == 1


def mlp_forward(segments, weights, biases, ln=None, activation: str = "ReLU", act_param: float = 0.0,
                residual: torch.Tensor | None = None, rows: int | None = None, modes=None, aggregate=None,
                save_act: list | None = None, save_need_dx: bool = True):
    """Fused MLP.  segments (in CONCAT order): list of (table [*, w] fp32, index int32 [rows] | None);
    ``modes[s]`` is SEG_MATMUL (default) or SEG_ADD.  Weights may be column slices of a larger
    matrix.  The segment that is also the residual is listed last for the kernel (its weight
    columns are carried in ``wcol``), so the residual comes from the staged rows.

    ``aggregate=(dst_of_row int32 [rows] non-decreasing, rowptr int32 [N+1], N)`` asks for the fused aggregation
    epilogue: returns ``(out, agg)`` with ``agg[v] = sum of out rows with dst v`` in row order,
    bit-identical to ``scatter_sum_csr(out, rowptr)``; returns ``(out, None)`` when the launch shape cannot
    carry it (the caller then runs K1).

    ``save_act`` (training forward): an empty list; when the kernel that serves the call can write the post-activation
    outputs of its hidden layers (gnc_mlp_save_act_supported) they are appended to it ([rows, H] each) for
    ``mlp_backward(saved_act=...)``, which then reads them instead of recomputing the forward of every tile;
    ``save_need_dx`` says whether that backward will want the input gradient (``need_dx``)."""
    lib = load_library()
    given = (segments, weights, biases, residual, rows, modes)
    small = _rows_of(segments, rows) <= _small_batch_rows(lib)
    # inference launches without extras: tables and weights go over as they lie first ([N, 3] inputs and [H, 3] first-layer
    # matrices cost a pad launch pair each per call otherwise) and are padded only when the kernel that will serve the launch
    # reads rows of 16-B pieces
    raw = not small and save_act is None and aggregate is None
    segs, weights, biases, residual, rows, modes = _prepare_mlp(*given, vector_rows=not (small or raw))
    dev = segs[0][0].device
    out = torch.empty(rows, weights[-1].size(0), dtype=torch.float32, device=dev)
    desc = make_mlp_desc(segs, weights, biases, ln, activation, act_param, residual, out, rows)
    if raw and any(t.data_ptr() % 16 or _ld(t) % 4 for t in [sg[0] for sg in segs] + weights) and \
            lib.gnc_mlp_operands_in_place_supported(ctypes.byref(desc)) != 0:
        segs, weights, biases, residual, rows, modes = _prepare_mlp(*given)  # a streaming kernel: rows of 16-B pieces
        desc = make_mlp_desc(segs, weights, biases, ln, activation, act_param, residual, out, rows)
    if small and lib.gnc_mlp_small_batch_supported(ctypes.byref(desc)) != 0:
        # every other kernel reads rows as 16-B pieces: 3-column inputs / [H, 3] weights go through a zero-padded copy
        # (one pad launch each); the small-batch kernel reads them where they lie
        segs, weights, biases, residual, rows, modes = _prepare_mlp(*given)
        desc = make_mlp_desc(segs, weights, biases, ln, activation, act_param, residual, out, rows)
    if (save_act is not None and SAVE_ACT and rows > 0 and len(weights) >= 2
            and lib.gnc_mlp_save_act_supported(ctypes.byref(desc)) == 0 and _backward_reads_saved_act(lib, desc, out, save_need_dx)):
        for l in range(len(weights) - 1):
            a = torch.empty(rows, weights[l].size(0), dtype=torch.float32, device=dev)
            desc.save_act[l] = a.data_ptr()
            save_act.append(a)
    agg = fix = None
    if aggregate is not None:
        dst_of_row, rowptr, num_nodes = aggregate
        _require_cuda(dst_of_row, rowptr)
        if rows > 0 and lib.gnc_mlp_agg_supported(ctypes.byref(desc)) == 0:
            agg = torch.empty(num_nodes, out.size(1), dtype=torch.float32, device=dev)  # fully defined after the fix-up
            fix = torch.empty(lib.gnc_mlp_agg_fix_len(), dtype=torch.int32, device=dev)
            desc.agg_out, desc.ld_agg = agg.data_ptr(), _ld(agg)
            desc.agg_index, desc.agg_fix = dst_of_row.contiguous().data_ptr(), fix.data_ptr()
    # executed FLOPs of this launch: 2 * rows * sum(in*out) over the Linear layers
    flops = 2.0 * rows * sum(w.size(0) * w.size(1) for w in weights)
    with torch.cuda.device(dev):
        nadd = sum(1 for m in modes if m == SEG_ADD)
        _check(_launch(f"mlp_fused_in{weights[0].size(1)}{'+%dadd' % nadd if nadd else ''}_h{weights[0].size(0)}"
                       f"_out{weights[-1].size(0)}_L{len(weights)}", out,
                       lambda: lib.gnc_mlp_forward_f32(ctypes.byref(desc), _stream(out)), flops),
               "gnc_mlp_forward_f32")
        if agg is not None:  # the destinations cut by a wave-range boundary, from the stored rows (same stream)
            _check(lib.gnc_agg_fixup_f32(out.data_ptr(), _ld(out), rowptr.data_ptr(), fix.data_ptr(), fix.numel(), num_nodes,
                                         out.size(1), agg.data_ptr(), _ld(agg), _stream(out)), "gnc_agg_fixup_f32")
    return (out, agg) if aggregate is not None else out


FOLD_EDGE_FEATURES = os.environ.get("GNC_NO_K6_FOLD") is None  # A/B switch: K6 as its own launch + a [E, 4] table


def mlp_forward_edge_features(pos: torch.Tensor, src: torch.Tensor, dst: torch.Tensor, weights, biases, ln=None,
                              activation: str = "ReLU", act_param: float = 0.0):
    """The edge encoder on edge features that are never stored (models/GNN.py:299-302 feeding :306): K6 runs as the
    prologue of the K4 launch (gnc_mlp_desc_t.ef_pos, ABI 19).  ``pos`` [N, 2] fp32, ``src`` / ``dst`` int32 [E] (the
    topology's sorted endpoint vectors).  Inference only.  Returns None when no kernel serves the shape this way
    (gnc_mlp_edge_features_supported): the caller then runs ``edge_features`` + ``mlp_forward``; bit-identical either way."""
    lib = load_library()
    if not FOLD_EDGE_FEATURES or pos.dim() != 2 or pos.size(1) != 2 or src.numel() == 0:
        return None
    _require_cuda(pos, src, dst)
    if src.dtype != torch.int32 or dst.dtype != torch.int32:
        raise TypeError("edge endpoint ids must be int32")
    pos = pos.float().contiguous()
    src, dst = src.contiguous(), dst.contiguous()
    rows, sd = src.numel(), pos.size(1)
    if weights[0].size(1) != sd + 1 or rows <= _small_batch_rows(lib):
        return None
    weights = [_rowmajor(w.detach()) for w in weights]  # read where they lie: only the weights-resident kernel serves this form
    biases = [b.contiguous() if b is not None else None for b in biases]
    dev = pos.device
    out = torch.empty(rows, weights[-1].size(0), dtype=torch.float32, device=dev)
    desc = make_mlp_desc([(pos, None, sd + 1, SEG_MATMUL, 0)], weights, biases, ln, activation, act_param, None, out, rows)
    desc.seg[0].ld = 4  # the segment's table is never read (computed rows); ld only has to cover the width
    desc.ef_pos, desc.ef_src, desc.ef_dst = pos.data_ptr(), src.data_ptr(), dst.data_ptr()
    desc.ef_nodes, desc.ef_space_dim = pos.size(0), sd
    if lib.gnc_mlp_edge_features_supported(ctypes.byref(desc)) != 0:
        return None
    flops = 2.0 * rows * sum(w.size(0) * w.size(1) for w in weights)
    with torch.cuda.device(dev):
        _check(_launch(f"mlp_fused_in{weights[0].size(1)}_h{weights[0].size(0)}_out{weights[-1].size(0)}_L{len(weights)}", out,
                       lambda: lib.gnc_mlp_forward_f32(ctypes.byref(desc), _stream(out)), flops), "gnc_mlp_forward_f32")
    return out


READOUT_MAX_HIDDEN, READOUT_MAX_CLASSES = 1024, 64
_readout_tickets: dict = {}


def _readout_ticket(dev) -> torch.Tensor:
    """One zeroed device counter per device AND stream (every launch leaves it at 0; launches on one stream are ordered, launches
    on different streams must not share a counter)."""
    key = (str(dev), torch.cuda.current_stream(dev).cuda_stream)
    if key not in _readout_tickets:
        _readout_tickets[key] = torch.zeros(1, dtype=torch.int32, device=dev)
    return _readout_tickets[key]


def readout_forward(y, w1, b1, w2, b2, w3, b3):
    """logits [C] = fc3(relu(fc2(relu(fc1(y))))) for ONE flattened graph output y [F] in one launch (gnc_readout_forward_f32);
    also returns the two post-ReLU hidden vectors for the backward."""
    lib = load_library()
    _require_cuda(y, w1, w2, w3)
    y = y.contiguous()
    w1, w2, w3 = _rowmajor(w1.detach()), _rowmajor(w2.detach()), _rowmajor(w3.detach())
    dev = y.device
    h1 = torch.empty(w1.size(0), dtype=torch.float32, device=dev)
    h2 = torch.empty(w2.size(0), dtype=torch.float32, device=dev)
    logits = torch.empty(w3.size(0), dtype=torch.float32, device=dev)
    bp = [b.detach().contiguous() if b is not None else None for b in (b1, b2, b3)]
    with torch.cuda.device(dev):
        _check(_launch("readout_forward", y,
                       lambda: lib.gnc_readout_forward_f32(y.data_ptr(), y.numel(), w1.data_ptr(), _ld(w1), bp[0].data_ptr() if bp[0] is not None else None,
                                                           w1.size(0), w2.data_ptr(), _ld(w2), bp[1].data_ptr() if bp[1] is not None else None, w2.size(0),
                                                           w3.data_ptr(), _ld(w3), bp[2].data_ptr() if bp[2] is not None else None, w3.size(0),
                                                           h1.data_ptr(), h2.data_ptr(), logits.data_ptr(), _readout_ticket(dev).data_ptr(), _stream(y)),
                       2.0 * (w1.numel() + w2.numel() + w3.numel())), "gnc_readout_forward_f32")
    return logits, h1, h2


def readout_backward(grad_logits, y, w1, w2, w3, h1, h2, need_dy: bool = True):
    """All gradients of ``readout_forward`` in one launch: (dy or None, dW1, db1, dW2, db2, dW3, db3)."""
    lib = load_library()
    g = grad_logits.contiguous()
    y = y.contiguous()
    w1, w2, w3 = _rowmajor(w1.detach()), _rowmajor(w2.detach()), _rowmajor(w3.detach())
    dev = y.device
    e = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)  # noqa: E731
    dw1, db1, dw2, db2, dw3, db3 = e(*w1.shape), e(w1.size(0)), e(*w2.shape), e(w2.size(0)), e(*w3.shape), e(w3.size(0))
    dy = e(y.numel()) if need_dy else None
    with torch.cuda.device(dev):
        _check(_launch("readout_backward", y,
                       lambda: lib.gnc_readout_backward_f32(g.data_ptr(), y.data_ptr(), y.numel(), w1.data_ptr(), _ld(w1), w1.size(0), w2.data_ptr(),
                                                            _ld(w2), w2.size(0), w3.data_ptr(), _ld(w3), w3.size(0), h1.data_ptr(), h2.data_ptr(),
                                                            dw1.data_ptr(), db1.data_ptr(), dw2.data_ptr(), db2.data_ptr(), dw3.data_ptr(),
                                                            db3.data_ptr(), dy.data_ptr() if dy is not None else None, _stream(y)),
                       4.0 * w1.numel()), "gnc_readout_backward_f32")
    return dy, dw1, db1, dw2, db2, dw3, db3


def dual_projection(x: torch.Tensor, wa: torch.Tensor, wb: torch.Tensor):
    """(x wa^T, x wb^T) over the same rows.  One launch for the small-batch projection shape (gnc_mlp_dual_projection_f32),
    two single-Linear launches otherwise."""
    lib = load_library()
    _require_cuda(x, wa, wb)
    x, wa, wb = _rowmajor(x), _rowmajor(wa.detach()), _rowmajor(wb.detach())
    rows = x.size(0)
    k, m = x.size(1), wa.size(0)
    small = 1 <= rows <= lib.gnc_mlp_small_batch_max_rows() and k == m == 128
    # a large batch at widths <= 64 (c3): the weights-resident kernel's dual instance - the rows are read once
    large = rows > lib.gnc_mlp_small_batch_max_rows() and k <= 64 and 32 < m <= 64 and m % 4 == 0 and _ld(x) % 4 == 0 \
        and x.data_ptr() % 16 == 0
    if ((small or large) and wa.shape == wb.shape and wa.size(1) == k and _ld(wa) == _ld(wb)
            and os.environ.get("GNC_NO_DUAL_PROJECTION") is None):
        oa = torch.empty(rows, m, dtype=torch.float32, device=x.device)
        ob = torch.empty(rows, m, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = _launch(f"mlp_fused_in{k}_h{m}_out{m}_L1x2", oa,
                         lambda: lib.gnc_mlp_dual_projection_f32(x.data_ptr(), _ld(x), rows, wa.data_ptr(), _ld(wa), wb.data_ptr(), _ld(wb),
                                                                 k, m, oa.data_ptr(), ob.data_ptr(), _ld(oa), _stream(x)),
                         2.0 * rows * 2 * k * m)
        if rc == 0:
            return oa, ob
    return mlp_forward([(x, None)], [wa], [None]), mlp_forward([(x, None)], [wb], [None])


def projection_t2(a: torch.Tensor, b: torch.Tensor, w: torch.Tensor, dn: int):
    """a W[:, :dn] + b W[:, dn:2 dn] with W [H, >= 2 dn] as nn.Linear holds it.  One launch on the transposed-read small-batch
    shape (gnc_mlp_projection_t2_f32); otherwise a projection launch over [a | b] with the transposed, stacked weight."""
    lib = load_library()
    _require_cuda(a, b, w)
    a, b, w = _rowmajor(a), _rowmajor(b), _rowmajor(w.detach())
    rows, h = a.size(0), w.size(0)
    if (1 <= rows <= lib.gnc_mlp_small_batch_max_rows() and h == dn == 128 and a.shape == b.shape == (rows, h) and w.size(1) >= 2 * dn
            and os.environ.get("GNC_NO_PROJECTION_T2") is None):
        out = torch.empty(rows, dn, dtype=torch.float32, device=a.device)
        with torch.cuda.device(a.device):
            rc = _launch("mlp_fused_in256_h128_out128_L1t", out,
                         lambda: lib.gnc_mlp_projection_t2_f32(a.data_ptr(), _ld(a), b.data_ptr(), _ld(b), rows, w.data_ptr(), _ld(w), h, dn,
                                                               out.data_ptr(), _ld(out), _stream(a)), 2.0 * rows * 2 * h * dn)
        if rc == 0:
            return out
    wt = torch.cat([w[:, :dn], w[:, dn:2 * dn]], dim=0).t().contiguous()  # [dn, 2h]
    return mlp_forward([(a, None), (b, None)], [wt], [None])


def small_batch_kernel_serves(segments, weights, biases, ln=None, activation: str = "ReLU", residual=None, rows=None,
                              modes=None) -> bool:
    """Would the small-batch (column-split) kernel run this ``mlp_forward`` call as its operands lie
    (gnc_mlp_small_batch_supported)?  The reference's one-graph-per-call regime is served by it."""
    lib = load_library()
    segs, w, b, res, rows, _ = _prepare_mlp(segments, weights, biases, residual, rows, modes, vector_rows=False)
    dummy = torch.empty(max(rows, 1), w[-1].size(0), device=segs[0][0].device)
    desc = make_mlp_desc(segs, w, b, ln, activation, 0.0, res, dummy, rows)
    return lib.gnc_mlp_small_batch_supported(ctypes.byref(desc)) == 0


# --------------------------------------------------------------------------- K8 backward
def mlp_backward_supported(segments, weights, biases, ln, activation, residual, rows, modes=None, saved_act=None) -> bool:
    """Shape query of the HIP backward kernel (ReLU, widths <= 64, 2..7 Linear layers ...).  With ``saved_act`` a small batch
    is asked about as its operands lie (its data kernel needs no padded copies, so the query makes none either)."""
    lib = load_library()
    if activation not in ACTIVATIONS or not (1 <= len(weights) <= GNC_MAX_LINEAR) or len(segments) > GNC_MAX_SEGMENTS:
        return False
    if saved_act and activation == "ReLU" and len(saved_act) == len(weights) - 1 and _rows_of(segments, rows) <= _small_batch_rows(lib):
        segs, w, b, res, rows_, _ = _prepare_mlp(segments, weights, biases, residual, rows, modes, vector_rows=False)
        dummy = torch.empty(1, w[-1].size(0), device=segs[0][0].device)
        desc = make_mlp_desc(segs, w, b, ln, activation, 0.0, res, dummy, rows_)
        for l, a in enumerate(saved_act):
            desc.save_act[l] = a.data_ptr()
        if lib.gnc_mlp_backward_small_batch_supported(ctypes.byref(desc)) == 1:
            return True
    segs, w, b, res, rows, _ = _prepare_mlp(segments, weights, biases, residual, rows, modes)
    dummy = torch.empty(1, w[-1].size(0), device=segs[0][0].device)
    desc = make_mlp_desc(segs, w, b, ln, activation, 0.0, res, dummy, rows)
    return lib.gnc_mlp_backward_supported(ctypes.byref(desc)) == 0


def mlp_backward(segments, weights, biases, ln, grad_out: torch.Tensor | None, rows: int | None = None, modes=None,
                 need_dx: bool = True, residual: torch.Tensor | None = None, fused: bool = True, grad_gather=None,
                 saved_act=None, defer_ln_sums: bool = False):
    """Data path of the MLP backward (see include/gnc_hip.h, K8).  Returns a dict with
    ``act`` (inputs of Linear 1..L-1), ``dz`` (grads of every pre-activation, dz[-1] = pre-LayerNorm),
    ``dx`` ([rows, in_dim0] in weight-column order, or None) and ``yhat`` (or None).  Shapes of the fused kernel
    (``fused=True`` and gnc_mlp_backward_fused_rows() > 0) return ``dw`` / ``db`` (one per Linear; ``dw[0]`` covers the
    columns of the MATMUL segment) instead of ``act`` / ``dz[1:]``.

    ``grad_gather = (table [N, out_dim], index int32 [rows])``: the gradient of output row r is
    ``(grad_out[r] if grad_out is not None else 0) + table[index[r]]`` - the backward of a scatter-sum that consumed the
    output rows (models/GNN.py:99).  Kernels that honour it (gnc_mlp_backward_grad_gather_honoured) gather inside the
    launch; otherwise the rows are gathered here first (K2).  ``r["grad_out"]`` is the effective row-ordered gradient when
    it had to be materialised, else None.

    ``defer_ln_sums``: leave the per-worker LayerNorm partial sums unsummed in ``r["ln_part"]`` ([P, 2 * out_dim]: d beta |
    d gamma) for the caller to fold into its ``xty_multi`` launch (split path only).

    ``saved_act``: the list ``mlp_forward(save_act=...)`` filled for the same call; kernels that honour it
    (gnc_mlp_backward_saved_act_honoured) read the post-activations instead of recomputing them."""
    lib = load_library()
    given = (segments, weights, biases, residual, rows, modes)
    bd = MlpBwdDesc()
    unpadded = False
    if saved_act and len(saved_act) == len(weights) - 1 and _rows_of(segments, rows) <= _small_batch_rows(lib):
        # a small batch with saved activations: its data kernel reads the operands where they lie (no padded copies of the
        # reference's 3-column inputs / first-layer weights: two pad launches each)
        segs, w, b, residual, rows, _ = _prepare_mlp(*given, vector_rows=False)
        dummy = torch.empty(1, w[-1].size(0), device=segs[0][0].device)
        bd.fwd = make_mlp_desc(segs, w, b, ln, "ReLU", 0.0, None, dummy, rows)
        for l, a in enumerate(saved_act):
            bd.fwd.save_act[l] = a.data_ptr()
        unpadded = lib.gnc_mlp_backward_small_batch_supported(ctypes.byref(bd.fwd)) == 1
    if not unpadded:
        segs, w, b, residual, rows, _ = _prepare_mlp(*given)
        dummy = torch.empty(1, w[-1].size(0), device=segs[0][0].device)
        bd.fwd = make_mlp_desc(segs, w, b, ln, "ReLU", 0.0, None, dummy, rows)
    dev = segs[0][0].device
    # residual given: ask the kernel to fold its gradient (grad_out) into the dx of the segment it came from
    mm = [sg for sg in segs if sg[3] == SEG_MATMUL]
    fold = (residual is not None and need_dx and mm[-1][1] is None and mm[-1][0].data_ptr() == residual.data_ptr()
            and mm[-1][2] == w[-1].size(0) and lib.gnc_mlp_backward_dx_add_honoured(ctypes.byref(bd.fwd)) == 1)
    bd.dx_add_grad_out = 1 if fold else 0
    n_lin = len(w)
    frows = lib.gnc_mlp_backward_fused_rows(ctypes.byref(bd.fwd)) if fused else 0
    parts = None
    if frows > 0:
        mk = [(w[l].size(0), w[l].size(1)) for l in range(n_lin)]
        parts = [torch.empty(frows, m * k + m, dtype=torch.float32, device=dev) for m, k in mk]
        for l in range(n_lin):
            bd.dw_partial[l] = parts[l].data_ptr()
    if saved_act:
        for l, a in enumerate(saved_act):
            bd.act[l] = a.data_ptr()
        bd.act_given = 1
        bd.dx = 1 if need_dx else None  # (the query looks at whether dx is wanted; the real pointer follows below)
        if len(saved_act) != n_lin - 1 or lib.gnc_mlp_backward_saved_act_honoured(ctypes.byref(bd)) != 1:
            for l in range(len(saved_act)):
                bd.act[l] = None
            bd.act_given = 0
        bd.dx = None
        if bd.act_given:  # the shape queries that only see the forward description learn about the saved rows this way
            for l, a in enumerate(saved_act):
                bd.fwd.save_act[l] = a.data_ptr()
            fold = (residual is not None and need_dx and mm[-1][1] is None and mm[-1][0].data_ptr() == residual.data_ptr()
                    and mm[-1][2] == w[-1].size(0) and lib.gnc_mlp_backward_dx_add_honoured(ctypes.byref(bd.fwd)) == 1)
            bd.dx_add_grad_out = 1 if fold else 0
    gt = gi = None
    if grad_gather is not None:
        gt, gi = _vector_rows(_rowmajor(grad_gather[0])), grad_gather[1]
        bd.grad_gather, bd.ld_grad_gather = gt.data_ptr(), _ld(gt)
        bd.grad_gather_index, bd.grad_gather_rows = gi.data_ptr(), gt.size(0)
        # the residual's gradient is the EFFECTIVE output gradient: when the kernel cannot fold it into dx itself (the
        # residual went through a padded copy: width % 4 != 0), the caller adds it and needs the rows as a tensor
        res_fold = (mm[-1][1] is None and residual is not None and mm[-1][0].data_ptr() == residual.data_ptr()
                    and mm[-1][2] == w[-1].size(0))
        caller_adds_residual = residual is not None and need_dx and not res_fold
        if caller_adds_residual or lib.gnc_mlp_backward_grad_gather_honoured(ctypes.byref(bd)) != 1:  # gather in front of the launch (K2)
            grad_out = gather_rows(gt, gi) if grad_out is None else gather_rows_add(gt, gi, _rowmajor(grad_out))
            bd.grad_gather = bd.grad_gather_index = None
            bd.ld_grad_gather = bd.grad_gather_rows = 0
            gt = gi = None
    # (the small-batch data kernel reads grad_out rows of any width where they lie: no padded copy of the decoder's [rows, 1])
    g = (_rowmajor(grad_out) if unpadded else _vector_rows(_rowmajor(grad_out))) if grad_out is not None else None
    if g is not None:
        bd.grad_out, bd.ld_grad_out = g.data_ptr(), _ld(g)
    g_sum = None
    g_eff = g if gt is None else None   # the effective output gradient as a tensor (None: part of it is gathered in the kernel)
    launch_ref = g if g is not None else gt
    if frows > 0:
        # fused data + weight-gradient kernel: per-wave partials of dW_l / db_l (and of the LayerNorm sums) instead of
        # the a_l / dz_l / y_hat tensors; only dz_0 (gradient of gathered ADD segments) and dx are written
        bd.dx_add_grad_out = 1 if (residual is not None and need_dx and mm[-1][1] is None
                                   and mm[-1][0].data_ptr() == residual.data_ptr() and mm[-1][2] == w[-1].size(0)) else 0
        if gt is not None and g is not None and bd.dx_add_grad_out:  # scratch for the summed rows (see gnc_hip.h, grad_sum)
            g_sum = torch.empty(rows, w[-1].size(0), dtype=torch.float32, device=dev)
            bd.grad_sum, bd.ld_grad_sum = g_sum.data_ptr(), _ld(g_sum)
        dz0 = torch.empty(rows, w[0].size(0), dtype=torch.float32, device=dev) if len(segs) > 1 else None
        if dz0 is not None:
            bd.dz[0] = dz0.data_ptr()
        dx = torch.empty(rows, w[0].size(1), dtype=torch.float32, device=dev) if need_dx else None
        if dx is not None:
            bd.dx, bd.ld_dx = dx.data_ptr(), _ld(dx)
        ln_part = None
        if ln is not None:
            ln_part = torch.empty(frows, 2 * w[-1].size(0), dtype=torch.float32, device=dev)
            bd.ln_partial = ln_part.data_ptr()
        flops = 2.0 * rows * (3 * sum(x.size(0) * x.size(1) for x in w))
        with torch.cuda.device(dev):
            _check(_launch(f"mlp_backward_fused_in{w[0].size(1)}_h{w[0].size(0)}_out{w[-1].size(0)}_L{n_lin}", launch_ref,
                           lambda: lib.gnc_mlp_backward_f32(ctypes.byref(bd), _stream(launch_ref)), flops), "gnc_mlp_backward_f32")
        dws, dbs = [], []
        for (m, k), part in zip(mk, parts):
            tot = part.sum(dim=0)  # fixed order: reproducible
            dws.append(tot[:m * k].view(m, k))
            dbs.append(tot[m * k:])
        ln_sums = None
        if ln_part is not None:
            tot = ln_part.sum(dim=0)
            ln_sums = (tot[:w[-1].size(0)], tot[w[-1].size(0):])
        return {"act": None, "dz": [dz0] + [None] * (n_lin - 1), "dx": dx, "yhat": None, "ln_sums": ln_sums, "dw": dws, "db": dbs,
                "residual_folded": bool(bd.dx_add_grad_out), "grad_out": g_eff, "saved_act_used": bool(bd.act_given),
                "_keep": (segs, w, b, g, gt, gi, g_sum, saved_act)}
    # split path: the data kernel emits a_l (unless the forward saved them: act_given) and dz_l for gnc_xty_f32
    act = list(saved_act) if bd.act_given else [torch.empty(rows, w[l].size(0), dtype=torch.float32, device=dev) for l in range(n_lin - 1)]
    dz = [torch.empty(rows, w[l].size(0), dtype=torch.float32, device=dev) for l in range(n_lin)]
    for l in range(n_lin):
        bd.dz[l] = dz[l].data_ptr()
        if l < n_lin - 1:
            bd.act[l] = act[l].data_ptr()
    dx = torch.empty(rows, w[0].size(1), dtype=torch.float32, device=dev) if need_dx else None
    if dx is not None:
        bd.dx, bd.ld_dx = dx.data_ptr(), _ld(dx)
    yhat = ln_part = None
    if ln is not None:
        prow = lib.gnc_mlp_backward_ln_partial_rows(ctypes.byref(bd.fwd))
        if prow > 0:  # d gamma / d beta sums formed inside the data kernel: no y_hat tensor, no colsum pass
            ln_part = torch.empty(prow, 2 * w[-1].size(0), dtype=torch.float32, device=dev)
            bd.ln_partial = ln_part.data_ptr()
        else:
            yhat = torch.empty(rows, w[-1].size(0), dtype=torch.float32, device=dev)
            bd.yhat = yhat.data_ptr()
    flops = 2.0 * rows * (2 * sum(x.size(0) * x.size(1) for x in w))
    with torch.cuda.device(dev):
        _check(_launch(f"mlp_backward_in{w[0].size(1)}_h{w[0].size(0)}_out{w[-1].size(0)}_L{n_lin}", launch_ref,
                       lambda: lib.gnc_mlp_backward_f32(ctypes.byref(bd), _stream(launch_ref)), flops), "gnc_mlp_backward_f32")
    ln_sums = None
    if ln_part is not None and not defer_ln_sums:
        tot = ln_part.sum(dim=0)  # fixed order: reproducible
        ln_sums = (tot[:w[-1].size(0)], tot[w[-1].size(0):])  # (d beta, d gamma)
    return {"act": act, "dz": dz, "dx": dx, "yhat": yhat, "ln_sums": ln_sums, "ln_part": ln_part if defer_ln_sums else None,
            "residual_folded": bool(bd.dx_add_grad_out), "grad_out": g_eff,
            "saved_act_used": bool(bd.act_given), "_keep": (segs, w, b, g, gt, gi, saved_act)}


def _xty_spans(n: int, step: int):
    return [(i, min(step, n - i)) for i in range(0, n, step)]


def xty(a: torch.Tensor, b: torch.Tensor):
    """(A^T B [M, K], column sums of A [M]) over the rows; A [rows, M], B [rows, K] fp32.  Blocks of up to
    256 x 256 per launch when both operands are wider than 64 columns (workgroup-shared row tiles), 128 x 128 next to
    a narrow operand; the per-worker partials are summed in a fixed order (bitwise reproducible)."""
    lib = load_library()
    _require_cuda(a, b)
    a, b = _rowmajor(a), _rowmajor(b)
    rows, m, k = a.size(0), a.size(1), b.size(1)
    dev = a.device
    c = torch.empty(m, k, dtype=torch.float32, device=dev)
    colsum = torch.empty(m, dtype=torch.float32, device=dev)
    blocks = []
    for m0, mm in _xty_spans(m, 256 if k > 64 else 128):
        for k0, kk in _xty_spans(k, 256 if mm > 64 else 128):
            if kk <= 64 and mm > 128:  # a narrow remainder of k next to a wide block of m: the per-wave kernel's limit
                blocks += [(m1 + m0, mm1, k0, kk) for m1, mm1 in _xty_spans(mm, 128)]
            else:
                blocks.append((m0, mm, k0, kk))
    with torch.cuda.device(dev):
        for m0, mm, k0, kk in blocks:
            p = lib.gnc_xty_partials_for(rows, mm, kk)
            part = torch.empty(p, mm * kk + mm, dtype=torch.float32, device=dev)
            av, bv = a[:, m0:m0 + mm], b[:, k0:k0 + kk]
            _check(_launch("xty", av, lambda: lib.gnc_xty_f32(av.data_ptr(), _ld(av), bv.data_ptr(), _ld(bv), rows, mm,
                                                              kk, part.data_ptr(), p, _stream(av)),
                           2.0 * rows * mm * kk), "gnc_xty_f32")
            # fixed-order sum of the partials straight into the block's place (one launch, no sum + copies)
            cblk = c[m0:m0 + mm, k0:k0 + kk]
            _check(lib.gnc_reduce_partials_f32(part.data_ptr(), p, mm * kk + mm, mm, kk, cblk.data_ptr(), c.stride(0),
                                               colsum[m0:m0 + mm].data_ptr() if k0 == 0 else None, _stream(av)),
                   "gnc_reduce_partials_f32")
    return c, colsum


def xty_multi(products, row_sums=()):
    """Several weight-gradient products at once.  ``products``: list of ``(a [rows, M], b [rows, K], out)`` with ``out`` an
    [M, K] view to write into (any row stride: a column block of a wider gradient) or None for a fresh tensor; returns the
    list of ``(a^T b, column sums of a)``.  ``row_sums``: list of [P, W] tensors (per-tile partial sums); their row sums
    [W] are returned as a second list.  Small batches (every operand within gnc_xty_small_max_rows rows: the reference's
    one-graph-per-step regime) run as ONE launch per 8 jobs (gnc_xty_small_f32); anything else goes product by product
    through ``xty``."""
    lib = load_library()
    products = [(_rowmajor(a), _rowmajor(b), out) for a, b, out in products]
    row_sums = [_rowmajor(p) for p in row_sums]
    lim = lib.gnc_xty_small_max_rows()
    small = (all(a.size(0) <= lim and a.size(0) >= 1 and a.size(1) <= 4096 and b.size(1) <= 4096 for a, b, _ in products)
             and all(1 <= p.size(0) <= lim for p in row_sums) and (products or row_sums))
    if not small:
        res = []
        for a, b, out in products:
            c, cs = xty(a, b)
            if out is not None:
                out.copy_(c)
                c = out
            res.append((c, cs))
        return res, [p.sum(dim=0) for p in row_sums]
    dev = (products[0][0] if products else row_sums[0]).device
    _require_cuda(*[t for a, b, _ in products for t in (a, b)], *row_sums)
    jobs, res, sums = [], [], []
    for a, b, out in products:
        m, k = a.size(1), b.size(1)
        if a.size(0) != b.size(0):
            raise ValueError("xty_multi: operands of one product must have the same number of rows")
        if out is None:
            out = torch.empty(m, k, dtype=torch.float32, device=dev)
        elif out.shape != (m, k) or out.stride(1) != 1 or out.dtype != torch.float32:
            raise ValueError("xty_multi: out must be a float32 [M, K] view with unit column stride")
        cs = torch.empty(m, dtype=torch.float32, device=dev)
        j = XtyJob()
        j.a, j.lda, j.b, j.ldb, j.rows, j.m, j.k = a.data_ptr(), _ld(a), b.data_ptr(), _ld(b), a.size(0), m, k
        j.dw, j.ld_dw, j.db, j.kind = out.data_ptr(), out.stride(0), cs.data_ptr(), 0
        jobs.append(j)
        res.append((out, cs))
    for p in row_sums:
        o = torch.empty(p.size(1), dtype=torch.float32, device=dev)
        j = XtyJob()
        j.a, j.lda, j.rows, j.m, j.dw, j.kind = p.data_ptr(), _ld(p), p.size(0), p.size(1), o.data_ptr(), 1
        jobs.append(j)
        sums.append(o)
    with torch.cuda.device(dev):
        for q in range(0, len(jobs), GNC_XTY_MAX_JOBS):
            chunk = jobs[q:q + GNC_XTY_MAX_JOBS]
            arr = (XtyJob * len(chunk))(*chunk)
            flops = sum(2.0 * j.rows * j.m * j.k for j in chunk if j.kind == 0)
            _check(_launch("xty_small", res[0][0] if res else sums[0],
                           lambda: lib.gnc_xty_small_f32(ctypes.byref(arr), len(chunk), torch.cuda.current_stream(dev).cuda_stream),
                           flops), "gnc_xty_small_f32")
    return res, sums


def colsum_pair(g: torch.Tensor, y: torch.Tensor):
    """(column sums of G, column sums of G*Y): d beta and d gamma of a LayerNorm."""
    lib = load_library()
    g, y = _vector_rows(_rowmajor(g)), _vector_rows(_rowmajor(y))
    rows, width = g.shape
    both = torch.empty(2, width, dtype=torch.float32, device=g.device)  # row 0: colsum(G), row 1: colsum(G * Y)
    with torch.cuda.device(g.device):
        p = lib.gnc_xty_partials(rows)
        for c0 in range(0, width, 64):  # 64-column slabs (a wide LayerNorm is a few launches over column slices)
            w = min(64, width - c0)
            gs, ys = g[:, c0:c0 + w], y[:, c0:c0 + w]
            part = torch.empty(p, 2 * w, dtype=torch.float32, device=g.device)
            _check(lib.gnc_colsum_pair_f32(gs.data_ptr(), _ld(g), ys.data_ptr(), _ld(y), rows, w, part.data_ptr(), p,
                                           _stream(g)), "gnc_colsum_pair_f32")
            _check(lib.gnc_reduce_partials_f32(part.data_ptr(), p, 2 * w, 2, w, both[:, c0:c0 + w].data_ptr(), width, None,
                                               _stream(g)), "gnc_reduce_partials_f32")
    return both[0], both[1]


# --------------------------------------------------------------------------- activations other than ReLU: backward pieces
def activation(z: torch.Tensor, name: str, param: float = 0.0) -> torch.Tensor:
    """act(z) elementwise (gnc_activation_f32)."""
    lib = load_library()
    _require_cuda(z)
    z = _rowmajor(z)
    out = torch.empty(z.size(0), z.size(1), dtype=torch.float32, device=z.device)
    with torch.cuda.device(z.device):
        _check(lib.gnc_activation_f32(z.data_ptr(), _ld(z), z.size(0), z.size(1), ACTIVATIONS[name], param, out.data_ptr(), _ld(out),
                                      _stream(z)), "gnc_activation_f32")
    return out


def activation_backward(z: torch.Tensor, grad_act: torch.Tensor, name: str, param: float = 0.0) -> torch.Tensor:
    """grad_act * act'(z) (gnc_activation_backward_f32)."""
    lib = load_library()
    _require_cuda(z, grad_act)
    z, grad_act = _rowmajor(z), _rowmajor(grad_act)
    out = torch.empty(z.size(0), z.size(1), dtype=torch.float32, device=z.device)
    with torch.cuda.device(z.device):
        _check(lib.gnc_activation_backward_f32(z.data_ptr(), _ld(z), grad_act.data_ptr(), _ld(grad_act), z.size(0), z.size(1),
                                               ACTIVATIONS[name], param, out.data_ptr(), _ld(out), _stream(z)),
               "gnc_activation_backward_f32")
    return out


def layer_norm_backward(y: torch.Tensor, gamma: torch.Tensor, grad_out: torch.Tensor, eps: float):
    """(grad_y, y_hat) of out = LayerNorm(y) * gamma + beta (gnc_layer_norm_backward_f32)."""
    lib = load_library()
    _require_cuda(y, gamma, grad_out)
    y, grad_out = _rowmajor(y), _rowmajor(grad_out)
    gy = torch.empty(y.size(0), y.size(1), dtype=torch.float32, device=y.device)
    yhat = torch.empty_like(gy)
    with torch.cuda.device(y.device):
        _check(lib.gnc_layer_norm_backward_f32(y.data_ptr(), _ld(y), gamma.contiguous().data_ptr(), grad_out.data_ptr(), _ld(grad_out),
                                               y.size(0), y.size(1), eps, gy.data_ptr(), _ld(gy), yhat.data_ptr(), _ld(yhat),
                                               _stream(y)), "gnc_layer_norm_backward_f32")
    return gy, yhat


# --------------------------------------------------------------------------- fused Adam
def adam_step(param: torch.Tensor, grad: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, step: torch.Tensor,
              scratch: torch.Tensor, lr: float, beta1: float, beta2: float, eps: float, weight_decay: float = 0.0) -> None:
    """One Adam update of the flat fp32 buffer ``param`` from ``grad`` (see include/gnc_hip.h); ``step`` is a device
    int64 [1] counter advanced by the call, ``scratch`` a device float32 [2]."""
    lib = load_library()
    _require_cuda(param, grad, exp_avg, exp_avg_sq, step, scratch)
    n = param.numel()
    for t in (param, grad, exp_avg, exp_avg_sq):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != n:
            raise ValueError("adam_step: flat contiguous float32 buffers of one size expected")
    if step.dtype != torch.int64 or scratch.dtype != torch.float32 or scratch.numel() < 2:
        raise ValueError("adam_step: step must be int64 [1], scratch float32 [2]")
    with torch.cuda.device(param.device):
        _check(_launch("adam_flat", param, lambda: lib.gnc_adam_step_f32(
            param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), n, lr, beta1, beta2, eps, weight_decay,
            step.data_ptr(), scratch.data_ptr(), _stream(param)), 28.0 * n), "gnc_adam_step_f32")
