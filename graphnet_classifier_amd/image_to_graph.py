"""Device-side image -> graph builders (drop-in counterparts of the reference's ``utils/image_to_graph``).

Same function names and arguments as the reference
(``image_to_graph_optimized.py:7,42,50``, ``image_to_graph_patch.py:6``, ``image_to_graph_superpixel.py:8``);
the image is decoded / resized on the host with PIL exactly as the reference does, then everything else
(node features, positions, edges) is produced in HBM by the kernels of ``csrc/graph_build.hip``.  Results
are the tensors ``utils/dataloader.py:49-51`` would build: ``x`` float32, ``pos`` float32, ``edge_index``
int64, already on the GPU, in the reference's node and edge order.
"""
from __future__ import annotations

import functools

import numpy as np
import torch

from . import native
from .MLP import default_device


def _device() -> torch.device:
    dev = default_device()
    if dev.type != "cuda":
        raise RuntimeError("image_to_graph: no GPU visible and no CPU fallback exists")
    return dev


def _load_resized(image_or_path, resize_value: int) -> np.ndarray:
    """optimized.py:65-70 / patch.py:18-24 / superpixel.py:21-26: PIL RGB, resize, uint8 [H, W, 3]."""
    from PIL import Image
    image = Image.open(image_or_path).convert("RGB") if isinstance(image_or_path, str) else image_or_path.convert("RGB")
    return np.array(image.resize((resize_value, resize_value)))


def create_grid_edges_optimized(H: int, W: int, diagonals: bool = False) -> torch.Tensor:
    """int64 [2, E] on the GPU; same edge order as optimized.py:7-39."""
    lib = native.load_library()
    dev = _device()
    e = lib.gnc_grid_num_edges(H, W, int(bool(diagonals)))
    ei = torch.empty(2, e, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        native._check(lib.gnc_grid_edges_i64(H, W, int(bool(diagonals)), ei.data_ptr(),
                                             torch.cuda.current_stream(dev).cuda_stream), "gnc_grid_edges_i64")
    return ei


@functools.lru_cache(maxsize=128)
def get_cached_edge_index(resize_value: int, diagonals: bool) -> torch.Tensor:
    """optimized.py:42-47: one topology per image size."""
    return create_grid_edges_optimized(resize_value, resize_value, diagonals)


def _to_device_u8(img: np.ndarray) -> torch.Tensor:
    if img.dtype != np.uint8 or img.ndim != 3:
        raise ValueError("expected a uint8 [H, W, C] image")
    return torch.from_numpy(np.ascontiguousarray(img)).to(_device())


def pixel_graph_from_array(img_u8: np.ndarray, diagonals: bool = False, use_cache: bool = True):
    lib = native.load_library()
    img = _to_device_u8(img_u8)
    H, W, C = img.shape
    x = torch.empty(H * W, C, dtype=torch.float32, device=img.device)
    pos = torch.empty(H * W, 2, dtype=torch.float32, device=img.device)
    with torch.cuda.device(img.device):
        native._check(lib.gnc_pixel_nodes_f32(img.data_ptr(), H, W, C, x.data_ptr(), pos.data_ptr(),
                                              torch.cuda.current_stream(img.device).cuda_stream), "gnc_pixel_nodes_f32")
    ei = get_cached_edge_index(H, bool(diagonals)) if (use_cache and H == W) else create_grid_edges_optimized(H, W, diagonals)
    return x, pos, ei


def image_to_graph_pixel_optimized(image_or_path, resize_value: int = 128, diagonals: bool = False, use_cache: bool = True):
    """optimized.py:50-87."""
    return pixel_graph_from_array(_load_resized(image_or_path, resize_value), diagonals, use_cache)


def patch_graph_from_array(img_u8: np.ndarray, patch_size: int = 8):
    lib = native.load_library()
    img = _to_device_u8(img_u8)
    H, W, C = img.shape
    nh, nw = H // patch_size, W // patch_size
    x = torch.empty(nh * nw, C, dtype=torch.float32, device=img.device)
    pos = torch.empty(nh * nw, 2, dtype=torch.float32, device=img.device)
    with torch.cuda.device(img.device):
        native._check(lib.gnc_patch_nodes_f32(img.data_ptr(), H, W, C, patch_size, x.data_ptr(), pos.data_ptr(),
                                              torch.cuda.current_stream(img.device).cuda_stream), "gnc_patch_nodes_f32")
    return x, pos, create_grid_edges_optimized(nh, nw, False)


def image_to_graph_patch(image_or_path, resize_value: int = 128, patch_size: int = 8):
    """patch.py:6-54."""
    return patch_graph_from_array(_load_resized(image_or_path, resize_value), patch_size)


def superpixel_graph_from_labels(img_u8: np.ndarray, segments: np.ndarray):
    """Everything of superpixel.py after the SLIC call (:33-71) for a given label image: per-segment mean
    colour (of img/255) and centroid, region adjacency, edges [i,j],[j,i] in lexicographic order."""
    lib = native.load_library()
    img = _to_device_u8(img_u8)
    H, W, C = img.shape
    if C != 3 or segments.shape != (H, W):
        raise ValueError("expected an RGB image and a label image of the same size")
    labels = torch.from_numpy(np.ascontiguousarray(segments.astype(np.int32))).to(img.device)
    n = H * W
    x = torch.empty(n, 3, dtype=torch.float32, device=img.device)
    pos = torch.empty(n, 2, dtype=torch.float32, device=img.device)
    ei = torch.empty(2, 4 * n, dtype=torch.int64, device=img.device)
    counts = torch.empty(3, dtype=torch.int32, device=img.device)
    with torch.cuda.device(img.device):
        nbytes = lib.gnc_rag_workspace_bytes(H, W)
        if nbytes == 0:
            native._check(-1, "gnc_rag_workspace_bytes")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=img.device)
        native._check(lib.gnc_rag_build(labels.data_ptr(), img.data_ptr(), H, W, x.data_ptr(), pos.data_ptr(), ei.data_ptr(),
                                        4 * n, counts.data_ptr(), ws.data_ptr(), nbytes,
                                        torch.cuda.current_stream(img.device).cuda_stream), "gnc_rag_build")
    s, e, bad = (int(v) for v in counts.tolist())  # one host sync: the sizes are data dependent
    if bad:
        raise ValueError("label image has values outside [0, H*W)")
    return x[:s], pos[:s], ei[:, :e]


def image_to_graph_superpixel(image_or_path, resize_value: int = 128, n_segments: int = 100, compactness: int = 10):
    """superpixel.py:8-73.  The SLIC segmentation itself is scikit-image's (superpixel.py:31) and stays on the
    host; it is imported lazily because that package is optional."""
    try:
        from skimage.segmentation import slic
        from skimage.util import img_as_float
    except ImportError as exc:  # same dependency the reference has
        raise ImportError("image_to_graph_superpixel needs scikit-image for SLIC (reference superpixel.py:4-5); "
                          "use superpixel_graph_from_labels with your own label image") from exc
    img = _load_resized(image_or_path, resize_value)
    segments = slic(img_as_float(img), n_segments=n_segments, compactness=compactness, start_label=0)
    return superpixel_graph_from_labels(img, segments)
