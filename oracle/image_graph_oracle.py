"""CPU oracle of the image -> graph builders that feed the hot path.  TEST INFRASTRUCTURE ONLY.

Restates (numpy, same arithmetic) the reference's
  utils/image_to_graph/image_to_graph_optimized.py:50-87  (pixel graph)
  utils/image_to_graph/image_to_graph_patch.py:6-54       (patch graph)
  utils/image_to_graph/image_to_graph_superpixel.py:33-71 (superpixel graph, everything AFTER the SLIC call)
on an already resized uint8 RGB array, in the tensor format utils/dataloader.py:49-51 hands to the model
(x float32, pos float32, edge_index int64).

Parity status: pixel and patch builders are PINNED by tests/golden/g7_image_graphs.npz (captured by running
the reference's own functions, tests/golden/make_golden.py).  The superpixel builder cannot be imported
(scikit-image is not installed; SLIC itself is out of reach), so its post-SLIC part is restated from the
source with the same scipy primitive the reference calls (scipy.ndimage.binary_dilation) and checked on
synthetic label images only: "parity unpinned" for that function.
"""
import numpy as np
from scipy.ndimage import binary_dilation

from oracle.graphnet_oracle import grid_edge_index


def pixel_graph(img_u8: np.ndarray, diagonals: bool = False):
    """optimized.py:69-87: x = pixels (raw 0..255), pos = (row, col), one-directional grid edges."""
    H, W, C = img_u8.shape
    x = img_u8.reshape(H * W, C).astype(np.float32)
    rows, cols = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    pos = np.stack([rows.flatten(), cols.flatten()], axis=1).astype(np.float32)
    return x, pos, grid_edge_index(H, W, diagonals)


def patch_graph(img_u8: np.ndarray, patch_size: int = 8):
    """patch.py:24-52: mean RGB (float64 mean of uint8) per patch, integer patch centres, grid over patches."""
    H, W, C = img_u8.shape
    nh, nw = H // patch_size, W // patch_size
    feats, poss = [], []
    for i in range(nh):
        for j in range(nw):
            patch = img_u8[i * patch_size:(i + 1) * patch_size, j * patch_size:(j + 1) * patch_size]
            feats.append(np.mean(patch, axis=(0, 1)))
            poss.append([i * patch_size + patch_size // 2, j * patch_size + patch_size // 2])
    return (np.array(feats, dtype=np.float64).astype(np.float32), np.array(poss).astype(np.float32),
            grid_edge_index(nh, nw, False))


def superpixel_graph_from_labels(img_u8: np.ndarray, segments: np.ndarray):
    """superpixel.py:28,33-71 with `segments` given: mean RGB of img/255 and centroid (y, x) per unique
    label; region adjacency by 4-connected dilation; each adjacent pair i<j emitted as [i,j],[j,i]."""
    img = img_u8.astype(np.float64) / 255.0  # skimage.util.img_as_float on uint8
    uniq = np.unique(segments)
    feats, poss = [], []
    for s in uniq:
        mask = segments == s
        feats.append(np.mean(img[mask], axis=0))
        ys, xs = np.where(mask)
        poss.append([np.mean(ys), np.mean(xs)])
    edges = []
    n = len(uniq)
    masks = [segments == s for s in uniq]
    dil = [binary_dilation(m) for m in masks]
    for i in range(n):
        for j in range(i + 1, n):
            if np.any(dil[i] & masks[j]):
                edges.append([i, j])
                edges.append([j, i])
    ei = np.array(edges, dtype=np.int64).T if edges else np.empty((2, 0), dtype=np.int64)
    return np.array(feats).astype(np.float32), np.array(poss).astype(np.float32), ei
