"""GPU parity of the device-side graph builders (csrc/graph_build.hip) against the golden vectors captured
from the reference's own pixel / patch builders (G6, G7) and against the oracle restatement of the
superpixel builder's post-SLIC part (unpinned: scikit-image is absent, see oracle/image_graph_oracle.py)."""
import numpy as np
import pytest
import torch

from oracle import image_graph_oracle as IO
from tests._util import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def I2G():
    from graphnet_classifier_amd import image_to_graph
    return image_to_graph


def test_grid_edges_match_reference_goldens(I2G):
    g = load_golden("g6_grid_edges.npz")
    for key, ref in g.items():
        _, hw, diag = key.split("_")
        h, w = (int(v) for v in hw.split("x"))
        ei = I2G.create_grid_edges_optimized(h, w, diag == "diag")
        assert ei.dtype == torch.int64 and np.array_equal(ei.cpu().numpy(), ref), key
    assert I2G.create_grid_edges_optimized(1, 1).shape == (2, 0)
    assert I2G.get_cached_edge_index(32, False) is I2G.get_cached_edge_index(32, False)  # lru_cache like optimized.py:42


@pytest.mark.parametrize("tag", ["muffin32", "chihuahua64"])
def test_pixel_and_patch_graphs_match_reference_goldens(I2G, tag):
    g = load_golden("g7_image_graphs.npz")
    img = g[f"{tag}/img"]
    for diag in (False, True):
        d = "diag" if diag else "nodiag"
        x, pos, ei = I2G.pixel_graph_from_array(img, diag)
        assert np.array_equal(x.cpu().numpy(), g[f"{tag}/pixel_{d}/x"])
        assert np.array_equal(pos.cpu().numpy(), g[f"{tag}/pixel_{d}/pos"])
        assert np.array_equal(ei.cpu().numpy(), g[f"{tag}/pixel_{d}/edge_index"])
    x, pos, ei = I2G.patch_graph_from_array(img, 8)
    assert np.array_equal(x.cpu().numpy(), g[f"{tag}/patch/x"])
    assert np.array_equal(pos.cpu().numpy(), g[f"{tag}/patch/pos"])
    assert np.array_equal(ei.cpu().numpy(), g[f"{tag}/patch/edge_index"])


def _voronoi_labels(rng, size, nseg, gaps=False):
    pts = rng.random((nseg, 2)) * size
    yy, xx = np.meshgrid(np.arange(size), np.arange(size), indexing="ij")
    d = (yy[..., None] - pts[:, 0]) ** 2 + (xx[..., None] - pts[:, 1]) ** 2
    lab = d.argmin(-1).astype(np.int32)
    return lab * 3 + 1 if gaps else lab  # non-contiguous label values exercise the np.unique compaction


@pytest.mark.parametrize("size,nseg,gaps", [(8, 5, False), (32, 40, False), (32, 100, True), (64, 100, False)])
def test_superpixel_graph_from_labels_matches_oracle(I2G, size, nseg, gaps):
    rng = np.random.default_rng(size * 1000 + nseg)
    img = rng.integers(0, 256, size=(size, size, 3), dtype=np.uint8)
    seg = _voronoi_labels(rng, size, nseg, gaps)
    x, pos, ei = I2G.superpixel_graph_from_labels(img, seg)
    rx, rpos, rei = IO.superpixel_graph_from_labels(img, seg)
    assert x.shape == rx.shape and ei.shape == rei.shape
    assert np.array_equal(ei.cpu().numpy(), rei)               # same pairs, same order
    assert np.array_equal(pos.cpu().numpy(), rpos)             # integer sums / count: exact
    assert float(np.abs(x.cpu().numpy() - rx).max()) <= 6e-8   # float64 mean of k/255 vs exact integer sum / 255: <= 1 ulp


def test_superpixel_graph_rejects_bad_labels(I2G):
    img = np.zeros((4, 4, 3), dtype=np.uint8)
    seg = np.zeros((4, 4), dtype=np.int32)
    seg[1, 1] = 99
    with pytest.raises(ValueError):
        I2G.superpixel_graph_from_labels(img, seg)


def test_builders_feed_the_model(I2G):
    """End to end on the device: shipped-image pixel graph -> CombinedModel logits, no host round trip."""
    from graphnet_classifier_amd.GNN import CombinedModel, GraphNet
    g = load_golden("g7_image_graphs.npz")
    x, pos, ei = I2G.pixel_graph_from_array(g["muffin32/img"])
    torch.manual_seed(0)
    m = CombinedModel(GraphNet(num_local_features=3, space_dim=2, out_channels=1, n_blocks=2, out_dim_node=32, out_dim_edge=32,
                               hidden_dim_node=32, hidden_dim_edge=32, hidden_dim_decoder=32, hidden_dim_processor_node=32,
                               hidden_dim_processor_edge=32), num_nodes=1024, classes=2)
    with torch.no_grad():
        logits = m((x, pos, ei))
    assert logits.is_cuda and logits.shape == (2,) and bool(torch.isfinite(logits).all())
