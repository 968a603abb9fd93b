"""CPU-only tests of the host side: C-ABI exports, ctypes struct layout, module surface and
state-dict compatibility, sharding, synthetic generators, loud failure without a GPU."""
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from tests._util import load_golden, sub_state_dict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_symbol_the_header_declares():
    from graphnet_classifier_amd import native
    lib = native.load_library()
    header = open(os.path.join(ROOT, "include", "gnc_hip.h")).read()
    declared = set(re.findall(r"\b(gnc_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(native.EXPORTED_SYMBOLS)
    nm = subprocess.run(["nm", "-D", "--defined-only", native.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (gnc_[a-z0-9_]+)", nm))
    assert declared <= exported, declared - exported
    assert lib.gnc_abi_version() == native.ABI_VERSION
    assert lib.gnc_target_arch() == b"gfx950"


def test_mlp_desc_layout_matches_c():
    import ctypes
    from graphnet_classifier_amd import native
    assert native.load_library().gnc_sizeof_mlp_desc() == ctypes.sizeof(native.MlpDesc)


def test_mlp_supported_shape_query_runs_without_gpu():
    import ctypes
    from graphnet_classifier_amd import native
    lib = native.load_library()
    d = native.MlpDesc()
    d.num_segments, d.num_linear, d.activation = 1, 2, 0
    d.seg[0].width, d.seg[0].ld = 8, 8
    d.in_dim[0], d.out_dim[0], d.in_dim[1], d.out_dim[1] = 8, 64, 64, 4
    d.ld_out, d.rows = 4, 10
    assert lib.gnc_mlp_supported(ctypes.byref(d)) == 0
    d.out_dim[0], d.in_dim[1] = 300, 300
    assert lib.gnc_mlp_supported(ctypes.byref(d)) == -2 and b"256" in lib.gnc_last_error_string()
    d.out_dim[0], d.in_dim[1] = 64, 32
    assert lib.gnc_mlp_supported(ctypes.byref(d)) == -1


def test_module_surface_and_state_dict_keys_match_reference_checkpoint():
    from graphnet_classifier_amd import GNN
    g = load_golden("g4_graphnet_ckpt.npz")
    sd = sub_state_dict(g, "sd/")
    m = GNN.CombinedModel(GNN.GraphNet(num_local_features=3, space_dim=2, out_channels=1, n_blocks=3), num_nodes=1024, classes=2)
    own = m.state_dict()
    assert list(own.keys()) == list(sd.keys())  # same names, same order
    assert all(tuple(own[k].shape) == tuple(sd[k].shape) for k in sd)
    m.load_state_dict(sd, strict=True)
    assert m.graph_net.name == "GraphNet" and m.graph_net.out_dim == 1
    for name in ("scatter_sum", "EdgeProcessor", "NodeProcessor", "build_GN_block", "GraphProcessor", "GraphNet",
                 "LinearClassifier", "CombinedModel", "MLP"):
        assert hasattr(GNN, name)


def test_graphnet_default_kwargs_match_reference_defaults():
    from graphnet_classifier_amd import GNN
    g = GNN.GraphNet()  # models/GNN.py:230-254 defaults: 10 blocks, widths 128, 3 features, space_dim 2
    assert len(g.graph_processor.blocks) == 10
    assert g.node_encoder.model[0].in_features == 3 and g.edge_encoder.model[0].in_features == 3
    assert g.graph_processor.blocks[0].edge_model.edge_processor.model[0].in_features == 384
    assert g.graph_processor.blocks[0].node_model.node_processor.model[0].in_features == 256
    assert g.node_decoder.model[-1].out_features == 1 and len(g.node_decoder.model) == 5  # no norm


def test_mlp_constructor_errors_match_reference():
    from graphnet_classifier_amd.MLP import MLP
    with pytest.raises(AssertionError):
        MLP(4, 4, norm_type="GroupNorm")  # models/MLP.py:30-33
    with pytest.raises(AttributeError):
        MLP(4, 4, activation="NotAnActivation")  # getattr(nn, activation) models/MLP.py:21
    m = MLP(12, 4, hidden_dim=8, hidden_layers=3, initializer="xavier_uniform_")
    assert [type(l).__name__ for l in m.model] == ["Linear", "ReLU", "Linear", "ReLU", "Linear", "ReLU", "Linear", "LayerNorm"]


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_forward_without_gpu_fails_loudly_instead_of_falling_back():
    from graphnet_classifier_amd import GNN
    m = GNN.GraphNet(n_blocks=1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(4, 3), torch.zeros(4, 2), torch.zeros(2, 3, dtype=torch.long))
    with pytest.raises(RuntimeError, match="no CPU fallback|No GPU|no GPU"):
        GNN.scatter_sum(torch.zeros(3, 2), torch.zeros(3, dtype=torch.long))
    with pytest.raises(NotImplementedError):
        GNN.scatter_sum(torch.zeros(3, 2), torch.zeros(3, dtype=torch.long), dim=1)


def test_shard_ranges_cover_every_graph_once_and_balance_edges():
    from graphnet_classifier_amd.sharding import shard_ranges
    rng = np.random.default_rng(0)
    edges = rng.integers(700, 950, size=6250)
    ep = np.concatenate([[0], np.cumsum(edges)])
    for world in (1, 2, 3, 4, 8):
        r = shard_ranges(ep, world)
        assert r[0][0] == 0 and r[-1][1] == 6250 and all(a[1] == b[0] for a, b in zip(r, r[1:]))
        loads = [ep[b] - ep[a] for a, b in r]
        assert max(loads) - min(loads) <= 2 * edges.max()
    # 6250 equal graphs over 8 ranks -> 781/782 graphs each
    r = shard_ranges(np.arange(6251) * 1600, 8)
    assert sorted({b - a for a, b in r}) == [781, 782]
    # fewer graphs than ranks: nothing lost, nothing duplicated
    r = shard_ranges(np.array([0, 10, 20, 30]), 8)
    assert sum(b - a for a, b in r) == 3 and all(b >= a for a, b in r)


def test_synthetic_workloads_are_deterministic_and_shaped_like_the_survey_says():
    from graphnet_classifier_amd import synthetic as S
    a, kw = S.make_workload("c3", 0.01)
    b, _ = S.make_workload("c3", 0.01)
    assert torch.equal(a.edge_index, b.edge_index) and torch.equal(a.x, b.x)
    assert a.num_nodes == 62 * 160 and a.num_edges == 62 * 1600 and kw["n_blocks"] == 2 and kw["out_dim_node"] == 64
    ei = a.edge_index
    assert torch.equal(ei[0, 0::2], ei[1, 1::2]) and torch.equal(ei[1, 0::2], ei[0, 1::2])  # [i,j],[j,i] interleaved
    assert bool((ei[0, 0::2] < ei[1, 0::2]).all())
    assert bool((ei // 160 == (torch.arange(a.num_edges) // 1600)).all())  # block diagonal
    c, kw2 = S.make_workload("c2", 0.002)
    deg = torch.bincount(c.edge_index[1], minlength=c.num_nodes)
    assert int(deg.min()) >= 2 and int(deg.max()) <= 8 and kw2["n_blocks"] == 3
    s = c.slice_graphs(3, 9)
    assert s.num_graphs == 6 and int(s.edge_index.min()) == 0 and int(s.edge_index.max()) == s.num_nodes - 1


def test_topology_cache_keys_host_tensors_by_content():
    from graphnet_classifier_amd.topology import TopologyCache
    ei = torch.tensor([[0, 1, 2], [1, 2, 0]])
    k1 = TopologyCache._key(ei, 3, "cuda:0")
    k2 = TopologyCache._key(ei.clone(), 3, "cuda:0")
    k3 = TopologyCache._key(torch.tensor([[0, 1, 2], [1, 2, 1]]), 3, "cuda:0")
    assert k1 == k2 and k1 != k3 and k1 != TopologyCache._key(ei, 4, "cuda:0")
