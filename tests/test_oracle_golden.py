"""Pins the CPU oracle (oracle/graphnet_oracle.py) to the golden vectors captured from the
reference's own classes (tests/golden/make_golden.py).  CPU only.

Tolerances: the oracle restates the same float32 arithmetic with differently-ordered
ATen/numpy kernels, so results agree to a few ulp; 2e-6 absolute on O(1) values is the
gate (SURVEY.md section 4 measured 3e-7 fp32 noise for the reference itself).
"""
import ast

import numpy as np
import pytest
import torch

from oracle import graphnet_oracle as O
from tests._util import load_golden, max_abs, sub_state_dict, t

TOL = 2e-6


def test_g1_scatter_sum():
    g = load_golden("g1_scatter.npz")
    src, index = t(g["src"]), t(g["index"])
    out = O.scatter_sum(src, index, dim=0)
    assert out.shape == g["out_infer"].shape
    assert torch.equal(out, t(g["out_infer"]))  # same edge-ordered fp32 sums -> bit exact
    assert torch.equal(O.scatter_sum(src, index, dim=0, dim_size=40), t(g["out_dimsize40"]))
    assert torch.equal(O.scatter_sum_fast(src, index, 37), t(g["out_infer"]))
    out1d = O.scatter_sum(t(g["src1d"]), index)
    assert out1d.shape == (37, 1) and torch.equal(out1d, t(g["out_1d"]))
    assert tuple(O.scatter_sum(torch.zeros(0, 8), torch.zeros(0, dtype=torch.long)).shape) == tuple(g["out_empty_shape"])
    with pytest.raises(NotImplementedError):
        O.scatter_sum(src, index, dim=1)
    assert max_abs(O.scatter_sum_index_add(src, index, 37), t(g["out_infer"])) < TOL


def test_g1_c_oracle():
    from oracle import c_oracle
    g = load_golden("g1_scatter.npz")
    assert np.array_equal(c_oracle.scatter_sum(g["src"], g["index"], 37), g["out_infer"])
    assert np.array_equal(c_oracle.scatter_sum(g["src"], g["index"], 40), g["out_dimsize40"])
    rowptr, perm = c_oracle.csr_build(g["index"], 37)
    assert np.array_equal(perm, np.argsort(g["index"], kind="stable").astype(np.int32))
    assert np.array_equal(np.diff(rowptr), np.bincount(g["index"], minlength=37))
    with pytest.raises(IndexError):
        c_oracle.scatter_sum(g["src"], g["index"], 5)


@pytest.mark.parametrize("tag", [f"{n}_hl{h}" for n in ("ln", "nonorm") for h in (1, 2, 3)])
def test_g2_mlp(tag):
    g = load_golden("g2_mlp.npz")
    sd = sub_state_dict(g, f"{tag}/sd/")
    y = O.mlp_forward(sd, "", t(g["x"])) if False else O.mlp_forward({"m." + k: v for k, v in sd.items()}, "m", t(g["x"]))
    assert max_abs(y, t(g[f"{tag}/y"])) < TOL


def test_g2_mlp_flatten_and_downcast():
    g = load_golden("g2_mlp.npz")
    sd = {"m." + k: v for k, v in sub_state_dict(g, "flat64/sd/").items()}
    y = O.mlp_forward(sd, "m", t(g["flat64/x"]))
    assert y.dtype == torch.float32 and max_abs(y, t(g["flat64/y"])) < TOL


def test_g3_processors_and_block():
    g = load_golden("g3_gnblock.npz")
    x, ei, ea = t(g["x"]), t(g["edge_index"]), t(g["edge_attr"])
    ep = {"p." + k: v for k, v in sub_state_dict(g, "ep/sd/").items()}
    npd = {"p." + k: v for k, v in sub_state_dict(g, "np/sd/").items()}
    blk = {"b." + k: v for k, v in sub_state_dict(g, "blk/sd/").items()}
    assert max_abs(O.edge_processor(ep, "p", x[ei[0]], x[ei[1]], ea), t(g["edge_out"])) < TOL
    assert max_abs(O.node_processor(npd, "p", x, ei, ea), t(g["node_out"])) < 4 * TOL
    bx, be = O.gn_block(blk, "b", x, ei, ea)
    assert max_abs(bx, t(g["block_x"])) < 4 * TOL and max_abs(be, t(g["block_e"])) < TOL


@pytest.mark.parametrize("name", ["g4_graphnet_tiny.npz", "g4_graphnet_default.npz"])
def test_g4_graphnet(name):
    g = load_golden(name)
    sd = sub_state_dict(g, "sd/")
    y = O.graphnet_forward(sd, t(g["x"]), t(g["pos"]), t(g["edge_index"]))
    assert y.shape == g["y"].shape
    assert max_abs(y, t(g["y"])) < 1e-5
    # fp64 cross-check of the restatement itself (noise floor of the fp32 reference)
    y64 = O.graphnet_forward(O.to_dtype(sd, torch.float64), t(g["x"]).double(), t(g["pos"]).double(), t(g["edge_index"]))
    assert max_abs(y64, t(g["y"])) < 1e-5


@pytest.mark.parametrize("name", ["g4_graphnet_tiny.npz", "g4_graphnet_default.npz"])
def test_padding_with_dummy_nodes_and_self_loops_leaves_the_graph_untouched(name):
    """What the any-topology captures rely on (graphnet_classifier_amd.train.CapturedTrainStep(edge_capacity=...),
    GNN.CapturedForward(edge_capacity=...)): extra nodes with zero features whose only edges are their own self-loops never
    reach a real node, so the reference's arithmetic on the padded graph gives the golden's outputs for the real nodes - the
    same bits, since every real destination still sums the same rows in the same order (the padding sorts behind them)."""
    g = load_golden(name)
    sd = sub_state_dict(g, "sd/")
    x, pos, ei = t(g["x"]), t(g["pos"]), t(g["edge_index"])
    n, e = x.size(0), ei.size(1)
    capacity = (e * 3 // 2 + 255) // 256 * 256
    dummies = (capacity + 7) // 8
    tail = n + torch.arange(capacity, dtype=ei.dtype) % dummies
    ei_pad = tail.repeat(2, 1)
    ei_pad[:, :e] = ei
    x_pad = torch.cat([x, torch.zeros(dummies, x.size(1))])
    pos_pad = torch.cat([pos, torch.zeros(dummies, pos.size(1))])
    y_pad = O.graphnet_forward(sd, x_pad, pos_pad, ei_pad)
    y = O.graphnet_forward(sd, x, pos, ei)
    assert torch.equal(y_pad[:n], y) and max_abs(y_pad[:n], t(g["y"])) < 1e-5
    assert bool(torch.isfinite(y_pad).all())
    assert int(torch.bincount(ei_pad[1, e:], minlength=n + dummies).max()) <= 8  # no dummy becomes a hub


def test_g4_graphnet_shipped_checkpoint():
    g = load_golden("g4_graphnet_ckpt.npz")
    sd = sub_state_dict(g, "sd/")
    assert len(sd) == 76 and sum(v.numel() for v in sd.values()) == 682339
    x, pos, ei = t(g["x"]), t(g["pos"]), t(g["edge_index"])
    y = O.graphnet_forward(sd, x, pos, ei, prefix="graph_net.")
    assert max_abs(y, t(g["y"])) < 1e-5
    assert max_abs(O.combined_forward(sd, x, pos, ei), t(g["logits"])) < 1e-5


def test_g5_forward_and_loss():
    g = load_golden("g5_train_step.npz")
    sd = sub_state_dict(g, "before/")
    logits = O.combined_forward(sd, t(g["x"]), t(g["pos"]), t(g["edge_index"]))
    assert logits.shape == (2,) and max_abs(logits, t(g["logits"])) < TOL
    # unbatched CrossEntropyLoss (utils/train_model.py:10,38): -log_softmax(logits)[label]
    loss = -(logits - torch.logsumexp(logits, 0))[int(g["label"])]
    assert abs(float(loss) - float(g["loss"])) < TOL
    kw = ast.literal_eval(bytes(g["kwargs_json"]).decode())
    assert kw["n_blocks"] == O.n_blocks_of(sd, "graph_net.")


def test_g6_grid_edges():
    g = load_golden("g6_grid_edges.npz")
    for key, ref in g.items():
        _, hw, diag = key.split("_")
        h, w = (int(v) for v in hw.split("x"))
        assert np.array_equal(O.grid_edge_index(h, w, diag == "diag"), ref), key
    assert O.grid_edge_index(2, 3).tolist() == [[0, 1, 3, 4, 0, 1, 2], [1, 2, 4, 5, 3, 4, 5]]


def test_edge_features_l1():
    pos = torch.tensor([[0.0, 0.0], [1.0, 3.0], [4.0, -1.0]])
    ei = torch.tensor([[0, 1, 2], [1, 2, 0]])
    e = O.edge_features(pos, ei)
    assert torch.equal(e, torch.tensor([[1.0, 3.0, 4.0], [3.0, -4.0, 7.0], [-4.0, 1.0, 5.0]]))


def test_g7_pixel_and_patch_builders():
    from oracle import image_graph_oracle as IO
    g = load_golden("g7_image_graphs.npz")
    for tag in ("muffin32", "chihuahua64"):
        img = g[f"{tag}/img"]
        for diag in (False, True):
            d = "diag" if diag else "nodiag"
            x, pos, ei = IO.pixel_graph(img, diag)
            assert np.array_equal(x, g[f"{tag}/pixel_{d}/x"]) and np.array_equal(pos, g[f"{tag}/pixel_{d}/pos"])
            assert np.array_equal(ei, g[f"{tag}/pixel_{d}/edge_index"])
        x, pos, ei = IO.patch_graph(img, 8)
        assert np.array_equal(x, g[f"{tag}/patch/x"]) and np.array_equal(pos, g[f"{tag}/patch/pos"])
        assert np.array_equal(ei, g[f"{tag}/patch/edge_index"])


def test_superpixel_builder_restatement_on_a_label_image():
    from oracle import image_graph_oracle as IO
    seg = np.array([[0, 0, 1, 1], [0, 2, 2, 1], [3, 3, 2, 5], [3, 3, 5, 5]])  # label 4 unused: np.unique compacts
    img = (np.arange(48).reshape(4, 4, 3) * 5).astype(np.uint8)
    x, pos, ei = IO.superpixel_graph_from_labels(img, seg)
    assert x.shape == (5, 3) and pos.shape == (5, 2)
    pairs = {(int(a), int(b)) for a, b in ei.T}
    und = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 4), (2, 3), (2, 4), (3, 4)]
    assert pairs == set(und) | {(b, a) for a, b in und}
    assert ei[:, 0].tolist() == [0, 1] and ei[:, 1].tolist() == [1, 0]  # [i,j],[j,i] interleaved, lexicographic
    assert np.allclose(pos[0], [1 / 3, 1 / 3]) and np.allclose(x[0], img[seg == 0].mean(0) / 255.0)


def test_g8_training_run_of_the_reference_is_reproduced_by_the_oracle():
    """G8 (make_golden.py::training_run, the reference's own train()): 16 steps of forward (oracle) + CE + autograd
    + torch.optim.Adam(lr=1e-3) from the recorded initial weights give the recorded per-step logits and final weights."""
    import ast
    g = load_golden("g8_training_run.npz")
    sd = {k: v.clone().requires_grad_(True) for k, v in sub_state_dict(g, "before/").items()}
    opt = torch.optim.Adam(list(sd.values()), lr=1e-3)
    crit = torch.nn.CrossEntropyLoss()
    pos, ei = t(g["pos"]), t(g["edge_index"])
    xs = [t(g["x0"]), t(g["x1"])]
    k = 0
    O.set_scatter_impl("index_add")  # the differentiable restatement of models/GNN.py:18-20 (the edge-ordered loop detaches)
    try:
        _g8_steps(g, sd, opt, crit, xs, pos, ei)
    finally:
        O.set_scatter_impl("sorted_loop")
    close = [float(((sd[n].detach() - t(g["after/" + n])).abs() < 2e-5).float().mean()) for n in sd]
    assert min(close) > 0.9 and sum(close) / len(close) > 0.97


def _g8_steps(g, sd, opt, crit, xs, pos, ei):
    k = 0
    for epoch in range(int(g["epochs"])):
        for s_ in range(2):
            logits = O.combined_forward(sd, xs[s_], pos, ei)
            assert max_abs(logits.detach(), t(g["step_logits"][k])) < 1e-5, k
            loss = crit(logits, torch.tensor(int(g["labels"][s_])))
            opt.zero_grad()
            loss.backward()
            opt.step()
            k += 1
    assert k == len(g["step_logits"])


def test_g9_default_model_gradients_are_reproduced_by_the_oracle():
    """G9 = logits, loss and all 76 parameter gradients of the reference's DEFAULT model (main.py:72-73: widths 128,
    3 blocks) on the R = 32 pixel graph of static/muffin/img_4_880_32.jpg, captured from the reference's own classes
    (make_golden.py::default_model_gradients).  The oracle + torch autograd on the CPU must reproduce them: that is
    what makes the oracle's autograd a valid checker for the full-size backward tests on the GPU."""
    g = load_golden("g9_default_train_grads.npz")
    sd = {k: v.clone().requires_grad_(True) for k, v in sub_state_dict(g, "before/").items()}
    O.set_scatter_impl("index_add")  # the differentiable restatement of models/GNN.py:18-20 (the edge-ordered loop detaches)
    try:
        logits = O.combined_forward(sd, t(g["x"]), t(g["pos"]), t(g["edge_index"]))
    finally:
        O.set_scatter_impl("sorted_loop")
    assert max_abs(logits.detach(), t(g["logits"])) < TOL
    loss = -(logits - torch.logsumexp(logits, 0))[int(g["label"])]
    assert abs(float(loss.detach()) - float(g["loss"])) < TOL
    loss.backward()
    assert sum(1 for k in g if k.startswith("grad/")) == 76 == len(sd)
    for k, v in sd.items():
        ref = t(g["grad/" + k])
        assert v.grad is not None and max_abs(v.grad, ref) < 2e-5 + 1e-4 * float(ref.abs().max()), k
