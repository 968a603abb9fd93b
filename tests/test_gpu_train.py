"""Row f3 of SURVEY.md section 8 on the GPU: the trainer (graphnet_classifier_amd.train.train) against runs of the
reference's own ``train()`` (utils/train_model.py:8-81) recorded in tests/golden/g8_*.npz by make_golden.py, the
fused flat Adam against torch.optim.Adam, the zero-copy flat gradient buffer, and the regression tests of the
round-1 advisor findings (stale padded weights, hipGraph replay after a weight change, topology cache key)."""
import ast
import os

import numpy as np
import pytest
import torch

from oracle import graphnet_oracle as O
from tests._util import load_golden, max_abs, sub_state_dict, t

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def G():
    from graphnet_classifier_amd import GNN
    return GNN


def _text(a):
    return bytes(a).decode().split("\n")


def _kwargs(g):
    return ast.literal_eval(bytes(g["kwargs_json"]).decode())


def _g8_dataset(g):
    pos, ei = t(g["pos"]), t(g["edge_index"])
    xs = [t(g["x0"]), t(g["x1"])]
    return [((xs[k], pos, ei), torch.tensor(int(g["labels"][k]))) for k in range(2)]


@pytest.mark.parametrize("capture", [False, True])
def test_train_reproduces_the_reference_run(G, tmp_path, capture):
    """16 optimizer steps of the reference's train() on two alternating samples of one pixel-grid topology: per-step
    logits (eager), the avg_loss lines of the log, the saved file names and the final weights."""
    from graphnet_classifier_amd.train import train
    g = load_golden("g8_training_run.npz")
    m = G.CombinedModel(G.GraphNet(**_kwargs(g)), num_nodes=64, classes=2)
    m.load_state_dict(sub_state_dict(g, "before/"), strict=True)
    seen = []
    if not capture:  # a device-to-host copy inside a hook is not capturable
        m.register_forward_hook(lambda mod, inp, out: seen.append(out.detach().cpu()))
    r = train(m, _g8_dataset(g), int(g["epochs"]), patience=int(g["patience"]), output_path=str(tmp_path), capture=capture)
    assert r["captured"] == capture
    if not capture:  # a replayed hipGraph does not run Python hooks; the eager run pins every step
        assert len(seen) == len(g["step_logits"])
        for k, lg in enumerate(seen):
            assert max_abs(lg, t(g["step_logits"][k])) < 1e-5, k
    files = sorted(f for f in os.listdir(tmp_path) if f.endswith(".pth"))
    assert files == _text(g["saved_files"])
    log = [f for f in os.listdir(tmp_path) if f.startswith("training_logs_")]
    assert len(log) == 1
    lines = open(tmp_path / log[0]).read().splitlines()
    mine = [l for l in lines if "avg_loss=" in l] + [l for l in lines if l.startswith("Best loss achieved")]
    assert mine == _text(g["log_lines"])
    assert lines[1] == f"Epochs: {int(g['epochs'])}, Patience: {int(g['patience'])}" and lines[3] == "-" * 50
    final = torch.load(tmp_path / "final_model.pth", map_location="cpu", weights_only=True)
    assert list(final.keys()) == [k[len("after/"):] for k in g if k.startswith("after/")]
    worst = 0.0
    for k, v in final.items():
        assert not v.is_cuda
        worst = max(worst, max_abs(v, t(g["after/" + k])))
    # 16 Adam steps: an entry whose gradient is at rounding level moves by ~lr per step in a direction fp32 noise
    # decides, everything else agrees to rounding; the loss sequence above is the tight check
    assert worst < 16 * 2.1e-3
    frac_close = np.mean([float(((v - t(g["after/" + k])).abs() < 2e-5).float().mean()) for k, v in final.items()])
    assert frac_close > 0.97, frac_close
    # the saved checkpoint loads strictly into a fresh module (and therefore into the reference's classes)
    G.CombinedModel(G.GraphNet(**_kwargs(g)), num_nodes=64, classes=2).load_state_dict(final, strict=True)


def test_captured_training_equals_eager_training(G, tmp_path):
    """The hipGraph-replayed step is the same arithmetic as the eager step: identical final weights, bit for bit."""
    from graphnet_classifier_amd.train import train
    g = load_golden("g8_training_run.npz")
    out = {}
    for capture in (False, True):
        m = G.CombinedModel(G.GraphNet(**_kwargs(g)), num_nodes=64, classes=2)
        m.load_state_dict(sub_state_dict(g, "before/"), strict=True)
        r = train(m, _g8_dataset(g), 3, patience=5, output_path=str(tmp_path / str(capture)), capture=capture)
        out[capture] = ({k: v.detach().cpu().clone() for k, v in m.state_dict().items()}, r["avg_loss"])
    assert out[True][1] == out[False][1]
    for k in out[True][0]:
        assert torch.equal(out[True][0][k], out[False][0][k]), k


def test_train_early_stopping_matches_the_reference(tmp_path, capsys):
    """utils/train_model.py:57-69 driven by the scripted module of make_golden.py (losses down, down, up, up with
    patience 2): same stop epoch, same saved files, same log and stdout lines."""
    from graphnet_classifier_amd.train import train
    g = load_golden("g8_early_stop.npz")

    class Scripted(torch.nn.Module):
        def __init__(self, seq):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(2, device=DEV))
            self.seq, self.i = seq, 0

        def forward(self, sample):
            out = self.seq[self.i].to(self.w.device) + 0.0 * self.w
            self.i += 1
            return out

    sm = Scripted([torch.tensor([float(v), 0.0]) for v in g["logits_first"]])
    dataset = [(torch.zeros(3), torch.tensor(0)), (torch.zeros(3), torch.tensor(0))]
    train(sm, dataset, int(g["epochs"]), patience=int(g["patience"]), output_path=str(tmp_path))
    assert sm.i == int(g["steps_run"])
    assert sorted(f for f in os.listdir(tmp_path) if f.endswith(".pth")) == _text(g["saved_files"])
    log = [f for f in os.listdir(tmp_path) if f.startswith("training_logs_")][0]
    lines = open(tmp_path / log).read().splitlines()
    keep = [l for l in lines if "avg_loss=" in l or l.startswith("Best loss achieved") or l.startswith("Epochs:")]
    assert keep == _text(g["log_lines"])
    stdout = [l for l in capsys.readouterr().out.splitlines() if l.startswith(("Early stopping", "Epoch "))]
    assert stdout == _text(g["stdout_lines"])


def test_fused_adam_matches_torch_adam(G):
    from graphnet_classifier_amd.train import FlatParameters, FusedAdam
    torch.manual_seed(0)
    a = torch.nn.Sequential(torch.nn.Linear(37, 19), torch.nn.Linear(19, 3)).to(DEV)
    b = torch.nn.Sequential(torch.nn.Linear(37, 19), torch.nn.Linear(19, 3)).to(DEV)
    b.load_state_dict(a.state_dict())
    ref = torch.optim.Adam(a.parameters(), lr=1e-3)
    flat = FlatParameters(b)
    mine = FusedAdam(flat, lr=1e-3)
    gen = torch.Generator(device=DEV).manual_seed(1)
    for step in range(25):
        grads = [torch.randn(p.shape, device=DEV, generator=gen) * (10.0 ** (step % 5 - 3)) for p in a.parameters()]
        for p, q, gr in zip(a.parameters(), b.parameters(), grads):
            p.grad = gr.clone()
            q.grad = gr.clone()
        ref.step()
        mine.step()
        for p, q in zip(a.parameters(), b.parameters()):
            assert q.grad.data_ptr() >= flat.grad.data_ptr() and q.grad.data_ptr() < flat.grad.data_ptr() + 4 * flat.grad.numel()
            assert max_abs(p.detach(), q.detach()) < 2e-7, step
    assert int(mine.step_count.item()) == 25
    # the padding between parameters stays exactly zero
    mask = torch.ones(flat.numel, dtype=torch.bool, device=DEV)
    for p, off in zip(flat.params, flat.offsets):
        mask[off:off + p.numel()] = False
    assert float(flat.flat[mask].abs().max()) == 0.0


def test_flat_gradients_are_views_and_the_reducer_is_a_noop_at_world_one(G):
    from graphnet_classifier_amd.sharding import FlatGradAllReduce
    from graphnet_classifier_amd import synthetic as S
    batch = S.superpixel_like_graphs(2, seed=5)
    m = G.GraphNet(**S.graphnet_kwargs(32, 1))
    red = FlatGradAllReduce(m.parameters())
    red.zero_grad()
    m(batch.x, batch.pos, batch.edge_index).sum().backward()
    before = [p.grad for p in m.parameters()]
    assert red() is None and all(p.grad is g for p, g in zip(m.parameters(), before))  # world 1: nothing happens
    flat = red.pack()
    lo, hi = flat.data_ptr(), flat.data_ptr() + 4 * flat.numel()
    off = 0
    for p, g in zip(m.parameters(), before):
        assert lo <= p.grad.data_ptr() < hi and p.grad.shape == p.shape
        assert torch.equal(flat[off:off + p.numel()].view_as(p), g)
        off += p.numel()
    # a second backward without zero_grad accumulates in place INTO the flat buffer
    m(batch.x, batch.pos, batch.edge_index).sum().backward()
    assert all(lo <= p.grad.data_ptr() < hi for p in m.parameters())
    assert max_abs(red.pack()[:before[0].numel()].view_as(before[0]), 2 * before[0]) < 1e-4 * float(before[0].abs().max() + 1)


# ---- regression tests of the round-1 advisor findings ---------------------------------------------------------
def test_in_place_edit_through_data_is_seen_by_the_next_forward(G):
    """`.data` edits do not bump `_version`; the [H, 3] first-layer encoder weights (ld % 4 != 0) are re-padded from
    the live tensor on every call."""
    from graphnet_classifier_amd import synthetic as S
    batch = S.superpixel_like_graphs(2, seed=9)
    torch.manual_seed(1)
    m = G.GraphNet(**S.graphnet_kwargs(32, 1))
    with torch.no_grad():
        m(batch.x, batch.pos, batch.edge_index)
        m.node_encoder.model[0].weight.data.mul_(2.0)
        m.edge_encoder.model[0].weight.data.add_(0.25)
        y = m(batch.x, batch.pos, batch.edge_index)
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    assert max_abs(y, O.graphnet_forward(sd, batch.x, batch.pos, batch.edge_index)) < 1e-5


def test_captured_forward_follows_weight_updates(G):
    """hipGraph replay after load_state_dict / an optimizer step must use the NEW weights everywhere, including the
    padded first-layer encoder weights."""
    rng = np.random.default_rng(0)
    ei = torch.from_numpy(O.grid_edge_index(8, 8))
    rr, cc = np.meshgrid(np.arange(8), np.arange(8), indexing="ij")
    pos = torch.from_numpy(np.stack([rr.ravel(), cc.ravel()], 1).astype(np.float32))
    torch.manual_seed(4)
    m = G.CombinedModel(G.GraphNet(num_local_features=3, space_dim=2, out_channels=1, n_blocks=2), num_nodes=64, classes=2)
    m.eval()
    x = torch.from_numpy(rng.random((64, 3)).astype(np.float32))
    cap = G.CapturedForward(m, x, pos, ei)
    torch.manual_seed(99)
    other = G.CombinedModel(G.GraphNet(num_local_features=3, space_dim=2, out_channels=1, n_blocks=2), num_nodes=64, classes=2)
    m.load_state_dict(other.state_dict())
    with torch.no_grad():
        eager = m((x.to(DEV), pos.to(DEV), ei.to(DEV)))
        theirs = other((x.to(DEV), pos.to(DEV), ei.to(DEV)))
    assert torch.equal(eager, theirs)
    assert torch.equal(cap(x).clone(), eager)


def test_topology_cache_tells_transposed_views_apart(G):
    from graphnet_classifier_amd.topology import TopologyCache, get_destination_csr
    pairs = torch.tensor([[0, 1], [1, 2], [2, 0], [3, 1]], device=DEV)  # [E, 2]
    a, b = pairs.t(), pairs.view(2, 4)
    assert a.data_ptr() == b.data_ptr() and a.shape == b.shape
    assert TopologyCache._key(a, 4, DEV) != TopologyCache._key(b, 4, DEV)
    cache = TopologyCache()
    ta, tb = cache.get(a, 4, DEV), cache.get(b, 4, DEV)
    assert ta is not tb and ta.dst_sorted.tolist() != tb.dst_sorted.tolist()
    # the operator-level scatter_sum caches on the index tensor itself: second call hits
    from graphnet_classifier_amd import topology
    idx = torch.tensor([2, 0, 2, 1], device=DEV)
    src = torch.arange(8, dtype=torch.float32, device=DEV).view(4, 2)
    h0 = topology._destination_cache.hits
    o1 = G.scatter_sum(src, idx, dim_size=3)
    o2 = G.scatter_sum(src, idx, dim_size=3)
    assert topology._destination_cache.hits == h0 + 1 and torch.equal(o1, o2)
    assert o1.tolist() == [[2.0, 3.0], [6.0, 7.0], [4.0, 6.0]]
    src.requires_grad_(True)
    G.scatter_sum(src, idx, dim_size=3).sum().backward()
    assert torch.equal(src.grad, torch.ones_like(src))


def test_gathered_id_outside_the_table_reads_zero_in_every_kernel_variant():
    """Uniform semantics across the K4 variants (round-1 advisor): generic kernel (Tanh), resident, streaming."""
    from graphnet_classifier_amd import native
    rng = np.random.default_rng(3)
    for width, act in ((64, "ReLU"), (64, "Tanh"), (128, "ReLU"), (256, "ReLU")):
        big = torch.from_numpy(rng.standard_normal((200, width)).astype(np.float32) + 3.0).to(DEV)
        table = big[:100]
        idx = torch.from_numpy(rng.integers(0, 100, size=300).astype(np.int32))
        idx[::5] = torch.from_numpy(rng.integers(100, 200, size=len(idx[::5])).astype(np.int32))
        w = torch.from_numpy((rng.standard_normal((width, width)) / np.sqrt(width)).astype(np.float32))
        b = torch.from_numpy(rng.standard_normal(width).astype(np.float32))
        y = native.mlp_forward([(table, idx.to(DEV))], [w.to(DEV)], [b.to(DEV)], activation=act)
        rows = big.cpu()[idx.long()] * (idx < 100)[:, None]
        assert max_abs(y.cpu(), rows @ w.t() + b) < 2e-5, (width, act)
    with pytest.raises(RuntimeError, match="must state its table"):
        d = native.make_mlp_desc([(table, idx.to(DEV), width, 0, 0)], [w.to(DEV)], [b.to(DEV)], None, "ReLU", 0.0, None,
                                 torch.empty(300, width, device=DEV), 300)
        d.seg[0].table_rows = 0
        import ctypes
        native._check(native.load_library().gnc_mlp_forward_f32(ctypes.byref(d), 0), "gnc_mlp_forward_f32")


def test_captured_step_after_eager_default_stream_training(G):
    """The sequence that used to take the process down at hipStreamEndCapture (tools/repro_capture2.py MODE=eager_first):
    one EAGER training step on the legacy default stream whose `loss` stays alive - so the parameters' AccumulateGrad
    nodes, bound to the default stream, stay alive too - and then a CapturedTrainStep built directly by the caller.
    The captured step differentiates private aliases of the parameters, so none of those nodes is reachable from the
    capture; the replayed step must equal the same step run eagerly."""
    from graphnet_classifier_amd.train import CapturedTrainStep, FlatParameters, FusedAdam
    g = load_golden("g8_training_run.npz")
    ds = _g8_dataset(g)
    crit = torch.nn.CrossEntropyLoss()
    finals = {}
    for mode in ("captured", "eager"):
        m = G.CombinedModel(G.GraphNet(**_kwargs(g)), num_nodes=64, classes=2)
        m.load_state_dict(sub_state_dict(g, "before/"), strict=True)
        opt = FusedAdam(FlatParameters(m))
        loss_sum = torch.zeros((), dtype=torch.float64, device=DEV)
        assert torch.cuda.current_stream() == torch.cuda.default_stream()
        (s0, l0), (s1, l1) = ds
        loss = crit(m((s0[0].to(DEV), s0[1].to(DEV), s0[2])), l0.to(DEV))  # eager, default stream; `loss` kept alive on purpose
        opt.zero_grad(); loss.backward(); opt.step()
        loss_sum += loss.detach().double()
        if mode == "captured":
            step = CapturedTrainStep(m, opt, crit, s1, l1, loss_sum)
            step(s1, l1)
            step(s0, l0)
        else:
            for s, l in ((s1, l1), (s0, l0)):
                lo = crit(m((s[0].to(DEV), s[1].to(DEV), s[2])), l.to(DEV))
                opt.zero_grad(); lo.backward(); opt.step()
                loss_sum += lo.detach().double()
        torch.cuda.synchronize()
        assert bool(torch.isfinite(loss)) and int(opt.step_count.item()) == 3
        finals[mode] = ({k: v.detach().cpu().clone() for k, v in m.state_dict().items()}, float(loss_sum))
        # the module's own .grad are the flat views holding the last step's gradient, as after an eager step
        assert all(p.grad is not None and p.grad.data_ptr() == v.data_ptr() for p, v in zip(opt.fp.params, opt.fp.reducer.views))
    assert finals["captured"][1] == finals["eager"][1]
    for k in finals["eager"][0]:
        assert torch.equal(finals["captured"][0][k], finals["eager"][0][k]), k


def test_captured_step_restores_batchnorm_buffers(G):
    """Constructing a CapturedTrainStep must not train: the warm-up steps' effect on BatchNorm running statistics is
    undone like their effect on the parameters and the Adam state (advisor finding, round 2)."""
    from graphnet_classifier_amd.train import CapturedTrainStep, FlatParameters, FusedAdam
    g = load_golden("g8_training_run.npz")
    kw = dict(_kwargs(g), norm_type="BatchNorm1d")
    torch.manual_seed(0)
    m = G.CombinedModel(G.GraphNet(**kw), num_nodes=64, classes=2)
    m.train()
    (s0, l0), _ = _g8_dataset(g)
    opt = FusedAdam(FlatParameters(m))
    before = {k: v.detach().clone() for k, v in m.state_dict().items()}
    loss_sum = torch.zeros((), dtype=torch.float64, device=DEV)
    step = CapturedTrainStep(m, opt, torch.nn.CrossEntropyLoss(), s0, l0, loss_sum)
    torch.cuda.synchronize()
    assert any("running_mean" in k for k in before)
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k]), k
    step(s0, l0)
    torch.cuda.synchronize()
    assert any(not torch.equal(v, before[k]) for k, v in m.state_dict().items() if "running_mean" in k)
    # the any-topology form pads with dummy rows, which batch statistics would see: refused, and train() keeps such a
    # model on eager steps when the topology changes
    with pytest.raises(NotImplementedError):
        CapturedTrainStep(m, opt, torch.nn.CrossEntropyLoss(), s0, l0, loss_sum, edge_capacity=1024)
    with pytest.raises(NotImplementedError):
        G.CapturedForward(m, *s0, edge_capacity=1024)
    G.CapturedForward(m.eval(), *s0, edge_capacity=1024)  # eval mode normalises row by row: fine


def _random_graph_dataset(n, count, seed, lo=150, hi=260):
    """Graphs over the same ``n`` nodes whose edge lists differ in content AND length (what a region-adjacency graph of a
    new image is: utils/image_to_graph/image_to_graph_superpixel.py:31-66)."""
    rng = np.random.default_rng(seed)
    ds = []
    for _ in range(count):
        e = int(rng.integers(lo, hi))
        ei = torch.from_numpy(rng.integers(0, n, size=(2, e)).astype(np.int64))
        x = torch.from_numpy(rng.random((n, 3), dtype=np.float32))
        pos = torch.from_numpy((rng.random((n, 2)) * 8).astype(np.float32))
        ds.append(((x, pos, ei), torch.tensor(int(rng.integers(0, 2)))))
    return ds


def test_any_topology_captured_training_matches_eager_training(G, tmp_path):
    """One graph per optimizer step with a NEW topology every step (main.py:60 on superpixel graphs): train() replays ONE
    captured step whose buffers hold a dummy node and spare edge slots and whose topology build is part of the graph.
    The padding contributes exact zeros: same per-epoch losses and final weights as the eager loop."""
    from graphnet_classifier_amd.train import train
    g = load_golden("g8_training_run.npz")
    ds = _random_graph_dataset(64, 10, seed=7)
    out = {}
    for capture in (False, True):
        m = G.CombinedModel(G.GraphNet(**_kwargs(g)), num_nodes=64, classes=2)
        m.load_state_dict(sub_state_dict(g, "before/"), strict=True)
        r = train(m, ds, 3, patience=5, output_path=str(tmp_path / str(capture)), capture=capture)
        assert r["captured_any_topology"] is capture  # (a topology that comes back also gets its own fixed capture)
        out[capture] = ({k: v.detach().cpu().clone() for k, v in m.state_dict().items()}, r["avg_loss"])
    for a, b in zip(out[True][1], out[False][1]):
        assert abs(a - b) <= 1e-5 * max(1.0, abs(b)), (out[True][1], out[False][1])
    close = total = 0
    for k in out[True][0]:
        d = (out[True][0][k] - out[False][0][k]).abs()
        close += int((d <= 2e-5).sum())
        total += d.numel()
    # an entry whose gradient is at rounding level moves by ~lr per step in a direction fp32 noise decides (as in G8)
    assert close >= 0.97 * total, (close, total)


def test_any_topology_captured_step_limits_and_flags(G):
    from graphnet_classifier_amd.train import CapturedTrainStep, FlatParameters, FusedAdam
    g = load_golden("g8_training_run.npz")
    m = G.CombinedModel(G.GraphNet(**_kwargs(g)), num_nodes=64, classes=2)
    m.load_state_dict(sub_state_dict(g, "before/"), strict=True)
    opt = FusedAdam(FlatParameters(m), lr=1e-3)
    loss_sum = torch.zeros((), dtype=torch.float64, device=DEV)
    ds = _random_graph_dataset(64, 3, seed=9)
    with pytest.raises(ValueError):
        CapturedTrainStep(m, opt, torch.nn.CrossEntropyLoss(), ds[0][0], ds[0][1], loss_sum, edge_capacity=16)
    step = CapturedTrainStep(m, opt, torch.nn.CrossEntropyLoss(), ds[0][0], ds[0][1], loss_sum, edge_capacity=512)
    before = {k: v.detach().clone() for k, v in m.state_dict().items()}
    assert float(loss_sum.item()) == 0.0  # capturing does not train
    for (sample, label) in ds:
        assert step.matches(sample)
        step(sample, label)
    step.check()
    assert np.isfinite(float(loss_sum.item())) and float(loss_sum.item()) > 0
    assert any(not torch.equal(before[k], v) for k, v in m.state_dict().items())
    # the eager step on the same sample gives the same loss
    x, pos, ei = ds[0][0]
    big = torch.cat([ei, torch.randint(0, 64, (2, 600))], dim=1)
    assert not step.matches((x, pos, big))
    with pytest.raises(ValueError):
        step((x, pos, big), ds[0][1])
    # device tensors in: the range test is a reduction launched OUTSIDE the graph right in front of the replay (regression: the
    # flags used to be cleared by a hipMemsetAsync node, which such a launch made write a foreign byte pattern on the next replay)
    for (sample, label) in ds:
        step(tuple(v.to(DEV) for v in sample), label)
        step.check()
    assert np.isfinite(float(loss_sum.item()))
    bad = ei.clone()
    bad[1, 5] = 64 + 7  # a dummy node's id: not a node of this graph
    with pytest.raises(IndexError):  # host tensor: raised at once, as models/GNN.py:18-20 does
        step((x, pos, bad), ds[0][1])
    step((x.to(DEV), pos.to(DEV), bad.to(DEV)), ds[0][1])  # device tensor: no sync in the step, the flag waits for check()
    with pytest.raises(IndexError):
        step.check()
    step.check()  # reported once
    bad[1, 5] = 100000  # beyond the dummies too: the topology build's own flag, and a poisoned forward
    step((x.to(DEV), pos.to(DEV), bad.to(DEV)), ds[0][1])
    with pytest.raises(IndexError):
        step.check()
    assert not np.isfinite(float(loss_sum.item()))  # never a plausible number
