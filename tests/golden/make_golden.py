#!/usr/bin/env python3
"""Capture golden input/output vectors by running the REFERENCE's own classes.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

Writes tests/golden/*.npz.  These files are data (inputs, explicit weights,
expected outputs); no reference source travels.  The reference imports
``torch_geometric.nn.MetaLayer`` (models/GNN.py:24), which is not installed and
cannot be installed offline; MetaLayer does no arithmetic (two row gathers and
a call order), so an in-memory module object that
implements exactly that call contract is registered under that name before the
reference module is imported.  ``torch_scatter`` is absent, so the reference
selects its own ``index_add_`` fallback (models/GNN.py:9-21).

Everything numerical below is computed by the reference's code: its
``scatter_sum``, ``MLP``, ``EdgeProcessor``, ``NodeProcessor``,
``build_GN_block``, ``GraphNet``, ``CombinedModel``, the training-step sequence
of utils/train_model.py:37-42 and ``create_grid_edges_optimized``.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _register_metalayer():
    tg = types.ModuleType("torch_geometric")
    tgnn = types.ModuleType("torch_geometric.nn")

    class MetaLayer(torch.nn.Module):
        def __init__(self, edge_model=None, node_model=None, global_model=None):
            super().__init__()
            self.edge_model, self.node_model, self.global_model = edge_model, node_model, global_model

        def forward(self, x, edge_index, edge_attr=None, u=None, batch=None):
            row, col = edge_index[0], edge_index[1]
            if self.edge_model is not None:
                edge_attr = self.edge_model(x[row], x[col], edge_attr, u, batch if batch is None else batch[row])
            if self.node_model is not None:
                x = self.node_model(x, edge_index, edge_attr, u, batch)
            return x, edge_attr, u

    tgnn.MetaLayer = MetaLayer
    tg.nn = tgnn
    sys.modules["torch_geometric"] = tg
    sys.modules["torch_geometric.nn"] = tgnn


def _np(t):
    return t.detach().cpu().numpy().copy()  # copy: optimizer steps update parameters in place


def _sd(module, tag="sd/"):
    return {tag + k: _np(v) for k, v in module.state_dict().items()}


def _save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrays)} arrays")


def random_graph(rng, n, e, ensure_empty=True):
    """Directed multigraph with repeated destinations and (optionally) a few
    nodes without in-edges; the last node always has an in-edge so that the
    reference's dim_size inference (models/GNN.py:16-17) equals n."""
    row = rng.integers(0, n, size=e)
    lo = 3 if ensure_empty else 0  # nodes 0..2 never receive
    col = rng.integers(lo, n, size=e)
    col[-1] = n - 1
    return np.stack([row, col]).astype(np.int64)


def main():
    sys.path.insert(0, REF)
    _register_metalayer()
    from models.GNN import (CombinedModel, EdgeProcessor, GraphNet, NodeProcessor, build_GN_block,
                            scatter_sum)
    from models.MLP import MLP
    from utils.image_to_graph.image_to_graph_optimized import (create_grid_edges_optimized,
                                                               image_to_graph_pixel_optimized)

    rng = np.random.default_rng(20250824)

    # ---- G1 scatter_sum ---------------------------------------------------
    torch.manual_seed(1)
    src = torch.randn(211, 8)
    index = torch.from_numpy(random_graph(rng, 37, 211)[1])
    src1d = torch.randn(211)
    _save("g1_scatter.npz",
          src=_np(src), index=_np(index),
          out_infer=_np(scatter_sum(src, index, dim=0)),
          out_dimsize40=_np(scatter_sum(src, index, dim=0, dim_size=40)),
          src1d=_np(src1d), out_1d=_np(scatter_sum(src1d, index, dim=0)),
          out_empty_shape=np.array(scatter_sum(torch.zeros(0, 8), torch.zeros(0, dtype=torch.long)).shape))

    # ---- G2 MLP -------------------------------------------------------------
    arrays = {}
    torch.manual_seed(2)
    xin = torch.randn(29, 7)
    arrays["x"] = _np(xin)
    for norm in ("LayerNorm", None):
        for hl in (1, 2, 3):
            m = MLP(7, 5, hidden_dim=16, hidden_layers=hl, norm_type=norm)
            if norm is not None:  # non-trivial affine so gamma/beta are exercised
                with torch.no_grad():
                    m.model[-1].weight.uniform_(0.5, 1.5)
                    m.model[-1].bias.uniform_(-0.5, 0.5)
            tag = f"{'ln' if norm else 'nonorm'}_hl{hl}"
            arrays.update(_sd(m, f"{tag}/sd/"))
            arrays[f"{tag}/y"] = _np(m(xin))
    # the .float() downcast of an fp64 input and the view(x.size(0), -1) flatten (models/MLP.py:46-47)
    m = MLP(12, 4, hidden_dim=8, hidden_layers=2)
    x3 = torch.randn(6, 3, 4, dtype=torch.float64)
    arrays.update(_sd(m, "flat64/sd/"))
    arrays["flat64/x"] = _np(x3)
    arrays["flat64/y"] = _np(m(x3))
    _save("g2_mlp.npz", **arrays)

    # ---- G3 processors / GN block ------------------------------------------
    torch.manual_seed(3)
    n, e, dn, de = 37, 211, 16, 8
    ei = torch.from_numpy(random_graph(rng, n, e))
    x = torch.randn(n, dn)
    ea = torch.randn(e, de)
    ep = EdgeProcessor(dn, de, hidden_dim=24, hidden_layers=2)
    npz = NodeProcessor(dn, de, hidden_dim=24, hidden_layers=2)
    blk = build_GN_block(dn, de, hidden_dim_node=24, hidden_dim_edge=24)
    e_out = ep(x[ei[0]], x[ei[1]], ea.clone())
    n_out = npz(x, ei, ea)
    bx, be, _ = blk(x, ei, ea.clone())
    _save("g3_gnblock.npz", x=_np(x), edge_index=_np(ei), edge_attr=_np(ea),
          edge_out=_np(e_out), node_out=_np(n_out), block_x=_np(bx), block_e=_np(be),
          **_sd(ep, "ep/sd/"), **_sd(npz, "np/sd/"), **_sd(blk, "blk/sd/"))

    # ---- G4 (i) tiny GraphNet, D=16, L=2 -----------------------------------
    torch.manual_seed(4)
    kw = dict(num_local_features=3, space_dim=2, out_channels=2, n_blocks=2, out_dim_node=16, out_dim_edge=16,
              hidden_dim_node=16, hidden_dim_edge=16, hidden_dim_decoder=16,
              hidden_dim_processor_node=16, hidden_dim_processor_edge=16)
    g = GraphNet(**kw)
    n, e = 53, 301
    ei = torch.from_numpy(random_graph(rng, n, e))
    x = torch.rand(n, 3)
    pos = torch.rand(n, 2) * 32
    _save("g4_graphnet_tiny.npz", x=_np(x), pos=_np(pos), edge_index=_np(ei), y=_np(g(x, pos, ei)),
          kwargs_json=np.frombuffer(repr(kw).encode(), dtype=np.uint8), **_sd(g))

    # ---- G4 (ii)/(iii) default model on the pixel graph of a shipped image --
    img = os.path.join(REF, "static/muffin/img_4_880_32.jpg")
    xi, posi, eii = image_to_graph_pixel_optimized(img, resize_value=32)
    x = torch.tensor(xi, dtype=torch.float32)
    pos = torch.tensor(posi, dtype=torch.float32)
    ei = torch.tensor(eii, dtype=torch.long)
    torch.manual_seed(5)
    g = GraphNet(num_local_features=3, space_dim=2, out_channels=1, n_blocks=3)  # main.py:72
    g.eval()
    with torch.no_grad():
        y = g(x, pos, ei)
    # weights as float16-exact values would lose information; keep float32
    _save("g4_graphnet_default.npz", x=_np(x), pos=_np(pos), edge_index=_np(ei), y=_np(y), **_sd(g))

    ck = torch.load(os.path.join(REF, "weights/GNN/dim32_3block/best_model_epoch5.pth"), map_location="cpu",
                    weights_only=True)
    m = CombinedModel(GraphNet(num_local_features=3, space_dim=2, out_channels=1, n_blocks=3), num_nodes=1024, classes=2)
    m.load_state_dict(ck)
    m.eval()
    with torch.no_grad():
        y = m.graph_net(x, pos, ei)
        logits = m((x, pos, ei))
    _save("g4_graphnet_ckpt.npz", x=_np(x), pos=_np(pos), edge_index=_np(ei), y=_np(y), logits=_np(logits), **_sd(m))

    # ---- G5 CombinedModel training step (utils/train_model.py:9-10,37-42) ---
    torch.manual_seed(6)
    kw = dict(num_local_features=3, space_dim=2, out_channels=1, n_blocks=2, out_dim_node=16, out_dim_edge=16,
              hidden_dim_node=16, hidden_dim_edge=16, hidden_dim_decoder=16,
              hidden_dim_processor_node=16, hidden_dim_processor_edge=16)
    n, e = 37, 211
    m = CombinedModel(GraphNet(**kw), num_nodes=n, classes=2)
    ei = torch.from_numpy(random_graph(rng, n, e))
    x = torch.rand(n, 3)
    pos = torch.rand(n, 2) * 32
    label = torch.tensor(1, dtype=torch.long)
    before = _sd(m, "before/")
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    crit = torch.nn.CrossEntropyLoss()
    logits = m((x, pos, ei))
    loss = crit(logits, label)
    opt.zero_grad()
    loss.backward()
    grads = {"grad/" + k: _np(p.grad) for k, p in m.named_parameters()}
    opt.step()
    _save("g5_train_step.npz", x=_np(x), pos=_np(pos), edge_index=_np(ei), label=_np(label), logits=_np(logits),
          loss=_np(loss), kwargs_json=np.frombuffer(repr(kw).encode(), dtype=np.uint8),
          **before, **grads, **_sd(m, "after/"))

    # ---- G6 grid topology (utils/image_to_graph/image_to_graph_optimized.py:7-39)
    arrays = {}
    for (h, w) in ((2, 3), (4, 4), (32, 32), (5, 3)):
        for diag in (False, True):
            arrays[f"grid_{h}x{w}_{'diag' if diag else 'nodiag'}"] = create_grid_edges_optimized(h, w, diag).astype(np.int64)
    _save("g6_grid_edges.npz", **arrays)


def image_graphs():
    """G7: the reference's own pixel and patch builders on shipped images (the formats of SURVEY 2.3)."""
    sys.path.insert(0, REF)
    from PIL import Image
    from utils.image_to_graph.image_to_graph_optimized import image_to_graph_pixel_optimized
    from utils.image_to_graph.image_to_graph_patch import image_to_graph_patch
    arrays = {}
    for tag, path, r in (("muffin32", "static/muffin/img_4_880_32.jpg", 32), ("chihuahua64", "static/chihuahua/img_4_799_64.jpg", 64)):
        img = Image.open(os.path.join(REF, path)).convert("RGB").resize((r, r))
        arrays[f"{tag}/img"] = np.array(img)
        for diag in (False, True):
            x, pos, ei = image_to_graph_pixel_optimized(img, resize_value=r, diagonals=diag)
            d = "diag" if diag else "nodiag"
            arrays[f"{tag}/pixel_{d}/x"] = np.asarray(x).astype(np.float32)      # dataloader.py:49
            arrays[f"{tag}/pixel_{d}/pos"] = np.asarray(pos).astype(np.float32)  # :50
            arrays[f"{tag}/pixel_{d}/edge_index"] = np.asarray(ei).astype(np.int64)
        x, pos, ei = image_to_graph_patch(img, resize_value=r, patch_size=8)
        arrays[f"{tag}/patch/x"] = np.asarray(x, dtype=np.float64).astype(np.float32)
        arrays[f"{tag}/patch/pos"] = np.asarray(pos).astype(np.float32)
        arrays[f"{tag}/patch/edge_index"] = np.asarray(ei).astype(np.int64)
    _save("g7_image_graphs.npz", **arrays)


def training_run():
    """G8: the reference's own ``train()`` (utils/train_model.py:8-81) on a two-sample dataset that alternates two
    images on ONE pixel-grid topology (8 x 8, the fixed-topology regime of image_to_graph_optimized.py:42-47):
    8 epochs x 2 samples = 16 optimizer steps unless its early stopping (patience 2) ends the run sooner.  Recorded:
    inputs, initial weights, the logits of every step (forward hook; the per-step loss follows from them and the
    labels), the avg_loss lines the reference wrote to its log, the names of the .pth files it saved, and the final
    weights (its final_model.pth)."""
    import contextlib
    import io
    import tempfile
    sys.path.insert(0, REF)
    _register_metalayer()
    from models.GNN import CombinedModel, GraphNet
    from utils.image_to_graph.image_to_graph_optimized import create_grid_edges_optimized
    from utils.train_model import train

    torch.manual_seed(8)
    kw = dict(num_local_features=3, space_dim=2, out_channels=1, n_blocks=2, out_dim_node=16, out_dim_edge=16,
              hidden_dim_node=16, hidden_dim_edge=16, hidden_dim_decoder=16,
              hidden_dim_processor_node=16, hidden_dim_processor_edge=16)
    h = w = 8
    ei = torch.from_numpy(create_grid_edges_optimized(h, w, False).astype(np.int64))
    rr, cc = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    pos = torch.from_numpy(np.stack([rr.ravel(), cc.ravel()], 1).astype(np.float32))   # optimized.py:76-79
    xs = [torch.rand(h * w, 3), torch.rand(h * w, 3)]
    labels = [torch.tensor(0, dtype=torch.long), torch.tensor(1, dtype=torch.long)]
    dataset = [((xs[0], pos, ei), labels[0]), ((xs[1], pos, ei), labels[1])]
    m = CombinedModel(GraphNet(**kw), num_nodes=h * w, classes=2)
    before = _sd(m, "before/")
    step_logits = []
    m.register_forward_hook(lambda mod, inp, out: step_logits.append(_np(out)))
    epochs, patience = 8, 2
    with tempfile.TemporaryDirectory() as tmp:
        with contextlib.redirect_stdout(io.StringIO()):
            train(m, dataset, epochs, patience=patience, output_path=tmp)
        files = sorted(f for f in os.listdir(tmp) if f.endswith(".pth"))
        log = [f for f in os.listdir(tmp) if f.startswith("training_logs_")]
        assert len(log) == 1
        with open(os.path.join(tmp, log[0])) as f:
            lines = f.read().splitlines()
        final = torch.load(os.path.join(tmp, "final_model.pth"), map_location="cpu", weights_only=True)
    loss_lines = [l for l in lines if "avg_loss=" in l] + [l for l in lines if l.startswith("Best loss achieved")]
    _save("g8_training_run.npz", x0=_np(xs[0]), x1=_np(xs[1]), pos=_np(pos), edge_index=_np(ei),
          labels=np.array([0, 1], dtype=np.int64), epochs=np.array(epochs), patience=np.array(patience),
          step_logits=np.stack(step_logits), log_lines=np.frombuffer("\n".join(loss_lines).encode(), dtype=np.uint8),
          saved_files=np.frombuffer("\n".join(files).encode(), dtype=np.uint8),
          kwargs_json=np.frombuffer(repr(kw).encode(), dtype=np.uint8), **before,
          **{"after/" + k: _np(v) for k, v in final.items()})
    print("G8:", len(step_logits), "steps;", files, loss_lines)

    # second run: the early-stopping branch (:57-69).  The GraphNet runs above only ever improve, so the loop is
    # driven by a scripted module (one dummy parameter, logits read from a list, input ignored): epoch losses go
    # down, down, up, up -> with patience 2 the reference stops after epoch 4 of 10.
    class Scripted(torch.nn.Module):
        def __init__(self, seq):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(2))
            self.seq, self.i = seq, 0

        def forward(self, sample):
            out = self.seq[self.i] + 0.0 * self.w
            self.i += 1
            return out

    a = [0.0, 0.0, 0.5, 0.6, 0.2, 0.3, 0.3, 0.35, 0.9, 0.9, 0.9, 0.9]
    seq = [torch.tensor([v, 0.0]) for v in a]
    sm = Scripted(seq)
    dataset = [(torch.zeros(3), torch.tensor(0)), (torch.zeros(3), torch.tensor(0))]
    with tempfile.TemporaryDirectory() as tmp:
        with contextlib.redirect_stdout(io.StringIO()) as out:
            train(sm, dataset, 10, patience=2, output_path=tmp)
        files = sorted(f for f in os.listdir(tmp) if f.endswith(".pth"))
        log = [f for f in os.listdir(tmp) if f.startswith("training_logs_")]
        with open(os.path.join(tmp, log[0])) as f:
            lines = f.read().splitlines()
    keep = [l for l in lines if "avg_loss=" in l or l.startswith("Best loss achieved") or l.startswith("Epochs:")]
    stdout = [l for l in out.getvalue().splitlines() if l.startswith(("Early stopping", "Epoch "))]
    _save("g8_early_stop.npz", logits_first=np.array(a, dtype=np.float32), epochs=np.array(10), patience=np.array(2),
          steps_run=np.array(sm.i), log_lines=np.frombuffer("\n".join(keep).encode(), dtype=np.uint8),
          saved_files=np.frombuffer("\n".join(files).encode(), dtype=np.uint8),
          stdout_lines=np.frombuffer("\n".join(stdout).encode(), dtype=np.uint8))
    print("G8b:", sm.i, "steps;", files, keep, stdout)


def default_model_gradients():
    """G9: the DEFAULT model of main.py:72-73 (all widths 128, 3 blocks, CombinedModel read-out over 1024 nodes) on the
    R = 32 pixel graph of a shipped image - the reference's real training regime (main.py:60: one such graph per
    optimizer step) - through utils/train_model.py:37-41: logits, loss and ALL 76 parameter gradients, computed by the
    reference's own classes.  Pins the 128-wide backward kernels against gradients the reference itself produced
    (VERDICT round 2, item 4a).  Inputs are those of g4_graphnet_default.npz (same image, same builder)."""
    sys.path.insert(0, REF)
    _register_metalayer()
    from models.GNN import CombinedModel, GraphNet
    from utils.image_to_graph.image_to_graph_optimized import image_to_graph_pixel_optimized
    img = os.path.join(REF, "static/muffin/img_4_880_32.jpg")
    xi, posi, eii = image_to_graph_pixel_optimized(img, resize_value=32)
    x = torch.tensor(xi, dtype=torch.float32)          # utils/dataloader.py:49-51
    pos = torch.tensor(posi, dtype=torch.float32)
    ei = torch.tensor(eii, dtype=torch.long)
    torch.manual_seed(9)
    m = CombinedModel(GraphNet(num_local_features=3, space_dim=2, out_channels=1, n_blocks=3), num_nodes=x.size(0), classes=2)
    label = torch.tensor(1, dtype=torch.long)          # muffin = class 1 under ImageFolder's sorted class names
    before = _sd(m, "before/")
    crit = torch.nn.CrossEntropyLoss()
    logits = m((x, pos, ei))                           # utils/train_model.py:37
    loss = crit(logits, label)                         # :38
    m.zero_grad()
    loss.backward()                                    # :41
    grads = {"grad/" + k: _np(p.grad) for k, p in m.named_parameters()}
    assert len(grads) == 76
    _save("g9_default_train_grads.npz", x=_np(x), pos=_np(pos), edge_index=_np(ei), label=_np(label), logits=_np(logits),
          loss=_np(loss), **before, **grads)
    print("G9: loss", float(loss), "logits", logits.tolist(), "max |grad|", max(float(np.abs(g).max()) for g in grads.values()))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "g8":
        training_run()
    elif len(sys.argv) > 1 and sys.argv[1] == "g9":
        default_model_gradients()
    else:
        main()
        image_graphs()
        training_run()
        default_model_gradients()
