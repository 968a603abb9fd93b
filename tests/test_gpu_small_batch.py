"""GPU parity tests of the small-batch kernels (the reference's one-graph-per-call regime, main.py:60,
utils/train_model.py:35-45): the column-split forward (mlp_col16.hip), its backward sibling (mlp_bwd_col16.hip), the
one-launch weight gradients (xty_small.hip) and the one-destination-per-lane-group K1 - each against float64 formulas on
the same inputs (tolerance 1e-5 on O(1) values, the north-star bound) and, where the contract says so, bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
TOL = 1e-5


@pytest.fixture(scope="module")
def native():
    from graphnet_classifier_amd import native as n
    n.load_library()
    return n


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)


def _lin(rng, o, i):
    b = 1.0 / np.sqrt(i)
    return _t(rng.uniform(-b, b, (o, i))), _t(rng.uniform(-b, b, (o,)))


def _D(x):
    return x.double().cpu()


def _ref_forward(segs, modes, ws, bs, ln, residual):
    """float64 definition of gnc_mlp_forward_f32; returns (out, hidden post-activations)."""
    modes = modes or [0] * len(segs)
    rows = [(_D(tb) if ix is None else _D(tb)[ix.cpu().long()]) for tb, ix in segs]
    x = torch.cat([r for r, m in zip(rows, modes) if m == 0], dim=1)
    z = x @ _D(ws[0]).t() + (_D(bs[0]) if bs[0] is not None else 0)
    for r, m in zip(rows, modes):
        if m == 1:
            z = z + r
    acts = []
    for w, b in zip(ws[1:], bs[1:]):
        z = torch.relu(z)
        acts.append(z)
        z = z @ _D(w).t() + (_D(b) if b is not None else 0)
    if ln is not None:
        z = torch.nn.functional.layer_norm(z, (z.size(1),), _D(ln[0]), _D(ln[1]), ln[2])
    if residual is not None:
        z = z + _D(residual)
    return z, acts


SHAPES = ["encoder3", "projection", "edge_wsplit", "node", "decoder", "h100", "concat_edge", "two_linears"]


@pytest.mark.parametrize("rows", [1, 15, 16, 17, 1000, 8192, 20011])  # (20011: more tiles than workgroups)
@pytest.mark.parametrize("shape", SHAPES)
def test_small_batch_forward_against_float64(native, shape, rows):
    rng = np.random.default_rng(len(shape) * 1000 + rows)
    n, D = 301, 128
    src = torch.from_numpy(rng.integers(0, n, size=rows).astype(np.int32)).to(DEV)
    dst = torch.from_numpy(np.sort(rng.integers(0, n, size=rows)).astype(np.int32)).to(DEV)
    ln = (_t(rng.uniform(0.5, 1.5, D)), _t(rng.uniform(-0.5, 0.5, D)), 1e-5)
    modes = residual = None
    if shape == "encoder3":  # nn.Linear(3, 128) on [rows, 3]: nothing is a 16-B piece
        segs = [(_t(rng.standard_normal((rows, 3))), None)]
        ws, bs = zip(_lin(rng, D, 3), _lin(rng, D, D), _lin(rng, D, D))
    elif shape == "projection":
        segs = [(_t(rng.standard_normal((rows, D))), None)]
        ws, bs, ln = [_lin(rng, D, D)[0]], [None], None
    elif shape == "edge_wsplit":
        e = _t(rng.standard_normal((rows, D)))
        segs = [(_t(rng.standard_normal((n, D))), src), (_t(rng.standard_normal((n, D))), dst), (e, None)]
        modes, residual = [native.SEG_ADD, native.SEG_ADD, native.SEG_MATMUL], e
        ws, bs = zip(_lin(rng, D, D), _lin(rng, D, D), _lin(rng, D, D))
    elif shape == "node":
        x = _t(rng.standard_normal((rows, D)))
        segs, residual = [(x, None), (_t(rng.standard_normal((rows, D))), None)], x
        ws, bs = zip(_lin(rng, D, 2 * D), _lin(rng, D, D), _lin(rng, D, D))
    elif shape == "decoder":
        segs, ln = [(_t(rng.standard_normal((rows, D))), None)], None
        ws, bs = zip(_lin(rng, D, D), _lin(rng, D, D), _lin(rng, 1, D))
    elif shape == "h100":  # widths that are no multiple of 16: masks on every path
        x = _t(rng.standard_normal((rows, 100)))
        segs, residual = [(x, None)], x
        ln = (_t(rng.uniform(0.5, 1.5, 100)), _t(rng.uniform(-0.5, 0.5, 100)), 1e-5)
        ws, bs = zip(_lin(rng, 100, 100), _lin(rng, 100, 100), _lin(rng, 100, 100))
    elif shape == "concat_edge":  # the concat form: two GATHERED matmul segments + e (three chunks of the first Linear)
        x, e = _t(rng.standard_normal((n, D))), _t(rng.standard_normal((rows, D)))
        segs, residual = [(x, src), (x, dst), (e, None)], e
        ws, bs = [_lin(rng, D, 3 * D)[0], _lin(rng, D, D)[0]], [None, _lin(rng, D, D)[1]]
    else:  # two_linears: 96 -> 128 -> 72
        segs, ln = [(_t(rng.standard_normal((rows, 96))), None)], None
        ws, bs = zip(_lin(rng, D, 96), _lin(rng, 72, D))
    ws, bs = list(ws), list(bs)
    assert native.small_batch_kernel_serves(segs, ws, bs, ln, "ReLU", residual, rows, modes)
    acts = []
    out = native.mlp_forward(segs, ws, bs, ln=ln, residual=residual, rows=rows, modes=modes, save_act=acts if len(ws) > 1 else None)
    want, want_acts = _ref_forward(segs, modes, ws, bs, ln, residual)
    scale = max(1.0, float(want.abs().max()))
    assert float((_D(out) - want).abs().max()) < TOL * scale
    for a, wa in zip(acts, want_acts):  # the training forward's saved post-activations
        assert float((_D(a) - wa).abs().max()) < TOL * max(1.0, float(wa.abs().max()))
    plain = native.mlp_forward(segs, ws, bs, ln=ln, residual=residual, rows=rows, modes=modes)
    assert torch.equal(out, plain)  # saving changes nothing about the output


@pytest.mark.parametrize("rows,n", [(5, 3), (16, 1), (33, 40), (1984, 1024), (4000, 77)])
def test_small_batch_edge_launch_leaves_aggregation_to_k1(native, rows, n):
    """A small-batch edge launch is served WITHOUT the aggregation epilogue (the fix-up and zero-fill launches it needs cost more
    than K1 at this size): `aggregate=` returns None for the sums, the rows are those of the plain launch, and K1 on them (the
    one-destination-per-lane-group kernel) adds in the reference's edge order, bit for bit."""
    rng = np.random.default_rng(rows * 3 + n)
    D = 128
    dst_np = np.sort(rng.integers(0, n, size=rows))
    if n > 4:
        dst_np = np.sort(np.where(dst_np == n // 2, n // 2 + 1, dst_np))  # at least one destination without rows
    dst = torch.from_numpy(dst_np.astype(np.int32)).to(DEV)
    src = torch.from_numpy(rng.integers(0, n, size=rows).astype(np.int32)).to(DEV)
    rowptr = torch.from_numpy(np.concatenate([[0], np.cumsum(np.bincount(dst_np, minlength=n))]).astype(np.int32)).to(DEV)
    e = _t(rng.standard_normal((rows, D)))
    segs = [(_t(rng.standard_normal((n, D))), src), (_t(rng.standard_normal((n, D))), dst), (e, None)]
    modes = [native.SEG_ADD, native.SEG_ADD, native.SEG_MATMUL]
    ws, bs = zip(_lin(rng, D, D), _lin(rng, D, D), _lin(rng, D, D))
    ln = (_t(rng.uniform(0.5, 1.5, D)), _t(rng.uniform(-0.5, 0.5, D)), 1e-5)
    out, agg = native.mlp_forward(segs, list(ws), list(bs), ln=ln, residual=e, rows=rows, modes=modes, aggregate=(dst, rowptr, n))
    assert agg is None
    plain = native.mlp_forward(segs, list(ws), list(bs), ln=ln, residual=e, rows=rows, modes=modes)
    assert torch.equal(out, plain)
    k1 = native.scatter_sum_csr(out, rowptr, None, n)
    # ... and K1 itself (the one-destination-per-lane-group kernel at this size) against the reference's order of additions
    o = out.cpu().numpy()
    want = np.zeros((n, D), dtype=np.float32)
    for k in range(rows):
        want[dst_np[k]] += o[k]
    assert np.array_equal(k1.cpu().numpy(), want)


@pytest.mark.parametrize("n,e,d", [(1, 1, 4), (40, 0, 64), (1024, 1984, 128), (300, 5000, 64), (16384, 30000, 128), (77, 900, 100)])
def test_small_graph_scatter_sum_with_permutation_bit_exact(native, n, e, d):
    rng = np.random.default_rng(n + e + d)
    index = torch.from_numpy(rng.integers(0, n, size=e).astype(np.int64))
    msgs = torch.from_numpy(rng.standard_normal((e, d)).astype(np.float32))
    rowptr, perm, _ = native.csr_build(index.to(DEV), n)
    got = native.scatter_sum_csr(msgs.to(DEV), rowptr, perm, n)
    want = np.zeros((n, d), dtype=np.float32)
    m, ix = msgs.numpy(), index.numpy()
    for k in range(e):  # the reference's edge order (stable sort)
        want[ix[k]] += m[k]
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("d,e,nadd,gather,lnorm", [(128, 1984, 2, 1, True), (128, 1000, 0, 0, True), (100, 999, 0, 0, True),
                                                   (96, 37, 2, 2, True), (128, 8192, 2, 1, True), (128, 500, 0, 0, False),
                                                   (72, 5, 2, 0, True),
                                                   # above the small-batch backward limit: the register-resident data kernel
                                                   (128, 9001, 2, 1, True), (128, 20011, 0, 0, True), (128, 12345, 2, 2, True)])
def test_small_batch_backward_against_float64_on_saved_activations(native, d, e, nadd, gather, lnorm):
    """K8 data kernel for small batches: dz of every layer, dx (with the residual's gradient folded in), the LayerNorm
    parameter sums and the gathered output gradient (`gather`: 0 none, 1 grad_out + gathered rows, 2 gathered rows only)
    against float64 formulas evaluated on the SAME saved post-activations (so no ReLU-side ambiguity)."""
    rng = np.random.default_rng(d * 7 + e + nadd + gather)
    n = 301
    ws, bs = zip(_lin(rng, d, d), _lin(rng, d, d), _lin(rng, d, d))
    ws, bs = list(ws), list(bs)
    ln = (_t(rng.uniform(0.5, 1.5, d)), _t(rng.uniform(-0.5, 0.5, d)), 1e-5) if lnorm else None
    ea = _t(rng.standard_normal((e, d)))
    src = torch.from_numpy(rng.integers(0, n, size=e).astype(np.int32)).to(DEV)
    dst = torch.from_numpy(np.sort(rng.integers(0, n, size=e)).astype(np.int32)).to(DEV)
    if nadd:
        segs = [(_t(rng.standard_normal((n, d))), src), (_t(rng.standard_normal((n, d))), dst), (ea, None)]
        modes = [native.SEG_ADD, native.SEG_ADD, native.SEG_MATMUL]
    else:
        segs, modes = [(ea, None)], None
    acts = []
    native.mlp_forward(segs, ws, bs, ln=ln, residual=ea, rows=e, modes=modes, save_act=acts)
    assert len(acts) == 2
    gout = _t(rng.standard_normal((e, d))) if gather != 2 else None
    gath = (_t(rng.standard_normal((n, d))), dst) if gather else None
    r = native.mlp_backward(segs, ws, bs, ln, gout, rows=e, modes=modes, need_dx=True, residual=ea, grad_gather=gath, saved_act=acts)
    assert r["saved_act_used"] and r["residual_folded"] and "dw" not in r
    assert gather == 0 or r["grad_out"] is None  # the launch gathered the rows itself
    a0, a1 = _D(acts[0]), _D(acts[1])
    g = (_D(gout) if gout is not None else 0) + (_D(gath[0])[dst.cpu().long()] if gath else 0)
    if lnorm:
        z2 = a1 @ _D(ws[2]).t() + _D(bs[2])
        mean = z2.mean(1, keepdim=True)
        rstd = 1 / torch.sqrt(((z2 - mean) ** 2).mean(1, keepdim=True) + 1e-5)
        yh = (z2 - mean) * rstd
        dy = g * _D(ln[0])
        dz2 = rstd * (dy - dy.mean(1, keepdim=True) - yh * (dy * yh).mean(1, keepdim=True))
    else:
        dz2 = g
    dz1 = (dz2 @ _D(ws[2])) * (a1 > 0)
    dz0 = (dz1 @ _D(ws[1])) * (a0 > 0)
    dx = dz0 @ _D(ws[0]) + g
    for got, want in ((r["dz"][2], dz2), (r["dz"][1], dz1), (r["dz"][0], dz0), (r["dx"], dx)):
        assert float((_D(got) - want).abs().max()) < TOL * max(1.0, float(want.abs().max()))
    if lnorm:
        sb, sg = r["ln_sums"]
        for got, want in ((sb, g.sum(0)), (sg, (g * yh).sum(0))):  # sums over all rows: relative to their size
            assert float((_D(got) - want).abs().max()) < TOL * max(1.0, float(want.abs().max()))


def test_xty_multi_products_and_row_sums(native):
    """Several dW / db products of different shapes (incl. a block written into a wider gradient) and the row sums of a
    partial-sum matrix in one launch, against float64; repeated calls give the same bits."""
    rng = np.random.default_rng(5)
    a1, b1 = _t(rng.standard_normal((1984, 128))), _t(rng.standard_normal((1984, 128)))
    a2, b2 = _t(rng.standard_normal((1024, 128))), _t(rng.standard_normal((1024, 256)))
    a3, b3 = _t(rng.standard_normal((777, 100))), _t(rng.standard_normal((777, 3)))
    a4, b4 = _t(rng.standard_normal((1, 1))), _t(rng.standard_normal((1, 72)))
    wide = torch.zeros(128, 300, device=DEV)
    part = _t(rng.standard_normal((124, 256)))
    prods = [(a1, b1, None), (a2, b2, wide[:, 20:276]), (a3, b3, None), (a4, b4, None)]
    res, sums = native.xty_multi(prods, [part])
    for (a, b, _), (c, cs) in zip(prods, res):
        want = _D(a).t() @ _D(b)
        assert float((_D(c) - want).abs().max()) < 1e-4 * max(1.0, float(want.abs().max()))
        assert float((_D(cs) - _D(a).sum(0)).abs().max()) < 1e-4 * max(1.0, float(_D(a).sum(0).abs().max()))
    assert res[1][0].data_ptr() == wide[:, 20:276].data_ptr()
    assert float(wide[:, :20].abs().max()) == 0.0 and float(wide[:, 276:].abs().max()) == 0.0  # nothing outside the block
    assert float((_D(sums[0]) - _D(part).sum(0)).abs().max()) < 1e-4
    res2, sums2 = native.xty_multi([(a1, b1, None), (a3, b3, None)], [part])
    assert torch.equal(res2[0][0], res[0][0]) and torch.equal(res2[1][0], res[2][0]) and torch.equal(sums2[0], sums[0])
    # more jobs than one launch carries
    many, _ = native.xty_multi([(a3, b3, None)] * 11)
    assert all(torch.equal(c, res[2][0]) for c, _ in many)


@pytest.mark.parametrize("rows", [1, 1000, 20011, 40000])
def test_projection_launches_of_the_wsplit(native, rows):
    """The W-split's node-side products in one launch each: (x Ws^T, x Wd^T) forward, d(ps) Ws + d(pd) Wd backward (W0 read
    transposed where it lies); 40000 rows: above the small-batch limit, the two-launch / copied-weight routes."""
    rng = np.random.default_rng(rows)
    D = 128
    w0 = _t(rng.uniform(-0.1, 0.1, (D, 3 * D)))
    x, a, b = (_t(rng.standard_normal((rows, D))) for _ in range(3))
    ps, pd = native.dual_projection(x, w0[:, :D], w0[:, D:2 * D])
    for got, want in ((ps, _D(x) @ _D(w0[:, :D]).t()), (pd, _D(x) @ _D(w0[:, D:2 * D]).t())):
        assert float((_D(got) - want).abs().max()) < TOL * max(1.0, float(want.abs().max()))
    dx = native.projection_t2(a, b, w0, D)
    want = _D(a) @ _D(w0[:, :D]) + _D(b) @ _D(w0[:, D:2 * D])
    assert float((_D(dx) - want).abs().max()) < TOL * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("rows", [8192 + 16, 30000])
def test_mid_size_launches_are_bitwise_reproducible(native, rows):
    """More tiles than workgroups (the tile loop of the small-batch kernels, the register-resident data kernel): ten launches
    of the edge / node / projection shapes and of their backward give the same bits - a missing barrier between tiles shows up
    here as run-to-run noise."""
    rng = np.random.default_rng(rows)
    n, D = 2001, 128
    src = torch.from_numpy(rng.integers(0, n, size=rows).astype(np.int32)).to(DEV)
    dst = torch.from_numpy(np.sort(rng.integers(0, n, size=rows)).astype(np.int32)).to(DEV)
    ln = (_t(rng.uniform(0.5, 1.5, D)), _t(rng.uniform(-0.5, 0.5, D)), 1e-5)
    e = _t(rng.standard_normal((rows, D)))
    x2 = _t(rng.standard_normal((rows, D)))
    edge = ([(_t(rng.standard_normal((n, D))), src), (_t(rng.standard_normal((n, D))), dst), (e, None)],
            [native.SEG_ADD, native.SEG_ADD, native.SEG_MATMUL], [_lin(rng, D, D) for _ in range(3)], e)
    node = ([(e, None), (x2, None)], None, [_lin(rng, D, 2 * D), _lin(rng, D, D), _lin(rng, D, D)], e)
    gout = _t(rng.standard_normal((rows, D)))
    for segs, modes, wb, res in (edge, node):
        ws, bs = [w for w, _ in wb], [b for _, b in wb]
        acts = []
        first = native.mlp_forward(segs, ws, bs, ln=ln, residual=res, rows=rows, modes=modes, save_act=acts)
        r0 = native.mlp_backward(segs, ws, bs, ln, gout, rows=rows, modes=modes, need_dx=True, residual=res, saved_act=acts)
        for _ in range(9):
            again = native.mlp_forward(segs, ws, bs, ln=ln, residual=res, rows=rows, modes=modes)
            assert torch.equal(first, again)
            r = native.mlp_backward(segs, ws, bs, ln, gout, rows=rows, modes=modes, need_dx=True, residual=res, saved_act=acts)
            assert torch.equal(r["dx"], r0["dx"]) and all(torch.equal(a, b) for a, b in zip(r["dz"], r0["dz"]))
            assert all(torch.equal(a, b) for a, b in zip(r["ln_sums"], r0["ln_sums"]))
    w0 = _t(rng.uniform(-0.1, 0.1, (D, 3 * D)))
    p0 = native.dual_projection(e, w0[:, :D], w0[:, D:2 * D])
    t0 = native.projection_t2(e, x2, w0, D)
    for _ in range(9):
        p = native.dual_projection(e, w0[:, :D], w0[:, D:2 * D])
        assert torch.equal(p[0], p0[0]) and torch.equal(p[1], p0[1])
        assert torch.equal(native.projection_t2(e, x2, w0, D), t0)


@pytest.mark.parametrize("F,h1,h2,c", [(1024, 128, 32, 2), (16384, 128, 32, 2), (150, 128, 32, 2), (1, 5, 3, 1), (4099, 300, 70, 10)])
def test_readout_classifier_kernels_against_float64(native, F, h1, h2, c):
    """LinearClassifier.forward of ONE graph (models/GNN.py:312-325) in one launch each way: logits, the saved hidden vectors and
    all seven gradients against float64; replays give the same bits (the ticket counter is left at 0)."""
    from graphnet_classifier_amd import functional as Fn
    rng = np.random.default_rng(F + h1)
    y = _t(rng.standard_normal(F)).requires_grad_()
    params = [p.requires_grad_() for p in (*_lin(rng, h1, F), *_lin(rng, h2, h1), *_lin(rng, c, h2))]
    logits = Fn.readout(y, *params)
    yd = _D(y.detach()).requires_grad_()
    pd = [_D(p.detach()).requires_grad_() for p in params]
    ref = torch.relu(torch.relu(yd @ pd[0].t() + pd[1]) @ pd[2].t() + pd[3]) @ pd[4].t() + pd[5]
    assert float((_D(logits.detach()) - ref.detach()).abs().max()) < TOL * max(1.0, float(ref.detach().abs().max()))
    gl = _t(rng.standard_normal(c))
    logits.backward(gl)
    ref.backward(_D(gl))
    for got, want in zip([y.grad] + [p.grad for p in params], [yd.grad] + [p.grad for p in pd]):
        assert got.shape == want.shape
        assert float((_D(got) - want).abs().max()) < TOL * max(1.0, float(want.abs().max()))
    again = Fn.readout(y.detach(), *[p.detach() for p in params])
    assert torch.equal(again, logits.detach())
