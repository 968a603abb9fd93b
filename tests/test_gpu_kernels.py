"""GPU parity tests of the individual HIP kernels, called through the C ABI
(graphnet_classifier_amd.native -> libgnc_hip.so), against the CPU oracle.

Bars: index work (CSR build) and the scatter-sum are checked BIT-EXACT (the CSR segment is
summed in the reference's edge order); the fp32-MFMA MLP is checked to 1e-5 absolute on O(1)
values (north_star tolerance), typically ~1e-6.
"""
import os

import numpy as np
import pytest
import torch

from oracle import graphnet_oracle as O
from tests._util import load_golden, max_abs, sub_state_dict, t

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def native():
    from graphnet_classifier_amd import native as n
    n.load_library()
    return n


def _random_index(rng, n, e, empty_frac=0.1):
    idx = rng.integers(int(n * empty_frac), n, size=e)
    return torch.from_numpy(idx.astype(np.int64))


# ------------------------------------------------------------------ topology
@pytest.mark.parametrize("n,e", [(1, 1), (7, 0), (37, 211), (1000, 12345), (5000, 300), (160, 1600), (100003, 1000003)])
def test_csr_build_matches_stable_sort(native, n, e):
    rng = np.random.default_rng(n * 7 + e)
    index = _random_index(rng, n, e) if e else torch.zeros(0, dtype=torch.int64)
    rowptr, perm, status = native.csr_build(index.to(DEV), n)
    assert status.tolist() == [0, 0]
    order = np.argsort(index.numpy(), kind="stable")
    counts = np.bincount(index.numpy(), minlength=n)
    ref_rowptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    assert np.array_equal(rowptr.cpu().numpy(), ref_rowptr)
    assert np.array_equal(perm.cpu().numpy(), order.astype(np.int32))


def test_csr_build_flags_out_of_range(native):
    index = torch.tensor([0, 5, 2, -1, 9, 3], dtype=torch.int64)
    rowptr, perm, status = native.csr_build(index.to(DEV), 5)
    assert status.tolist() == [1, 0]
    # valid edges (0, 2, 3) still form a correct CSR; invalid ones sit behind rowptr[N]
    assert rowptr.cpu().tolist() == [0, 1, 1, 2, 3, 3]
    assert perm.cpu().tolist()[:3] == [0, 2, 5]


# ------------------------------------------------------------------ topology in one call (gnc_topology_build)
def _batch_of_graphs(rng, sizes, edges_per_graph, gap=0):
    """Graph-ordered edge list: graph g owns a contiguous node range (plus `gap` isolated ids behind it), edges contiguous."""
    src, dst, base = [], [], 0
    for n, e in zip(sizes, edges_per_graph):
        src.append(base + rng.integers(0, n, size=e))
        dst.append(base + rng.integers(0, n, size=e))
        base += n + gap
    return np.concatenate(src).astype(np.int64), np.concatenate(dst).astype(np.int64), base


def _topology_cases():
    rng = np.random.default_rng(2025)
    cases = {}
    s, d, n = _batch_of_graphs(rng, [160] * 50, [1600] * 50)
    cases["c3_like"] = (s, d, n, 0)
    sizes = rng.choice([144, 156, 169], size=120)
    s, d, n = _batch_of_graphs(rng, sizes, rng.integers(770, 913, size=120))
    cases["c2_like_ragged"] = (s, d, n, 0)
    s, d, n = _batch_of_graphs(rng, [40] * 9, [300] * 9, gap=500)  # isolated node ids between the graphs and behind them
    cases["gaps_of_isolated_nodes"] = (s, d, n + 777, 0)
    s, d, n = _batch_of_graphs(rng, [40] * 9, [300] * 9, gap=5000)  # ... so many that a range spans more than 4096 ids
    cases["gaps_wider_than_the_span_limit"] = (s, d, n + 777, 1)
    ei = O.grid_edge_index(32, 32)
    cases["pixel_grid_32"] = (ei[0], ei[1], 1024, 0)
    cases["pixel_grid_32_x7"] = (np.concatenate([ei[0] + 1024 * k for k in range(7)]), np.concatenate([ei[1] + 1024 * k for k in range(7)]), 7168, 0)
    d = np.sort(rng.integers(0, 30000, size=200000)).astype(np.int64)
    cases["sorted_destinations"] = (rng.integers(0, 30000, size=200000).astype(np.int64), d, 30000, 0)
    cases["one_edge"] = (np.array([0], dtype=np.int64), np.array([0], dtype=np.int64), 1, 0)
    cases["exactly_one_tile"] = _batch_of_graphs(rng, [100] * 2, [1024] * 2) + (0,)
    cases["random_small"] = (rng.integers(0, 37, size=211).astype(np.int64), rng.integers(3, 37, size=211).astype(np.int64), 37, 0)
    # ---- edge lists the LDS path must hand to the general path (status[2] == 1)
    cases["random_large"] = (rng.integers(0, 100003, size=1000003).astype(np.int64), rng.integers(0, 100003, size=1000003).astype(np.int64), 100003, 1)
    cases["random_medium"] = (rng.integers(0, 1000, size=12345).astype(np.int64), rng.integers(0, 1000, size=12345).astype(np.int64), 1000, 1)
    ei = O.grid_edge_index(128, 128)
    cases["pixel_grid_128"] = (ei[0], ei[1], 128 * 128, 1)  # one graph of 32,512 edges
    s, d, n = _batch_of_graphs(rng, [160] * 6, [1600, 1600, 5000, 1600, 1600, 1600])
    cases["one_big_graph_among_small"] = (s, d, n, 1)
    s, d, n = _batch_of_graphs(rng, [160] * 5, [900] * 5)
    d[2000:2300] = 400  # a hub: in-degree 300 inside the third graph
    cases["hub_destination"] = (s, d, n, 0)
    s, d, n = _batch_of_graphs(rng, [50] * 4, [2000] * 4)
    d[:] = (d // 50) * 50 + 7  # every edge of a graph into ONE node: in-degree 2000
    cases["star_graphs"] = (s, d, n, 0)
    s, d, n = _batch_of_graphs(rng, [9000] * 3, [2000] * 3)  # destination span of a range above 4096
    cases["wide_span"] = (s, d, n, 1)
    return cases


_TOPO = _topology_cases()


@pytest.mark.parametrize("name", sorted(_TOPO))
def test_topology_build_matches_stable_sort(native, name):
    """gnc_topology_build: rowptr / perm bit-equal to a stable argsort by destination, endpoint vectors equal to
    src[perm] / dst[perm], for graph-ordered batches (LDS path, status[2] == 0) and for edge lists that are not
    (device-gated general path, status[2] == 1); without the gated path the flag alone must tell."""
    src, dst, n, general = _TOPO[name]
    order = np.argsort(dst, kind="stable")
    ref_rowptr = np.concatenate([[0], np.cumsum(np.bincount(dst, minlength=n))]).astype(np.int32)
    rowptr, perm, ss, ds, status = native.topology_build(torch.from_numpy(src).to(DEV), torch.from_numpy(dst).to(DEV), n, gated_fallback=True)
    assert status.tolist() == [0, 0, general]
    assert np.array_equal(rowptr.cpu().numpy(), ref_rowptr)
    assert np.array_equal(perm.cpu().numpy(), order.astype(np.int32))
    assert np.array_equal(ss.cpu().numpy(), src[order].astype(np.int32))
    assert np.array_equal(ds.cpu().numpy(), dst[order].astype(np.int32))
    # int32 ids, no endpoints (what the source-sorted CSR of the backward uses), twice: bitwise reproducible
    for _ in range(2):
        rp2, pm2, s2, d2, st2 = native.topology_build(None, torch.from_numpy(dst.astype(np.int32)).to(DEV), n, gated_fallback=True)
        assert s2 is None and d2 is None and st2.tolist() == [0, 0, general]
        assert torch.equal(rp2, rowptr) and torch.equal(pm2, perm)
    # without the gated general path: the flag tells the caller to sort some other way
    rp3, pm3, _, _, st3 = native.topology_build(torch.from_numpy(src).to(DEV), torch.from_numpy(dst).to(DEV), n, gated_fallback=False)
    assert st3.tolist() == [0, 0, general]
    if not general:
        assert torch.equal(rp3, rowptr) and torch.equal(pm3, perm)


def test_topology_build_empty_and_bad_ids(native):
    rowptr, perm, ss, ds, status = native.topology_build(torch.zeros(0, dtype=torch.int64, device=DEV), torch.zeros(0, dtype=torch.int64, device=DEV), 7)
    assert rowptr.tolist() == [0] * 8 and perm.numel() == 0 and status.tolist() == [0, 0, 0]
    src = torch.tensor([3, 0, 7, 2, 1, 9], dtype=torch.int64, device=DEV)
    dst = torch.tensor([0, 5, 2, -1, 4, 3], dtype=torch.int64, device=DEV)
    rowptr, perm, ss, ds, status = native.topology_build(src, dst, 8)
    assert status.tolist()[:2] == [1, 1]
    # ids are sanitised: nothing downstream can index out of range before the host has raised
    assert int(ss.min()) >= 0 and int(ss.max()) < 8 and int(ds.min()) >= 0 and int(ds.max()) < 8
    assert int(rowptr.min()) >= 0 and int(rowptr.max()) <= 6 and sorted(perm.tolist()) == list(range(6))
    # the module API turns the flags into the reference's IndexError (models/GNN.py:18-20)
    from graphnet_classifier_amd.topology import GraphTopology
    with pytest.raises(IndexError):
        GraphTopology(torch.stack([src, dst.clamp(0, 7)]), 8, device=DEV)
    with pytest.raises(IndexError):
        GraphTopology(torch.stack([src.clamp(0, 7), dst]), 8, device=DEV)


def test_graph_topology_takes_the_general_path_for_edge_lists_that_are_not_graph_ordered(native):
    """Module level: a synchronous build of a random edge list reruns through rocPRIM, a deferred one through the gated
    kernels; both equal the stable sort, and so does the source-sorted CSR of the backward."""
    from graphnet_classifier_amd import topology
    rng = np.random.default_rng(5)
    n, e = 3000, 50000
    ei = torch.from_numpy(rng.integers(0, n, size=(2, e)).astype(np.int64))
    order = np.argsort(ei[1].numpy(), kind="stable")
    topos = [topology.GraphTopology(ei, n, device=DEV)]
    topology.set_validation("deferred")
    try:
        topos.append(topology.GraphTopology(ei, n, device=DEV))
        topology.check_deferred()
    finally:
        topology.set_validation("sync")
    for tp in topos:
        assert np.array_equal(tp.perm.cpu().numpy(), order.astype(np.int32))
        assert np.array_equal(tp.src_sorted.cpu().numpy(), ei[0].numpy()[order].astype(np.int32))
        assert np.array_equal(tp.dst_sorted.cpu().numpy(), ei[1].numpy()[order].astype(np.int32))
        rp, pm = tp.csc
        so = np.argsort(ei[0].numpy()[order], kind="stable")
        assert np.array_equal(pm.cpu().numpy(), so.astype(np.int32))


# ------------------------------------------------------------------ K1 scatter-sum
def test_permute_index_checked_flags_and_sanitises(native):
    src = torch.tensor([3, 0, 7, -1, 2, 9], dtype=torch.int64, device=DEV)
    perm = torch.tensor([5, 4, 3, 2, 1, 0], dtype=torch.int32, device=DEV)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    out = native.permute_index_checked(src, perm, 8, status)
    assert out.tolist() == [0, 2, 0, 7, 0, 3] and status.item() == 1  # 9 and -1 flagged, stored as 0
    status.zero_()
    out = native.permute_index_checked(src.clamp(0, 7), None, 8, status)
    assert out.tolist() == [3, 0, 7, 0, 2, 7] and status.item() == 0


@pytest.mark.parametrize("d", [1, 3, 4, 8, 12, 16, 32, 64, 100, 128, 256, 260])
@pytest.mark.parametrize("use_perm", [True, False])
def test_scatter_sum_bit_exact(native, d, use_perm):
    rng = np.random.default_rng(d)
    n, e = 997, 9001
    index = _random_index(rng, n, e)
    src = torch.from_numpy(rng.standard_normal((e, d)).astype(np.float32))
    rowptr, perm, _ = native.csr_build(index.to(DEV), n)
    if use_perm:
        out = native.scatter_sum_csr(src.to(DEV), rowptr, perm, n)
    else:  # messages already in destination-sorted order
        out = native.scatter_sum_csr(src.to(DEV)[perm.long()], rowptr, None, n)
    ref = O.scatter_sum_fast(src, index, n)
    assert torch.equal(out.cpu(), ref)
    # run-to-run determinism
    out2 = native.scatter_sum_csr(src.to(DEV), rowptr, perm, n) if use_perm else out
    assert torch.equal(out, out2)


def test_scatter_sum_golden_g1(native):
    g = load_golden("g1_scatter.npz")
    src, index = t(g["src"], DEV), t(g["index"], DEV)
    rowptr, perm, _ = native.csr_build(index, 37)
    assert torch.equal(native.scatter_sum_csr(src, rowptr, perm, 37).cpu(), t(g["out_infer"]))
    rowptr40, perm40, _ = native.csr_build(index, 40)
    assert torch.equal(native.scatter_sum_csr(src, rowptr40, perm40, 40).cpu(), t(g["out_dimsize40"]))


def test_scatter_sum_skewed_degrees(native):
    """One hub destination with thousands of in-edges next to empty and degree-1 nodes."""
    rng = np.random.default_rng(5)
    n, e, d = 300, 20000, 64
    idx = rng.integers(0, n, size=e)
    idx[: e // 2] = 17
    index = torch.from_numpy(idx.astype(np.int64))
    src = torch.from_numpy(rng.standard_normal((e, d)).astype(np.float32))
    rowptr, perm, _ = native.csr_build(index.to(DEV), n)
    out = native.scatter_sum_csr(src.to(DEV), rowptr, perm, n)
    assert torch.equal(out.cpu(), O.scatter_sum_fast(src, index, n))


def test_scatter_sum_strided_rows(native):
    """src/out given as column slices of wider buffers (leading dimension > width)."""
    rng = np.random.default_rng(6)
    n, e, d = 257, 3000, 64
    index = _random_index(rng, n, e)
    wide = torch.from_numpy(rng.standard_normal((e, 192)).astype(np.float32)).to(DEV)
    rowptr, perm, _ = native.csr_build(index.to(DEV), n)
    out = native.scatter_sum_csr(wide[:, 64:128], rowptr, perm, n)
    assert torch.equal(out.cpu(), O.scatter_sum_fast(wide[:, 64:128].cpu().contiguous(), index, n))


# ------------------------------------------------------------------ K2 / K6
@pytest.mark.parametrize("d", [1, 3, 16, 64, 128, 200, 256])
def test_gather_rows(native, d):
    rng = np.random.default_rng(d + 100)
    n, e = 513, 7777
    table = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32))
    idx = torch.from_numpy(rng.integers(0, n, size=e).astype(np.int32))
    out = native.gather_rows(table.to(DEV), idx.to(DEV))
    assert torch.equal(out.cpu(), table[idx.long()])


@pytest.mark.parametrize("d", [3, 16, 64, 100, 128])
def test_gather_rows_add(native, d):
    """out[r] = table[index[r]] + addend[r] (gradient of an edge latent that feeds the aggregation and the next block)."""
    rng = np.random.default_rng(d)
    table = torch.from_numpy(rng.standard_normal((77, d)).astype(np.float32))
    add = torch.from_numpy(rng.standard_normal((1001, d)).astype(np.float32))
    idx = torch.from_numpy(rng.integers(0, 77, size=1001).astype(np.int32))
    out = native.gather_rows_add(table.to(DEV), idx.to(DEV), add.to(DEV))
    assert torch.equal(out.cpu(), table[idx.long()] + add)


@pytest.mark.parametrize("space_dim", [1, 2, 3])
def test_edge_features(native, space_dim):
    rng = np.random.default_rng(space_dim)
    n, e = 211, 3001
    pos = torch.from_numpy((rng.random((n, space_dim)) * 32).astype(np.float32))
    ei = torch.from_numpy(rng.integers(0, n, size=(2, e)).astype(np.int64))
    out = native.edge_features(pos.to(DEV), ei[0].int().to(DEV), ei[1].int().to(DEV))
    ref = O.edge_features(pos, ei)
    assert max_abs(out.cpu(), ref) <= 4e-6  # |.|-sum order may differ for space_dim == 3
    assert torch.equal(out.cpu()[:, :space_dim], ref[:, :space_dim])


@pytest.mark.parametrize("hidden,out_dim,hl,ln", [(64, 64, 2, True), (64, 64, 2, False), (48, 40, 2, True), (32, 48, 1, True), (64, 64, 3, True)])
@pytest.mark.parametrize("rows", [40000, 33 * 1024 + 17])
def test_edge_encoder_with_k6_as_its_prologue_is_bit_identical(native, hidden, out_dim, hl, ln, rows):
    """ABI 19 (gnc_mlp_desc_t.ef_pos): the edge encoder computing models/GNN.py:299-302 inside its own launch gives, bit for
    bit, what gnc_edge_features_f32 + the plain launch give (and that pair is checked against the oracle above / below);
    row counts with a ragged last tile and fewer tiles than waves."""
    rng = np.random.default_rng(rows + hidden)
    n = 5003
    pos = torch.from_numpy((rng.random((n, 2)) * 32).astype(np.float32)).to(DEV)
    src = torch.from_numpy(rng.integers(0, n, size=rows).astype(np.int32)).to(DEV)
    dst = torch.from_numpy(rng.integers(0, n, size=rows).astype(np.int32)).to(DEV)
    sd = _mlp_sd(rng, 3, hidden, out_dim, hl, ln)
    keys = sorted(sd, key=lambda k: (int(k.split(".")[2]), k))
    lin = [(sd[k].to(DEV), sd[k.replace("weight", "bias")].to(DEV)) for k in keys if k.endswith("weight") and sd[k].dim() == 2]
    lnp = None
    if ln:
        last = max(int(k.split(".")[2]) for k in sd)
        lnp = (sd[f"m.model.{last}.weight"].to(DEV), sd[f"m.model.{last}.bias"].to(DEV), 1e-5)
    ws, bs = [w for w, _ in lin], [b for _, b in lin]
    ea = native.edge_features(pos, src, dst)
    ref = native.mlp_forward([(ea, None)], ws, bs, ln=lnp)
    got = native.mlp_forward_edge_features(pos, src, dst, ws, bs, ln=lnp)
    assert got is not None, "the weights-resident kernel should serve this shape"
    assert torch.equal(got, ref)
    # against the oracle too (the reference's arithmetic on the reference's features)
    ei = torch.stack([src.long().cpu(), dst.long().cpu()])
    assert max_abs(got.cpu(), O.mlp_forward(sd, "m", O.edge_features(pos.cpu(), ei))) <= 1e-5


@pytest.mark.parametrize("rows", [70001, 33 * 1024, 40000])
def test_three_column_input_and_weights_are_read_where_they_lie(native, rows):
    """ABI 19 (gnc_mlp_operands_in_place_supported): the node encoder's [N, 3] input and [H, 3] first-layer matrix
    (models/GNN.py:251-253, :305) go to the weights-resident kernel unpadded; bit-identical to the launch on zero-padded copies."""
    rng = np.random.default_rng(rows)
    x = torch.from_numpy(rng.standard_normal((rows, 3)).astype(np.float32)).to(DEV)
    sd = _mlp_sd(rng, 3, 64, 64, 2, True)
    ws = [sd[f"m.model.{i}.weight"].to(DEV) for i in (0, 2, 4)]
    bs = [sd[f"m.model.{i}.bias"].to(DEV) for i in (0, 2, 4)]
    lnp = (sd["m.model.5.weight"].to(DEV), sd["m.model.5.bias"].to(DEV), 1e-5)
    pad_launches = []
    real = native._vector_rows
    native._vector_rows = lambda t: (pad_launches.append(tuple(t.shape)), real(t))[1]
    try:
        got = native.mlp_forward([(x, None)], ws, bs, ln=lnp)
    finally:
        native._vector_rows = real
    assert not pad_launches, f"padded copies were made: {pad_launches}"
    xp = torch.nn.functional.pad(x, (0, 1))[:, :3]
    w0p = torch.nn.functional.pad(ws[0], (0, 1))[:, :3]
    ref = native.mlp_forward([(xp, None)], [w0p] + ws[1:], bs, ln=lnp)
    assert torch.equal(got, ref)
    assert max_abs(got.cpu(), O.mlp_forward(sd, "m", x.cpu())) <= 1e-5


@pytest.mark.parametrize("k,m,rows", [(64, 64, 100000), (64, 64, 33 * 1024 + 5), (48, 40, 70001), (64, 36, 40000)])
def test_dual_projection_of_a_large_batch_is_bit_identical_to_two_launches(native, k, m, rows):
    """The W-split's node-side products x Ws^T and x Wd^T (models/GNN.py:58-61) in ONE launch of the weights-resident kernel
    (rows read once, both matrices resident; column slices of the nn.Linear matrix read where they lie)."""
    rng = np.random.default_rng(rows + k)
    x = torch.from_numpy(rng.standard_normal((rows, k)).astype(np.float32)).to(DEV)
    w0 = torch.from_numpy(rng.standard_normal((m, 2 * k + 8)).astype(np.float32)).to(DEV)
    wa, wb = w0[:, :k], w0[:, k:2 * k]
    names = []
    native.set_kernel_timers(type("T", (), {"launch": lambda self, name, t, fn, work=0.0: (names.append(name), fn())[1]})())
    try:
        a, b = native.dual_projection(x, wa, wb)
    finally:
        native.set_kernel_timers(None)
    assert names == [f"mlp_fused_in{k}_h{m}_out{m}_L1x2"], names
    ra = native.mlp_forward([(x, None)], [wa], [None])
    rb = native.mlp_forward([(x, None)], [wb], [None])
    assert torch.equal(a, ra) and torch.equal(b, rb)
    assert max_abs(a.cpu(), x.cpu() @ wa.cpu().t()) <= 1e-4 and max_abs(b.cpu(), x.cpu() @ wb.cpu().t()) <= 1e-4


def test_edge_encoder_k6_prologue_declines_what_it_does_not_serve(native):
    rng = np.random.default_rng(5)
    n = 100
    pos = torch.from_numpy(rng.random((n, 2)).astype(np.float32)).to(DEV)
    pos3 = torch.from_numpy(rng.random((n, 3)).astype(np.float32)).to(DEV)
    mk = lambda rows: (torch.from_numpy(rng.integers(0, n, size=rows).astype(np.int32)).to(DEV),
                       torch.from_numpy(rng.integers(0, n, size=rows).astype(np.int32)).to(DEV))
    w = lambda o, i: torch.from_numpy(rng.standard_normal((o, i)).astype(np.float32)).to(DEV)
    b = lambda o: torch.zeros(o, device=DEV)
    s, d = mk(1984)  # a single small graph: the column-split kernel on a stored table is the route
    assert native.mlp_forward_edge_features(pos, s, d, [w(64, 3), w(64, 64), w(64, 64)], [b(64)] * 3) is None
    s, d = mk(300000)
    assert native.mlp_forward_edge_features(pos3, s, d, [w(64, 4), w(64, 64), w(64, 64)], [b(64)] * 3) is None      # space_dim 3
    assert native.mlp_forward_edge_features(pos, s, d, [w(128, 3), w(128, 128), w(128, 128)], [b(128)] * 3) is None  # width 128
    assert native.mlp_forward_edge_features(pos, s, d, [w(64, 3), w(64, 64), w(64, 64)], [b(64)] * 3) is not None


# ------------------------------------------------------------------ K4 fused MLP
def _mlp_sd(rng, in_dim, hidden, out_dim, hidden_layers, ln):
    dims = [in_dim] + [hidden] * hidden_layers + [out_dim]
    sd, i = {}, 0
    for a, b in zip(dims[:-1], dims[1:]):
        bound = 1.0 / np.sqrt(a)
        sd[f"m.model.{i}.weight"] = torch.from_numpy(rng.uniform(-bound, bound, (b, a)).astype(np.float32))
        sd[f"m.model.{i}.bias"] = torch.from_numpy(rng.uniform(-bound, bound, (b,)).astype(np.float32))
        i += 2
    if ln:
        sd[f"m.model.{i - 1}.weight"] = torch.from_numpy(rng.uniform(0.5, 1.5, (out_dim,)).astype(np.float32))
        sd[f"m.model.{i - 1}.bias"] = torch.from_numpy(rng.uniform(-0.5, 0.5, (out_dim,)).astype(np.float32))
    return sd


def _run_mlp(native, sd, segments, residual=None, activation="ReLU"):
    ws = [k for k in sd if k.endswith("weight") and sd[k].ndim == 2]
    ws.sort(key=lambda k: int(k.split(".")[2]))
    weights = [sd[k].to(DEV) for k in ws]
    biases = [sd[k.replace("weight", "bias")].to(DEV) for k in ws]
    lnk = [k for k in sd if k.endswith("weight") and sd[k].ndim == 1]
    ln = (sd[lnk[0]].to(DEV), sd[lnk[0].replace("weight", "bias")].to(DEV), 1e-5) if lnk else None
    segs = [(tab.to(DEV), idx.to(DEV) if idx is not None else None) for tab, idx in segments]
    return native.mlp_forward(segs, weights, biases, ln=ln, activation=activation,
                              residual=residual.to(DEV) if residual is not None else None)


@pytest.mark.parametrize("tag", [f"{n}_hl{h}" for n in ("ln", "nonorm") for h in (1, 2, 3)])
def test_mlp_golden_g2(native, tag):
    g = load_golden("g2_mlp.npz")
    sd = {"m." + k: v for k, v in sub_state_dict(g, f"{tag}/sd/").items()}
    y = _run_mlp(native, sd, [(t(g["x"]), None)])
    assert max_abs(y.cpu(), t(g[f"{tag}/y"])) < 1e-5


@pytest.mark.parametrize("rows", [1, 31, 32, 33, 127, 128, 129, 1000])
def test_mlp_row_tails(native, rows):
    rng = np.random.default_rng(rows)
    sd = _mlp_sd(rng, 64, 64, 64, 2, True)
    x = torch.from_numpy(rng.standard_normal((rows, 64)).astype(np.float32))
    y = _run_mlp(native, sd, [(x, None)], residual=x)
    ref = O.mlp_forward(sd, "m", x) + x
    assert y.shape == ref.shape and max_abs(y.cpu(), ref) < 1e-5


@pytest.mark.parametrize("in_dim,hidden,out_dim,hl,ln", [
    (3, 128, 128, 2, True),      # node / edge encoder (main.py:72 defaults)
    (3, 64, 64, 2, True),
    (128, 128, 1, 2, False),     # node decoder
    (64, 64, 1, 2, False),
    (7, 16, 5, 1, True),
    (50, 48, 40, 3, True),       # nothing a multiple of 32
    (96, 96, 96, 2, False),
    (256, 256, 256, 2, True),    # config 5 width
    (130, 200, 70, 2, True),
    (64, 32, 256, 2, True),      # out wider than hidden
])
def test_mlp_shapes(native, in_dim, hidden, out_dim, hl, ln):
    rng = np.random.default_rng(in_dim * 1000 + hidden + out_dim)
    rows = 777
    sd = _mlp_sd(rng, in_dim, hidden, out_dim, hl, ln)
    x = torch.from_numpy(rng.standard_normal((rows, in_dim)).astype(np.float32))
    y = _run_mlp(native, sd, [(x, None)])
    ref = O.mlp_forward(sd, "m", x)
    assert max_abs(y.cpu(), ref) < 1e-5


@pytest.mark.parametrize("d", [16, 64, 128, 256])
def test_mlp_edge_processor_fused_gather_concat_residual(native, d):
    """EdgeProcessor (models/GNN.py:57-64): MLP(cat[x[row], x[col], e]) + e with the gathers,
    the concat and the residual fused into the kernel."""
    rng = np.random.default_rng(d)
    n, e = 301, 2111
    sd = _mlp_sd(rng, 3 * d, d, d, 2, True)
    x = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32))
    ea = torch.from_numpy(rng.standard_normal((e, d)).astype(np.float32))
    ei = torch.from_numpy(rng.integers(0, n, size=(2, e)).astype(np.int64))
    y = _run_mlp(native, sd, [(x, ei[0].int()), (x, ei[1].int()), (ea, None)], residual=ea)
    ref = O.mlp_forward(sd, "m", torch.cat([x[ei[0]], x[ei[1]], ea], -1)) + ea
    assert max_abs(y.cpu(), ref) < 1e-5


@pytest.mark.parametrize("act", ["Tanh", "Sigmoid", "SiLU", "GELU", "LeakyReLU", "ELU", "Identity"])
def test_mlp_other_activations(native, act):
    rng = np.random.default_rng(11)
    sd = _mlp_sd(rng, 20, 32, 8, 2, True)
    x = torch.from_numpy(rng.standard_normal((100, 20)).astype(np.float32))
    ws = [sd[f"m.model.{i}.weight"].to(DEV) for i in (0, 2, 4)]
    bs = [sd[f"m.model.{i}.bias"].to(DEV) for i in (0, 2, 4)]
    ln = (sd["m.model.5.weight"].to(DEV), sd["m.model.5.bias"].to(DEV), 1e-5)
    p = 0.01 if act == "LeakyReLU" else 1.0
    y = native.mlp_forward([(x.to(DEV), None)], ws, bs, ln=ln, activation=act, act_param=p)
    f = getattr(torch.nn, act)()
    h = f(O.linear(x, sd["m.model.0.weight"], sd["m.model.0.bias"]))
    h = f(O.linear(h, sd["m.model.2.weight"], sd["m.model.2.bias"]))
    ref = O.layer_norm(O.linear(h, sd["m.model.4.weight"], sd["m.model.4.bias"]), sd["m.model.5.weight"], sd["m.model.5.bias"])
    assert max_abs(y.cpu(), ref) < 2e-5


@pytest.mark.parametrize("d,out", [(16, 16), (64, 64), (64, 128), (128, 128), (96, 40)])
def test_single_linear_projection(native, d, out):
    """num_linear == 1: plain projection x W^T (+ b), used by the W-split of the edge processor."""
    rng = np.random.default_rng(d + out)
    x = torch.from_numpy(rng.standard_normal((333, d)).astype(np.float32))
    w = torch.from_numpy((rng.standard_normal((out, d)) / np.sqrt(d)).astype(np.float32))
    b = torch.from_numpy(rng.standard_normal(out).astype(np.float32))
    y = native.mlp_forward([(x.to(DEV), None)], [w.to(DEV)], [None])
    assert max_abs(y.cpu(), x @ w.t()) < 1e-5
    yb = native.mlp_forward([(x.to(DEV), None)], [w.to(DEV)], [b.to(DEV)])
    assert max_abs(yb.cpu(), x @ w.t() + b) < 1e-5
    # weight given as a column slice of a wider matrix (ld_weight > in_dim)
    wide = torch.from_numpy((rng.standard_normal((out, 3 * d)) / np.sqrt(d)).astype(np.float32)).to(DEV)
    ys = native.mlp_forward([(x.to(DEV), None)], [wide[:, d:2 * d]], [None])
    assert max_abs(ys.cpu(), x @ wide[:, d:2 * d].cpu().t()) < 1e-5


@pytest.mark.parametrize("d", [16, 64, 128])
def test_mlp_additive_segments_equal_concat_form(native, d):
    """W-split: Ws x[src] + Wd x[dst] gathered and ADDED == first Linear on cat[x[src], x[dst], e]."""
    rng = np.random.default_rng(d * 3)
    n, e = 211, 1500
    sd = _mlp_sd(rng, 3 * d, d, d, 2, True)
    x = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32))
    ea = torch.from_numpy(rng.standard_normal((e, d)).astype(np.float32))
    ei = torch.from_numpy(rng.integers(0, n, size=(2, e)).astype(np.int32))
    w0 = sd["m.model.0.weight"].to(DEV)
    ws = [w0[:, 2 * d:], sd["m.model.2.weight"].to(DEV), sd["m.model.4.weight"].to(DEV)]
    bs = [sd[f"m.model.{i}.bias"].to(DEV) for i in (0, 2, 4)]
    ln = (sd["m.model.5.weight"].to(DEV), sd["m.model.5.bias"].to(DEV), 1e-5)
    xd = x.to(DEV)
    ps = native.mlp_forward([(xd, None)], [w0[:, :d]], [None])
    pd = native.mlp_forward([(xd, None)], [w0[:, d:2 * d]], [None])
    y = native.mlp_forward([(ps, ei[0].to(DEV)), (pd, ei[1].to(DEV)), (ea.to(DEV), None)], ws, bs, ln=ln,
                           residual=ea.to(DEV), modes=[native.SEG_ADD, native.SEG_ADD, native.SEG_MATMUL])
    ref = O.mlp_forward(sd, "m", torch.cat([x[ei[0].long()], x[ei[1].long()], ea], -1)) + ea
    assert max_abs(y.cpu(), ref) < 1e-5


@pytest.mark.parametrize("d", [64, 128, 256])
@pytest.mark.parametrize("n,e,hot", [(211, 1500, 0), (40, 33, 0), (5000, 70001, 0), (3000, 90000, 30000), (7, 4096, 0)])
def test_fused_aggregation_epilogue_bit_equals_k1(native, n, e, hot, d):
    """the edge kernel's segmented-sum epilogue (gnc_mlp_desc_t.agg_out) + gnc_agg_fixup_f32 give,
    bit for bit, what K1 gives on the stored rows: empty destinations, tails, more waves than tiles, one
    destination spanning many waves' ranges (`hot` rows of destination 5).  d = 64: weights-resident kernel,
    d = 128: streaming kernel (two 64-column chunks per tile), d = 256: 16-row streaming kernel (four slabs per tile)."""
    rng = np.random.default_rng(n + e + d)
    sd = _mlp_sd(rng, 3 * d, d, d, 2, True)
    dst = rng.integers(0, n, size=e)
    if hot:
        dst[:hot] = 5
    if n > 20:
        dst[dst == 11] = 12  # a destination without edges
    dst = np.sort(dst).astype(np.int32)
    src = rng.integers(0, n, size=e).astype(np.int32)
    x = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).to(DEV)
    ea = torch.from_numpy(rng.standard_normal((e, d)).astype(np.float32)).to(DEV)
    rowptr = torch.from_numpy(np.concatenate([[0], np.cumsum(np.bincount(dst, minlength=n))]).astype(np.int32)).to(DEV)
    w0 = sd["m.model.0.weight"].to(DEV)
    ws = [w0[:, 2 * d:], sd["m.model.2.weight"].to(DEV), sd["m.model.4.weight"].to(DEV)]
    bs = [sd[f"m.model.{i}.bias"].to(DEV) for i in (0, 2, 4)]
    ln = (sd["m.model.5.weight"].to(DEV), sd["m.model.5.bias"].to(DEV), 1e-5)
    ps = native.mlp_forward([(x, None)], [w0[:, :d]], [None])
    pd = native.mlp_forward([(x, None)], [w0[:, d:2 * d]], [None])
    segs = [(ps, torch.from_numpy(src).to(DEV)), (pd, torch.from_numpy(dst).to(DEV)), (ea, None)]
    modes = [native.SEG_ADD, native.SEG_ADD, native.SEG_MATMUL]
    y0 = native.mlp_forward(segs, ws, bs, ln=ln, residual=ea, modes=modes)
    y, agg = native.mlp_forward(segs, ws, bs, ln=ln, residual=ea, modes=modes,
                                aggregate=(torch.from_numpy(dst).to(DEV), rowptr, n))
    if agg is None and (os.environ.get("GNC_MLP_NO_RESIDENT") or os.environ.get("GNC_MLP_NO_STREAM2") or os.environ.get("GNC_MLP_NO_STREAM16")):
        pytest.skip("the kernel that carries the epilogue at this width is switched off by an A/B variable")
    if agg is None:  # a small batch at 65..128 features: the column-split kernel leaves the aggregation to K1 (cheaper there)
        assert native.small_batch_kernel_serves(segs, ws, bs, ln, "ReLU", ea, e, modes)
        agg = native.scatter_sum_csr(y, rowptr, None, n)
    assert torch.equal(y, y0)
    ref = native.scatter_sum_csr(y, rowptr, None, n)
    assert torch.equal(agg, ref)
    # and the oracle's index_add_ order, for completeness
    assert torch.equal(agg.cpu(), O.scatter_sum(y.cpu(), torch.from_numpy(dst).long(), dim_size=n))


@pytest.mark.parametrize("d", [64, 128, 256])
def test_fused_aggregation_random_shapes(native, d):
    """Twelve random (nodes, edges, degree law) draws per width, including power-law degrees, all edges on one
    destination and row counts around multiples of the tile and of the grid's wave count."""
    rng = np.random.default_rng(4242 + d)
    sd = _mlp_sd(rng, 3 * d, d, d, 2, True)
    w0 = sd["m.model.0.weight"].to(DEV)
    ws = [w0[:, 2 * d:], sd["m.model.2.weight"].to(DEV), sd["m.model.4.weight"].to(DEV)]
    bs = [sd[f"m.model.{i}.bias"].to(DEV) for i in (0, 2, 4)]
    ln = (sd["m.model.5.weight"].to(DEV), sd["m.model.5.bias"].to(DEV), 1e-5)
    modes = [native.SEG_ADD, native.SEG_ADD, native.SEG_MATMUL]
    sizes = [(1, 1), (3, 31), (3, 32), (9, 33), (100, 2047), (100, 2048 * 32), (100, 2048 * 32 + 1), (4000, 65537),
             (50000, 50000), (17, 70000), (1000, 123457), (2, 99999)]
    for k, (n, e) in enumerate(sizes):
        if k % 3 == 0:
            dst = rng.integers(0, n, size=e)
        elif k % 3 == 1:
            dst = np.minimum((rng.pareto(1.2, size=e)).astype(np.int64), n - 1)   # heavy tail on low ids
        else:
            dst = np.full(e, n - 1)                                                # one destination takes everything
        dst = np.sort(dst).astype(np.int32)
        src = rng.integers(0, n, size=e).astype(np.int32)
        x = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).to(DEV)
        ea = torch.from_numpy(rng.standard_normal((e, d)).astype(np.float32)).to(DEV)
        rowptr = torch.from_numpy(np.concatenate([[0], np.cumsum(np.bincount(dst, minlength=n))]).astype(np.int32)).to(DEV)
        ps = native.mlp_forward([(x, None)], [w0[:, :d]], [None])
        pd = native.mlp_forward([(x, None)], [w0[:, d:2 * d]], [None])
        dst_t = torch.from_numpy(dst).to(DEV)
        y, agg = native.mlp_forward([(ps, torch.from_numpy(src).to(DEV)), (pd, dst_t), (ea, None)], ws, bs, ln=ln, residual=ea,
                                    modes=modes, aggregate=(dst_t, rowptr, n))
        if agg is None and (os.environ.get("GNC_MLP_NO_RESIDENT") or os.environ.get("GNC_MLP_NO_STREAM2") or os.environ.get("GNC_MLP_NO_STREAM16")):
            pytest.skip("the kernel that carries the epilogue at this width is switched off by an A/B variable")
        if agg is None:  # a small batch at 65..128 features: served without the epilogue, K1 follows (see the test above)
            assert e <= 32768 and d <= 128, (n, e, d)  # (d <= 64 only under the A/B switch GNC_COL16_D64=1)
            continue
        assert torch.equal(agg, native.scatter_sum_csr(y, rowptr, None, n)), (n, e, k)


def test_fused_aggregation_falls_back_when_the_shape_cannot_carry_it(native):
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.standard_normal((100, 256)).astype(np.float32)).to(DEV)  # 256 wide: the 16-row kernel
    w = torch.from_numpy(rng.standard_normal((256, 256)).astype(np.float32)).to(DEV)
    dst = torch.arange(100, dtype=torch.int32, device=DEV)
    rowptr = torch.arange(101, dtype=torch.int32, device=DEV)
    y, agg = native.mlp_forward([(x, None)], [w], [None], aggregate=(dst, rowptr, 100))
    assert agg is None and max_abs(y.cpu(), x.cpu() @ w.cpu().t()) < 2e-4


def test_gather_window_id_outside_the_stated_table_reads_zero(native):
    """gnc_mlp_segment_t.table_rows: the weights-resident kernel gathers through a bounds-checked buffer window,
    so an id at or beyond the stated table reads zeros.  The table is the first half of a larger allocation, so a
    kernel that ignored the bound would read real (non-zero) memory, not fault."""
    rng = np.random.default_rng(77)
    big = torch.from_numpy(rng.standard_normal((200, 64)).astype(np.float32) + 3.0).to(DEV)
    table = big[:100]
    idx = torch.from_numpy(rng.integers(0, 100, size=500).astype(np.int32))
    idx[::7] = torch.from_numpy(rng.integers(100, 200, size=len(idx[::7])).astype(np.int32))
    w = torch.from_numpy((rng.standard_normal((64, 64)) / 8).astype(np.float32))
    b = torch.from_numpy(rng.standard_normal(64).astype(np.float32))
    y = native.mlp_forward([(table, idx.to(DEV))], [w.to(DEV)], [b.to(DEV)])
    rows = big.cpu()[idx.long()] * (idx < 100)[:, None]
    assert max_abs(y.cpu(), rows @ w.t() + b) < 1e-5


def test_mlp_rejects_unsupported(native):
    x = torch.zeros(4, 8, device=DEV)
    w = [torch.zeros(300, 8, device=DEV), torch.zeros(4, 300, device=DEV)]
    b = [torch.zeros(300, device=DEV), torch.zeros(4, device=DEV)]
    with pytest.raises(RuntimeError, match="256"):
        native.mlp_forward([(x, None)], w, b)
    with pytest.raises(NotImplementedError):
        native.mlp_forward([(x, None)], [w[0][:8], w[1][:, :8]], [b[0][:8], b[1]], activation="Softplus")


# ------------------------------------------------------------------ K8 backward kernels
@pytest.mark.parametrize("m,k", [(64, 64), (64, 3), (16, 40), (128, 64), (64, 192), (128, 128), (256, 256), (130, 70), (96, 256),
                                 (256, 4), (65, 65), (200, 296), (100, 296), (256, 512), (1, 256), (128, 256), (256, 128),
                                 (129, 129), (300, 90)])
@pytest.mark.parametrize("rows", [4099, 31])
def test_xty_matches_matmul(native, m, k, rows):
    """Every block shape of the weight-gradient product: the per-wave kernel (a narrow operand), the four workgroup-shared
    instances (128 x 128, 128 x 256, 256 x 128, 256 x 256), operands split into several blocks, ragged column counts,
    fewer rows than one tile."""
    rng = np.random.default_rng(m * 7 + k)
    a = torch.from_numpy(rng.standard_normal((rows, m)).astype(np.float32))
    b = torch.from_numpy(rng.standard_normal((rows, k)).astype(np.float32))
    c, cs = native.xty(a.to(DEV), b.to(DEV))
    ref = a.double().t() @ b.double()
    assert max_abs(c.cpu(), ref.float()) < 2e-3  # sums of ~4k products of O(1) values in fp32
    assert float((c.cpu().double() - ref).abs().max() / ref.abs().max()) < 2e-5
    assert max_abs(cs.cpu(), a.double().sum(0).float()) < 1e-3
    c2, _ = native.xty(a.to(DEV), b.to(DEV))
    assert torch.equal(c, c2)  # fixed summation order
    # operands that are column slices of wider tensors (leading dimension > width, as the backward hands them over)
    wide_a = torch.randn(rows, m + 8, device=DEV)
    wide_b = torch.randn(rows, k + 12, device=DEV)
    wide_a[:, 4:4 + m] = a.to(DEV)
    wide_b[:, 8:8 + k] = b.to(DEV)
    c3, cs3 = native.xty(wide_a[:, 4:4 + m], wide_b[:, 8:8 + k])
    assert torch.equal(c3, c) and torch.equal(cs3, cs)


@pytest.mark.parametrize("width", [64, 40, 128, 200, 256])
def test_colsum_pair(native, width):
    rng = np.random.default_rng(9)
    g = torch.from_numpy(rng.standard_normal((5003, width)).astype(np.float32))
    y = torch.from_numpy(rng.standard_normal((5003, width)).astype(np.float32))
    sg, sgy = native.colsum_pair(g.to(DEV), y.to(DEV))
    assert max_abs(sg.cpu(), g.double().sum(0).float()) < 1e-3
    assert max_abs(sgy.cpu(), (g.double() * y.double()).sum(0).float()) < 1e-3


def _torch_mlp(sd, x_cat, residual=None):
    h = x_cat
    keys = sorted({int(k.split(".")[2]) for k in sd})
    lin = [i for i in keys if sd[f"m.model.{i}.weight"].ndim == 2]
    for n, i in enumerate(lin):
        h = torch.nn.functional.linear(h, sd[f"m.model.{i}.weight"], sd[f"m.model.{i}.bias"])
        if n + 1 < len(lin):
            h = torch.relu(h)
    norm = [i for i in keys if sd[f"m.model.{i}.weight"].ndim == 1]
    for i in norm:
        h = torch.nn.functional.layer_norm(h, (h.size(-1),), sd[f"m.model.{i}.weight"], sd[f"m.model.{i}.bias"], 1e-5)
    return h + residual if residual is not None else h


@pytest.mark.parametrize("in_dims,hidden,out_dim,hl,ln,res", [
    ((64,), 64, 64, 2, True, False),
    ((64, 64), 64, 64, 2, True, True),     # node processor: [x | agg], residual x
    ((64, 64, 64), 64, 64, 2, True, True),  # edge processor, concat form
    ((4,), 64, 64, 2, True, False),        # encoder
    ((64,), 64, 1, 2, False, False),       # decoder
    ((16, 16), 16, 16, 1, True, True),
    ((20,), 32, 8, 3, True, False),
    ((128,), 128, 128, 2, True, False),       # widths 65..128: streamed-weights variant
    ((128, 128), 128, 128, 2, True, True),
    ((128, 128, 128), 128, 128, 2, True, True),
    ((4,), 128, 128, 2, True, False),
    ((128,), 128, 1, 2, False, False),
    ((96, 40), 100, 72, 3, True, False),
    ((256,), 256, 256, 2, True, False),       # widths 129..256: 16-row streamed-weights variant (mlp_backward16.hip)
    ((256, 256), 256, 256, 2, True, True),
    ((256, 256, 256), 256, 256, 2, True, True),
    ((4,), 256, 256, 2, True, False),
    ((256,), 256, 1, 2, False, False),
    ((200, 72), 136, 160, 3, True, False),
    ((64,), 192, 64, 1, True, False),
])
@pytest.mark.parametrize("fused", [False, True])
def test_mlp_backward_kernel_matches_autograd(native, in_dims, hidden, out_dim, hl, ln, res, fused):
    """dz / act / dx / yhat of the K8 data kernel and the xty weight gradients against torch.autograd
    of the same MLP in float64; ``fused``: weight gradients from the fused data + weight-gradient kernel where the
    shape allows it (three Linear layers, one row-ordered segment, widths <= 64)."""
    rng = np.random.default_rng(sum(in_dims) + hidden + out_dim)
    rows = 1000
    in_dim = sum(in_dims)
    sd = _mlp_sd(rng, in_dim, hidden, out_dim, hl, ln)
    tabs = [torch.from_numpy(rng.standard_normal((rows, w)).astype(np.float32)) for w in in_dims]
    residual = tabs[0] if res else None
    gout = torch.from_numpy(rng.standard_normal((rows, out_dim)).astype(np.float32))
    # float64 reference
    sd64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    t64 = [t.double().requires_grad_(True) for t in tabs]
    y = _torch_mlp(sd64, torch.cat(t64, -1), t64[0] if res else None)
    y.backward(gout.double())
    # HIP
    ws = [sd[k].to(DEV) for k in sorted((k for k in sd if sd[k].ndim == 2), key=lambda k: int(k.split(".")[2]))]
    bs = [sd[k.replace("weight", "bias")].to(DEV) for k in sorted((k for k in sd if sd[k].ndim == 2), key=lambda k: int(k.split(".")[2]))]
    lnk = [k for k in sd if k.endswith("weight") and sd[k].ndim == 1]
    lnp = (sd[lnk[0]].to(DEV), sd[lnk[0].replace("weight", "bias")].to(DEV), 1e-5) if lnk else None
    segs = [(t.to(DEV), None) for t in tabs]
    assert native.mlp_backward_supported(segs, ws, bs, lnp, "ReLU", residual.to(DEV) if res else None, rows)
    r = native.mlp_backward(segs, ws, bs, lnp, gout.to(DEV), rows=rows, need_dx=True, fused=fused)
    dx_ref = torch.cat([t.grad for t in t64], -1)
    if res:
        dx_ref[:, :in_dims[0]] -= gout.double()  # the kernel's dx excludes the residual path
    assert max_abs(r["dx"].cpu(), dx_ref.float()) < 2e-5
    lin_keys = sorted((k for k in sd if sd[k].ndim == 2), key=lambda k: int(k.split(".")[2]))
    for li, k in enumerate(lin_keys):
        if "dw" in r:
            dw, db = r["dw"][li], r["db"][li]
        else:
            inp = torch.cat(tabs, -1).to(DEV) if li == 0 else r["act"][li - 1]
            dw, db = native.xty(r["dz"][li], inp)
        gw, gb = sd64[k].grad, sd64[k.replace("weight", "bias")].grad
        assert float((dw.cpu().double() - gw).abs().max()) < 1e-4 * max(1.0, float(gw.abs().max())), k
        assert float((db.cpu().double() - gb).abs().max()) < 1e-4 * max(1.0, float(gb.abs().max())), k
    if lnk:
        # weights-resident shapes form both sums inside the data kernel; the others hand y_hat to colsum_pair
        dbeta, dgamma = r["ln_sums"] if r["ln_sums"] is not None else native.colsum_pair(gout.to(DEV), r["yhat"])
        assert float((dgamma.cpu().double() - sd64[lnk[0]].grad).abs().max()) < 1e-3
        assert float((dbeta.cpu().double() - sd64[lnk[0].replace("weight", "bias")].grad).abs().max()) < 1e-3


@pytest.mark.parametrize("d,e,with_rows", [(64, 2111, True), (64, 2111, False), (48, 1000, True), (64, 32 * 1024 * 3 + 7, True),
                                           (64, 32 * 1024 * 3 + 7, False), (128, 999, True), (256, 777, False)])
def test_mlp_backward_gathered_output_gradient(native, d, e, with_rows):
    """ABI 16: the backward of the scatter-sum that consumed the edge rows (a row gather of grad_agg by destination,
    models/GNN.py:99) folded into the K8 launch (`grad_gather`).  The launch with the gathered part must equal the same
    launch on the materialised gradient gather(grad_agg, dst) [+ grad_out] - bit for bit where the arithmetic is the same
    (dz_0, weight gradients, LayerNorm sums), to rounding for dx (its residual path adds the two parts one after the other) -
    for both forms (rows + gathered, gathered alone), a last partial tile, and the width classes whose kernels do not
    honour the field (the binding then gathers in front of the launch)."""
    rng = np.random.default_rng(d * 7 + e)
    n = 301
    sd = _mlp_sd(rng, d, d, d, 2, True)
    t = lambda a: torch.from_numpy(a.astype(np.float32)).to(DEV)  # noqa: E731
    ps, pd_, ea = t(rng.standard_normal((n, d))), t(rng.standard_normal((n, d))), t(rng.standard_normal((e, d)))
    src = torch.from_numpy(rng.integers(0, n, size=e).astype(np.int32)).to(DEV)
    dst = torch.from_numpy(np.sort(rng.integers(0, n, size=e)).astype(np.int32)).to(DEV)
    gout = t(rng.standard_normal((e, d))) if with_rows else None
    gagg = t(rng.standard_normal((n, d)))
    ws = [sd[f"m.model.{i}.weight"].to(DEV) for i in (0, 2, 4)]
    bs = [sd[f"m.model.{i}.bias"].to(DEV) for i in (0, 2, 4)]
    ln = (sd["m.model.5.weight"].to(DEV), sd["m.model.5.bias"].to(DEV), 1e-5)
    segs = [(ps, src), (pd_, dst), (ea, None)]
    modes = [native.SEG_ADD, native.SEG_ADD, native.SEG_MATMUL]
    g_eff = native.gather_rows(gagg, dst) if gout is None else native.gather_rows_add(gagg, dst, gout)
    ref = native.mlp_backward(segs, ws, bs, ln, g_eff, rows=e, modes=modes, need_dx=True, residual=ea)
    r = native.mlp_backward(segs, ws, bs, ln, gout, rows=e, modes=modes, need_dx=True, residual=ea, grad_gather=(gagg, dst))
    torch.cuda.synchronize()
    folded = r["grad_out"] is None
    assert folded == (d <= 64 and os.environ.get("GNC_NO_GRAD_GATHER_FOLD") is None and os.environ.get("GNC_NO_FUSED_BACKWARD") is None)
    assert r["residual_folded"] == ref["residual_folded"]
    assert torch.equal(r["dz"][0], ref["dz"][0])
    assert max_abs(r["dx"].cpu(), ref["dx"].cpu()) < 2e-6
    if "dw" in ref:
        for a, b_ in zip(r["dw"] + r["db"] + list(r["ln_sums"]), ref["dw"] + ref["db"] + list(ref["ln_sums"])):
            assert torch.equal(a, b_)
    # and against the definition, on the host
    want = gagg.cpu()[dst.cpu().long()] + (gout.cpu() if gout is not None else 0)
    assert torch.equal(g_eff.cpu(), want)


@pytest.mark.parametrize("d,e,nadd,gg", [(64, 2111, 2, 0), (64, 2111, 2, 1), (64, 2111, 2, 2), (48, 1000, 2, 1), (40, 333, 0, 0),
                                         (64, 32 * 1024 * 3 + 7, 2, 1), (64, 32 * 1024 * 2 + 31, 0, 0), (64, 5, 2, 2),
                                         (128, 2111, 2, 0), (128, 70001, 2, 1), (100, 999, 0, 0), (96, 1500, 2, 2),
                                         (256, 2111, 2, 0), (256, 40001, 2, 1), (200, 999, 0, 0), (192, 700, 2, 2), (132, 300, 2, 0),
                                         (64, 4099, -2, 0), (48, 777, -2, 0),
                                         # 128 features above the small-batch backward limit: the register-resident data kernel, behind the
                                         # column-split forward (20011 rows) and behind the streaming forward; -2: the node processor's two dx chunks
                                         (128, 20011, 2, 1), (128, 20011, -2, 0), (128, 70001, -2, 0), (128, 9000, 0, 0)])
def test_mlp_backward_saved_activations(native, d, e, nadd, gg):
    """ABI 16: the training forward keeps the hidden layers' post-activations (`save_act`, written by the weights-resident
    kernel straight from its accumulators) and the fused K8 kernel reads them (`act_given`) instead of recomputing the first
    two Linear layers of every tile.  (i) the saved tensors equal the oracle's hidden activations; (ii) the backward on them
    equals the recomputing backward - same arithmetic from there on - for the plain shape, the W-split shape and both forms
    of the gathered output gradient, incl. a last partial tile and a batch smaller than one tile."""
    rng = np.random.default_rng(d * 11 + e + nadd + gg)
    n = 301
    sd = _mlp_sd(rng, d, d, d, 2, True)
    t = lambda a: torch.from_numpy(a.astype(np.float32)).to(DEV)  # noqa: E731
    ea = t(rng.standard_normal((e, d)))
    ws = [sd[f"m.model.{i}.weight"].to(DEV) for i in (0, 2, 4)]
    bs = [sd[f"m.model.{i}.bias"].to(DEV) for i in (0, 2, 4)]
    ln = (sd["m.model.5.weight"].to(DEV), sd["m.model.5.bias"].to(DEV), 1e-5)
    src = torch.from_numpy(rng.integers(0, n, size=e).astype(np.int32)).to(DEV)
    dst = torch.from_numpy(np.sort(rng.integers(0, n, size=e)).astype(np.int32)).to(DEV)
    if nadd == -2:  # the node processor's shape: two row-ordered MATMUL segments [x | agg], residual x (split K8 path)
        agg_in = t(rng.standard_normal((e, d)))
        ws[0] = t(rng.uniform(-1, 1, (d, 2 * d)) / np.sqrt(2 * d))
        segs, modes, nadd = [(ea, None), (agg_in, None)], None, 0
    elif nadd:
        ps, pd_ = t(rng.standard_normal((n, d))), t(rng.standard_normal((n, d)))
        segs, modes = [(ps, src), (pd_, dst), (ea, None)], [native.SEG_ADD, native.SEG_ADD, native.SEG_MATMUL]
    else:
        segs, modes = [(ea, None)], None
    acts = []
    out = native.mlp_forward(segs, ws, bs, ln=ln, residual=ea, rows=e, modes=modes, save_act=acts)
    assert len(acts) == 2 and acts[0].shape == (e, d)
    plain = native.mlp_forward(segs, ws, bs, ln=ln, residual=ea, rows=e, modes=modes)
    assert torch.equal(out, plain)  # saving changes nothing about the output
    z0 = torch.cat([sg[0].cpu() for sg in segs if sg[1] is None], dim=1) @ ws[0].cpu().t() + bs[0].cpu()
    if nadd:
        z0 = z0 + ps.cpu()[src.cpu().long()] + pd_.cpu()[dst.cpu().long()]
    a0 = torch.relu(z0)
    a1 = torch.relu(a0 @ ws[1].cpu().t() + bs[1].cpu())
    assert max_abs(acts[0].cpu(), a0) < 1e-5 and max_abs(acts[1].cpu(), a1) < 1e-5
    gout = t(rng.standard_normal((e, d))) if gg != 2 else None
    gath = (t(rng.standard_normal((n, d))), dst) if gg else None
    kw = dict(rows=e, modes=modes, need_dx=True, residual=ea, grad_gather=gath)
    ref = native.mlp_backward(segs, ws, bs, ln, gout, **kw)
    r = native.mlp_backward(segs, ws, bs, ln, gout, saved_act=acts, **kw)
    torch.cuda.synchronize()
    assert r["saved_act_used"] and not ref["saved_act_used"]
    tol = 1e-5  # the forward's saved a_l and the recomputed ones differ by rounding (summation order of the first Linear)
    # ... and a pre-activation within rounding of 0 may land on either side of the ReLU in the two runs: that row's
    # gradients then differ legitimately (the derivative is not defined there).  At most a few rows per case.
    odd = torch.zeros(e, dtype=torch.bool)

    def close_rows(a, b_):
        nonlocal odd
        bad = (a.cpu() - b_.cpu()).abs().amax(dim=1) >= tol
        odd |= bad

    if nadd:
        close_rows(r["dz"][0], ref["dz"][0])
    close_rows(r["dx"], ref["dx"])
    if "dw" not in ref:  # split path (streaming kernels): the data kernel hands the SAVED tensors on to the weight-gradient products
        assert all(a.data_ptr() == s_.data_ptr() for a, s_ in zip(r["act"], acts))
        for a, b_ in zip(r["dz"], ref["dz"]):
            close_rows(a, b_)
        for a, b_ in zip(r["act"], ref["act"]):  # what the recomputing kernel emits = what the forward saved
            assert max_abs(a.cpu(), b_.cpu()) < tol
    assert int(odd.sum()) <= max(4, e // 25), int(odd.sum())  # (forward and recomputing backward may be different kernel families:
    # a row counts as soon as ONE of its 2 x d hidden units lands on the other side of the ReLU)
    wtol = 1e-5 if not odd.any() else 1e-2  # sums over the rows: a flipped row moves them by about its own gradient
    sums_r = (r["dw"] + r["db"] if "dw" in r else []) + list(r["ln_sums"])
    sums_ref = (ref["dw"] + ref["db"] if "dw" in ref else []) + list(ref["ln_sums"])
    for a, b_ in zip(sums_r, sums_ref):
        assert float((a - b_).abs().max()) <= wtol * max(1.0, float(b_.abs().max()))


@pytest.mark.parametrize("d", [64, 128, 256])
def test_mlp_backward_wsplit_shape_matches_autograd(native, d):
    """The W-split edge-processor shape in the K8 data kernels of every width class (weights-resident, 32-row streamed,
    16-row streamed): two GATHERED additive segments + the row-ordered e table.  dz_0 (the gradient of both gathered
    projections), dx (gradient through W_e, residual excluded) and the weight gradients against float64 autograd."""
    rng = np.random.default_rng(d + 5)
    n, e = 301, 2111
    sd = _mlp_sd(rng, d, d, d, 2, True)   # first Linear = W_e (the e columns of W0); the projections arrive pre-multiplied
    ps = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32))
    pd_ = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32))
    ea = torch.from_numpy(rng.standard_normal((e, d)).astype(np.float32))
    src = torch.from_numpy(rng.integers(0, n, size=e).astype(np.int32))
    dst = torch.from_numpy(np.sort(rng.integers(0, n, size=e)).astype(np.int32))
    gout = torch.from_numpy(rng.standard_normal((e, d)).astype(np.float32))
    sd64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    ps64, pd64, e64 = (t_.double().requires_grad_(True) for t_ in (ps, pd_, ea))
    z0 = torch.nn.functional.linear(e64, sd64["m.model.0.weight"], sd64["m.model.0.bias"]) + ps64[src.long()] + pd64[dst.long()]
    z0.retain_grad()
    h = torch.relu(z0)
    h = torch.relu(torch.nn.functional.linear(h, sd64["m.model.2.weight"], sd64["m.model.2.bias"]))
    h = torch.nn.functional.linear(h, sd64["m.model.4.weight"], sd64["m.model.4.bias"])
    h = torch.nn.functional.layer_norm(h, (d,), sd64["m.model.5.weight"], sd64["m.model.5.bias"], 1e-5) + e64
    h.backward(gout.double())
    ws = [sd[f"m.model.{i}.weight"].to(DEV) for i in (0, 2, 4)]
    bs = [sd[f"m.model.{i}.bias"].to(DEV) for i in (0, 2, 4)]
    ln = (sd["m.model.5.weight"].to(DEV), sd["m.model.5.bias"].to(DEV), 1e-5)
    segs = [(ps.to(DEV), src.to(DEV)), (pd_.to(DEV), dst.to(DEV)), (ea.to(DEV), None)]
    modes = [native.SEG_ADD, native.SEG_ADD, native.SEG_MATMUL]
    assert native.mlp_backward_supported(segs, ws, bs, ln, "ReLU", ea.to(DEV), e, modes)
    for fused in (False, True):
        r = native.mlp_backward(segs, ws, bs, ln, gout.to(DEV), rows=e, modes=modes, need_dx=True, residual=ea.to(DEV), fused=fused)
        assert max_abs(r["dz"][0].cpu(), z0.grad.float()) < 2e-5
        dx_ref = e64.grad - (0 if r["residual_folded"] else gout.double())
        assert max_abs(r["dx"].cpu(), dx_ref.float()) < 2e-5
        if "dw" in r:
            for li, k in enumerate(("m.model.0.weight", "m.model.2.weight", "m.model.4.weight")):
                gw = sd64[k].grad
                assert float((r["dw"][li].cpu().double() - gw).abs().max()) < 1e-4 * max(1.0, float(gw.abs().max())), (k, fused)


def test_gathered_segment_gradient_runs_through_k1_and_is_reproducible(native):
    """Operator-level concat form (MetaLayer's x[row], x[col] fused as gathered MATMUL segments): the gradient of a
    gathered table is the per-row sum of the edge gradients.  It is formed by K1 through the index's own destination CSR
    (no float atomics): equal to float64 autograd and bit-identical from run to run."""
    from graphnet_classifier_amd import functional as Fn
    rng = np.random.default_rng(77)
    n, e, d = 97, 1501, 32
    sd = _mlp_sd(rng, 3 * d, d, d, 2, True)
    xs = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32))
    ea = torch.from_numpy(rng.standard_normal((e, d)).astype(np.float32))
    src = torch.from_numpy(rng.integers(0, n, size=e).astype(np.int32))
    dst = torch.from_numpy(rng.integers(0, n - 5, size=e).astype(np.int32))  # the last rows receive nothing
    gout = torch.from_numpy(rng.standard_normal((e, d)).astype(np.float32))
    x64, e64 = xs.double().requires_grad_(True), ea.double().requires_grad_(True)
    h = torch.cat([x64[src.long()], x64[dst.long()], e64], dim=-1)
    for i in (0, 2, 4):
        h = torch.nn.functional.linear(h, sd[f"m.model.{i}.weight"].double(), sd[f"m.model.{i}.bias"].double())
        h = torch.relu(h) if i < 4 else h
    h = torch.nn.functional.layer_norm(h, (d,), sd["m.model.5.weight"].double(), sd["m.model.5.bias"].double(), 1e-5) + e64
    h.backward(gout.double())
    ws = [sd[f"m.model.{i}.weight"].to(DEV) for i in (0, 2, 4)]
    bs = [sd[f"m.model.{i}.bias"].to(DEV) for i in (0, 2, 4)]
    ln = (sd["m.model.5.weight"].to(DEV), sd["m.model.5.bias"].to(DEV), 1e-5)
    got = []
    for _ in range(2):
        xd, ed = xs.to(DEV).requires_grad_(True), ea.to(DEV).requires_grad_(True)
        y = Fn.fused_mlp([(xd, src.to(DEV)), (xd, dst.to(DEV)), (ed, None)], ws, bs, ln=ln, residual=ed)
        y.backward(gout.to(DEV))
        got.append((xd.grad.clone(), ed.grad.clone()))
    assert torch.equal(got[0][0], got[1][0]) and torch.equal(got[0][1], got[1][1])
    assert max_abs(got[0][0].cpu(), x64.grad.float()) < 5e-5 * max(1.0, float(x64.grad.abs().max()))
    assert max_abs(got[0][1].cpu(), e64.grad.float()) < 2e-5 * max(1.0, float(e64.grad.abs().max()))
