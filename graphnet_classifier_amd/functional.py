"""Autograd-aware operators over the HIP kernels.

Forward passes run in ``libgnc_hip.so`` (see :mod:`graphnet_classifier_amd.native`).  The
reference's only caller that needs gradients is ``utils/train_model.py:41``
(``loss.backward()``):

* scatter-sum backward is a row gather (HIP kernel K2);
* the fused-MLP backward runs on the hand-written K8 kernels (every width class up to 256); the training
  forward of the widths <= 64 kernels keeps the hidden layers' post-activations for it, the wider ones
  recompute the forward of each tile from the inputs (``GNC_TORCH_BACKWARD=1`` switches to a PyTorch-ROCm
  recompute for A/B runs; nothing here ever runs on the CPU).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

import os

from . import native

# GNC_TORCH_BACKWARD=1 keeps the PyTorch-ROCm recompute backward everywhere (A/B and parity checks)
HIP_BACKWARD = os.environ.get("GNC_TORCH_BACKWARD") is None


def _node_projections(x, w0, dn):
    """(x Ws^T, x Wd^T) [N, H] each: the per-node halves of the W-split first Linear (models/GNN.py:58-61).  ONE launch with two
    output tables for a small batch at 128 features and for a large batch at widths <= 64 (c3: rows read once, both column
    slices resident), two launches otherwise.  (One launch over the stacked weight [Ws ; Wd] writing ONE [N, 2H] table, the
    halves as gather tables, measured no faster - c3 8.14 / 8.29 vs 8.13 / 8.25 ms per step, c2 41.51 vs 41.55 - and was dropped.)"""
    return native.dual_projection(x, w0[:, :dn], w0[:, dn:2 * dn])


class _ScatterSumCSR(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, rowptr, perm, dst_of_row, num_nodes):
        ctx.save_for_backward(dst_of_row)
        return native.scatter_sum_csr(src, rowptr, perm, num_nodes)

    @staticmethod
    def backward(ctx, grad_out):
        (dst_of_row,) = ctx.saved_tensors  # d out[dst(e)] / d src[e] = I
        return native.gather_rows(grad_out.contiguous(), dst_of_row), None, None, None, None


def scatter_sum_csr(src: torch.Tensor, rowptr: torch.Tensor, perm, dst_of_row: torch.Tensor, num_nodes: int):
    """out[v] = sum of src rows whose destination is v.  ``perm`` maps sorted position ->
    row of ``src`` (None when ``src`` is already destination-sorted); ``dst_of_row[r]`` is the
    destination of ``src`` row r (int32), used by the backward gather."""
    return _ScatterSumCSR.apply(src, rowptr, perm, dst_of_row, num_nodes)


class _PermuteRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, perm, inv_perm):
        ctx.save_for_backward(inv_perm)
        return native.gather_rows(table, perm)

    @staticmethod
    def backward(ctx, grad_out):
        (inv_perm,) = ctx.saved_tensors
        return native.gather_rows(grad_out.contiguous(), inv_perm), None, None


def permute_rows(table: torch.Tensor, perm: torch.Tensor, inv_perm: torch.Tensor) -> torch.Tensor:
    """out[k] = table[perm[k]] for a permutation ``perm`` (int32) with inverse ``inv_perm``."""
    return _PermuteRows.apply(table, perm, inv_perm)


def _torch_activation(name: str, param: float):
    if name == "LeakyReLU":
        return lambda v: F.leaky_relu(v, param)
    if name == "ELU":
        return lambda v: F.elu(v, param)
    return {"ReLU": F.relu, "Identity": lambda v: v, "Tanh": torch.tanh, "Sigmoid": torch.sigmoid, "SiLU": F.silu,
            "GELU": F.gelu}[name]


class _MlpMeta:
    """Non-tensor arguments of one fused-MLP call."""
    __slots__ = ("indices", "num_linear", "activation", "act_param", "ln_eps", "has_ln", "has_residual", "rows", "training")

    def __init__(self, indices, num_linear, activation, act_param, ln_eps, has_ln, has_residual, rows):
        self.indices, self.num_linear, self.activation, self.act_param = indices, num_linear, activation, act_param
        self.ln_eps, self.has_ln, self.has_residual, self.rows = ln_eps, has_ln, has_residual, rows
        # the caller's grad mode (inside Function.forward it is always off): a backward can only follow when it was on
        self.training = torch.is_grad_enabled()


class _FusedMLP(torch.autograd.Function):
    """args = tables[S] + weights[L] + biases[L] + (gamma, beta if LN) + (residual if any)."""

    @staticmethod
    def forward(ctx, meta: _MlpMeta, *args):
        s, l = len(meta.indices), meta.num_linear
        tables, weights, biases = args[:s], args[s:s + l], args[s + l:s + 2 * l]
        rest = list(args[s + 2 * l:])
        ln = (rest.pop(0), rest.pop(0), meta.ln_eps) if meta.has_ln else None
        residual = rest.pop(0) if meta.has_residual else None
        # training forward: the kernel also leaves the hidden layers' post-activations (what autograd would keep) for K8
        acts = [] if (meta.training and any(ctx.needs_input_grad) and HIP_BACKWARD) else None
        out = native.mlp_forward(list(zip(tables, meta.indices)), list(weights), list(biases), ln=ln,
                                 activation=meta.activation, act_param=meta.act_param, residual=residual, rows=meta.rows,
                                 save_act=acts, save_need_dx=any(ctx.needs_input_grad[1:1 + s]))
        ctx.meta = meta
        # the saved post-activations go through save_for_backward like the inputs: autograd then frees them as soon as this
        # node's backward has run (held as plain attributes they lived until the whole graph died: +30 GB at c5)
        ctx.n_acts = len(acts) if acts else 0
        ctx.save_for_backward(*args, *(acts or []))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        meta, saved = ctx.meta, ctx.saved_tensors
        args = saved[:len(saved) - ctx.n_acts]
        acts = list(saved[len(saved) - ctx.n_acts:]) if ctx.n_acts else None
        s, l = len(meta.indices), meta.num_linear
        need = ctx.needs_input_grad[1:]
        if meta.rows == 0:  # no rows: every gradient is a zero of its argument's shape (an edge-less graph, superpixel.py:70-71)
            return (None,) + tuple(torch.zeros_like(a) if n else None for a, n in zip(args, need))
        if HIP_BACKWARD:
            hip = _fused_mlp_backward_hip(meta, args, need, grad_out, acts)
            if hip is None:  # an activation other than ReLU (or a shape outside K8): layer by layer, still on this library
                hip = _layerwise_mlp_backward_hip(meta, args, need, grad_out)
            return (None,) + hip
        # GNC_TORCH_BACKWARD=1 (A/B and parity runs only): PyTorch-ROCm recompute backward of the same ops on the GPU
        leaves = [a.detach().requires_grad_(bool(n)) for a, n in zip(args, need)]
        tables, weights, biases = leaves[:s], leaves[s:s + l], leaves[s + l:s + 2 * l]
        rest = leaves[s + 2 * l:]
        act = _torch_activation(meta.activation, meta.act_param)
        with torch.enable_grad():
            h = torch.cat([t if i is None else t[i.long()] for t, i in zip(tables, meta.indices)], dim=-1)
            for k in range(l):
                h = F.linear(h, weights[k], biases[k])
                if k + 1 < l:
                    h = act(h)
            if meta.has_ln:
                h = F.layer_norm(h, (h.size(-1),), rest[0], rest[1], meta.ln_eps)
            if meta.has_residual:
                h = h + rest[-1]
            wanted = [x for x, n in zip(leaves, need) if n]
            grads = iter(torch.autograd.grad(h, wanted, grad_out.contiguous(), allow_unused=True))
        return (None,) + tuple(next(grads) if n else None for n in need)


def _fused_mlp_backward_hip(meta: _MlpMeta, args, need, grad_out, saved_act=None):
    """Backward of one fused-MLP call on the HIP kernels (K8): data path + one skinny GEMM per Linear.
    Returns the gradient tuple in argument order, or None when the shape is outside the kernels."""
    s, l = len(meta.indices), meta.num_linear
    tables, weights, biases = list(args[:s]), list(args[s:s + l]), list(args[s + l:s + 2 * l])
    rest = list(args[s + 2 * l:])
    ln = (rest[0], rest[1], meta.ln_eps) if meta.has_ln else None
    residual = rest[-1] if meta.has_residual else None
    segments = list(zip(tables, meta.indices))
    if (meta.activation != "ReLU" or l < 2 or not grad_out.is_cuda
            or not native.mlp_backward_supported(segments, weights, biases, ln, meta.activation, residual, meta.rows,
                                                 saved_act=saved_act)):
        return None
    grad_out = grad_out.contiguous()
    need_tables = any(need[:s])
    r = native.mlp_backward(segments, weights, biases, ln, grad_out, rows=meta.rows, need_dx=need_tables, saved_act=saved_act,
                            defer_ln_sums=True)
    grads = [None] * len(args)
    # inputs: dx is in concat (= weight column) order
    off = 0
    layer0_inputs = []
    for k, (t, idx) in enumerate(segments):
        w = t.size(1)
        rows_k = t if idx is None else native.gather_rows(t, idx)
        layer0_inputs.append(rows_k)
        if need[k]:
            gk = r["dx"][:, off:off + w]
            if idx is None:
                grads[k] = gk
            else:  # rows gathered by index (operator-level API only): sum the row gradients per table row with K1 through
                # the index's own destination CSR - stable order, no float atomics, bitwise reproducible
                from .topology import get_destination_csr
                csr = get_destination_csr(idx, t.size(0), t.device)
                grads[k] = native.scatter_sum_csr(gk, csr.rowptr, csr.perm, t.size(0))
        off += w
    # weights and biases: dW_l = dz_l^T (input of Linear l), db_l = column sums of dz_l - every product of this MLP (and
    # the row sums of its LayerNorm partials) in one xty_multi call: one launch for a small batch
    products, slots = [], []
    for k in range(l):
        if not (need[s + k] or need[s + l + k]):
            continue
        if "dw" in r:  # fused kernel: weight gradients came out of the data kernel (single row-ordered segment)
            grads[s + k] = r["dw"][k] if need[s + k] else None
            grads[s + l + k] = r["db"][k] if need[s + l + k] else None
            continue
        dz = r["dz"][k]
        if k == 0:
            dw = torch.empty(weights[0].size(0), sum(t.size(1) for t in layer0_inputs), dtype=torch.float32, device=dz.device)
            off = 0
            for rows_k in layer0_inputs:  # column blocks of dW_0, in concat order
                products.append((dz, rows_k, dw[:, off:off + rows_k.size(1)]))
                slots.append((k, off == 0))
                off += rows_k.size(1)
            grads[s + k] = dw if need[s + k] else None
        else:
            products.append((dz, r["act"][k - 1], None))
            slots.append((k, True))
    ln_part = r.get("ln_part")
    res, sums = native.xty_multi(products, [ln_part] if ln_part is not None else []) if (products or ln_part is not None) else ([], [])
    for (k, first), (c, cs) in zip(slots, res):
        if k > 0:
            grads[s + k] = c if need[s + k] else None
        if first:
            grads[s + l + k] = cs if need[s + l + k] else None
    pos = s + 2 * l
    if meta.has_ln:
        if ln_part is not None:
            od = ln_part.size(1) // 2
            dbeta, dgamma = sums[0][:od], sums[0][od:]
        else:
            dbeta, dgamma = r["ln_sums"] if r["ln_sums"] is not None else native.colsum_pair(grad_out, r["yhat"])
        grads[pos] = dgamma if need[pos] else None
        grads[pos + 1] = dbeta if need[pos + 1] else None
        pos += 2
    if meta.has_residual and need[pos]:
        grads[pos] = grad_out
    return tuple(grads)


def _layerwise_mlp_backward_hip(meta: _MlpMeta, args, need, grad_out):
    """Backward of one fused-MLP call for the shapes the K8 kernels do not take - an activation other than ReLU
    (models/MLP.py:21 accepts any nn.<Name>) - layer by layer on the library's own kernels: the pre-activations are
    recomputed with single-Linear K4 launches, act / act' / the LayerNorm backward are row-wise kernels
    (csrc/elementwise.hip), da = dz W is a single-Linear launch on the transposed weight, dW = dz^T a and db come from
    gnc_xty_f32, gathered segments get their gradient through K1.  Nothing here is on the path of a ReLU model."""
    s, l = len(meta.indices), meta.num_linear
    tables, weights, biases = list(args[:s]), list(args[s:s + l]), list(args[s + l:s + 2 * l])
    rest = list(args[s + 2 * l:])
    act, ap = meta.activation, meta.act_param
    if act not in native.ACTIVATIONS or not grad_out.is_cuda:
        raise NotImplementedError(f"backward of an MLP with activation nn.{act} has no HIP kernel")
    segments = list(zip(tables, meta.indices))
    grad_out = grad_out.contiguous()
    # forward recompute: z_k (pre-activations) and a_k = act(z_k)
    zs, acts = [], []
    z = native.mlp_forward(segments, [weights[0]], [biases[0]], activation="Identity", rows=meta.rows)
    for k in range(l):
        zs.append(z)
        if k + 1 < l:
            a = native.activation(z, act, ap)
            acts.append(a)
            z = native.mlp_forward([(a, None)], [weights[k + 1]], [biases[k + 1]], activation="Identity")
    grads = [None] * len(args)
    pos = s + 2 * l
    if meta.has_ln:
        dz, yhat = native.layer_norm_backward(zs[-1], rest[0], grad_out, meta.ln_eps)
        if need[pos] or need[pos + 1]:
            dbeta, dgamma = native.colsum_pair(grad_out, yhat)
            grads[pos] = dgamma if need[pos] else None
            grads[pos + 1] = dbeta if need[pos + 1] else None
        pos += 2
    else:
        dz = grad_out
    if meta.has_residual and need[pos]:
        grads[pos] = grad_out
    for k in range(l - 1, -1, -1):
        want_w = need[s + k] or need[s + l + k]
        if k > 0:
            if want_w:
                dw, db = native.xty(dz, acts[k - 1])
                grads[s + k] = dw if need[s + k] else None
                grads[s + l + k] = db if need[s + l + k] else None
            da = native.mlp_forward([(dz, None)], [weights[k].t().contiguous()], [None], activation="Identity")  # dz W_k
            dz = native.activation_backward(zs[k - 1], da, act, ap)
            continue
        # first Linear: its input is the virtual concat of the segments (gathered ones materialised for the products)
        off, parts, db0 = 0, [], None
        for q, (t, idx) in enumerate(segments):
            w = t.size(1)
            if want_w:
                c, cs = native.xty(dz, t if idx is None else native.gather_rows(t, idx))
                parts.append(c)
                db0 = cs if db0 is None else db0
            if need[q]:
                gq = native.mlp_forward([(dz, None)], [weights[0][:, off:off + w].t().contiguous()], [None], activation="Identity")
                if idx is None:
                    grads[q] = gq
                else:
                    from .topology import get_destination_csr
                    csr = get_destination_csr(idx, t.size(0), t.device)
                    grads[q] = native.scatter_sum_csr(gq, csr.rowptr, csr.perm, t.size(0))
            off += w
        if want_w:
            grads[s] = (torch.cat(parts, dim=1) if len(parts) > 1 else parts[0]) if need[s] else None
            grads[s + l] = db0 if need[s + l] else None
    return tuple(grads)


def fused_mlp(segments, weights, biases, ln=None, activation: str = "ReLU", act_param: float = 0.0, residual=None,
              rows: int | None = None) -> torch.Tensor:
    """segments: list of (table fp32 [*, w], index int32 [rows] | None); see native.mlp_forward."""
    tables = [t for t, _ in segments]
    indices = tuple(i for _, i in segments)
    if rows is None:
        rows = indices[0].numel() if indices[0] is not None else tables[0].size(0)
    meta = _MlpMeta(indices, len(weights), activation, float(act_param), float(ln[2]) if ln is not None else 0.0,
                    ln is not None, residual is not None, int(rows))
    args = list(tables) + list(weights) + list(biases)
    if ln is not None:
        args += [ln[0], ln[1]]
    if residual is not None:
        args.append(residual)
    return _FusedMLP.apply(meta, *args)


class _EdgeProcessorWSplit(torch.autograd.Function):
    """EdgeProcessor (models/GNN.py:57-64) on destination-sorted edges with the first Linear
    split algebraically:  W0 [x_src | x_dst | e]^T = Ws x[src] + Wd x[dst] + We e.
    Two per-NODE projection launches (num_linear == 1, no bias) + one per-EDGE launch whose
    first two segments are gathered and ADDED.  args = x, e, weights[L], biases[L], gamma, beta."""

    @staticmethod
    def forward(ctx, meta, topo, x, e, *params):
        ctx.topo = topo
        src, dst, num_linear, activation, act_param, ln_eps, has_ln = meta[:7]
        with_agg = len(meta) > 7 and meta[7]
        weights, biases = list(params[:num_linear]), list(params[num_linear:2 * num_linear])
        ln = (params[2 * num_linear], params[2 * num_linear + 1], ln_eps) if has_ln else None
        dn = x.size(1)
        w0 = weights[0]
        ps, pd = _node_projections(x, w0, dn)                                 # [N, H] = x Ws^T, x Wd^T
        training = len(meta) > 8 and meta[8] and any(ctx.needs_input_grad) and HIP_BACKWARD
        acts = [] if training else None   # the hidden layers' post-activations, kept for K8 where the kernel can
        out = native.mlp_forward([(ps, src), (pd, dst), (e, None)], [w0[:, 2 * dn:]] + weights[1:], biases, ln=ln,
                                 activation=activation, act_param=act_param, residual=e, rows=e.size(0),
                                 modes=[native.SEG_ADD, native.SEG_ADD, native.SEG_MATMUL],
                                 aggregate=(topo.dst_sorted, topo.rowptr, topo.num_nodes) if with_agg else None,
                                 save_act=acts, save_need_dx=bool(ctx.needs_input_grad[3]))
        # kept for the backward through save_for_backward (freed when this node's backward has run): the projections - its
        # gathered inputs again, [N, H] each - and the post-activations
        extra = ([ps, pd] + acts) if training else []
        ctx.n_extra = len(extra)
        ctx.meta = meta[:7]
        ctx.with_agg = with_agg
        ctx.set_materialize_grads(False)  # an unused output (the last block's e') arrives as None, not as zeros
        ctx.save_for_backward(x, e, *params, *extra)
        if not with_agg:
            return out
        out, agg = out
        if agg is None:  # the launch shape cannot carry the epilogue: K1 as a separate launch
            agg = native.scatter_sum_csr(out, topo.rowptr, None, topo.num_nodes)
        return out, agg

    @staticmethod
    def backward(ctx, grad_out, grad_agg=None):
        src, dst, num_linear, activation, act_param, ln_eps, has_ln = ctx.meta
        need = ctx.needs_input_grad[2:]
        if not ctx.with_agg:
            grad_agg = None
        if grad_out is None and grad_agg is None:
            return (None, None) + tuple(None for _ in need)
        if ctx.saved_tensors[1].size(0) == 0:  # an edge-less graph (superpixel.py:70-71): zeros of every argument's shape
            saved0 = ctx.saved_tensors[:len(ctx.saved_tensors) - getattr(ctx, "n_extra", 0)]
            return (None, None) + tuple(torch.zeros_like(a) if n else None for a, n in zip(saved0, need))
        # e' feeds the aggregation (its backward is a row gather by destination: grad_agg[dst]) AND whatever consumed e'
        # itself (grad_out): the K8 launch adds the gathered rows itself where its kernel can
        if HIP_BACKWARD:
            hip = _edge_wsplit_backward_hip(ctx, grad_out, grad_agg)
            if hip is None:  # (EdgeProcessor.forward_sorted only takes the W-split route for shapes K8 serves)
                raise NotImplementedError("W-split edge processor: this shape has no HIP backward kernel (use the concat form)")
            return (None, None) + hip
        # GNC_TORCH_BACKWARD=1 (A/B and parity runs only): PyTorch-ROCm recompute backward of the same ops on the GPU
        if grad_agg is not None and grad_out is not None:
            grad_out = native.gather_rows_add(grad_agg.contiguous(), dst, grad_out.contiguous())
        elif grad_agg is not None:
            grad_out = native.gather_rows(grad_agg.contiguous(), dst)
        saved = ctx.saved_tensors
        saved = saved[:len(saved) - getattr(ctx, "n_extra", 0)]
        leaves = [a.detach().requires_grad_(bool(n)) for a, n in zip(saved, need)]
        x, e, params = leaves[0], leaves[1], leaves[2:]
        act = _torch_activation(activation, act_param)
        with torch.enable_grad():
            h = torch.cat([x[src.long()], x[dst.long()], e], dim=-1)
            for k in range(num_linear):
                h = F.linear(h, params[k], params[num_linear + k])
                if k + 1 < num_linear:
                    h = act(h)
            if has_ln:
                h = F.layer_norm(h, (h.size(-1),), params[2 * num_linear], params[2 * num_linear + 1], ln_eps)
            h = h + e
            wanted = [t for t, n in zip(leaves, need) if n]
            grads = iter(torch.autograd.grad(h, wanted, grad_out.contiguous(), allow_unused=True))
        return (None, None) + tuple(next(grads) if n else None for n in need)


def _edge_wsplit_backward_hip(ctx, grad_out, grad_agg=None):
    """HIP backward of the W-split edge processor.  Per edge: the K8 data kernel (recompute + chain),
    then dz0 is (i) the gradient of both gathered projections - summed per node through the two CSRs
    (destination-sorted: the forward's; source-sorted: topo.csc) - and (ii) the left operand of dWe."""
    src, dst, num_linear, activation, act_param, ln_eps, has_ln = ctx.meta
    topo = ctx.topo
    need = ctx.needs_input_grad[2:]
    saved = ctx.saved_tensors
    n_extra = getattr(ctx, "n_extra", 0)
    extra = list(saved[len(saved) - n_extra:]) if n_extra else []
    saved = saved[:len(saved) - n_extra]
    x, e = saved[0], saved[1]
    params = list(saved[2:])
    weights, biases = params[:num_linear], params[num_linear:2 * num_linear]
    ln = (params[2 * num_linear], params[2 * num_linear + 1], ln_eps) if has_ln else None
    if topo is None or activation != "ReLU" or not (grad_out if grad_out is not None else grad_agg).is_cuda:
        return None
    dn = x.size(1)
    w0 = weights[0]
    h = w0.size(0)
    ws_, wd_, we_ = w0[:, :dn], w0[:, dn:2 * dn], w0[:, 2 * dn:]
    # the forward's projections are needed again as the gathered additive inputs (kept by the training forward)
    if extra:
        ps, pd = extra[0], extra[1]
    else:
        ps, pd = _node_projections(x, w0, dn)
    segments = [(ps, src), (pd, dst), (e, None)]
    modes = [native.SEG_ADD, native.SEG_ADD, native.SEG_MATMUL]
    wl = [we_] + weights[1:]
    if not native.mlp_backward_supported(segments, wl, biases, ln, activation, e, e.size(0), modes):
        return None
    r = native.mlp_backward(segments, wl, biases, ln, grad_out.contiguous() if grad_out is not None else None, rows=e.size(0),
                            modes=modes, need_dx=bool(need[1]), residual=e,
                            grad_gather=(grad_agg.contiguous(), dst) if grad_agg is not None else None,
                            saved_act=extra[2:] or None, defer_ln_sums=True)
    grad_out = r["grad_out"]  # effective row-ordered gradient (None when the launch gathered part of it itself)
    dz0 = r["dz"][0]
    grads = [None] * (2 + len(params))
    if need[1]:  # through We, plus the residual path (folded into the kernel's dx when it can)
        grads[1] = r["dx"] if r["residual_folded"] else r["dx"] + grad_out
    # node side: d(ps)[v] = sum of dz0 over edges with src == v, d(pd)[v] = ... dst == v
    csc_rowptr, csc_perm = topo.csc
    dps = native.scatter_sum_csr(dz0, csc_rowptr, csc_perm, topo.num_nodes)
    dpd = native.scatter_sum_csr(dz0, topo.rowptr, None, topo.num_nodes)
    if need[0]:  # dx = dps Ws + dpd Wd: one projection launch over [dps | dpd] with the weight [Ws ; Wd]^T
        grads[0] = native.projection_t2(dps, dpd, w0, dn)  # (a small batch reads W0 transposed, in one launch, without the copy)
    # every weight-gradient product of this processor (and the row sums of its LayerNorm partials) in one xty_multi call:
    # one launch for a small batch; dW_0 = [dps^T x | dpd^T x | dz0^T e] is written block by block in place
    products, slots = [], []
    if need[2] or need[2 + num_linear]:
        dw0 = torch.empty(h, 2 * dn + e.size(1), dtype=torch.float32, device=e.device)
        products += [(dps, x, dw0[:, :dn]), (dpd, x, dw0[:, dn:2 * dn])]
        slots += [None, None]
        if "dw" in r:
            dw0[:, 2 * dn:] = r["dw"][0]
            grads[2 + num_linear] = r["db"][0]
        else:
            products.append((dz0, e, dw0[:, 2 * dn:]))
            slots.append(("b", 0))
        grads[2] = dw0
    for k in range(1, num_linear):
        if "dw" in r:
            grads[2 + k], grads[2 + num_linear + k] = r["dw"][k], r["db"][k]
        else:
            products.append((r["dz"][k], r["act"][k - 1], None))
            slots.append(("wb", k))
    ln_part = r.get("ln_part")
    res, sums = native.xty_multi(products, [ln_part] if ln_part is not None else [])
    for slot, (c, cs) in zip(slots, res):
        if slot is None:
            continue
        kind, k = slot
        if kind == "wb":
            grads[2 + k] = c
        grads[2 + num_linear + k] = cs
    if has_ln:
        if ln_part is not None:
            od = ln_part.size(1) // 2
            dbeta, dgamma = sums[0][:od], sums[0][od:]
        else:
            dbeta, dgamma = r["ln_sums"] if r["ln_sums"] is not None else native.colsum_pair(grad_out, r["yhat"])
        grads[2 + 2 * num_linear] = dgamma
        grads[2 + 2 * num_linear + 1] = dbeta
    return tuple(g if n else None for g, n in zip(grads, need))


def edge_processor_wsplit(x, e, topo, weights, biases, ln, activation="ReLU", act_param=0.0, with_agg=False):
    """``with_agg=True`` returns ``(e', agg)``: the per-destination sums come from the edge launch's epilogue (or
    from K1 when that launch cannot carry it) and both outputs are differentiable."""
    meta = (topo.src_sorted, topo.dst_sorted, len(weights), activation, float(act_param),
            float(ln[2]) if ln is not None else 0.0, ln is not None, bool(with_agg), torch.is_grad_enabled())
    args = list(weights) + list(biases) + ([ln[0], ln[1]] if ln is not None else [])
    return _EdgeProcessorWSplit.apply(meta, topo, x, e, *args)


def edge_processor_wsplit_aggregated(x, e, topo, weights, biases, ln, activation="ReLU", act_param=0.0):
    """Inference form of ``edge_processor_wsplit`` (no autograd graph): the edge launch also forms the
    per-destination sums of its output rows in its epilogue (include/gnc_hip.h, ``agg_out``).
    Returns ``(e', agg)``; ``agg`` is None when the launch shape cannot carry the epilogue (the caller then
    runs K1).  ``agg`` is bit-identical to ``scatter_sum_csr(e', topo.rowptr)``."""
    dn = x.size(1)
    w0 = weights[0]
    ps, pd = _node_projections(x, w0, dn)
    return native.mlp_forward([(ps, topo.src_sorted), (pd, topo.dst_sorted), (e, None)], [w0[:, 2 * dn:]] + list(weights[1:]),
                              list(biases), ln=ln, activation=activation, act_param=act_param, residual=e, rows=e.size(0),
                              modes=[native.SEG_ADD, native.SEG_ADD, native.SEG_MATMUL],
                              aggregate=(topo.dst_sorted, topo.rowptr, topo.num_nodes))


class _Readout(torch.autograd.Function):
    """LinearClassifier.forward for ONE graph (models/GNN.py:312-325) as one launch each way (csrc/readout.hip)."""

    @staticmethod
    def forward(ctx, y, w1, b1, w2, b2, w3, b3):
        logits, h1, h2 = native.readout_forward(y, w1, b1, w2, b2, w3, b3)
        ctx.save_for_backward(y, w1, w2, w3, h1, h2)
        ctx.has_bias = (b1 is not None, b2 is not None, b3 is not None)
        return logits

    @staticmethod
    def backward(ctx, grad_logits):
        y, w1, w2, w3, h1, h2 = ctx.saved_tensors
        need = ctx.needs_input_grad
        dy, dw1, db1, dw2, db2, dw3, db3 = native.readout_backward(grad_logits, y, w1, w2, w3, h1, h2, need_dy=need[0])
        hb = ctx.has_bias
        return (dy.view_as(y) if need[0] else None, dw1 if need[1] else None, db1 if (need[2] and hb[0]) else None,
                dw2 if need[3] else None, db2 if (need[4] and hb[1]) else None, dw3 if need[5] else None,
                db3 if (need[6] and hb[2]) else None)


def readout(y, w1, b1, w2, b2, w3, b3):
    """Single-graph read-out MLP; ``y`` 1-D float32 on the GPU."""
    return _Readout.apply(y, w1, b1, w2, b2, w3, b3)


def edge_features(pos: torch.Tensor, src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """[pos[dst]-pos[src], L1 norm] per edge (models/GNN.py:299-302).  ``pos`` is input data
    (utils/dataloader.py:50); gradients with respect to it are not provided."""
    if pos.requires_grad and torch.is_grad_enabled():
        raise NotImplementedError("gradients with respect to node positions are not implemented")
    return native.edge_features(pos, src, dst)
