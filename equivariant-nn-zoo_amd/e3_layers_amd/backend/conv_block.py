"""One MessagePassing layer (FactorizedConvolution + Gate) as ONE autograd node.

The arithmetic is the kernels' (``csrc/``); between them sits the host.  Composed from one ``autograd.Function`` per
kernel, a convolution layer costs ~12 Function applies forward and ~12 graph nodes backward, each with its own
argument checks, context bookkeeping, stream switches and engine hand-offs: 0.45 + 0.5 ms of Python per layer and
step -- at 256 molecules the host enqueue time (6.0 ms per step) had caught up with the GPU time (7.1 ms), so faster
kernels stopped paying.  Here the launches of a layer are issued back to back from one forward and one backward
function, on the same three streams (radial MLP | self-connection | node features -> tensor product) plus the
weight-gradient stream, with explicit events instead of autograd's per-node stream hand-offs.

What it replaces (reference, per layer): ``FactorizedConvolution.forward`` + ``Gate``
(``e3_layers/nn/message_passing.py:91-124, 249``).  The composed path stays -- it is the definition, it serves
double backward (force training), un-keyed node attributes, the 'norm' nonlinearity -- and ``E3K_CONV_BLOCK=0``
forces it; ``tests/test_gpu_model.py::test_conv_block_equals_composed_layers`` pins the two against each other.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch

from .tuning import knob as _knob

from . import lib as L
from . import ops, radial_table
from .graph import GraphTopo

ENABLED = _knob("E3K_CONV_BLOCK")
LOOK_AHEAD = _knob("E3K_BLOCK_LOOK_AHEAD")     # the next layer's radial branch issued one layer early
AHEAD_STATS = [0]      # look-ahead results consumed (tests)
# The GEMMs of a layer that share an operand go out in one e3k_gemm_multi call each -- backward: the input gradients of the
# trailing Linear and of the self-connection (both read the gradient of the convolution output), then linear_1's input
# gradient accumulated on top, then the weight gradients (forked: trailing Linear + self-connection right behind the gate
# backward, linear_1's after the tensor product; on one stream: all three together); forward, on one stream: keyed
# self-connection + linear_1 (both read the node features).  Measured against one call per operator: 256 molecules 5.73
# vs 5.79 ms, 32 molecules 3.45 vs 3.38 ms (host-bound either way) -- launches per step 263 -> 228.
MERGE = 1


class ConvBlockPlan:
    """Static description of one layer (specs own their cached launch descriptors)."""

    def __init__(self, *, in_blocks, lin1_spec, mlp_alphas, mlp_act, mlp_cst, mlp_k0, last_spec, tp_plan, post_spec, scale,
                 sc_spec, sc_m_off, sc_ld_m, gate_spec, addend: bool = False):
        self.in_blocks, self.lin1_spec = in_blocks, lin1_spec
        # addend: the self-connection is NOT part of the block (general, un-keyed node attributes: the outer-product form of
        # ops.fctp); its output [N, d_conv] (cf) is handed in, the trailing Linear accumulates on top of it, and the backward
        # hands the gradient of the convolution output back for it (native executor only; travels in the ``m_pre`` slot)
        self.addend = bool(addend)
        self.mlp_alphas, self.mlp_act, self.mlp_cst, self.last_spec = tuple(mlp_alphas), mlp_act, float(mlp_cst), last_spec
        self.mlp_k0 = int(mlp_k0)
        self.tp_plan, self.post_spec, self.scale = tp_plan, post_spec, float(scale)
        self.sc_spec, self.sc_m_off, self.sc_ld_m = sc_spec, tuple(sc_m_off) if sc_m_off is not None else None, sc_ld_m
        self.gate_spec = gate_spec
        self.prefetched = None     # radial branch of THIS layer issued by the previous layer's forward (look-ahead)
        self.guard_key = None      # the radial MLP's last-layer Parameter: key of its knot-table guard (radial_table.guard)


class _on:
    """``with _on(stream, main)``: launches go to ``stream`` (no-op when it is ``main``)."""

    __slots__ = ("st", "main")

    def __init__(self, st, main):
        self.st, self.main = st, main

    def __enter__(self):
        if self.st is not self.main:
            torch.cuda.set_stream(self.st)

    def __exit__(self, *exc):
        if self.st is not self.main:
            torch.cuda.set_stream(self.main)
        return False


def _wait(consumer, producer):
    if consumer is not producer:
        consumer.wait_stream(producer)


def _rec(t, st, main):
    if t is not None and st is not main:
        t.record_stream(st)


def _grad_buffer(weight, need: bool):
    """(buffer to accumulate into, tensor to return to autograd) for one parameter."""
    if not need:
        return None, None
    sink = ops._sink_for(weight)
    if sink is not None:
        return sink, None
    buf = torch.zeros_like(weight)
    return buf.view(-1), buf


def _radial_branch(rows, plan: ConvBlockPlan, w_last, w_hidden, keep: bool, table):
    """Hidden chain + last layer on ``rows`` (edges, or the knots) -> per-edge path weights (call on the radial stream)."""
    h, zs = ops._mlp_fwd_raw(rows, w_hidden, plan.mlp_alphas, plan.mlp_act, plan.mlp_cst, keep)
    w = torch.empty(h.shape[0], plan.last_spec.d_out, device=rows.device, dtype=torch.float32)
    with ops.timed_launch("radial_last_fwd", (h.shape[0], plan.last_spec.d_in, plan.last_spec.d_out)):
        ops._lin_fwd_raw(h, w_last, None, w, plan.last_spec, 1.0, False)
    if table is not None:      # w so far: the MLP on the knots; every edge interpolates between its three knots
        radial_table.guard(plan.guard_key if plan.guard_key is not None else w_last, w, blocks=table.blocks)
        w = radial_table.interp_fwd_raw(w, table)
    return h, zs, w


class ConvBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, node_attrs, edge_radial, sh, plan: ConvBlockPlan, topo: GraphTopo, groups, in_cf: bool, out_cf: bool,
                fork: bool, table, nxt, w_lin1, w_post, w_sc, w_last, *w_hidden):
        """``table`` = (centre knot int32 [E], offset [E], knot CSR pointers, knot CSR edge ids, knots) when the radial MLP
        is evaluated on a knot table (backend/radial_table.py): ``edge_radial`` is then the radial basis ON THE KNOTS
        ([knots + 1, n_basis]), the MLP runs on those rows and every edge interpolates its weights.
        ``nxt`` = (plan, last-layer weight, hidden weights) of the NEXT layer when it reads the same radial rows: its
        radial branch is issued here, on the radial stream, behind this layer's tensor product -- it then runs under this
        layer's trailing Linear and gate and the next layer's node-side launches instead of in front of its product."""
        L.require_cuda(x, edge_radial, sh)
        x, edge_radial, sh = L.f32c(x), L.f32c(edge_radial), L.f32c(sh)
        dev = x.device
        main = torch.cuda.current_stream(dev)
        fork = bool(fork) and not torch.cuda.is_current_stream_capturing()
        side = ops.side_stream(dev, 0) if fork else main      # radial MLP
        side2 = ops.side_stream(dev, 1) if fork else main     # self-connection
        keep = any(ctx.needs_input_grad)
        has_sc = plan.sc_spec is not None
        n = x.shape[0]

        # --- radial branch: hidden chain + last layer -> per-edge path weights [E, W]
        pref, plan.prefetched = plan.prefetched, None
        if pref is not None and pref[0][0] is edge_radial and pref[0][1] is table and pref[0][2:] == (keep, fork):
            h, zs, w = pref[1]                      # issued by the previous layer (look-ahead); `_wait(main, side)` below
            AHEAD_STATS[0] += 1
        else:
            _wait(side, main)
            with _on(side, main):
                h, zs, w = _radial_branch(edge_radial, plan, w_last, w_hidden, keep, table)
            _rec(edge_radial, side, main)
        # --- node side
        x_cf = x if in_cf else ops._relayout_raw(x, plan.in_blocks, True)
        a_rep = m = None
        x1 = torch.empty(n, plan.lin1_spec.d_out, device=dev, dtype=torch.float32)
        if has_sc and MERGE and not fork:      # one stream anyway: the self-connection and linear_1 in one launch
            node_attrs = L.f32c(node_attrs)
            a_rep = node_attrs.index_select(0, groups.reps)
            m = ops._kw_fwd_raw(a_rep, w_sc, plan.sc_spec, plan.sc_m_off, plan.sc_ld_m)
            conv = (torch.empty if plan.sc_spec.out_covered else torch.zeros)(n, plan.sc_spec.d_out, device=dev, dtype=torch.float32)
            ops._run_segments([ops._grp_segs("fwd", x_cf, m, conv, groups, plan.sc_spec, plan.sc_m_off),
                               ops._lin_fwd_segs(x_cf, w_lin1, x1, plan.lin1_spec, 1.0, False)])
            side2 = main                                 # (the self-connection is already on this stream)
        else:
            if has_sc:
                node_attrs = L.f32c(node_attrs)
                _wait(side2, main)
                with _on(side2, main):
                    a_rep = node_attrs.index_select(0, groups.reps)
                    m = ops._kw_fwd_raw(a_rep, w_sc, plan.sc_spec, plan.sc_m_off, plan.sc_ld_m)
                    conv = ops._grp_fwd_raw(x_cf, m, groups, plan.sc_spec, plan.sc_m_off)       # [N, conv_out] cf
                _rec(x_cf, side2, main)
                _rec(node_attrs, side2, main)
            ops._lin_fwd_raw(x_cf, w_lin1, None, x1, plan.lin1_spec, 1.0, False)
        _wait(main, side)
        _rec(w, main, side)
        mid = ops._tp_fwd_raw(x1, sh, w, topo, plan.tp_plan)
        if nxt is not None and fork and LOOK_AHEAD:
            plan_n, w_last_n, w_hidden_n = nxt
            side.wait_stream(main)                   # behind this layer's tensor product: both are HBM streams
            with _on(side, main):
                # (the rows and the table themselves are the key: held here, their identity cannot be reused by a later batch)
                plan_n.prefetched = ((edge_radial, table, keep, fork),
                                     _radial_branch(edge_radial, plan_n, w_last_n, w_hidden_n, keep, table))
        if has_sc:
            _wait(main, side2)
            _rec(conv, main, side2)
            ops._lin_fwd_raw(mid, w_post, None, conv, plan.post_spec, plan.scale, True)     # conv += scale * Linear(mid)
        else:
            conv = (torch.empty if plan.post_spec.out_covered else torch.zeros)(n, plan.post_spec.d_out, device=dev, dtype=torch.float32)
            ops._lin_fwd_raw(mid, w_post, None, conv, plan.post_spec, plan.scale, False)
        y = ops._gate_fwd_raw(conv, plan.gate_spec, out_cf)
        if keep:
            ctx.save_for_backward(x_cf, edge_radial, sh, h, w, x1, mid, conv, a_rep, m, w_lin1, w_post, w_sc, w_last,
                                  *w_hidden, *zs)
            ctx.cfg = (plan, topo, groups, bool(in_cf), bool(out_cf), fork, len(w_hidden), table)
            ctx.attrs_shape = tuple(node_attrs.shape) if has_sc else None
        return y

    @staticmethod
    def backward(ctx, gy):
        plan, topo, groups, in_cf, out_cf, fork, n_hidden, table = ctx.cfg
        saved = ctx.saved_tensors
        x_cf, edge_radial, sh, h, w, x1, mid, conv, a_rep, m, w_lin1, w_post, w_sc, w_last = saved[:14]
        w_hidden, zs = saved[14:14 + n_hidden], saved[14 + n_hidden:]
        need = ctx.needs_input_grad
        need_x, need_attrs, need_radial, need_sh = need[0], need[1], need[2], need[3]
        p0 = 12
        need_lin1, need_post, need_sc, need_last = need[p0], need[p0 + 1], need[p0 + 2], need[p0 + 3]
        need_hidden = need[p0 + 4:]
        if torch.is_grad_enabled() or need_sh:
            raise RuntimeError(
                "the fused convolution block serves first-order training only (no gradient w.r.t. the spherical harmonics, "
                "no create_graph=True): MessagePassing takes the composed path for those by itself; set E3K_CONV_BLOCK=0 "
                "if this was reached another way")
        has_sc = plan.sc_spec is not None
        dev = gy.device
        main = torch.cuda.current_stream(dev)
        fork = fork and not torch.cuda.is_current_stream_capturing()
        side = ops.side_stream(dev, 0) if fork else main
        side2 = ops.side_stream(dev, 1) if fork else main
        side3 = ops.side_stream(dev, 2) if (fork and ops.WGRAD_SIDE) else main
        gy = L.f32c(gy)

        return ConvBlockFn._backward_grouped(ctx, gy, main, side, side2, side3, fork)

    @staticmethod
    def _backward_grouped(ctx, gy, main, side, side2, side3, fork):
        """The backward, its GEMMs grouped by shared operand: four or five e3k_gemm_multi calls (round 2: nine calls,
        fifteen launches)."""
        plan, topo, groups, in_cf, out_cf, _, n_hidden, table = ctx.cfg
        saved = ctx.saved_tensors
        x_cf, edge_radial, sh, h, w, x1, mid, conv, a_rep, m, w_lin1, w_post, w_sc, w_last = saved[:14]
        w_hidden, zs = saved[14:14 + n_hidden], saved[14 + n_hidden:]
        need = ctx.needs_input_grad
        need_x, need_attrs, need_radial = need[0], need[1], need[2]
        p0 = 12
        need_lin1, need_post, need_sc, need_last = need[p0], need[p0 + 1], need[p0 + 2], need[p0 + 3]
        need_hidden = need[p0 + 4:]
        has_sc = plan.sc_spec is not None
        dev = gy.device
        n = x_cf.shape[0]
        need_radial_side = need_last or need_radial or any(need_hidden)
        need_x1 = need_x or need_lin1

        g_conv = ops._gate_bwd_raw(conv, gy, plan.gate_spec, out_cf)
        # (A) both readers of g_conv: the trailing Linear's input gradient and the self-connection's
        g_mid = (torch.empty if plan.post_spec.in_covered else torch.zeros)(n, plan.post_spec.d_in, device=dev, dtype=torch.float32)
        segs = [ops._lin_dgrad_segs(g_conv, w_post, g_mid, plan.post_spec, plan.scale, False)]
        g_xcf = None
        if need_x:
            if has_sc:
                g_xcf = (torch.empty if plan.sc_spec.in_covered else torch.zeros)(n, plan.sc_spec.d_in, device=dev, dtype=torch.float32)
                segs.append(ops._grp_segs("dgrad", g_conv, m, g_xcf, groups, plan.sc_spec, plan.sc_m_off))
            else:
                g_xcf = (torch.empty if plan.lin1_spec.in_covered else torch.zeros)(n, plan.lin1_spec.d_in, device=dev, dtype=torch.float32)
        ops._run_segments(segs)
        ret_post = ret_lin1 = ret_sc = g_attrs = None
        want_sc = has_sc and (need_sc or need_attrs)
        gm = None

        def weight_grads(with_g_conv: bool, with_lin1: bool):
            """(C) weight gradients off the critical path, one call: trailing Linear + self-connection (per key) need
            g_conv, linear_1 needs g_x1.  Forked, the first two start right behind the gate backward and linear_1's
            follows the tensor product; on one stream all three go out together."""
            nonlocal ret_post, ret_lin1, gm
            segs = []
            if with_g_conv and need_post:
                gb_post, ret_post = _grad_buffer(w_post, True)
                segs.append(ops._lin_wgrad_segs(mid, g_conv, gb_post, plan.post_spec, plan.scale))
            if with_g_conv and want_sc:
                gm = torch.zeros(tuple(m.shape), device=dev, dtype=torch.float32)
                segs.append(ops._grp_segs("wgrad", x_cf, gm, g_conv, groups, plan.sc_spec, plan.sc_m_off))
            if with_lin1 and need_lin1:
                gb_lin1, ret_lin1 = _grad_buffer(w_lin1, True)
                segs.append(ops._lin_wgrad_segs(x_cf, g_x1, gb_lin1, plan.lin1_spec, 1.0))
            if segs:
                ops._run_segments(segs, wgrad=True)

        def keyed_weight_grads():
            nonlocal ret_sc, g_attrs
            # per-key weight gradients -> the flat weight's and the attributes': on the self-connection stream, where
            # the consumer of g_attrs (the stream alias of node_attrs) lives
            _wait(side2, side3)
            with _on(side2, main):
                gb_sc, ret_sc = _grad_buffer(w_sc, need_sc)
                acc = 1 if (gb_sc is not None and ret_sc is None) else 0
                ga = ops._kw_bwd_raw(a_rep, w_sc, gm, plan.sc_spec, plan.sc_m_off, plan.sc_ld_m, bool(need_attrs), gb_sc, acc)
                if need_attrs:
                    g_attrs = torch.zeros(ctx.attrs_shape, device=dev, dtype=torch.float32)
                    g_attrs.index_add_(0, groups.reps, ga)
            _rec(gm, side2, side3)
            _rec(a_rep, side2, main)

        if fork and (need_post or want_sc):
            _wait(side3, main)
            with _on(side3, main):
                weight_grads(True, False)
            for t in (g_conv, mid, x_cf):
                _rec(t, side3, main)
            if want_sc:
                keyed_weight_grads()
        # tensor product: features on this stream, per-edge weights handed to the radial stream
        g_x1 = ops._tp_bwd_x_raw(sh, w, g_mid, topo, plan.tp_plan) if need_x1 else None
        g_radial = ret_last = None
        ret_hidden: List[Optional[torch.Tensor]] = [None] * n_hidden
        if need_radial_side:
            g_w, _ = ops._tp_bwd_w_raw(x1, sh, w, g_mid, topo, plan.tp_plan, False, True)
            _wait(side, main)
            _rec(g_w, side, main)
            with _on(side, main):
                if table is not None:      # transpose of the interpolation: the gradient of the MLP's output on the knots
                    g_w = radial_table.interp_bwd_raw(g_w, table)
                if need_last:
                    gb_last, ret_last = _grad_buffer(w_last, True)
                    ops._lin_wgrad_raw(h, g_w, gb_last, plan.last_spec, 1.0)
                if need_radial or any(need_hidden):
                    g_h = ops._lin_dgrad_raw(g_w, w_last, plan.last_spec, 1.0)
                    gws = []
                    for i, wh in enumerate(w_hidden):
                        buf, ret = _grad_buffer(wh, need_hidden[i])
                        gws.append(buf)
                        ret_hidden[i] = ret
                    g_radial = torch.empty_like(edge_radial) if need_radial else None
                    ops._mlp_bwd_raw(edge_radial, w_hidden, zs, plan.mlp_alphas, plan.mlp_act, plan.mlp_cst, g_h, gws, g_radial)
            for t in (g_mid, x1, sh, edge_radial):
                _rec(t, side, main)
        # (B) linear_1's input gradient on top of the self-connection's
        g_x = None
        if need_x:
            ops._run_segments([ops._lin_dgrad_segs(g_x1, w_lin1, g_xcf, plan.lin1_spec, 1.0, has_sc)])
            g_x = g_xcf if in_cf else ops._relayout_raw(g_xcf, plan.in_blocks, False)
        if fork:
            if need_lin1:
                _wait(side3, main)
                with _on(side3, main):
                    weight_grads(False, True)
                _rec(g_x1, side3, main)
                _rec(x_cf, side3, main)
        elif need_post or need_lin1 or want_sc:
            weight_grads(True, True)
            if want_sc:
                keyed_weight_grads()
        # parameter gradients handed back to autograd (no gradient sink) are consumed on THIS stream
        if ret_post is not None or ret_lin1 is not None:
            _wait(main, side3)
        if ret_sc is not None:
            _wait(main, side2)
        if ret_last is not None or any(r is not None for r in ret_hidden):
            _wait(main, side)      # (g_radial is consumed by the radial stream's alias of the edge embedding: no wait)
        if fork:      # buffers zero-filled on a side stream and handed to autograd are read (and later freed) on this one
            for r in (ret_post, ret_lin1, ret_sc, ret_last, *ret_hidden):
                if r is not None:
                    r.record_stream(main)
            if table is not None:
                for t in table:      # the knot bins were built on the main stream and read on the radial one
                    if isinstance(t, torch.Tensor):
                        _rec(t, side, main)
        if ops.GRAD_READY is not None:
            rets = (ret_lin1, ret_post, ret_sc if has_sc else None, ret_last, *ret_hidden)
            needs = (need_lin1, need_post, need_sc or not has_sc, need_last, *need_hidden)
            if all(needs) and all(r is None for r in rets):          # every weight gradient of the layer went to the sink
                ops.GRAD_READY([w_lin1, w_post, w_last, *w_hidden] + ([w_sc] if has_sc else []))
        return (g_x, g_attrs, g_radial, None, None, None, None, None, None, None, None, None,
                ret_lin1, ret_post, ret_sc, ret_last, *ret_hidden)


def conv_block(x, node_attrs, edge_radial, sh, plan: ConvBlockPlan, topo, groups, in_cf: bool, out_cf: bool, fork: bool,
               w_lin1, w_post, w_sc, w_last, w_hidden: Sequence[torch.Tensor], table=None, nxt=None, pre=None, m_pre=None):
    from . import conv_native

    if m_pre is not None:    # the per-key self-connection weights of this layer come from conv_native.KwStackFn
        node_attrs = w_sc = None
    if pre is not None:      # stack mode: the radial MLP's rows of this layer come from conv_native.RadialStackFn
        return conv_native.NativeConvBlockFn.apply(x, node_attrs, None, sh, plan, topo, groups, in_cf, out_cf, fork, table, nxt,
                                                   pre, m_pre, w_lin1, w_post, w_sc, None)
    if conv_native.ENABLED and conv_native.native_layer(plan) is not None:      # the same sequence issued by csrc/e3k_layer.hip
        out = conv_native.NativeConvBlockFn.apply(x, node_attrs, edge_radial, sh, plan, topo, groups, in_cf, out_cf, fork, table,
                                                  nxt, None, m_pre, w_lin1, w_post, w_sc, w_last, *w_hidden)
        return out[0] if isinstance(out, tuple) else out      # (addend form: the dirtied addend rides along as a second output)
    return ConvBlockFn.apply(x, node_attrs, edge_radial, sh, plan, topo, groups, in_cf, out_cf, fork, table, nxt,
                             w_lin1, w_post, w_sc, w_last, *w_hidden)
