"""One MessagePassing layer (FactorizedConvolution + Gate) for FORCE training: differentiable twice, on the knot table.

Reference: ``GradientOutput.forward`` (``e3_layers/nn/output.py:31-53``) takes ``-dE/dpos`` by autograd with
``create_graph=self.training`` and the loss on the forces is differentiated again (``configs/config_energy_force.py:18``);
per layer that is ``FactorizedConvolution.forward`` + ``Gate`` (``nn/message_passing.py:91-124, 249``) differentiated twice.

Rounds 1-3 served this with the composed path -- one ``autograd.Function`` per kernel, the radial MLP evaluated per edge
(the radii require grad, so the knot table was refused) together with its first and second derivatives, autograd adding
``[E, W]`` tensors: 668 launches and 11.9 ms for 64 molecules.  Here the layer is THREE autograd nodes whose
bodies issue the kernels back to back:

  ``ForceBlockFn.forward``    x -> y: keyed self-connection + linear_1 -> table-form tensor product -> trailing Linear -> gate
                              (the same sequence as ``conv_block``; outputs the layer's intermediates x1 / conv as well, so that
                              the second-order pass below can send cotangents to them)
  ``ForceBlockBwdFn.forward`` the FIRST backward (forces): g_y -> g_x, g_sh, g_r.  Inputs-only (``ops.inputs_only_backward``:
                              no parameter gradient is formed).  The per-edge weights and their slope come from the tables
                              T and D (``radial_table``): ``w[E, W]`` and ``dw/dr[E, W]`` never exist -- ``e3k_tp_bwd_e_table``
                              reduces d/dsh to nine and d/dr to one number per edge in registers.
  ``ForceBlockBwdFn.backward``  the adjoint of that pass ("u-sweep": runs layer 0 -> L-1 in the final backward): every
                              contraction is multilinear, so the adjoint of each kernel is one of its siblings with operands
                              exchanged -- the three tensor-product terms of the product rule in ONE walk
                              (``e3k_tp_fwd_jvp_table``), likewise ``e3k_tp_bwd_x_dual_table`` / ``e3k_tp_bwd_w_dual`` -- and
                              the gradients of T and D are transposed interpolations of ``[E, W]`` weight gradients.
  ``ForceBlockFn.backward``   the ordinary first-order backward ("v-sweep", L-1 -> 0) with the u-sweep's cotangents of x1 and
                              conv added in: parameter gradients straight into the gradient sink.

Launches per layer: forward 6, first backward 6, u-sweep 15, v-sweep 12 -- against ~130 composed.
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from .tuning import knob as _knob
from torch.autograd.function import once_differentiable

from . import lib as L
from . import ops, radial_table
from .conv_block import ConvBlockPlan, _grad_buffer

ENABLED = _knob("E3K_FORCE_BLOCK")
# 1: a layer interpolates its per-edge weights w [E, W] (forward) and their slope dw/dr [E, W] (first backward) ONCE and every
# tensor-product kernel of the three passes streams those rows; 0: every kernel gathers four rows per table and edge itself (the
# in-kernel form of the energy step).  Five or six kernels per layer read them: 64 molecules 7.03 -> see DESIGN.md ms per step.
MATERIALIZE = _knob("E3K_FORCE_MATERIALIZE")
EDGE_ATOMICS = _knob("E3K_FORCE_EDGE_ATOMICS")      # 1: rounds 4-5's atomics on g_sh / g_r (not reproducible run to run)
FUSE_XW = _knob("E3K_TP_BWD_FUSED")      # the input and the weight gradient of the tensor product in one walk (as conv_native)
STATS = [0, 0, 0]      # forwards, first backwards (create_graph), u-sweeps (tests)
_WARNED = [False]
_DECLINE = [0]


class declined:
    """``with conv_force.declined():`` -- layers built inside take the composed path.  ``GradientOutput`` uses it when the
    differentiation variable required grad BEFORE it was called (the caller wants d loss / d pos of a force loss: relaxation,
    adversarial or structure gradients -- third derivatives of the layer, which the block does not form; ADVICE r4: it used to
    hand back a partial gradient behind a one-time warning)."""

    def __enter__(self):
        _DECLINE[0] += 1

    def __exit__(self, *exc):
        _DECLINE[0] -= 1


def supported(plan: ConvBlockPlan, dev) -> bool:
    """Layers the force block serves: keyed self-connection inside the block, a tensor-product plan with the second-order kernels
    (channel-complete: the 64-channel models, l_max <= 3)."""
    if not ENABLED or _DECLINE[0] or plan is None or plan.addend or plan.sc_spec is None:
        return False
    hit = plan.__dict__.get("_force_ok")
    if hit is None:
        h = plan.tp_plan.handle(dev)
        # (with the weights materialised -- the default -- the second-order kernels also exist for the l_max 3 plans that are walked
        #  by two waves per group)
        hit = plan.__dict__["_force_ok"] = bool(L.load().e3k_tp_table2_supported(h)
                                                or (MATERIALIZE and L.load().e3k_tp_second_order_streamed_supported(h)))
    return hit


# ---- raw launches of the table-form tensor-product kernels ------------------------------------------------------------------
def _tp_fwd_table(x1, sh, T, bins, topo, tp):
    n, e = x1.shape[0], sh.shape[0]
    out = torch.empty(n, tp.d_mid, device=x1.device, dtype=torch.float32)
    L.check(L.load().e3k_tp_fwd_table(tp.handle(x1.device), L.ptr(x1), L.ptr(sh), L.ptr(T), L.ptr(bins.bin), L.ptr(bins.coef),
                                      L.ptr(topo.src), L.ptr(topo.dst_ptr), L.ptr(topo.dst_perm), n, e, L.ptr(out), L.stream_ptr()),
            "e3k_tp_fwd_table")
    return out


def _tp_bwd_x_table(sh, T, bins, g_mid, topo, tp):
    n, e = g_mid.shape[0], sh.shape[0]
    gx = (torch.empty if tp.bwd_x_overwrites(sh.device) else torch.zeros)(n, tp.d_in, device=sh.device, dtype=torch.float32)
    L.check(L.load().e3k_tp_bwd_x_table(tp.handle(sh.device), L.ptr(sh), L.ptr(T), L.ptr(bins.bin), L.ptr(bins.coef), L.ptr(g_mid),
                                        L.ptr(topo.dst), L.ptr(topo.src_ptr), L.ptr(topo.src_perm), n, e, L.ptr(gx), L.stream_ptr()),
            "e3k_tp_bwd_x_table")
    return gx


def _tp_fwd_ptable(x1, sh, packed, bins, topo, tp):
    """the forward from the PACKED table (``radial_table.pack_raw``)"""
    n, e = x1.shape[0], sh.shape[0]
    out = torch.empty(n, tp.d_mid, device=x1.device, dtype=torch.float32)
    L.check(L.load().e3k_tp_fwd_ptable(tp.handle(x1.device), L.ptr(x1), L.ptr(packed), L.ptr(bins.records(topo, sh, "dst")),
                                       L.ptr(topo.dst_ptr), n, e, L.ptr(out), L.stream_ptr()), "e3k_tp_fwd_ptable")
    return out


def _tp_bwd_x_ptable(sh, packed, bins, g_mid, topo, tp):
    n, e = g_mid.shape[0], sh.shape[0]
    gx = (torch.empty if tp.bwd_x_overwrites(sh.device) else torch.zeros)(n, tp.d_in, device=sh.device, dtype=torch.float32)
    L.check(L.load().e3k_tp_bwd_x_ptable(tp.handle(sh.device), L.ptr(packed), L.ptr(bins.records(topo, sh, "src")), L.ptr(g_mid),
                                         L.ptr(topo.src_ptr), n, e, L.ptr(gx), L.stream_ptr()), "e3k_tp_bwd_x_ptable")
    return gx


def _tp_bwd_xw_ptable(x1, sh, packed, bins, g_mid, topo, tp):
    """(g_x1 [N, d_in], g_w [E, W]): the input gradient and every edge's weight gradient in ONE walk of the source CSR"""
    n, e = g_mid.shape[0], sh.shape[0]
    gx = (torch.empty if tp.bwd_x_overwrites(sh.device) else torch.zeros)(n, tp.d_in, device=sh.device, dtype=torch.float32)
    gw = torch.empty(e, tp.w_numel, device=sh.device, dtype=torch.float32)
    L.check(L.load().e3k_tp_bwd_xw_ptable(tp.handle(sh.device), L.ptr(x1), L.ptr(packed), L.ptr(bins.records(topo, sh, "src")),
                                          L.ptr(g_mid), L.ptr(topo.src_ptr), n, e, L.ptr(gx), L.ptr(gw), L.stream_ptr()),
            "e3k_tp_bwd_xw_ptable")
    return gx, gw


def _edge_grad_buffers(tp, e: int, dev):
    """(g_sh [E, d_sh], g_r [E], per-item partials or None).  DETERMINISTIC (default): every (node, group) work item stores its
    share of an edge's sums into its own slice and the library adds the slices in item order -- forces are bit-reproducible run to
    run and no zero fill is needed; ``E3K_FORCE_EDGE_ATOMICS=1`` restores rounds 4-5's float atomics onto a zero-filled pair."""
    if EDGE_ATOMICS:
        buf = torch.zeros(e * (tp.d_sh + 1), device=dev, dtype=torch.float32)      # one fill for both (atomics accumulate)
        part = None
    else:
        buf = torch.empty(e * (tp.d_sh + 1), device=dev, dtype=torch.float32)
        part = torch.empty(max(int(L.load().e3k_tp_edge_partials_floats(tp.handle(dev), e)), 1), device=dev, dtype=torch.float32)
    return buf[:e * tp.d_sh].view(e, tp.d_sh), buf[e * tp.d_sh:], part


def _b(bins):
    """(knot, weights) pointers of the table form, or (None, None): T / D are then the materialised rows w / dw [E, W]"""
    return (None, None) if bins is None else (L.ptr(bins.bin), L.ptr(bins.coef))


def _tp_bwd_e_table(x1, sh, T, D, bins, g_mid, topo, tp):
    """(g_sh [E, d_sh], g_r [E]): the gradient w.r.t. the spherical harmonics and the radius"""
    n, e = x1.shape[0], sh.shape[0]
    g_sh, g_r, part = _edge_grad_buffers(tp, e, x1.device)
    L.check(L.load().e3k_tp_bwd_e_table(tp.handle(x1.device), L.ptr(x1), L.ptr(sh), L.ptr(T), L.ptr(D), *_b(bins),
                                        L.ptr(g_mid), L.ptr(topo.src), L.ptr(topo.dst_ptr), L.ptr(topo.dst_perm), n, e, L.ptr(g_sh),
                                        L.ptr(g_r), None, L.ptr(part), L.stream_ptr()), "e3k_tp_bwd_e_table")
    return g_sh, g_r


def _tp_fwd_jvp(x1, x2, sh, sh2, T, D, bins, s2, topo, tp):
    n, e = x1.shape[0], sh.shape[0]
    out = torch.empty(n, tp.d_mid, device=x1.device, dtype=torch.float32)
    L.check(L.load().e3k_tp_fwd_jvp_table(tp.handle(x1.device), L.ptr(x1), L.ptr(x2), L.ptr(sh), L.ptr(sh2), L.ptr(T), L.ptr(D),
                                          *_b(bins), L.ptr(s2), L.ptr(topo.src), L.ptr(topo.dst_ptr),
                                          L.ptr(topo.dst_perm), n, e, L.ptr(out), L.stream_ptr()), "e3k_tp_fwd_jvp_table")
    return out


def _tp_bwd_x_dual(sh, sh2, T, D, bins, s2, g_mid, topo, tp):
    n, e = g_mid.shape[0], sh.shape[0]
    gx = (torch.empty if tp.bwd_x_overwrites(sh.device) else torch.zeros)(n, tp.d_in, device=sh.device, dtype=torch.float32)
    L.check(L.load().e3k_tp_bwd_x_dual_table(tp.handle(sh.device), L.ptr(sh), L.ptr(sh2), L.ptr(T), L.ptr(D), *_b(bins), L.ptr(s2), L.ptr(g_mid), L.ptr(topo.dst), L.ptr(topo.src_ptr),
                                             L.ptr(topo.src_perm), n, e, L.ptr(gx), L.stream_ptr()), "e3k_tp_bwd_x_dual_table")
    return gx


def _tp_bwd_xe(x1, sh, w, dw, g_mid, topo, tp, want_w: bool):
    """(g_x1, g_sh, g_r, g_w or None): ``tp_bwd_x`` + ``_tp_bwd_e_table`` (+ ``tp_bwd_w``) in one walk (streamed rows w, dw [E, W])"""
    n, e = g_mid.shape[0], sh.shape[0]
    gx = (torch.empty if tp.bwd_x_overwrites(sh.device) else torch.zeros)(n, tp.d_in, device=sh.device, dtype=torch.float32)
    g_sh, g_r, part = _edge_grad_buffers(tp, e, sh.device)
    gw = torch.empty(e, tp.w_numel, device=sh.device, dtype=torch.float32) if want_w else None
    L.check(L.load().e3k_tp_bwd_xe(tp.handle(sh.device), L.ptr(x1), L.ptr(sh), L.ptr(w), L.ptr(dw), L.ptr(g_mid), L.ptr(topo.dst),
                                   L.ptr(topo.src_ptr), L.ptr(topo.src_perm), n, e, L.ptr(gx), L.ptr(g_sh), L.ptr(g_r), L.ptr(gw),
                                   L.ptr(part), L.stream_ptr()), "e3k_tp_bwd_xe")
    return gx, g_sh, g_r, gw


def _tp_bwd_xw_dual(x1, x2, sh, sh2, w, dw, s2, g_mid, topo, tp, want_plain: bool):
    """(cot_x1, g_w dual, g_w plain or None): ``_tp_bwd_x_dual`` + ``_tp_bwd_w_dual`` (+ ``tp_bwd_w``) in one walk (streamed rows)"""
    n, e = g_mid.shape[0], sh.shape[0]
    gx = (torch.empty if tp.bwd_x_overwrites(sh.device) else torch.zeros)(n, tp.d_in, device=sh.device, dtype=torch.float32)
    gw = torch.empty(e, tp.w_numel, device=sh.device, dtype=torch.float32)
    gwp = torch.empty(e, tp.w_numel, device=sh.device, dtype=torch.float32) if want_plain else None
    L.check(L.load().e3k_tp_bwd_xw_dual(tp.handle(sh.device), L.ptr(x1), L.ptr(x2), L.ptr(sh), L.ptr(sh2), L.ptr(w), L.ptr(dw), L.ptr(s2),
                                        L.ptr(g_mid), L.ptr(topo.dst), L.ptr(topo.src_ptr), L.ptr(topo.src_perm), n, e, L.ptr(gx), L.ptr(gw),
                                        L.ptr(gwp), L.stream_ptr()), "e3k_tp_bwd_xw_dual")
    return gx, gw, gwp


def _tp_bwd_w_dual(x1, x2, sh, sh2, g_mid, topo, tp):
    n, e = x1.shape[0], sh.shape[0]
    gw = torch.empty(e, tp.w_numel, device=x1.device, dtype=torch.float32)
    L.check(L.load().e3k_tp_bwd_w_dual(tp.handle(x1.device), L.ptr(x1), L.ptr(x2), L.ptr(sh), L.ptr(sh2), L.ptr(g_mid), L.ptr(topo.src),
                                       L.ptr(topo.dst_ptr), L.ptr(topo.dst_perm), n, e, L.ptr(gw), L.stream_ptr()), "e3k_tp_bwd_w_dual")
    return gw


def _gate_bwd2(conv, gy, h, spec, out_cf: bool, want_gy: bool = True, want_x: bool = True):
    """(cot g_y, cot conv) of g_conv = gate'(conv) g_y for the cotangent h of g_conv"""
    g_gy = torch.empty_like(gy) if want_gy else None
    g_x = torch.empty_like(conv) if want_x else None
    L.check(L.load().e3k_gate_bwd2(L.ptr(conv), L.ptr(gy), L.ptr(h), conv.shape[0], spec.in_dim, spec.out_dim, spec.c_array(),
                                   len(spec.segs), int(out_cf), L.ptr(g_gy), L.ptr(g_x), L.stream_ptr()), "e3k_gate_bwd2")
    return g_gy, g_x


class _Cfg:
    """The non-tensor arguments of a block (one object: a single ``None`` in the backward's return)."""

    __slots__ = ("plan", "topo", "groups", "bins", "in_cf", "out_cf", "sh_data", "w", "dw")

    def __init__(self, plan, topo, groups, bins, in_cf, out_cf, sh_data):
        self.plan, self.topo, self.groups, self.bins = plan, topo, groups, bins
        self.in_cf, self.out_cf, self.sh_data = bool(in_cf), bool(out_cf), bool(sh_data)
        self.w = self.dw = None      # MATERIALIZE: the layer's per-edge weights / their slope [E, W], interpolated once

    def weights(self, T, D=None):
        """``D``: the slope will be needed too (the geometry requires grad): both tables in one pass"""
        if self.w is None:
            if D is not None and self.dw is None:
                bins, e, width = self.bins, self.bins.bin.numel(), T.shape[1]
                self.w = torch.empty(e, width, device=T.device, dtype=torch.float32)
                self.dw = torch.empty(e, width, device=T.device, dtype=torch.float32)
                L.check(L.load().e3k_rtable_interp_fwd2(L.ptr(T), L.ptr(D), L.ptr(bins.perm), L.ptr(bins.bin), L.ptr(bins.coef), e, bins.knots,
                                                        width, L.ptr(self.w), L.ptr(self.dw), L.stream_ptr()), "e3k_rtable_interp_fwd2")
            else:
                self.w = radial_table.interp_fwd_raw(T, self.bins)
        return self.w

    def slopes(self, D):
        if self.dw is None:
            self.dw = radial_table.interp_fwd_raw(D, self.bins)
        return self.dw


def _to_cf(x, plan, in_cf):
    return x if in_cf else ops._relayout_raw(x, plan.in_blocks, True)


class ForceBlockFn(torch.autograd.Function):
    """y, x1, conv = layer(x, M, T, D, sh, r; linear_1, trailing Linear).  ``r`` (the edge lengths) enters through ``cfg.bins``
    (knot and interpolation weights per edge); it is an argument so that its gradient has somewhere to go."""

    @staticmethod
    def forward(ctx, x, m, T, D, sh, r, w_lin1, w_post, cfg: _Cfg):
        L.require_cuda(x, m, T, sh)
        x, m, T, D, sh = L.f32c(x), L.f32c(m), L.f32c(T), L.f32c(D), L.f32c(sh)
        plan, topo, groups, bins = cfg.plan, cfg.topo, cfg.groups, cfg.bins
        n, dev = x.shape[0], x.device
        x_cf = _to_cf(x, plan, cfg.in_cf)
        x1 = torch.empty(n, plan.lin1_spec.d_out, device=dev, dtype=torch.float32)
        conv = (torch.empty if plan.sc_spec.out_covered else torch.zeros)(n, plan.sc_spec.d_out, device=dev, dtype=torch.float32)
        # both readers of x_cf in one launch: the keyed self-connection and linear_1
        ops._run_segments([ops._grp_segs("fwd", x_cf, m, conv, groups, plan.sc_spec, plan.sc_m_off),
                           ops._lin_fwd_segs(x_cf, w_lin1, x1, plan.lin1_spec, 1.0, False)])
        if MATERIALIZE:
            mid = ops._tp_fwd_raw(x1, sh, cfg.weights(T, D if ctx.needs_input_grad[4] else None), topo, plan.tp_plan)
        else:
            mid = _tp_fwd_table(x1, sh, T, bins, topo, plan.tp_plan)
        ops._lin_fwd_raw(mid, w_post, None, conv, plan.post_spec, plan.scale, True)      # conv += scale * Linear(mid)
        y = ops._gate_fwd_raw(conv, plan.gate_spec, cfg.out_cf)
        STATS[0] += 1
        ctx.save_for_backward(x_cf, m, T, D, sh, w_lin1, w_post, x1, conv, mid)
        ctx.cfg = cfg
        ctx.set_materialize_grads(False)
        return y, x1, conv

    @staticmethod
    def backward(ctx, g_y, g_x1_in, g_conv_in):
        cfg: _Cfg = ctx.cfg
        x_cf, m, T, D, sh, w_lin1, w_post, x1, conv, mid = ctx.saved_tensors
        need = ctx.needs_input_grad
        plan = cfg.plan
        if torch.is_grad_enabled():
            # ---- the force evaluation: differentiable, inputs only
            if not ops.INPUTS_ONLY:
                raise NotImplementedError(
                    "the force block differentiates twice only under ops.inputs_only_backward() (GradientOutput asks for d y / d pos "
                    "alone): a create_graph=True backward that also wants parameter gradients takes the composed path -- set "
                    "E3K_FORCE_BLOCK=0")
            if g_x1_in is not None or g_conv_in is not None:
                raise NotImplementedError("third derivatives are not built")
            if g_y is None:
                return (None,) * 9
            g_x, g_sh, g_r = ForceBlockBwdFn.apply(g_y, x1, conv, m, T, D, sh, w_lin1, w_post, cfg)
            return (g_x if need[0] else None), None, None, None, (g_sh if need[4] else None), (g_r if need[5] else None), None, None, None
        # ---- first-order backward (the final pass of a training step: "v-sweep"; or forces without create_graph)
        topo, groups, bins, tp = cfg.topo, cfg.groups, cfg.bins, plan.tp_plan
        params = not ops.INPUTS_ONLY
        # (behind a u-sweep -- cotangents of x1 / conv arrive -- the geometry gradient would lack the third-derivative terms: none is
        #  handed back rather than a partial one; the u-sweep has said so)
        want_e = (need[4] or need[5]) and not (ops.PARAMS_ONLY and cfg.sh_data) and g_x1_in is None and g_conv_in is None
        dev, n = conv.device, conv.shape[0]
        if g_y is not None:
            g_conv = ops._gate_bwd_raw(conv, L.f32c(g_y), plan.gate_spec, cfg.out_cf)
            if g_conv_in is not None:
                g_conv.add_(g_conv_in)
        elif g_conv_in is not None:
            g_conv = L.f32c(g_conv_in)
        else:
            g_conv = None
        g_x = g_m = g_T = g_sh = g_r = ret_lin1 = ret_post = None
        g_x1 = None
        if g_conv is not None:
            g_mid = (torch.empty if plan.post_spec.in_covered else torch.zeros)(n, plan.post_spec.d_in, device=dev, dtype=torch.float32)
            g_xcf = (torch.empty if plan.sc_spec.in_covered else torch.zeros)(n, plan.sc_spec.d_in, device=dev, dtype=torch.float32)
            ops._run_segments([ops._lin_dgrad_segs(g_conv, w_post, g_mid, plan.post_spec, plan.scale, False),
                               ops._grp_segs("dgrad", g_conv, m, g_xcf, groups, plan.sc_spec, plan.sc_m_off)])
            g_w = None
            if MATERIALIZE:
                if want_e and FUSE_XW:                    # input, edge (and weight) gradients in one walk (csrc/e3k_tp.hip MODE 8)
                    g_x1, g_sh, g_r, g_w = _tp_bwd_xe(x1, sh, cfg.weights(T), cfg.slopes(D), g_mid, topo, tp, bool(params and need[2]))
                elif params and need[2] and FUSE_XW:      # both gradients of the product in one walk (MODE 6)
                    g_x1, g_w = ops._tp_bwd_xw_raw(x1, sh, cfg.weights(T), g_mid, topo, tp)
                else:
                    g_x1 = ops._tp_bwd_x_raw(sh, cfg.weights(T), g_mid, topo, tp)
                    if want_e:
                        g_sh, g_r = _tp_bwd_e_table(x1, sh, cfg.weights(T), cfg.slopes(D), None, g_mid, topo, tp)
            else:
                g_x1 = _tp_bwd_x_table(sh, T, bins, g_mid, topo, tp)
                if want_e:
                    g_sh, g_r = _tp_bwd_e_table(x1, sh, T, D, bins, g_mid, topo, tp)
            if params and need[2]:
                if g_w is None:
                    g_w, _ = ops._tp_bwd_w_raw(x1, sh, None, g_mid, topo, tp, False, True)
                g_T = radial_table.interp_bwd_raw(g_w, bins)
        if g_x1_in is not None:
            g_x1 = L.f32c(g_x1_in) if g_x1 is None else g_x1.add_(g_x1_in)
        if g_x1 is not None:
            if g_conv is None:
                g_xcf = torch.zeros(n, plan.lin1_spec.d_in, device=dev, dtype=torch.float32)
            ops._run_segments([ops._lin_dgrad_segs(g_x1, w_lin1, g_xcf, plan.lin1_spec, 1.0, True)])
        if need[0] and (g_conv is not None or g_x1 is not None):
            g_x = g_xcf if cfg.in_cf else ops._relayout_raw(g_xcf, plan.in_blocks, False)
        if params:
            segs = []
            if need[7] and g_conv is not None:
                gb_post, ret_post = _grad_buffer(w_post, True)
                segs.append(ops._lin_wgrad_segs(mid, g_conv, gb_post, plan.post_spec, plan.scale))
            if need[1] and g_conv is not None:
                g_m = torch.zeros(tuple(m.shape), device=dev, dtype=torch.float32)
                segs.append(ops._grp_segs("wgrad", x_cf, g_m, g_conv, groups, plan.sc_spec, plan.sc_m_off))
            if need[6] and g_x1 is not None:
                gb_lin1, ret_lin1 = _grad_buffer(w_lin1, True)
                segs.append(ops._lin_wgrad_segs(x_cf, g_x1, gb_lin1, plan.lin1_spec, 1.0))
            if segs:
                ops._run_segments(segs, wgrad=True)
        return g_x, g_m, g_T, None, g_sh, g_r, ret_lin1, ret_post, None


class ForceBlockBwdFn(torch.autograd.Function):
    """(g_x, g_sh, g_r) = d <g_y, layer(x, sh, r)> / d (x, sh, r): the first backward of a force evaluation as an op whose own
    backward is the hand-written adjoint (the u-sweep)."""

    @staticmethod
    def forward(ctx, g_y, x1, conv, m, T, D, sh, w_lin1, w_post, cfg: _Cfg):
        plan, topo, groups, bins, tp = cfg.plan, cfg.topo, cfg.groups, cfg.bins, cfg.plan.tp_plan
        g_y = L.f32c(g_y)
        dev, n = conv.device, conv.shape[0]
        g_conv = ops._gate_bwd_raw(conv, g_y, plan.gate_spec, cfg.out_cf)
        g_mid = (torch.empty if plan.post_spec.in_covered else torch.zeros)(n, plan.post_spec.d_in, device=dev, dtype=torch.float32)
        g_xcf = (torch.empty if plan.sc_spec.in_covered else torch.zeros)(n, plan.sc_spec.d_in, device=dev, dtype=torch.float32)
        ops._run_segments([ops._lin_dgrad_segs(g_conv, w_post, g_mid, plan.post_spec, plan.scale, False),
                           ops._grp_segs("dgrad", g_conv, m, g_xcf, groups, plan.sc_spec, plan.sc_m_off)])
        if MATERIALIZE and FUSE_XW:
            g_x1, g_sh, g_r, _ = _tp_bwd_xe(x1, sh, cfg.weights(T), cfg.slopes(D), g_mid, topo, tp, False)
        elif MATERIALIZE:
            g_x1 = ops._tp_bwd_x_raw(sh, cfg.weights(T), g_mid, topo, tp)
            g_sh, g_r = _tp_bwd_e_table(x1, sh, cfg.weights(T), cfg.slopes(D), None, g_mid, topo, tp)
        else:
            g_x1 = _tp_bwd_x_table(sh, T, bins, g_mid, topo, tp)
            g_sh, g_r = _tp_bwd_e_table(x1, sh, T, D, bins, g_mid, topo, tp)
        ops._run_segments([ops._lin_dgrad_segs(g_x1, w_lin1, g_xcf, plan.lin1_spec, 1.0, True)])
        g_x = g_xcf if cfg.in_cf else ops._relayout_raw(g_xcf, plan.in_blocks, False)
        STATS[1] += 1
        ctx.save_for_backward(g_y, x1, conv, m, T, D, sh, w_lin1, w_post, g_conv, g_mid, g_x1)
        ctx.cfg = cfg
        ctx.set_materialize_grads(False)
        return g_x, g_sh, g_r

    @staticmethod
    @once_differentiable
    def backward(ctx, u_x, u_sh, u_r):
        cfg: _Cfg = ctx.cfg
        g_y, x1, conv, m, T, D, sh, w_lin1, w_post, g_conv, g_mid, g_x1 = ctx.saved_tensors
        need = ctx.needs_input_grad
        if need[6] and not (ops.PARAMS_ONLY and cfg.sh_data) and not _WARNED[0]:
            # A plain ``loss.backward()`` (the reference's trainer, run/trainer.py:371) also walks towards ``pos``: GradientOutput has
            # set ``pos.requires_grad`` back by then, so AccumulateGrad drops what arrives there -- nothing reads d loss / d pos.
            # Forming it would take third derivatives of the layer, which this block does not have: it hands back no gradient for
            # the geometry inputs, says so once, and the parameter gradients are complete.
            import warnings

            _WARNED[0] = True
            warnings.warn("force block: d(loss)/d(pos) of a force loss (third derivatives) is not formed; parameter gradients are "
                          "complete.  Differentiate with run.parallel.backward_parameters() to skip that branch altogether, or set "
                          "E3K_FORCE_BLOCK=0 if the position gradient of a force loss is needed.")
        plan, topo, groups, bins, tp = cfg.plan, cfg.topo, cfg.groups, cfg.bins, cfg.plan.tp_plan
        dev, n, e = conv.device, conv.shape[0], sh.shape[0]
        STATS[2] += 1
        if u_x is None and u_sh is None and u_r is None:
            return (None,) * 10
        # the partners of (x1, sh, r) in the product rule; an absent cotangent is a zero operand
        u_sh = L.f32c(u_sh) if u_sh is not None else torch.zeros(e, tp.d_sh, device=dev, dtype=torch.float32)
        u_r = L.f32c(u_r).reshape(-1) if u_r is not None else torch.zeros(e, device=dev, dtype=torch.float32)
        ret_lin1 = ret_post = g_m = None
        cot_gconv = None
        wsegs = []
        if u_x is not None:
            a = _to_cf(L.f32c(u_x), plan, cfg.in_cf)                                   # cotangent of g_xcf
            u_gx1 = torch.empty(n, plan.lin1_spec.d_out, device=dev, dtype=torch.float32)
            cot_gconv = (torch.empty if plan.sc_spec.out_covered else torch.zeros)(n, plan.sc_spec.d_out, device=dev, dtype=torch.float32)
            # adjoints of the two input-gradient GEMMs w.r.t. their gradient operand = the forward GEMMs on `a`
            ops._run_segments([ops._grp_segs("fwd", a, m, cot_gconv, groups, plan.sc_spec, plan.sc_m_off),
                               ops._lin_fwd_segs(a, w_lin1, u_gx1, plan.lin1_spec, 1.0, False)])
            # ... and w.r.t. their weights = weight-gradient GEMMs pairing `a` with the gradients they multiplied (issued below,
            # together with the trailing Linear's)
            if need[3]:
                g_m = torch.zeros(tuple(m.shape), device=dev, dtype=torch.float32)
                wsegs.append(ops._grp_segs("wgrad", a, g_m, g_conv, groups, plan.sc_spec, plan.sc_m_off))
            if need[7]:
                gb_lin1, ret_lin1 = _grad_buffer(w_lin1, True)
                wsegs.append(ops._lin_wgrad_segs(a, g_x1, gb_lin1, plan.lin1_spec, 1.0))
        else:
            u_gx1 = torch.zeros(n, plan.lin1_spec.d_out, device=dev, dtype=torch.float32)
        # tensor-product family: Phi = <u_gx1, g_x1> + <u_sh, g_sh> + <u_r, g_r> is the derivative of <g_mid, TP(x1, sh, w(r))> along
        # (u_gx1, u_sh, u_r)
        if MATERIALIZE:
            wv, dwv, kb = cfg.weights(T), cfg.slopes(D), None
        else:
            wv, dwv, kb = T, D, bins
        cot_gmid = _tp_fwd_jvp(x1, u_gx1, sh, u_sh, wv, dwv, kb, u_r, topo, tp)
        g_T = g_D = None
        if MATERIALIZE and FUSE_XW and need[1] and need[4]:      # the three walks below as one (csrc/e3k_tp.hip MODE 7)
            cot_x1, gw_dual, gw_plain = _tp_bwd_xw_dual(x1, u_gx1, sh, u_sh, wv, dwv, u_r, g_mid, topo, tp, bool(need[5]))
            g_T = radial_table.interp_bwd_raw(gw_dual, bins)
            if need[5]:
                g_D = radial_table.interp_bwd_raw(gw_plain, bins, scale=u_r)
        else:
            cot_x1 = _tp_bwd_x_dual(sh, u_sh, wv, dwv, kb, u_r, g_mid, topo, tp) if need[1] else None
            if need[4]:
                g_T = radial_table.interp_bwd_raw(_tp_bwd_w_dual(x1, u_gx1, sh, u_sh, g_mid, topo, tp), bins)
            if need[5]:
                g_w, _ = ops._tp_bwd_w_raw(x1, sh, None, g_mid, topo, tp, False, True)
                g_D = radial_table.interp_bwd_raw(g_w, bins, scale=u_r)
        # trailing Linear: g_mid = scale * Linear^T(g_conv)
        if cot_gconv is None:
            cot_gconv = (torch.empty if plan.post_spec.out_covered else torch.zeros)(n, plan.post_spec.d_out, device=dev, dtype=torch.float32)
            ops._lin_fwd_raw(cot_gmid, w_post, None, cot_gconv, plan.post_spec, plan.scale, False)
        else:
            ops._lin_fwd_raw(cot_gmid, w_post, None, cot_gconv, plan.post_spec, plan.scale, True)
        if need[8]:
            gb_post, ret_post = _grad_buffer(w_post, True)
            wsegs.append(ops._lin_wgrad_segs(cot_gmid, g_conv, gb_post, plan.post_spec, plan.scale))
        if wsegs:
            ops._run_segments(wsegs, wgrad=True)
        cot_gy, cot_conv = _gate_bwd2(conv, g_y, cot_gconv, plan.gate_spec, cfg.out_cf, bool(need[0]), bool(need[2]))
        return cot_gy, cot_x1, cot_conv, g_m, g_T, g_D, None, ret_lin1, ret_post, None


def force_block(x, m, T, D, sh, r, plan: ConvBlockPlan, topo, groups, bins, in_cf: bool, out_cf: bool, w_lin1, w_post) -> torch.Tensor:
    cfg = _Cfg(plan, topo, groups, bins, in_cf, out_cf, ops.is_data_only(sh))
    y, _x1, _conv = ForceBlockFn.apply(x, m, T, D, sh, r, w_lin1, w_post, cfg)
    return y
