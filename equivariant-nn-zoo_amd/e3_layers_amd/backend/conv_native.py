"""The fused convolution block through the native layer executor (``csrc/e3k_layer.hip``).

Same arithmetic, same kernels and the same stream layout as ``backend/conv_block.py`` -- which stays as the readable
definition of the sequence and serves whatever this path declines -- but the launches of a layer's forward (and of its
backward) are issued by ONE C call: Python allocates the outputs (two or three buffers per pass instead of twenty
tensors), fills one argument struct and returns.  Reference: ``FactorizedConvolution.forward`` + ``Gate``
(``e3_layers/nn/message_passing.py:91-124, 249``).

Host time per layer: 0.22 -> 0.05 ms forward, 0.30 -> 0.07 ms backward (32 molecules, where the step is host-bound).
``E3K_LAYER_NATIVE=0`` keeps the Python sequence.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional

import torch

from .tuning import knob as _knob
from torch.autograd.function import once_differentiable

from . import lib as L
from . import ops, radial_table

ENABLED = _knob("E3K_LAYER_NATIVE")
AHEAD_STATS = [0]
PROF_KINDS = {"tp_fwd": 0, "tp_bwd_x": 1, "tp_bwd_w": 2, "rtable_fwd": 3, "rtable_bwd": 4, "radial_last_fwd": 5}
_LAYERS: List["NativeLayer"] = []      # every layer object created (bench.py arms / reads their per-kernel timers)


MAX_ROUNDS = 4


class NativeLayer:
    """The static description of one layer handed to ``e3k_layer_create`` (the ctypes arrays stay referenced here)."""

    def __init__(self, plan):
        self.plan = plan
        self.handles = {}          # device index -> e3k_layer*
        self.keep = []             # ctypes arrays referenced by the descriptor until create() has copied them
        self.ok = True
        try:
            self.desc = self._describe(plan)
        except NotImplementedError:
            self.ok = False
        _LAYERS.append(self)

    def _set(self, rounds) -> L.GemmSet:
        """``rounds``: [(ctypes problem array, n)] -- concatenated round-major (problems of one round write distinct
        blocks, a later round accumulates on top of the earlier ones: the executor issues them in order)."""
        if not 1 <= len(rounds) <= MAX_ROUNDS:
            raise NotImplementedError("template set with more rounds than the executor takes")
        total = sum(n for _, n in rounds)
        arr = (L.GemmProblem * total)()
        gs = L.GemmSet()
        pos = 0
        for r, (src, n) in enumerate(rounds):
            gs.round_start[r] = pos
            for i in range(n):
                C.memmove(C.byref(arr, (pos + i) * C.sizeof(L.GemmProblem)), C.byref(src, i * C.sizeof(L.GemmProblem)),
                          C.sizeof(L.GemmProblem))
            pos += n
        gs.round_start[len(rounds)] = pos
        self.keep.append(arr)
        gs.p, gs.n, gs.n_rounds = arr, total, len(rounds)
        return gs

    def _describe(self, plan) -> L.LayerDesc:
        d = L.LayerDesc()
        lin1, post, last, sc = plan.lin1_spec, plan.post_spec, plan.last_spec, plan.sc_spec
        has_sc = sc is not None
        if plan.addend and has_sc:
            raise NotImplementedError("addend with a keyed self-connection in the block")
        acc_post = has_sc or plan.addend      # the trailing Linear adds to what the self-connection left in the buffer
        T = ops._templates
        d.lin1_fwd = self._set(T(lin1, ("fwd", 1.0, False, 0, 1.0, False), lambda: ops._lin_fwd_templates(lin1, 1.0, False, 0, 1.0, False)).rounds)
        d.lin1_dgrad = self._set(T(lin1, ("dgrad", 1.0, False), lambda: ops._lin_dgrad_templates(lin1, 1.0, False)).rounds)
        d.lin1_dgrad_acc = self._set(T(lin1, ("dgrad", 1.0, True), lambda: ops._lin_dgrad_templates(lin1, 1.0, True)).rounds)
        d.lin1_wgrad = self._set(T(lin1, ("wgrad", 1.0), lambda: ops._lin_wgrad_templates(lin1, 1.0)).rounds)
        sc_ = plan.scale
        d.post_fwd = self._set(T(post, ("fwd", sc_, acc_post, 0, 1.0, False), lambda: ops._lin_fwd_templates(post, sc_, acc_post, 0, 1.0, False)).rounds)
        d.post_dgrad = self._set(T(post, ("dgrad", sc_, False), lambda: ops._lin_dgrad_templates(post, sc_, False)).rounds)
        d.post_wgrad = self._set(T(post, ("wgrad", sc_), lambda: ops._lin_wgrad_templates(post, sc_)).rounds)
        if has_sc:
            d.sc_fwd = self._set(ops._grouped_templates(sc, plan.sc_m_off, "fwd"))
            d.sc_dgrad = self._set(ops._grouped_templates(sc, plan.sc_m_off, "dgrad"))
            d.sc_wgrad = self._set(ops._grouped_templates(sc, plan.sc_m_off, "wgrad"))
            kw = ops._kw_array(sc, plan.sc_m_off)
            self.keep.append(kw)
            d.kw, d.n_kw, d.V, d.ld_m = kw, len(sc.instr), sc.v, plan.sc_ld_m
        d.last_fwd = self._set(T(last, ("fwd", 1.0, False, 0, 1.0, False), lambda: ops._lin_fwd_templates(last, 1.0, False, 0, 1.0, False)).rounds)
        d.last_dgrad = self._set(T(last, ("dgrad", 1.0, False), lambda: ops._lin_dgrad_templates(last, 1.0, False)).rounds)
        d.last_wgrad = self._set(T(last, ("wgrad", 1.0), lambda: ops._lin_wgrad_templates(last, 1.0)).rounds)
        gate = plan.gate_spec.c_array()
        self.keep.append(gate)
        d.gate, d.n_gate = gate, len(plan.gate_spec.segs)
        blocks = tuple(b for b in plan.in_blocks if b[1] > 1 and b[2] > 1)
        if blocks:
            arr = ops._blocks(blocks)
            self.keep.append(arr)
            d.in_blocks, d.n_in_blocks = arr, len(blocks)
        n_hidden = len(plan.mlp_alphas)
        if not (1 <= n_hidden <= 4) or plan.mlp_act not in ops.ACT_IDS:
            raise NotImplementedError("radial MLP shape")
        d.k0, d.h, d.n_hidden, d.act, d.cst = plan.mlp_k0, last.d_in, n_hidden, ops.ACT_IDS[plan.mlp_act], plan.mlp_cst
        for i, al in enumerate(plan.mlp_alphas):
            d.alphas[i] = al
        d.d_in, d.d_x1, d.d_mid, d.d_conv, d.d_out, d.W = lin1.d_in, lin1.d_out, post.d_in, post.d_out, plan.gate_spec.out_dim, last.d_out
        d.post_in_covered, d.lin1_in_covered, d.post_out_covered = int(post.in_covered), int(lin1.in_covered), int(post.out_covered)
        if plan.addend:
            d.post_out_covered = 1      # (no zero fill: the buffer holds the addend)
        d.sc_in_covered = int(sc.in_covered) if has_sc else 1
        d.sc_out_covered = int(sc.out_covered) if has_sc else 1
        if has_sc and not sc.out_covered and not plan.addend:
            # The self-connection does not reach every column of the pre-gate buffer (layer 0: scalars in, so only the 0e blocks; the
            # trailing Linear fills 1o and 2e) and the executor zero-filled the buffer for the Linear to accumulate on: a fill launch
            # per layer in front of the first GEMM.  Instead the FIRST writer of a block nobody wrote yet overwrites -- when the two
            # together reach every column (checked here: the widths of the distinct blocks add up to the row).
            widths = {ins.out_off: ins.mul_out * ins.dim for ins in list(sc.instr) + list(post.instr)}
            if sum(widths.values()) == post.d_out:
                written = {ins.out_off for ins in sc.instr}
                for i in range(d.post_fwd.n):
                    off = (d.post_fwd.p[i].C or 0) // 4
                    if off not in written:
                        d.post_fwd.p[i].accumulate = 0
                        written.add(off)
                d.sc_out_covered = 1
        if plan.gate_spec.in_dim != post.d_out or (has_sc and (sc.d_out != post.d_out or sc.d_in != lin1.d_in)):
            raise NotImplementedError("layer dims")
        return d

    def handle(self, device) -> Optional[int]:
        if not self.ok:
            return None
        idx = device.index if device.index is not None else torch.cuda.current_device()
        h = self.handles.get(idx)
        if h is None:
            self.desc.tp = self.plan.tp_plan.handle(device)
            self.desc.tp_bwd_x_overwrites = int(self.plan.tp_plan.bwd_x_overwrites(device))
            out = C.c_void_p()
            with torch.cuda.device(idx):
                L.check(L.load().e3k_layer_create(C.byref(self.desc), C.byref(out)), "e3k_layer_create")
            h = self.handles[idx] = out.value
        return h

    # ---- per-kernel timers (bench.py) ----
    def profile(self, capacity: int, kinds=None) -> None:
        """Arms ``capacity`` event pairs per kernel kind (0 disarms); ``kinds``: only these (names of ``PROF_KINDS``)."""
        mask = 0xFFFFFFFF if kinds is None else sum(1 << PROF_KINDS[k] for k in kinds)
        for h in self.handles.values():
            L.check(L.load().e3k_layer_profile_mask(h, capacity, mask), "e3k_layer_profile_mask")

    def profile_read(self, kind: str):
        out = []
        cap = 4096
        ms, n, e = (C.c_float * cap)(), (C.c_int64 * cap)(), (C.c_int64 * cap)()
        for h in self.handles.values():
            cnt = L.load().e3k_layer_profile_read(h, PROF_KINDS[kind], ms, n, e, cap)
            if cnt < 0:
                L.check(cnt, "e3k_layer_profile_read")
            out += [(float(ms[i]), int(n[i]), int(e[i])) for i in range(cnt)]
        return out

    def __del__(self):
        try:
            lib = L.load()
            for h in self.handles.values():
                lib.e3k_layer_destroy(h)
        except Exception:
            pass


def native_layer(plan) -> Optional[NativeLayer]:
    nl = plan.__dict__.get("_native")
    if nl is None:
        nl = plan.__dict__["_native"] = NativeLayer(plan)
    return nl if nl.ok else None


def _record_once(t, stream) -> None:
    """``t.record_stream(stream)`` the first time this tensor meets this stream (tensors shared by every layer of a pass)."""
    seen = getattr(t, "_e3k_rec", None)
    key = stream.cuda_stream
    if seen is None:
        t._e3k_rec = {key}
    elif key in seen:
        return
    else:
        seen.add(key)
    t.record_stream(stream)


def _ptr(t, off: int = 0):
    return None if t is None else t.data_ptr() + 4 * off


class _Carve:
    """Offsets (in floats, 64-float aligned) of the pieces of one buffer."""

    __slots__ = ("total", "off")

    def __init__(self):
        self.total, self.off = 0, {}

    def add(self, name: str, numel: int) -> None:
        self.off[name] = self.total
        self.total += -(-int(numel) // 64) * 64

    def alloc(self, dev) -> torch.Tensor:
        return torch.empty(max(self.total, 64), device=dev, dtype=torch.float32)


# 1: layers on the knot table whose tensor-product plan has the in-kernel form (e3k_tp_table_supported: the l_max 2 models)
# interpolate their path weights inside tp_fwd / tp_bwd_x: no interpolation pass, no w[E, W] (0.2-0.3 GB a layer at 256
# molecules) written, read twice and kept for the backward
TP_TABLE = _knob("E3K_TP_TABLE")
# 1: the addend of an addend-form layer (ConvBlockPlan.addend) is accumulated on in place instead of being copied into the block's buffer
ADDEND_INPLACE = _knob("E3K_ADDEND_INPLACE")


# 1: ... from the table PACKED into one 12-byte record per (knot, weight) (e3k_rtable_pack: the cubic's Taylor coefficients about the
# middle of the knot interval, the two small ones in fp16): one dwordx3 load per path slot instead of four dword loads out of four
# rows; 0: the four-row form of round 4
TP_TABLE_PACKED = _knob("E3K_TP_TABLE_PACKED")
TP_BWD_FUSED = _knob("E3K_TP_BWD_FUSED")


def _packed_buffer(rows: int, width: int, dev) -> torch.Tensor:
    return torch.empty(rows, width, 3, device=dev, dtype=torch.int32)


def in_kernel_table(plan, table, dev) -> bool:
    if not TP_TABLE or table is None:
        return False
    hit = plan.__dict__.get("_tp_table_ok")
    if hit is None:
        hit = plan.__dict__["_tp_table_ok"] = bool(L.load().e3k_tp_table_supported(plan.tp_plan.handle(dev)))
        plan.tp_plan._e3k_table_form = hit      # (bench.py: which byte model describes this plan's launches)
    return hit


def _table_fields(rad: L.LayerRadial, table) -> None:
    """``table``: radial_table.KnotBins"""
    rad.knots = table.knots
    rad.bin, rad.bin_ptr, rad.bin_perm = table.bin.data_ptr(), table.ptr.data_ptr(), table.perm.data_ptr()
    rad.bin_coef, rad.bin_seg = table.coef.data_ptr(), table.seg.data_ptr()


def _radial_struct(rad: L.LayerRadial, plan, edge_radial, table, n_edges: int, keep: bool, w_last, w_hidden, buf, carve, w, t_tab, p_tab=None):
    r = edge_radial.shape[0]
    rad.R, rad.E, rad.keep = r, n_edges, int(keep)
    rad.use_table = int(table is not None)
    rad.radial = edge_radial.data_ptr()
    if table is not None:
        _table_fields(rad, table)
        rad.T = t_tab.data_ptr()
    rad.w_last = w_last.data_ptr()
    for i, wh in enumerate(w_hidden):
        rad.w_hidden[i] = wh.data_ptr()
    rad.h = _ptr(buf, carve.off["h"])
    if keep:
        for i in range(len(w_hidden)):
            rad.z[i] = _ptr(buf, carve.off[f"z{i}"])
    rad.w = _ptr(w)
    rad.in_kernel = 1 if w is None else 0
    rad.P = p_tab.data_ptr() if (p_tab is not None and w is None) else None


def _stack_radial_struct(rad: L.LayerRadial, plan, pre, table, n_edges: int, w, p_tab=None, packed: bool = False):
    """Radial argument block of a layer whose MLP rows ``pre`` came from the stack: with the table the layer interpolates
    ``pre`` (= T) into ``w``; without, ``pre`` is ``w``.  ``packed``: ``p_tab`` was filled by the stack (one launch for all layers)."""
    rad.R, rad.E, rad.keep, rad.have_rows = pre.shape[0], n_edges, 0, 1
    rad.packed = int(bool(packed and p_tab is not None))
    rad.use_table = int(table is not None)
    if table is not None:
        _table_fields(rad, table)
        rad.T = pre.data_ptr()
        rad.w = _ptr(w)
        rad.in_kernel = 1 if w is None else 0
        rad.P = p_tab.data_ptr() if (p_tab is not None and w is None) else None
    else:
        rad.w = pre.data_ptr()


def _consumed_on(stream, *tensors):
    """Gradients allocated on the side stream a stack's backward runs on are read on ``stream`` (AccumulateGrad, the next
    node): the caching allocator must not hand their blocks back to the side stream before that reader is done."""
    if stream is None:
        return
    for t in tensors:
        if t is not None:
            t.record_stream(stream)


def _slope_ctx(slope, knots_r, bessel_w) -> L.SlopeCtx:
    _, r_max, r_min, p, one_over_r, kind = slope
    sl = L.SlopeCtx()
    sl.knots, sl.bessel_w = knots_r.data_ptr(), bessel_w.data_ptr()
    sl.r_max, sl.r_min, sl.p, sl.one_over_r, sl.cutoff_kind = float(r_max), float(r_min), float(p), int(one_over_r), int(kind)
    return sl


STACK_STATS = [0, 0]      # stack evaluations so far, layers in the most recent one (tests)
# E3K_HOST_TIMING=1: host seconds inside the layer functions, split into the C call and the Python around it
# (tools/host_split.py --layer-timing prints them): [fwd total, fwd C call, bwd total, bwd C call, calls]
HOST_TIMING = [0.0, 0.0, 0.0, 0.0, 0] if _knob("E3K_HOST_TIMING") == 1 else None


class RadialStackFn(torch.autograd.Function):
    """The radial MLPs of ALL the layers that read one edge embedding, in one op: ``rows`` [R, k0] (the radial basis on
    the knots, or per edge) -> per layer its output rows [R, W_l] (``fc(edge_radial)`` of ``nn/message_passing.py:74-79,93``,
    evaluated for every layer at once).  Forward: the hidden chains in one launch, the last layers in one call; backward
    (runs once the gradients of all layers' rows have arrived, i.e. behind the first layer's backward): last-layer weight
    gradients in one call, their input gradients in one, the hidden chains in one.  Per layer that was 2 launches forward
    and 4-5 backward over the same 4 097 knot rows: latency, not work."""

    @staticmethod
    def forward(ctx, rows, plans, use_table: bool, main, slope, bessel_w, *weights):
        """``main``: the stream the rest of the network runs on when this op was put on the radial stream (else None):
        gradients handed back to autograd are consumed there.
        ``slope`` (force training, table only): (knot radii [R], r_max, r_min, p, one_over_r, cutoff kind) and ``bessel_w`` (the
        basis' frequencies: a differentiable input then, else None) -- the op returns the layers' SLOPE tables D_l = d T_l / d r
        behind their tables T_l (``e3k_radial_slope_fwd``: forward-mode derivative of the hidden chains per knot in float64, last
        layers in fp32)."""
        L.require_cuda(rows)
        blocks = int(getattr(rows, "_e3k_blocks", 1))      # (a keyed source's stacked knot basis: its tables are guarded block by block)
        rows = L.f32c(rows)
        dev = rows.device
        n = len(plans)
        n_hidden = len(plans[0].mlp_alphas)
        per = 1 + n_hidden
        assert len(weights) == n * per
        keep = any(ctx.needs_input_grad)
        ctx.n_slope = 0
        r, hdim = rows.shape[0], plans[0].last_spec.d_in
        handles = (C.c_void_p * n)(*[native_layer(p).handle(dev) for p in plans])
        rads = (L.LayerRadial * n)()
        outs, bufs, carves = [], [], []
        for i, plan in enumerate(plans):
            w_last, w_hidden = weights[i * per], weights[i * per + 1:(i + 1) * per]
            carve = _Carve()
            carve.add("h", r * hdim)
            if keep:
                for l in range(n_hidden):
                    carve.add(f"z{l}", r * hdim)
            buf = carve.alloc(dev)
            out = torch.empty(r, plan.last_spec.d_out, device=dev, dtype=torch.float32)
            rad = rads[i]
            rad.R, rad.E, rad.keep, rad.use_table = r, r, int(keep), int(use_table)
            rad.radial, rad.w_last = rows.data_ptr(), w_last.data_ptr()
            for l, wh in enumerate(w_hidden):
                rad.w_hidden[l] = wh.data_ptr()
                if keep:
                    rad.z[l] = _ptr(buf, carve.off[f"z{l}"])
            rad.h = _ptr(buf, carve.off["h"])
            if use_table:
                rad.T = out.data_ptr()
                if TP_TABLE_PACKED and slope is None and blocks == 1 and in_kernel_table(plan, True, dev):
                    # the in-kernel form's packed table: written by the stack's own launch for all layers (e3k_rtable_pack_multi)
                    out._e3k_packed = _packed_buffer(r, plan.last_spec.d_out, dev)
                    rad.in_kernel, rad.P = 1, out._e3k_packed.data_ptr()
            else:
                rad.w = out.data_ptr()
            outs.append(out)
            bufs.append(buf)
            carves.append(carve)
        L.check(L.load().e3k_radial_stack_fwd(handles, rads, n, L.stream_ptr()), "e3k_radial_stack_fwd")
        STACK_STATS[0] += 1
        STACK_STATS[1] = n
        if use_table:      # one launch for the tables of all the layers (recorded with every build while a graph is captured)
            radial_table.guard_many([(plan.guard_key if plan.guard_key is not None else weights[i * per], out, False, blocks)
                                     for i, (plan, out) in enumerate(zip(plans, outs))])
        hps, slopes = [], []
        if slope is not None:
            if not use_table:
                raise ValueError("slope tables exist on the knot table only")
            knots_r = L.f32c(slope[0])
            bw = L.f32c(bessel_w.detach())
            hps = [torch.empty(r, hdim, device=dev, dtype=torch.float32) for _ in plans]
            slopes = [torch.empty(r, plan.last_spec.d_out, device=dev, dtype=torch.float32) for plan in plans]
            sl = _slope_ctx(slope, knots_r, bw)
            L.check(L.load().e3k_radial_slope_fwd(handles, rads, n, C.byref(sl), (C.c_void_p * n)(*[t.data_ptr() for t in hps]),
                                                  (C.c_void_p * n)(*[t.data_ptr() for t in slopes]), L.stream_ptr()),
                    "e3k_radial_slope_fwd")
            ctx.n_slope = n
            radial_table.guard_many([(plan.guard_key if plan.guard_key is not None else weights[i * per], d_tab, True)
                                     for i, (plan, d_tab) in enumerate(zip(plans, slopes))])
        if keep:
            extra = (knots_r, bw) if slope is not None else ()
            ctx.save_for_backward(rows, *bufs, *weights, *hps, *extra)
            ctx.cfg = (plans, use_table, carves, n_hidden)
            ctx.main = main
            ctx.slope = slope
        return tuple(outs) + tuple(slopes)

    @staticmethod
    @once_differentiable
    def backward(ctx, *g_all):
        plans, use_table, carves, n_hidden = ctx.cfg
        n, per = len(plans), 1 + n_hidden
        saved = ctx.saved_tensors
        rows, bufs, weights = saved[0], saved[1:1 + n], saved[1 + n:1 + n + n * per]
        tail = saved[1 + n + n * per:]                      # (slope mode: the layers' H' rows, the knot radii, the frequencies)
        hps = tail[:n] if ctx.n_slope else ()
        g_rows, g_slopes = g_all[:n], (g_all[n:] if hps else ())
        dev = rows.device
        r, hdim, k0 = rows.shape[0], plans[0].last_spec.d_in, rows.shape[1]
        need = ctx.needs_input_grad
        need_rows = need[0]
        if hps:      # a layer whose slope table received a gradient takes part even when its table did not
            g_rows = tuple(g if (g is not None or g_slopes[i] is None) else torch.zeros_like(g_slopes[i]) for i, g in enumerate(g_rows))
        live = [i for i in range(n) if g_rows[i] is not None]
        rets = [None] * (n * per)
        g_in = g_bessel = None
        if live:
            handles = (C.c_void_p * len(live))(*[native_layer(plans[i]).handle(dev) for i in live])
            items = (L.RadialStackItem * len(live))()
            g_h = torch.empty(len(live), r, hdim, device=dev, dtype=torch.float32)
            g_rad = torch.empty(len(live), r, k0, device=dev, dtype=torch.float32) if need_rows else None
            keepalive = []
            for j, i in enumerate(live):
                it = items[j]
                g = L.f32c(g_rows[i])
                keepalive.append(g)
                rad = it.rad
                rad.R, rad.E, rad.keep, rad.use_table = r, r, 1, int(use_table)
                rad.radial, rad.w_last = rows.data_ptr(), weights[i * per].data_ptr()
                rad.h = _ptr(bufs[i], carves[i].off["h"])
                for l in range(n_hidden):
                    rad.w_hidden[l] = weights[i * per + 1 + l].data_ptr()
                    rad.z[l] = _ptr(bufs[i], carves[i].off[f"z{l}"])
                it.g_rows = g.data_ptr()
                for l in range(per):
                    if not need[6 + i * per + l]:
                        continue
                    w = weights[i * per + l]
                    sink = ops._sink_for(w)
                    if sink is None:
                        rets[i * per + l] = torch.zeros_like(w)
                        sink = rets[i * per + l].view(-1)
                    if l == 0:
                        it.gb_last = sink.data_ptr()
                    else:
                        it.gb_hidden[l - 1] = sink.data_ptr()
                it.g_h = g_h[j].data_ptr()
                if g_rad is not None:
                    it.g_radial = g_rad[j].data_ptr()
                if hps and g_slopes[i] is not None:
                    gs = L.f32c(g_slopes[i])
                    g_hp = torch.empty(r, hdim, device=dev, dtype=torch.float32)
                    keepalive += [gs, g_hp]
                    it.g_slope, it.hp, it.g_hp = gs.data_ptr(), hps[i].data_ptr(), g_hp.data_ptr()
            sl_ref = None
            if hps and any(g_slopes[i] is not None for i in live):
                knots_r, bw = tail[n], tail[n + 1]
                sl = _slope_ctx(ctx.slope, knots_r, bw)
                acc = torch.empty(int(L.load().e3k_slope_tangent_bwd_scratch(len(live), n_hidden, k0, hdim, r)), device=dev, dtype=torch.float64)
                sl.acc = acc.data_ptr()
                if need[5]:
                    g_bessel = torch.zeros_like(bw)
                    sl.g_bessel = g_bessel.data_ptr()
                keepalive.append(acc)
                sl_ref = C.byref(sl)
            L.check(L.load().e3k_radial_stack_bwd(handles, items, len(live), sl_ref, L.stream_ptr()), "e3k_radial_stack_bwd")
            if g_rad is not None:
                g_in = g_rad[0] if len(live) == 1 else g_rad.sum(0)
            del keepalive
        _consumed_on(ctx.main, g_in, g_bessel, *rets)
        return (g_in, None, None, None, None, g_bessel, *rets)


KW_STACK_STATS = [0, 0]      # stack evaluations so far, layers in the most recent one (tests)


class KwStackFn(torch.autograd.Function):
    """The per-key contracted self-connection weights M_l[t] = sum_v a_t[v] W_l[:, v, :] of ALL the layers that read one
    ``node_attrs`` tensor (``self.sc(x, node_attrs)`` of ``nn/message_passing.py:81-87,100``, keyed form), in one op:
    forward = one gather of the keys' representative attribute rows + ONE launch for all layers; backward (behind the first
    layer's backward, with the gM every layer handed back) = ONE launch for the weight gradients, one + a reduction for the
    attribute gradient summed over the layers, one scatter to the representatives.  Per layer that was 2 launches forward
    and 6 backward, plus an autograd add of the attribute gradients."""

    @staticmethod
    def forward(ctx, node_attrs, groups, plans, main, *w_sc):
        L.require_cuda(node_attrs)
        node_attrs = L.f32c(node_attrs)
        dev = node_attrs.device
        n, n_keys, v = len(plans), groups.n_keys, plans[0].sc_spec.v
        a_rep = torch.empty(n_keys, v, device=dev, dtype=torch.float32)
        ms = [torch.empty(n_keys, p.sc_ld_m, device=dev, dtype=torch.float32) for p in plans]
        if any(ctx.needs_input_grad):
            # the layers' keyed weight-gradient GEMMs ADD into gM_l: zero-filled here for all layers by ONE launch (each layer's
            # backward takes its slice exactly once -- a second backward through the same graph allocates and fills its own)
            sizes = [n_keys * p.sc_ld_m for p in plans]
            gm_all = torch.zeros(sum(sizes), device=dev, dtype=torch.float32)
            for m, g in zip(ms, torch.split(gm_all, sizes)):
                m._e3k_gm = [g.view(n_keys, -1)]
        handles = (C.c_void_p * n)(*[native_layer(p).handle(dev) for p in plans])
        items = (L.KwStackItem * n)()
        for it, w, m in zip(items, w_sc, ms):
            it.w_sc, it.m = w.data_ptr(), m.data_ptr()
        L.check(L.load().e3k_kw_stack_fwd(handles, items, n, node_attrs.data_ptr(), groups.reps.data_ptr(), n_keys, a_rep.data_ptr(),
                                          L.stream_ptr()), "e3k_kw_stack_fwd")
        KW_STACK_STATS[0] += 1
        KW_STACK_STATS[1] = n
        if any(ctx.needs_input_grad):
            ctx.save_for_backward(a_rep, *w_sc)
            ctx.cfg = (plans, groups, tuple(node_attrs.shape))
            ctx.main = main
        return tuple(ms)

    @staticmethod
    @once_differentiable
    def backward(ctx, *g_ms):
        plans, groups, attrs_shape = ctx.cfg
        saved = ctx.saved_tensors
        a_rep, w_sc = saved[0], saved[1:]
        dev = a_rep.device
        need = ctx.needs_input_grad
        live = [i for i in range(len(plans)) if g_ms[i] is not None]
        rets = [None] * len(plans)
        g_attrs = None
        if live:
            n, n_keys, v = len(live), groups.n_keys, plans[0].sc_spec.v
            handles = (C.c_void_p * n)(*[native_layer(plans[i]).handle(dev) for i in live])
            items = (L.KwStackItem * n)()
            keep = []
            for it, i in zip(items, live):
                g = L.f32c(g_ms[i])
                keep.append(g)
                it.w_sc, it.m = w_sc[i].data_ptr(), g.data_ptr()
                if need[4 + i]:
                    sink = ops._sink_for(w_sc[i])
                    if sink is not None:
                        it.gb_sc, it.acc_sc = sink.data_ptr(), 1
                    else:
                        rets[i] = ops._kw_weight_buffer(w_sc[i], plans[i].sc_spec)
                        it.gb_sc, it.acc_sc = rets[i].data_ptr(), 0
            ga = ws = None
            if need[0]:
                both = torch.empty(n_keys * v + attrs_shape[0] * attrs_shape[1], device=dev, dtype=torch.float32)      # adjacent: one zero fill
                ga, g_attrs = both[:n_keys * v].view(n_keys, v), both[n_keys * v:].view(attrs_shape)
                ws = torch.empty(max(int(L.load().e3k_kw_stack_bwd_workspace(handles, n, n_keys)), 1), device=dev, dtype=torch.float32)
            L.check(L.load().e3k_kw_stack_bwd(handles, items, n, a_rep.data_ptr(), groups.reps.data_ptr(), groups.bounds.data_ptr(),
                                              attrs_shape[0], n_keys, _ptr(ga), _ptr(g_attrs), _ptr(ws), L.stream_ptr()),
                    "e3k_kw_stack_bwd")
            del keep
        _consumed_on(ctx.main, g_attrs, *rets)
        return (g_attrs, None, None, None, *rets)


def _radial_alloc(plan, edge_radial, table, n_edges: int, keep: bool, dev):
    """Buffers of one radial branch: (activations buffer, its carve, w [E, W], table rows [R, W] or None); the packed table of the
    in-kernel form rides on the table rows as ``t_tab._e3k_packed``."""
    r, hdim, width = edge_radial.shape[0], plan.last_spec.d_in, plan.last_spec.d_out
    carve = _Carve()
    carve.add("h", r * hdim)
    if keep:
        for i in range(len(plan.mlp_alphas)):
            carve.add(f"z{i}", r * hdim)
    buf = carve.alloc(dev)
    w = None if in_kernel_table(plan, table, dev) else torch.empty(n_edges, width, device=dev, dtype=torch.float32)
    t_tab = torch.empty(r, width, device=dev, dtype=torch.float32) if table is not None else None
    if t_tab is not None and w is None and TP_TABLE_PACKED:
        t_tab._e3k_packed = _packed_buffer(r, width, dev)
    return buf, carve, w, t_tab


class NativeConvBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, node_attrs, edge_radial, sh, plan, topo, groups, in_cf: bool, out_cf: bool, fork: bool, table, nxt,
                pre, m_pre, w_lin1, w_post, w_sc, w_last, *w_hidden):
        """``pre`` (stack mode, ``RadialStackFn``): the radial MLP's output rows of this layer, computed for all layers of
        the network at once -- the knot table [knots + 1, W] (``table`` given: the layer interpolates) or the per-edge
        weights [E, W] themselves; ``edge_radial`` / ``w_last`` / ``w_hidden`` are then unused (None) and ``nxt`` =
        (next plan, next layer's ``pre``) lets this layer issue the next one's interpolation early.
        ``m_pre`` (``KwStackFn``): the per-key contracted self-connection weights M [n_keys, ld_m] of this layer, computed
        for all layers at once; ``node_attrs`` / ``w_sc`` are then unused (None) and the backward hands back gM."""
        from . import conv_block

        stack = pre is not None
        if stack:
            pre = L.f32c(pre)
            edge_radial = pre          # (stands in wherever only identity / residency matters)
        L.require_cuda(x, edge_radial, sh)
        x, edge_radial, sh = L.f32c(x), L.f32c(edge_radial), L.f32c(sh)
        dev = x.device
        layer = native_layer(plan).handle(dev)
        main = ops.current_stream(dev)
        fork = bool(fork) and not torch.cuda.is_current_stream_capturing()
        side = ops.side_stream(dev, 0) if fork else main
        side2 = ops.side_stream(dev, 1) if fork else main
        keep = any(ctx.needs_input_grad)
        has_sc = plan.sc_spec is not None
        addend = None
        addend_inplace = False
        if plan.addend:      # the self-connection's output, computed outside the block (travels in the m_pre slot)
            given = m_pre
            addend, m_pre = L.f32c(m_pre), None
            L.require_cuda(addend)
            # the tensor itself becomes the block's pre-gate buffer (as Linear(base=...) does): the trailing Linear accumulates into
            # it, the gate's backward reads it -- no copy.  (A caller that handed in something that had to be converted gets the copy.)
            addend_inplace = bool(ADDEND_INPLACE) and addend is given and not (addend.is_leaf and addend.requires_grad)
        n, e = x.shape[0], sh.shape[0]
        if addend is not None and tuple(addend.shape) != (n, plan.post_spec.d_out):
            raise ValueError(f"addend {tuple(addend.shape)} is not [N, d_conv] = ({n}, {plan.post_spec.d_out})")
        a = L.LayerFwdArgs()
        a.N, a.E = n, e
        a.in_cf, a.out_cf, a.keep, a.fork = int(in_cf), int(out_cf), int(keep), int(fork)
        a.main, a.side, a.side2 = main.cuda_stream, side.cuda_stream, side2.cuda_stream
        a.x, a.sh = x.data_ptr(), sh.data_ptr()
        a.src, a.dst_ptr, a.dst_perm = topo.src.data_ptr(), topo.dst_ptr.data_ptr(), topo.dst_perm.data_ptr()
        a.w_lin1, a.w_post = w_lin1.data_ptr(), w_post.data_ptr()
        # --- radial branch: this layer's (or the look-ahead's result), and the next layer's look-ahead
        pref, plan.prefetched = plan.prefetched, None
        mode = "stack" if stack else "native"
        own_table = rbuf = rcarve = t_tab = p_tab = None
        inker = in_kernel_table(plan, table, dev)
        if stack and inker:
            w = None                                 # the tensor-product kernels read the table rows themselves
            p_ready = False
            if TP_TABLE_PACKED:
                p_tab = getattr(pre, "_e3k_packed", None)      # packed by the stack, with the tables of the other layers
                p_ready = p_tab is not None
                if p_tab is None:
                    with conv_block._on(side, main):     # (packed behind the stack's rows, on the radial stream)
                        p_tab = _packed_buffer(pre.shape[0], plan.last_spec.d_out, dev)
            _stack_radial_struct(a.rad, plan, pre, table, e, None, p_tab, packed=p_ready)
        elif stack and table is None:
            w = pre                                  # per-edge weights straight from the stack
            _stack_radial_struct(a.rad, plan, pre, None, e, None)
        elif pref is not None and pref[0][0] is edge_radial and pref[0][1] is table and pref[0][2:] == (keep, fork, mode):
            rbuf, rcarve, w, t_tab = pref[1]
            a.has_w = 1
            AHEAD_STATS[0] += 1
            conv_block.AHEAD_STATS[0] += 1
            if stack:
                _stack_radial_struct(a.rad, plan, pre, table, e, w)
        elif stack:
            with conv_block._on(side, main):
                w = torch.empty(e, plan.last_spec.d_out, device=dev, dtype=torch.float32)
            _stack_radial_struct(a.rad, plan, pre, table, e, w)
        else:
            with conv_block._on(side, main):       # (buffers used on the radial stream are allocated on it)
                rbuf, rcarve, w, t_tab = _radial_alloc(plan, edge_radial, table, e, keep, dev)
            own_table = t_tab
        if not stack:
            p_tab = getattr(t_tab, "_e3k_packed", None) if t_tab is not None else None
            _radial_struct(a.rad, plan, edge_radial, table, e, keep, w_last, w_hidden, rbuf, rcarve, w, t_tab, p_tab)
        if p_tab is not None:      # the packed kernels walk edge records (once per batch: the first layer builds them, on this stream)
            a.rad.erec_dst = table.records(topo, sh, "dst").data_ptr()
        nxt_keep = None
        if stack and nxt is not None and fork and conv_block.LOOK_AHEAD and table is not None and not inker:
            plan_n, pre_n = nxt
            nl_n = native_layer(plan_n)
            if nl_n is not None:
                with conv_block._on(side, main):
                    w_n = torch.empty(e, plan_n.last_spec.d_out, device=dev, dtype=torch.float32)
                rad_n = L.LayerRadial()
                _stack_radial_struct(rad_n, plan_n, pre_n, table, e, w_n)
                a.next, a.next_rad = nl_n.handle(dev), C.pointer(rad_n)
                nxt_keep = (rad_n, None, None, w_n, None, plan_n, pre_n)
        elif nxt is not None and fork and conv_block.LOOK_AHEAD and not stack:
            plan_n, w_last_n, w_hidden_n = nxt
            nl_n = native_layer(plan_n)
            if nl_n is not None:
                with conv_block._on(side, main):
                    nbuf, ncarve, w_n, t_n = _radial_alloc(plan_n, edge_radial, table, e, keep, dev)
                rad_n = L.LayerRadial()
                _radial_struct(rad_n, plan_n, edge_radial, table, e, keep, w_last_n, w_hidden_n, nbuf, ncarve, w_n, t_n,
                               getattr(t_n, "_e3k_packed", None) if t_n is not None else None)
                a.next, a.next_rad = nl_n.handle(dev), C.pointer(rad_n)
                nxt_keep = (rad_n, nbuf, ncarve, w_n, t_n, plan_n, w_last_n)
        # --- node side buffers: one allocation
        carve = _Carve()
        need_relayout = (not in_cf) and bool(tuple(b for b in plan.in_blocks if b[1] > 1 and b[2] > 1))
        if need_relayout:
            carve.add("x_cf", n * plan.lin1_spec.d_in)
        have_m = m_pre is not None
        if has_sc and not have_m:
            carve.add("a_rep", groups.n_keys * plan.sc_spec.v)
            carve.add("m", groups.n_keys * plan.sc_ld_m)
        if not addend_inplace:
            carve.add("conv", n * plan.post_spec.d_out)
        carve.add("x1", n * plan.lin1_spec.d_out)
        carve.add("mid", n * plan.post_spec.d_in)
        buf = carve.alloc(dev)
        y = torch.empty(n, plan.gate_spec.out_dim, device=dev, dtype=torch.float32)
        off = carve.off
        if need_relayout:
            a.x_cf = _ptr(buf, off["x_cf"])
        if has_sc:
            a.perm, a.bounds, a.reps, a.n_keys = groups.perm.data_ptr(), groups.bounds.data_ptr(), groups.reps.data_ptr(), groups.n_keys
            if have_m:
                m_pre = L.f32c(m_pre)
                a.have_m, a.m = 1, m_pre.data_ptr()
            else:
                node_attrs = L.f32c(node_attrs)
                a.node_attrs, a.w_sc = node_attrs.data_ptr(), w_sc.data_ptr()
                a.a_rep, a.m = _ptr(buf, off["a_rep"]), _ptr(buf, off["m"])
        a.x1, a.mid, a.y = _ptr(buf, off["x1"]), _ptr(buf, off["mid"]), y.data_ptr()
        if addend_inplace:
            ctx.mark_dirty(addend)
            ctx.set_materialize_grads(False)
            a.conv = addend.data_ptr()
        else:
            a.conv = _ptr(buf, off["conv"])
            if addend is not None:
                buf[off["conv"]:off["conv"] + n * plan.post_spec.d_out].view(n, plan.post_spec.d_out).copy_(addend)
        if HOST_TIMING is not None:
            import time

            t_c = time.perf_counter()
            L.check(L.load().e3k_layer_fwd(layer, C.byref(a)), "e3k_layer_fwd")
            HOST_TIMING[1] += time.perf_counter() - t_c
        else:
            L.check(L.load().e3k_layer_fwd(layer, C.byref(a)), "e3k_layer_fwd")
        if own_table is not None:      # the a-posteriori error guard of the table just built (radial stream)
            with conv_block._on(side, main):
                radial_table.guard(plan.guard_key if plan.guard_key is not None else w_last, own_table, blocks=table.blocks)
        if fork:      # what a stream other than the allocating one touched must not return to the allocator before that
            # stream is done with it.  record_stream costs ~5 us a call: only the pairs that need it -- the radial buffers
            # live on the radial stream (w is also read by the tensor product on this one), tensors every layer of a
            # forward shares (edge embedding, attributes, knot bins) are recorded once per tensor, not once per layer
            if has_sc:
                buf.record_stream(side2)
                x.record_stream(side2)
                if not have_m:
                    _record_once(node_attrs, side2)
                else:
                    m_pre.record_stream(main)      # (allocated by KwStackFn on the self-connection stream; the backward's
                                                   #  input-gradient GEMM reads it on this one)
            if w is None:
                (pre if stack else t_tab).record_stream(main)      # (the table: allocated on the radial stream, read by the tensor product)
                if p_tab is not None:
                    p_tab.record_stream(main)
            elif w is not pre:
                w.record_stream(main)
            else:
                pre.record_stream(main)        # (allocated by the stack on the radial stream, read by the tensor product here)
            _record_once(edge_radial, side)
            if table is not None:
                for t in table:
                    if isinstance(t, torch.Tensor):
                        _record_once(t, side)
        if nxt_keep is not None:
            rad_n, nbuf, ncarve, w_n, t_n, plan_n, w_last_n = nxt_keep
            if t_n is not None:
                with conv_block._on(side, main):
                    radial_table.guard(plan_n.guard_key if plan_n.guard_key is not None else w_last_n, t_n, blocks=table.blocks)
            # (the rows and the table themselves are the key: held here, their identity cannot be reused by a later batch;
            #  in stack mode the key is the next layer's own rows, w_last_n stands for them)
            plan_n.prefetched = ((w_last_n if stack else edge_radial, table, keep, fork, mode), (nbuf, ncarve, w_n, t_n))
        if keep:
            ctx.save_for_backward(x if (in_cf or not need_relayout) else None, edge_radial, sh, buf, rbuf, w, w_lin1, w_post, w_sc,
                                  w_last, t_tab if (w is None and not stack) else None, addend if addend_inplace else m_pre, *w_hidden,
                                  p_tab)
            ctx.cfg = (plan, topo, groups, bool(in_cf), bool(out_cf), fork, len(w_hidden), table, carve, rcarve, need_relayout)
            ctx.stack = stack
            ctx.attrs_shape = tuple(node_attrs.shape) if (has_sc and not have_m) else None
        ctx.addend_inplace = addend_inplace
        if addend_inplace:      # (a dirty input has to be an output; nothing reads it downstream)
            return y, addend
        return y

    @staticmethod
    def backward(ctx, gy, *unused):
        from . import conv_block

        plan, topo, groups, in_cf, out_cf, fork, n_hidden, table, carve, rcarve, need_relayout = ctx.cfg
        saved = ctx.saved_tensors
        x_in, edge_radial, sh, buf, rbuf, w, w_lin1, w_post, w_sc, w_last, t_keep, m_pre = saved[:12]
        conv_ext = None
        if ctx.addend_inplace:      # slot 11 holds the pre-gate buffer (the addend, accumulated on in place)
            conv_ext, m_pre = m_pre, None
            if gy is None:
                gy = torch.zeros(conv_ext.shape[0], plan.gate_spec.out_dim, device=conv_ext.device, dtype=torch.float32)
        w_hidden = saved[12:12 + n_hidden]
        p_tab = saved[12 + n_hidden]                 # the packed table of the in-kernel form (or None)
        need = ctx.needs_input_grad
        need_x, need_attrs, need_radial, need_sh = need[0], need[1], need[2], need[3]
        stack, need_pre = ctx.stack, need[12]
        have_m, need_m = m_pre is not None, need[13]
        need_addend = bool(plan.addend and need[13])
        p0 = 14
        need_lin1, need_post, need_sc, need_last = need[p0], need[p0 + 1], need[p0 + 2], need[p0 + 3]
        need_hidden = need[p0 + 4:]
        if stack:
            need_radial = need_last = False
            need_hidden = ()
        if torch.is_grad_enabled() or need_sh:
            raise RuntimeError(
                "the fused convolution block serves first-order training only (no gradient w.r.t. the spherical harmonics, "
                "no create_graph=True): MessagePassing takes the composed path for those by itself; set E3K_CONV_BLOCK=0 "
                "if this was reached another way")
        has_sc = plan.sc_spec is not None
        dev = gy.device
        layer = native_layer(plan).handle(dev)
        main = ops.current_stream(dev)
        fork = fork and not torch.cuda.is_current_stream_capturing()
        side = ops.side_stream(dev, 0) if fork else main
        side2 = ops.side_stream(dev, 1) if fork else main
        side3 = ops.side_stream(dev, 2) if (fork and ops.WGRAD_SIDE) else main
        gy = L.f32c(gy)
        n, e, r = gy.shape[0], sh.shape[0], edge_radial.shape[0]
        lin1, post, last, sc = plan.lin1_spec, plan.post_spec, plan.last_spec, plan.sc_spec
        off = carve.off
        a = L.LayerBwdArgs()
        a.N, a.E = n, e
        a.in_cf, a.out_cf, a.fork = int(in_cf), int(out_cf), int(fork)
        a.need_x, a.need_attrs, a.need_radial = int(need_x), int(bool(need_attrs and has_sc)), int(need_radial)
        a.fuse_xw = int(TP_BWD_FUSED)
        a.main, a.side, a.side2, a.side3 = main.cuda_stream, side.cuda_stream, side2.cuda_stream, side3.cuda_stream
        a.x_cf = _ptr(buf, off["x_cf"]) if need_relayout else x_in.data_ptr()
        a.sh, a.x1, a.mid = sh.data_ptr(), _ptr(buf, off["x1"]), _ptr(buf, off["mid"])
        a.conv = conv_ext.data_ptr() if conv_ext is not None else _ptr(buf, off["conv"])
        a.src, a.dst, a.dst_ptr, a.dst_perm = topo.src.data_ptr(), topo.dst.data_ptr(), topo.dst_ptr.data_ptr(), topo.dst_perm.data_ptr()
        a.src_ptr, a.src_perm = topo.src_ptr.data_ptr(), topo.src_perm.data_ptr()
        a.w_lin1, a.w_post = w_lin1.data_ptr(), w_post.data_ptr()
        if has_sc:
            if have_m:
                a.have_m, a.m = 1, m_pre.data_ptr()
                need_sc = need_attrs = False      # (formed by KwStackFn's backward from the gm this pass hands back)
            else:
                a.a_rep, a.m, a.w_sc = _ptr(buf, off["a_rep"]), _ptr(buf, off["m"]), w_sc.data_ptr()
            a.perm, a.bounds, a.reps, a.n_keys = groups.perm.data_ptr(), groups.bounds.data_ptr(), groups.reps.data_ptr(), groups.n_keys
        if stack:
            _stack_radial_struct(a.rad, plan, edge_radial, table, e, w, p_tab)      # (edge_radial: the layer's pre-computed rows)
        else:
            _radial_struct(a.rad, plan, edge_radial, table, e, True, w_last, w_hidden, rbuf, rcarve, w,
                           None if table is None else (w if w is not None else t_keep), p_tab)
            if w is not None:
                a.rad.T = None
        if p_tab is not None:
            a.rad.erec_src = table.records(topo, sh, "src").data_ptr()
        a.gy = gy.data_ptr()
        # ---- gradient buffers: the flat gradient buffer (sink) or zero-filled temporaries handed back to autograd
        rets = {}

        def grad_buffer(name, weight, needed):
            if not needed:
                return None
            sink = ops._sink_for(weight)
            if sink is not None:
                return sink
            rets[name] = torch.zeros_like(weight)
            return rets[name].view(-1)

        gb_lin1, gb_post = grad_buffer("lin1", w_lin1, need_lin1), grad_buffer("post", w_post, need_post)
        gb_last = grad_buffer("last", w_last, need_last)
        gb_hidden = [grad_buffer(f"h{i}", wh, need_hidden[i]) for i, wh in enumerate(w_hidden)]
        gb_sc = None
        if has_sc and need_sc:
            sink = ops._sink_for(w_sc)
            if sink is not None:
                gb_sc, a.acc_sc = sink, 1
            else:
                rets["sc"] = ops._kw_weight_buffer(w_sc, sc)
                gb_sc, a.acc_sc = rets["sc"].view(-1), 0
        a.gb_lin1, a.gb_post, a.gb_sc, a.gb_last = _ptr(gb_lin1), _ptr(gb_post), _ptr(gb_sc), _ptr(gb_last)
        for i, g in enumerate(gb_hidden):
            a.gb_hidden[i] = _ptr(g)
        need_radial_side = need_last or need_radial or any(need_hidden) or (stack and need_pre)
        want_sc = has_sc and (need_sc or need_attrs) and not have_m
        g_pre = g_m = None
        if has_sc and have_m and need_m:      # gM: written by the keyed weight-gradient GEMM (weight-gradient stream), read by
            # KwStackFn's backward on the self-connection stream (the executor orders the two)
            ready = getattr(m_pre, "_e3k_gm", None)
            if ready:                           # zero-filled by KwStackFn's forward with the other layers' (taken once)
                g_m = ready.pop()
                a.have_m = 2
                if fork:
                    g_m.record_stream(side3)
            else:
                with conv_block._on(side3, main):
                    g_m = torch.empty(groups.n_keys, plan.sc_ld_m, device=dev, dtype=torch.float32)
            a.gm = g_m.data_ptr()
        # ---- scratch: one allocation
        sc_ = _Carve()
        sc_.add("g_conv", n * post.d_out)
        sc_.add("g_mid", n * post.d_in)
        if need_x or need_lin1:
            sc_.add("g_x1", n * lin1.d_out)
        g_x = None
        if need_x:
            if need_relayout:
                sc_.add("g_xcf", n * lin1.d_in)
                g_x = torch.empty(n, lin1.d_in, device=dev, dtype=torch.float32)
            else:
                g_x = torch.empty(n, lin1.d_in, device=dev, dtype=torch.float32)
        if need_radial_side:
            if stack and table is None:
                g_pre = torch.empty(e, last.d_out, device=dev, dtype=torch.float32)      # g_w IS the gradient of the layer's rows
            else:
                sc_.add("g_w", e * last.d_out)
            if table is not None:
                if stack:
                    with conv_block._on(side, main):
                        g_pre = torch.empty(r, last.d_out, device=dev, dtype=torch.float32)
                else:
                    sc_.add("g_T", r * last.d_out)
                sc_.add("table_ws", int(L.load().e3k_rtable_bwd_workspace_floats(e, table.knots, last.d_out)))
            if need_radial or any(need_hidden):
                sc_.add("g_h", r * last.d_in)
        if want_sc:
            sc_.add("gm", groups.n_keys * plan.sc_ld_m)
            if need_attrs:
                sc_.add("ga", groups.n_keys * sc.v)
                sc_.add("kw_ws", int(L.load().e3k_keyed_weights_bwd_workspace(ops._kw_array(sc, plan.sc_m_off), len(sc.instr),
                                                                               groups.n_keys, sc.v)))
        work = sc_.alloc(dev)
        so = sc_.off
        a.g_conv, a.g_mid = _ptr(work, so["g_conv"]), _ptr(work, so["g_mid"])
        if need_addend:      # d(gate)/d(conv) is the addend's gradient: its own tensor instead of the scratch slot
            g_m = torch.empty(n, post.d_out, device=dev, dtype=torch.float32)
            a.g_conv = g_m.data_ptr()
        if "g_x1" in so:
            a.g_x1 = _ptr(work, so["g_x1"])
        if need_x:
            if need_relayout:
                a.g_xcf, a.g_x = _ptr(work, so["g_xcf"]), g_x.data_ptr()
            else:
                a.g_xcf = g_x.data_ptr()
        for k in ("g_w", "g_T", "table_ws", "g_h", "gm", "ga", "kw_ws"):
            if k in so:
                setattr(a, k, _ptr(work, so[k]))
        if g_pre is not None:
            if table is None:
                a.g_w = g_pre.data_ptr()
            else:
                a.g_T = g_pre.data_ptr()
        g_attrs = g_radial = None
        if need_attrs and has_sc:
            with conv_block._on(side2, main):
                g_attrs = torch.empty(ctx.attrs_shape, device=dev, dtype=torch.float32)
            a.g_attrs = g_attrs.data_ptr()
        if need_radial:
            with conv_block._on(side, main):
                g_radial = torch.empty_like(edge_radial)
            a.g_radial = g_radial.data_ptr()
        if HOST_TIMING is not None:
            import time

            t_c = time.perf_counter()
            L.check(L.load().e3k_layer_bwd(layer, C.byref(a)), "e3k_layer_bwd")
            HOST_TIMING[3] += time.perf_counter() - t_c
        else:
            L.check(L.load().e3k_layer_bwd(layer, C.byref(a)), "e3k_layer_bwd")
        if fork:      # (see the forward: only the stream / tensor pairs that need it)
            for st in (side, side2, side3):
                if st is not main:
                    work.record_stream(st)
            buf.record_stream(side)               # x1: operand of tp_bwd_w when it runs on the radial stream
            if side3 is not main:
                buf.record_stream(side3)          # mid, x_cf: operands of the weight gradients
                if x_in is not None:
                    x_in.record_stream(side3)
            _record_once(sh, side)                # (tp_bwd_w runs on this stream; the radial chain only reads rows it wrote)
        # parameter gradients handed back to autograd (no gradient sink) are consumed on THIS stream
        if "post" in rets or "lin1" in rets:
            conv_block._wait(main, side3)
        if "sc" in rets:
            conv_block._wait(main, side2)
        if "last" in rets or any(f"h{i}" in rets for i in range(n_hidden)):
            conv_block._wait(main, side)
        if fork:
            for t in rets.values():
                t.record_stream(main)
        if ops.GRAD_READY is not None:
            needs = (need_lin1, need_post, need_sc or not has_sc or have_m) + (() if stack else (need_last, *need_hidden))
            if all(needs) and not rets:          # every weight gradient of the layer went to the sink
                # (stack mode: the radial MLP's gradients arrive with RadialStackFn's backward, after the last layer -- the
                #  layer's slice of the all-reduce schedule is then its node-side weights, see run/parallel.py)
                ops.GRAD_READY([w_lin1, w_post] + ([] if stack else [w_last, *w_hidden]) + ([w_sc] if (has_sc and not have_m) else []))
        if g_pre is not None and fork and table is None:
            g_pre.record_stream(side)          # (allocated here, consumed by the stack's backward on the radial stream)
        if g_m is not None and fork:
            g_m.record_stream(side3 if need_addend else side2)      # (addend: read by the weight-gradient GEMMs there)
        return (g_x, g_attrs, g_radial, None, None, None, None, None, None, None, None, None, g_pre, g_m,
                rets.get("lin1"), rets.get("post"), rets.get("sc"), rets.get("last"), *[rets.get(f"h{i}") for i in range(n_hidden)])
