"""autograd wrappers around the libe3k C ABI.

Every op here is ``torch.autograd.Function`` glue: it allocates outputs with torch, fills the
C structs of ``include/e3k.h`` with raw device pointers and enqueues the HIP kernels on the
current stream.  Backward passes call the matching backward kernels.

Double backward (force training: ``GradientOutput`` differentiates with ``create_graph=True``,
``e3_layers/nn/output.py:42-50``, and the loss on the forces is differentiated again): when a
backward runs with grad mode enabled it is expressed through further ``autograd.Function``s
(``*DgradFn``, ``*WgradFn``, ``TpBwdXFn``, ``ActBwdFn`` ...) whose own backward passes reuse the same
kernels with the operands' roles exchanged (every contraction here is multilinear) or the
``*_bwd2`` kernels (activations, gate, spherical harmonics, radial basis).  Third derivatives are
not built and fail loudly.
"""
from __future__ import annotations

import ctypes as C
import math
import os as _os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from .tuning import knob as _knob
from torch.autograd.function import once_differentiable

import weakref

from . import lib as L
from .graph import GraphTopo

# bench.py sets this to a dict to collect, per named kernel, (start_event, end_event, meta) of every launch: HIP
# events recorded on the LAUNCHING stream (the convolution branches run on side streams, which an event pair on
# torch's current stream taken by the caller would not see); None = no profiling.
PROFILE: Optional[Dict[str, list]] = None
PROFILE_ONLY = None      # a set of kernel names: only these are recorded (the timed region times the dominant kernel alone)


class timed_launch:
    """``with timed_launch("tp_fwd", meta): <one kernel launch>`` -- a no-op unless ``PROFILE`` is a dict."""

    __slots__ = ("name", "meta", "ev0")

    def __init__(self, name: str, meta):
        self.name, self.meta, self.ev0 = name, meta, None

    def __enter__(self):
        if PROFILE is not None and (PROFILE_ONLY is None or self.name in PROFILE_ONLY):
            self.ev0 = torch.cuda.Event(enable_timing=True)
            self.ev0.record()

    def __exit__(self, *exc):
        if self.ev0 is not None:
            ev1 = torch.cuda.Event(enable_timing=True)
            ev1.record()
            PROFILE.setdefault(self.name, []).append((self.ev0, ev1, self.meta))
        return False

_side_streams: Dict[int, "torch.cuda.Stream"] = {}
# E3K_FWD_FORK=2: also fork while a HIP graph is being captured (multi-stream capture).  ROCm's graph executor did not
# run the captured branches concurrently (8.45 vs 8.35 ms at 256 molecules), so the default keeps captures single-stream.
FORK_IN_CAPTURE = _knob("E3K_FWD_FORK") == 2


_stream_objects: Dict[tuple, "torch.cuda.Stream"] = {}


def current_stream(device=None) -> "torch.cuda.Stream":
    """``torch.cuda.current_stream(device)`` without building a new Stream object per call (~8 us each, a dozen and a half calls per
    training step): the objects are memoised on (device index, raw stream handle) -- torch's streams live for the whole process."""
    if L._raw_stream is None or L._cur_device is None:
        return torch.cuda.current_stream(device)
    idx = device.index if (device is not None and getattr(device, "index", None) is not None) else L._cur_device()
    raw = L._raw_stream(idx)
    st = _stream_objects.get((idx, raw))
    if st is None:
        st = _stream_objects[(idx, raw)] = torch.cuda.current_stream(idx)
    return st


def side_stream(device, which: int = 0) -> "torch.cuda.Stream":
    """A per-device side stream for an independent branch (``which`` = 0: radial MLP of a convolution; 1: the
    self-connection; 2: sunk weight gradients).  Work enqueued there is joined by ``join_side_streams()`` before
    anything outside autograd (optimizer, all-reduce) reads its results."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = idx if which == 0 else (idx, which)
    st = _side_streams.get(key)
    if st is None:
        st = _side_streams[key] = torch.cuda.Stream(device=device)
    return st


class on_stream:
    """``with on_stream(side, main): ...`` -- make ``side`` the current stream, ``main`` again afterwards.  Both are
    streams of the current device, so this is two ``set_stream`` calls; ``torch.cuda.stream()`` re-derives the
    current stream and device on entry and exit (≈ 15 us of Python per use, a dozen uses per step)."""

    __slots__ = ("side", "main")

    def __init__(self, side, main):
        self.side, self.main = side, main

    def __enter__(self):
        torch.cuda.set_stream(self.side)

    def __exit__(self, *exc):
        torch.cuda.set_stream(self.main)
        return False


def side_streams_of(device_index: int):
    return [st for key, st in _side_streams.items() if (key if isinstance(key, int) else key[0]) == device_index]


def join_side_streams() -> None:
    """Make the current stream wait for every side stream of its device (cheap when they are idle).  The gradient
    sink writes weight gradients without an AccumulateGrad node, so autograd's end-of-backward stream sync does not
    know about them: the optimizer / all-reduce call this first."""
    if not _side_streams or not torch.cuda.is_available():
        return
    if not (FORK_IN_CAPTURE or not torch.cuda.is_current_stream_capturing()):
        return
    cur_dev = torch.cuda.current_device()
    for key, st in _side_streams.items():
        if (key if isinstance(key, int) else key[0]) == cur_dev:
            torch.cuda.current_stream().wait_stream(st)


# Gradient sink: run/parallel.FlatGradients registers (data_ptr, numel) -> view of its flat gradient
# buffer for every parameter.  A backward that finds its weight there accumulates the weight
# gradient straight into the (pre-zeroed) flat buffer and returns None for it: no zero-fill of a
# temporary and no autograd "+=" per parameter.
# Entries are (buffer slice, weak reference to the Parameter): an address is only trusted while the Parameter that
# registered it is alive and still lives there (``model.to()``, ``load_state_dict(assign=True)`` move parameters, and
# the allocator may hand the old address to an unrelated tensor).
GRAD_SINK: Dict[Tuple[int, int], Tuple[torch.Tensor, "weakref.ref"]] = {}
# Set by run/parallel.FlatGradients.enable_overlapped_all_reduce: called (from the autograd engine's thread) with the
# weight tensors of a layer once every kernel that adds into their sunk gradients has been ENQUEUED (on whichever
# stream) -- the listener starts that slice's all-reduce behind them while the backward of the earlier layers goes on.
GRAD_READY = None
WGRAD_SIDE = _knob("E3K_WGRAD_SIDE")            # sunk weight gradients of the Linears run on a side stream
WGRAD_SIDE_MIN_ROWS = _knob("E3K_WGRAD_SIDE_MIN_ROWS")
class _Mode:
    """A process-wide mode flag with an owner.  These flags steer the e3k autograd functions from OUTSIDE a forward or
    backward call (a fork scope around a convolution, a gradient-selection scope around ``autograd.grad`` /
    ``backward``).  They cannot be thread-local -- the autograd engine runs backward functions on its own threads, not on
    the caller's -- so they are process-wide, and therefore guarded: a second thread entering a scope that another
    thread holds raises instead of silently inheriting (or clobbering) the other's mode.  Truthiness = inside a scope."""

    __slots__ = ("name", "depth", "owner", "_lock")

    def __init__(self, name: str):
        import threading

        self.name, self.depth, self.owner, self._lock = name, 0, None, threading.Lock()

    def __bool__(self) -> bool:
        return self.depth > 0

    def __enter__(self):
        import threading

        me = threading.get_ident()
        with self._lock:
            if self.depth and self.owner != me:
                raise RuntimeError(f"ops.{self.name} is held by another thread: the e3k backward modes are process-wide "
                                   "(one model step at a time per process; use one process per GPU)")
            self.owner = me
            self.depth += 1
        return self

    def __exit__(self, *exc):
        with self._lock:
            self.depth -= 1
            if self.depth == 0:
                self.owner = None
        return False


# Inside ``with ops.IN_FORK:`` a convolution runs its forked (multi-stream) forward: only ops recorded then move their
# sunk weight gradients to the side stream -- small, host-bound batches keep everything on one stream.  (Each autograd
# function copies the flag into its ``ctx`` at forward time; the backward reads the copy, never the flag.)
IN_FORK = _Mode("IN_FORK")
# Inside ``with ops.inputs_only_backward():`` GradientOutput differentiates its function w.r.t. a data tensor (forces =
# -dE/dpos): that backward pass needs no parameter gradients, yet ``ctx.needs_input_grad`` of a Python autograd.Function
# is fixed at forward time and says True for every Parameter -- the weight-gradient GEMMs of the force pass would all
# be computed and dropped.  (torch's own ops ask the engine per call; custom Functions cannot.)
INPUTS_ONLY = _Mode("INPUTS_ONLY")
# Inside ``with ops.params_only_backward(): loss.backward(inputs=params)`` the caller differentiates w.r.t. Parameters
# only: in a force-training step ``pos`` still requires grad when the loss is differentiated, so every backward would
# also produce d loss / d pos through the spherical harmonics -- three extra grad_sh edge passes per layer that nothing
# consumes.  Tensors the geometry layers mark as functions of the input data alone get no gradient then.
PARAMS_ONLY = _Mode("PARAMS_ONLY")


def inputs_only_backward() -> _Mode:
    """``with ops.inputs_only_backward(): torch.autograd.grad(y, data_tensor, ...)``: gradients of Parameters (and of
    tensors computed from Parameters alone) are skipped inside the e3k backward functions."""
    return INPUTS_ONLY


def params_only_backward() -> _Mode:
    return PARAMS_ONLY


def is_data_only(t) -> bool:
    """Set by computeEdgeVector / SphericalEncoding on tensors computed from the input positions alone."""
    return bool(getattr(t, "_e3k_data_only", False))


def mark_data_only(t, flag: bool = True):
    if flag and t is not None:
        t._e3k_data_only = True
    return t


def _param_only(t) -> bool:
    return isinstance(t, torch.nn.Parameter) or getattr(t, "_e3k_param_only", False)


def _param_slots(*inputs):
    return tuple(i for i, t in enumerate(inputs) if isinstance(t, torch.Tensor) and _param_only(t))


def _needs(ctx):
    need = ctx.needs_input_grad
    if INPUTS_ONLY:
        slots = getattr(ctx, "param_slots", ())
        if slots:
            need = tuple(False if i in slots else v for i, v in enumerate(need))
    return need


def _sink_for(t: torch.Tensor):
    if not GRAD_SINK:
        return None
    key = (t.data_ptr(), t.numel())
    hit = GRAD_SINK.get(key)
    if hit is None:
        return None
    buf, ref = hit
    p = ref()
    if p is None or p.data_ptr() != key[0] or p.numel() != key[1]:
        del GRAD_SINK[key]       # stale: the parameter was freed or re-allocated after enable_direct_accumulation()
        return None
    return buf


ACT_IDS = {None: 0, "identity": 0, "ssp": 1, "silu": 2, "tanhlu": 3, "tanh": 4, "abs": 5}


def _c(t):
    """Contiguous fp32 through *differentiable* torch ops, applied before ``Function.apply`` so that the
    tensors a Function saves are its true inputs (double backward differentiates through them)."""
    if t is None or (t.dtype == torch.float32 and t.is_contiguous()):
        return t
    return t.float().contiguous()


def _addr(t: torch.Tensor, elem_off: int = 0) -> int:
    return t.data_ptr() + 4 * int(elem_off)


# tools/gemm_profile.py sets this to a list: (start_event, end_event, wgrad, [(M1, M2, N, K, V), ...]) per launch group
PROFILE_GEMM = None


def _run_gemm(problems: List[L.GemmProblem], wgrad: bool = False) -> None:
    if not problems:
        return
    arr = (L.GemmProblem * len(problems))(*problems)
    lib = L.load()
    fn = lib.e3k_gemm_wgrad if wgrad else lib.e3k_gemm
    prof = PROFILE_GEMM
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    L.check(fn(arr, len(problems), L.stream_ptr()), "e3k_gemm_wgrad" if wgrad else "e3k_gemm")
    if prof is not None:
        ev1.record()
        prof.append((ev0, ev1, wgrad, [(p.M1, p.M2, p.N, p.K, p.V) for p in problems]))


def _layout_strides(layout: str, mul: int, dim: int) -> Tuple[int, int]:
    """(stride over m = r2, stride over the channel) of one irrep block."""
    if layout == "e3nn":
        return 1, dim
    if layout == "cf":
        return mul, 1
    raise ValueError(layout)


# --------------------------------------------------------------------------------------
# o3.Linear as grouped strided GEMMs
# --------------------------------------------------------------------------------------
@dataclass
class LinInstr:
    in_off: int
    out_off: int
    mul_in: int
    mul_out: int
    dim: int
    w_off: int
    alpha: float
    i_in: int = 0
    i_out: int = 0


@dataclass
class LinearSpec:
    d_in: int
    d_out: int
    instr: List[LinInstr]
    in_layout: str = "e3nn"
    out_layout: str = "e3nn"
    bias_blocks: List[Tuple[int, int, int]] = field(default_factory=list)  # (out_off, mul, bias_off)
    out_covered: bool = True  # every output element is written by some instruction
    in_covered: bool = True   # every input element feeds some instruction
    weight_numel: int = 0

    def rounds(self, key: str) -> List[List[LinInstr]]:
        """Group instructions so that inside one launch no two write the same block."""
        seen: Dict[int, int] = {}
        out: List[List[LinInstr]] = []
        for ins in self.instr:
            k = getattr(ins, key)
            r = seen.get(k, 0)
            seen[k] = r + 1
            while len(out) <= r:
                out.append([])
            out[r].append(ins)
        return out


def _chain_rounds(rounds):
    """Rounds of GEMM problems (round r + 1 accumulates onto blocks round r wrote) -> the same sums with the later rounds' problems
    hung behind their round-0 problem as K-CHAIN followers (``e3k.h``: ``e3k_gemm_problem.chain``): one pass over C, one launch.
    A problem that finds no head of its shape stays in its round."""
    if len(rounds) < 2 or not _knob("E3K_GEMM_CHAIN"):
        return rounds
    heads = [[p] for p in rounds[0]]
    where = {}
    for i, (p,) in enumerate(heads):
        if not p.V and not p.chain:
            where[(p.C, p.M2, p.N, p.c_r1, p.c_r2, p.c_n)] = i
    left = []
    for r, group in enumerate(rounds[1:], 1):
        rest = []
        for p in group:
            i = where.get((p.C, p.M2, p.N, p.c_r1, p.c_r2, p.c_n))
            if i is None or p.V or p.bias or p.act != heads[i][0].act:
                rest.append(p)
            else:
                heads[i].append(p)
        left.append(rest)
    first = []
    for chain in heads:
        chain[0].chain = len(chain) - 1
        for f in chain[1:]:
            f.chain = 0
        first += chain
    return [first] + [g for g in left if g]


class _GemmTemplates:
    """Descriptor arrays of one (layer, pass), built once: pointer fields hold byte offsets (``e3k_gemm_rebased``)."""

    __slots__ = ("rounds", "shapes", "loose_bias")

    def __init__(self, rounds: List[List[L.GemmProblem]], loose_bias=(), wgrad: bool = False):
        if not wgrad:
            rounds = _chain_rounds([g for g in rounds if g])
        self.rounds = [((L.GemmProblem * len(g))(*g), len(g)) for g in rounds if g]
        self.shapes = [[(p.M2, p.N, p.K, p.V) for p in g] for g in rounds if g]
        self.loose_bias = tuple(loose_bias)

    def run(self, a, a2, b, c, bias, rows: int, wgrad: bool = False) -> None:
        lib, st = L.load(), L.stream_ptr()
        prof = PROFILE_GEMM
        for (arr, n), shapes in zip(self.rounds, self.shapes):
            if prof is not None:
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
            L.check(lib.e3k_gemm_rebased(arr, n, a, a2, b, c, bias, rows, int(wgrad), st), "e3k_gemm_rebased")
            if prof is not None:
                ev1.record()
                prof.append((ev0, ev1, wgrad, [(rows,) + sh for sh in shapes]))


def _templates(spec, key, build) -> _GemmTemplates:
    cache = spec.__dict__.get("_gemm_templates")
    if cache is None:
        cache = spec.__dict__["_gemm_templates"] = {}
    t = cache.get(key)
    if t is None:
        t = cache[key] = build()
    return t


def _lin_fwd_templates(spec: "LinearSpec", scale: float, accumulate: bool, act: int, act_cst: float, has_bias: bool):
    bias_at = {off: 4 * boff + 1 for off, mul, boff in spec.bias_blocks} if has_bias else {}
    done_bias = set()
    rounds = spec.rounds("i_out")
    if act and len(rounds) != 1:
        raise NotImplementedError("fused activation needs single-round linears")
    out = []
    for r, group in enumerate(rounds):
        probs = []
        for ins in group:
            a_r2, a_k = _layout_strides(spec.in_layout, ins.mul_in, ins.dim)
            c_r2, c_n = _layout_strides(spec.out_layout, ins.mul_out, ins.dim)
            p = L.GemmProblem()
            p.A, p.A2, p.B, p.C = 4 * ins.in_off, None, 4 * ins.w_off, 4 * ins.out_off
            p.bias = None
            if r == 0 and ins.out_off in bias_at:
                p.bias = bias_at[ins.out_off]
                done_bias.add(ins.out_off)
            p.M1, p.M2, p.N, p.K, p.V = 0, ins.dim, ins.mul_out, ins.mul_in, 0
            p.accumulate = 1 if (r > 0 or accumulate) else 0
            p.a_r1, p.a_r2, p.a_k = spec.d_in, a_r2, a_k
            p.b_k, p.b_n = ins.mul_out, 1
            p.c_r1, p.c_r2, p.c_n = spec.d_out, c_r2, c_n
            p.alpha = ins.alpha * scale
            if act:
                p.act, p.act_cst = act, act_cst
            probs.append(p)
        out.append(probs)
    # biased blocks without any incoming path
    loose = [(off, mul, boff) for off, mul, boff in spec.bias_blocks if has_bias and off not in done_bias]
    return _GemmTemplates(out, loose)


def _lin_fwd_raw(x, weight, bias, y, spec: LinearSpec, scale: float, accumulate: bool, act: int = 0, act_cst: float = 1.0):
    """y (+)= sum over instructions alpha*scale * x_block @ W_block (+ bias); raw launch, no autograd."""
    has_bias = bias is not None
    t = _templates(spec, ("fwd", scale, bool(accumulate), act, act_cst, has_bias),
                   lambda: _lin_fwd_templates(spec, scale, bool(accumulate), act, act_cst, has_bias))
    t.run(x.data_ptr(), None, weight.data_ptr(), y.data_ptr(), bias.data_ptr() if has_bias else None, x.shape[0])
    for off, mul, boff in t.loose_bias:
        y[:, off:off + mul] += bias[boff:boff + mul]


def _lin_dgrad_templates(spec: "LinearSpec", scale: float, accumulate: bool = False):
    out = []
    for r, group in enumerate(spec.rounds("i_in")):
        probs = []
        for ins in group:
            a_r2, a_k = _layout_strides(spec.out_layout, ins.mul_out, ins.dim)
            c_r2, c_n = _layout_strides(spec.in_layout, ins.mul_in, ins.dim)
            p = L.GemmProblem()
            p.A, p.A2, p.B, p.C, p.bias = 4 * ins.out_off, None, 4 * ins.w_off, 4 * ins.in_off, None
            p.M1, p.M2, p.N, p.K, p.V = 0, ins.dim, ins.mul_in, ins.mul_out, 0
            p.accumulate = 1 if (r > 0 or accumulate) else 0
            p.a_r1, p.a_r2, p.a_k = spec.d_out, a_r2, a_k
            p.b_k, p.b_n = 1, ins.mul_out  # W^T
            p.c_r1, p.c_r2, p.c_n = spec.d_in, c_r2, c_n
            p.alpha = ins.alpha * scale
            probs.append(p)
        out.append(probs)
    return _GemmTemplates(out)


def _lin_dgrad_raw(gy, weight, spec: LinearSpec, scale: float, out=None, accumulate: bool = False):
    """gx (+)= sum over instructions alpha*scale * gy_block @ W_block^T  (``out``: write / accumulate into this buffer)."""
    rows = gy.shape[0]
    gx = out if out is not None else (torch.empty if spec.in_covered else torch.zeros)(rows, spec.d_in, device=gy.device, dtype=torch.float32)
    accumulate = bool(accumulate and out is not None)
    t = _templates(spec, ("dgrad", scale, accumulate), lambda: _lin_dgrad_templates(spec, scale, accumulate))
    t.run(gy.data_ptr(), None, weight.data_ptr(), gx.data_ptr(), None, rows)
    return gx


def _lin_wgrad_templates(spec: "LinearSpec", scale: float):
    probs = []
    for ins in spec.instr:
        a_r2, a_k = _layout_strides(spec.in_layout, ins.mul_in, ins.dim)
        c_r2, c_n = _layout_strides(spec.out_layout, ins.mul_out, ins.dim)
        p = L.GemmProblem()
        p.A, p.A2, p.B, p.C, p.bias = 4 * ins.in_off, None, 4 * ins.w_off, 4 * ins.out_off, None
        p.M1, p.M2, p.N, p.K, p.V = 0, ins.dim, ins.mul_out, ins.mul_in, 0
        p.accumulate = 1
        p.a_r1, p.a_r2, p.a_k = spec.d_in, a_r2, a_k
        p.b_k, p.b_n = ins.mul_out, 1
        p.c_r1, p.c_r2, p.c_n = spec.d_out, c_r2, c_n
        p.alpha = ins.alpha * scale
        probs.append(p)
    return _GemmTemplates([probs], wgrad=True)


def _lin_wgrad_raw(x, gy, gw, spec: LinearSpec, scale: float) -> None:
    """gw += alpha*scale * x_block^T @ gy_block per instruction (gw pre-zeroed or a gradient sink)."""
    t = _templates(spec, ("wgrad", scale), lambda: _lin_wgrad_templates(spec, scale))
    t.run(x.data_ptr(), None, gw.data_ptr(), gy.data_ptr(), None, x.shape[0], wgrad=True)


def _bias_grad_diff(gy, spec: LinearSpec):
    """Differentiable bias gradient (double-backward mode only): column sums of the biased blocks."""
    blocks = sorted(spec.bias_blocks, key=lambda b: b[2])
    pos = 0
    parts = []
    for off, mul, boff in blocks:
        assert boff == pos, "bias blocks must tile the bias vector"
        parts.append(gy[:, off:off + mul].sum(0))
        pos += mul
    return torch.cat(parts)


class StridedLinearFn(torch.autograd.Function):
    """y[rows, d_out] = (base +) sum over instructions  alpha * x_block @ W_block  (+ bias)."""

    @staticmethod
    def forward(ctx, x, weight, bias, base, spec: LinearSpec, scale: float, act: int = 0, act_cst: float = 1.0):
        ctx.param_slots = _param_slots(x, weight, bias, base)
        L.require_cuda(x, weight)
        x = L.f32c(x)
        weight = L.f32c(weight)
        rows = x.shape[0]
        assert x.shape[1] == spec.d_in, (x.shape, spec.d_in)
        if base is not None:
            if act:
                raise NotImplementedError("fused activation cannot accumulate into a base tensor")
            y = base
            ctx.mark_dirty(base)
        elif spec.out_covered:
            y = torch.empty(rows, spec.d_out, device=x.device, dtype=torch.float32)
        else:
            y = torch.zeros(rows, spec.d_out, device=x.device, dtype=torch.float32)
        if bias is not None:
            bias = L.f32c(bias)
        _lin_fwd_raw(x, weight, bias, y, spec, scale, base is not None, act, act_cst)
        if act:
            ctx.save_for_backward(x, weight, y)
        else:
            ctx.save_for_backward(x, weight)
        ctx.act, ctx.act_cst = act, act_cst
        ctx.spec, ctx.scale, ctx.has_bias, ctx.has_base = spec, scale, bias is not None, base is not None
        ctx.bias_param = bias      # (a parameter: looked up in the gradient sink by the backward)
        ctx.in_fork = bool(IN_FORK)
        return y

    @staticmethod
    def backward(ctx, gy):
        need = _needs(ctx)
        spec: LinearSpec = ctx.spec
        scale = ctx.scale
        if torch.is_grad_enabled():   # double backward: stay differentiable
            if ctx.act:
                raise NotImplementedError("double backward through a fused linear+activation layer is not built; "
                                          "use the unfused layers (default)")
            x, weight = ctx.saved_tensors
            gx = LinearDgradFn.apply(gy, weight, spec, scale) if need[0] else None
            gw = LinearWgradFn.apply(x, gy, spec, scale, weight.numel()).view_as(weight) if need[1] else None
            gb = _bias_grad_diff(gy, spec) if (ctx.has_bias and need[2]) else None
            gbase = gy if (ctx.has_base and need[3]) else None
            return gx, gw, gb, gbase, None, None, None, None
        if ctx.act:
            x, weight, y = ctx.saved_tensors
            gy = L.f32c(gy)
            gz = torch.empty_like(gy)
            L.check(L.load().e3k_act_bwd_from_output(L.ptr(y), L.ptr(gy), gy.numel(), ctx.act, ctx.act_cst, L.ptr(gz),
                                                     L.stream_ptr()), "e3k_act_bwd_from_output")
            gy = gz
        else:
            x, weight = ctx.saved_tensors
        gy = L.f32c(gy)
        rows = x.shape[0]
        gx = gw = gb = None
        sunk = False
        if need[1]:
            gw = _sink_for(weight)
            sunk = gw is not None
            if not sunk:
                gw = torch.zeros_like(weight)
            if (sunk and WGRAD_SIDE and ctx.in_fork and need[0] and rows >= WGRAD_SIDE_MIN_ROWS
                    and not torch.cuda.is_current_stream_capturing()):
                # a weight gradient that lands in the gradient sink is off the critical path: nothing in the
                # backward consumes it, so it goes to a side stream that only the optimizer / all-reduce joins
                cur = current_stream(x.device)
                st = side_stream(x.device, 2)
                st.wait_stream(cur)
                with on_stream(st, cur):
                    _lin_wgrad_raw(x, gy, gw, spec, scale)
                x.record_stream(st)
                gy.record_stream(st)
            else:
                _lin_wgrad_raw(x, gy, gw, spec, scale)
        if need[0]:
            gx = _lin_dgrad_raw(gy, weight, spec, scale)
        if sunk:
            gw = None   # already accumulated into the flat gradient buffer
        if ctx.has_bias and need[2]:
            nb = sum(m for _, m, _ in spec.bias_blocks)
            # (the column sums are accumulated with atomics: straight into the flat gradient buffer when the bias has a slot there --
            #  no zero-filled temporary, no add by autograd)
            gb_out = _sink_for(ctx.bias_param) if ctx.bias_param.numel() == nb else None
            if gb_out is None:
                gb = gb_out = torch.zeros(nb, device=x.device, dtype=torch.float32)
            lib = L.load()
            for off, mul, boff in spec.bias_blocks:
                L.check(lib.e3k_colsum(_addr(gy, off), rows, mul, spec.d_out, _addr(gb_out, boff), L.stream_ptr()), "e3k_colsum")
        gbase = gy if (ctx.has_base and need[3]) else None
        return gx, gw, gb, gbase, None, None, None, None


class LinearDgradFn(torch.autograd.Function):
    """gx = dgrad(gy, W); bilinear, so its backward is again (forward, wgrad) with roles exchanged."""

    @staticmethod
    def forward(ctx, gy, weight, spec: LinearSpec, scale: float):
        gy, weight = L.f32c(gy), L.f32c(weight)
        ctx.save_for_backward(gy, weight)
        ctx.spec, ctx.scale = spec, scale
        return _lin_dgrad_raw(gy, weight, spec, scale)

    @staticmethod
    def backward(ctx, h):   # h: cotangent of gx, shaped like x
        gy, weight = ctx.saved_tensors
        g_gy = StridedLinearFn.apply(h, weight, None, None, ctx.spec, ctx.scale) if ctx.needs_input_grad[0] else None
        g_w = None
        if ctx.needs_input_grad[1]:
            sink = None if torch.is_grad_enabled() else _sink_for(weight)
            if sink is not None:    # last backward of a force-training step: straight into the flat gradient buffer
                _lin_wgrad_raw(L.f32c(h), gy, sink, ctx.spec, ctx.scale)
            else:
                g_w = LinearWgradFn.apply(h, gy, ctx.spec, ctx.scale, weight.numel())
        return g_gy, g_w, None, None


class LinearWgradFn(torch.autograd.Function):
    """gw = wgrad(x, gy); its backward: g_x = dgrad(gy, h), g_gy = forward(x, h) for the cotangent h of gw."""

    @staticmethod
    def forward(ctx, x, gy, spec: LinearSpec, scale: float, w_numel: int):
        x, gy = L.f32c(x), L.f32c(gy)
        ctx.save_for_backward(x, gy)
        ctx.spec, ctx.scale = spec, scale
        gw = torch.zeros(w_numel, device=x.device, dtype=torch.float32)
        _lin_wgrad_raw(x, gy, gw, spec, scale)
        return gw

    @staticmethod
    def backward(ctx, h):
        x, gy = ctx.saved_tensors
        h = h.reshape(-1)
        g_x = LinearDgradFn.apply(gy, h, ctx.spec, ctx.scale) if ctx.needs_input_grad[0] else None
        g_gy = StridedLinearFn.apply(x, h, None, None, ctx.spec, ctx.scale) if ctx.needs_input_grad[1] else None
        return g_x, g_gy, None, None, None


def strided_linear(x, weight, bias, spec: LinearSpec, base=None, scale: float = 1.0, act: Optional[str] = None,
                   act_cst: float = 1.0):
    """``act='ssp'`` fuses ``act_cst * ssp(.)`` into the GEMM epilogue (radial MLP layers)."""
    x, weight, bias = _c(x), _c(weight), _c(bias)
    if act is None:
        return StridedLinearFn.apply(x, weight, bias, base, spec, float(scale))
    if act != "ssp":
        return activation(StridedLinearFn.apply(x, weight, bias, base, spec, float(scale)), act, act_cst)
    return StridedLinearFn.apply(x, weight, bias, base, spec, float(scale), 1, float(act_cst))


# --------------------------------------------------------------------------------------
# fused hidden chain of the radial MLP
# --------------------------------------------------------------------------------------
def mlp_hidden_supported(k0: int, hs: Sequence[int], act: Optional[str]) -> bool:
    """Shapes the fused kernels take: input width <= 64, 1-4 hidden layers of one width 32 or 64."""
    return (0 < k0 <= 64 and 1 <= len(hs) <= 4 and len(set(hs)) == 1 and hs[0] in (32, 64)
            and act in ACT_IDS and act is not None)


def _ptr_array(tensors):
    return (C.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


_DENSE_SPECS: Dict[Tuple[int, int, float], "LinearSpec"] = {}   # specs own the cached descriptor templates


def _mlp_unfused(x, weights, alphas, act: str, cst: float):
    """The same chain through the per-layer ops (used to build a double-backward graph)."""
    cur = x
    for w, al in zip(weights, alphas):
        k, n = w.shape
        spec = _DENSE_SPECS.get((k, n, al))
        if spec is None:
            spec = _DENSE_SPECS[(k, n, al)] = LinearSpec(k, n, [LinInstr(0, 0, k, n, 1, 0, al)], "e3nn", "e3nn", [], True, True, k * n)
        cur = activation(strided_linear(cur, w.reshape(-1), None, spec), act, cst)
    return cur


def _mlp_fwd_raw(x, weights, alphas, act: str, cst: float, keep: bool):
    """h = (act o linear)^L (x); ``keep``: also return the pre-activations the backward needs."""
    e, k0 = x.shape
    h, n = weights[0].shape[1], len(weights)
    out = torch.empty(e, h, device=x.device, dtype=torch.float32)
    zs = [torch.empty(e, h, device=x.device, dtype=torch.float32) for _ in range(n)] if keep else []
    L.check(L.load().e3k_mlp_hidden_fwd(L.ptr(x), e, k0, h, n, _ptr_array(weights), (C.c_float * n)(*alphas), ACT_IDS[act],
                                        cst, _ptr_array(zs) if zs else None, L.ptr(out), L.stream_ptr()),
            "e3k_mlp_hidden_fwd")
    return out, zs


def _mlp_bwd_raw(x, weights, zs, alphas, act: str, cst: float, g, gws, gx):
    """gws[l] (accumulated; None = skipped), gx ([E,k0] or None) from g = gradient wrt the chain's output."""
    e, k0 = x.shape
    h, n = weights[0].shape[1], len(weights)
    L.check(L.load().e3k_mlp_hidden_bwd(L.ptr(x), e, k0, h, n, _ptr_array(weights), (C.c_float * n)(*alphas), ACT_IDS[act],
                                        cst, _ptr_array(zs), L.ptr(g), _ptr_array(gws), L.ptr(gx), L.stream_ptr()),
            "e3k_mlp_hidden_bwd")


class MlpHiddenFn(torch.autograd.Function):
    """h = (act o linear)^L (x): one launch forward, one backward (csrc/e3k_mlp.hip)."""

    @staticmethod
    def forward(ctx, x, alphas: Tuple[float, ...], act: str, cst: float, *weights):
        ctx.param_slots = _param_slots(x, None, None, None, *weights)
        L.require_cuda(x, *weights)
        x = L.f32c(x)
        weights = tuple(L.f32c(w) for w in weights)
        e, k0 = x.shape
        h, n = weights[0].shape[1], len(weights)
        need_grad = any(ctx.needs_input_grad)
        out = torch.empty(e, h, device=x.device, dtype=torch.float32)
        zs = [torch.empty(e, h, device=x.device, dtype=torch.float32) for _ in range(n)] if need_grad else []
        L.check(L.load().e3k_mlp_hidden_fwd(L.ptr(x), e, k0, h, n, _ptr_array(weights), (C.c_float * n)(*alphas), ACT_IDS[act],
                                            cst, _ptr_array(zs) if zs else None, L.ptr(out), L.stream_ptr()),
                "e3k_mlp_hidden_fwd")
        ctx.save_for_backward(x, *weights, *zs)
        ctx.cfg = (alphas, act, cst, n)
        return out

    @staticmethod
    def backward(ctx, g):
        need = _needs(ctx)
        alphas, act, cst, n = ctx.cfg
        saved = ctx.saved_tensors
        x, weights, zs = saved[0], saved[1:1 + n], saved[1 + n:]
        if torch.is_grad_enabled():   # double backward: differentiate the per-layer ops instead
            wrt = [t for t, nd in zip((x,) + tuple(weights), (need[0],) + tuple(need[4:])) if nd]
            with torch.enable_grad():
                y = _mlp_unfused(x, weights, alphas, act, cst)
                grads = list(torch.autograd.grad(y, wrt, g, create_graph=True, allow_unused=True))
            gx = grads.pop(0) if need[0] else None
            gws = [grads.pop(0) if nd else None for nd in need[4:]]
            return (gx, None, None, None, *gws)
        g = L.f32c(g)
        e, k0 = x.shape
        h = weights[0].shape[1]
        gx = torch.empty_like(x) if need[0] else None
        gws, ret = [], []
        for w, nd in zip(weights, need[4:]):
            if not nd:
                gws.append(None)
                ret.append(None)
                continue
            sink = _sink_for(w)
            if sink is not None:
                gws.append(sink)
                ret.append(None)
            else:
                buf = torch.zeros_like(w)
                gws.append(buf)
                ret.append(buf)
        L.check(L.load().e3k_mlp_hidden_bwd(L.ptr(x), e, k0, h, n, _ptr_array(weights), (C.c_float * n)(*alphas), ACT_IDS[act],
                                            cst, _ptr_array(zs), L.ptr(g), _ptr_array(gws), L.ptr(gx), L.stream_ptr()),
                "e3k_mlp_hidden_bwd")
        return (gx, None, None, None, *ret)


def mlp_hidden(x, weights: Sequence[torch.Tensor], alphas: Sequence[float], act: str, cst: float):
    """weights[l]: [k_l, h] parameters (2-D); returns the activations of the last hidden layer."""
    return MlpHiddenFn.apply(_c(x), tuple(float(a) for a in alphas), act, float(cst), *[_c(w) for w in weights])


# --------------------------------------------------------------------------------------
# FullyConnectedTensorProduct with scalar second operand (the self-connection)
# --------------------------------------------------------------------------------------
@dataclass
class FctpInstr:
    in_off: int
    out_off: int
    mul_in: int
    mul_out: int
    dim: int
    w_off: int
    alpha: float
    i_in: int = 0
    i_out: int = 0


@dataclass
class FctpSpec:
    d_in: int
    d_out: int
    v: int
    instr: List[FctpInstr]
    in_layout: str = "cf"
    out_layout: str = "cf"
    out_covered: bool = True
    in_covered: bool = True

    rounds = LinearSpec.rounds


def _fctp_problems(x, attrs, b, c, spec: FctpSpec, rows: int, wgrad: bool):
    """Outer-mode problems  C[(n,k), w] (+)= alpha * sum_{u,v} x[n,u,k] attrs[n,v] B[(u,v), w]  (forward: B = weight, C = out;
    wgrad: B = the weight gradient, accumulated, C = the incoming gradient), one list per round."""
    rounds = spec.rounds("i_out") if not wgrad else [list(spec.instr)]
    out = []
    for r, group in enumerate(rounds):
        probs = []
        for ins in group:
            a_r2, a_k = _layout_strides(spec.in_layout, ins.mul_in, ins.dim)
            c_r2, c_n = _layout_strides(spec.out_layout, ins.mul_out, ins.dim)
            p = L.GemmProblem()
            p.A, p.A2, p.B, p.C, p.bias = _addr(x, ins.in_off), _addr(attrs), _addr(b, ins.w_off), _addr(c, ins.out_off), None
            p.M1, p.M2, p.N, p.K, p.V = rows, ins.dim, ins.mul_out, ins.mul_in * spec.v, spec.v
            p.accumulate = 1 if (wgrad or r > 0) else 0
            p.a_r1, p.a_r2, p.a_k, p.a2_r1 = spec.d_in, a_r2, a_k, spec.v
            p.b_k, p.b_n = ins.mul_out, 1
            p.c_r1, p.c_r2, p.c_n = spec.d_out, c_r2, c_n
            p.alpha = ins.alpha
            probs.append(p)
        out.append(probs)
    return out


def _fctp_fwd_raw(x, attrs, weight, spec: FctpSpec):
    rows = x.shape[0]
    y = (torch.empty if spec.out_covered else torch.zeros)(rows, spec.d_out, device=x.device, dtype=torch.float32)
    for probs in _fctp_problems(x, attrs, weight, y, spec, rows, False):
        _run_gemm(probs)
    return y


def _fctp_wgrad_raw(x, attrs, gy, gw, spec: FctpSpec) -> None:
    """gw += alpha * sum_n (x (x) attrs)^T gy  (gw zero-filled or a gradient sink)."""
    for probs in _fctp_problems(x, attrs, gw, gy, spec, x.shape[0], True):
        _run_gemm(probs, wgrad=True)


def _fctp_bwd_inputs_raw(gy, x, attrs, weight, spec: FctpSpec, want_x: bool, want_a: bool):
    """(gx, ga): H = gy . W^T per instruction ([rows, (u, v)], one plain GEMM), then gx[n,u,k] = sum_v attrs[n,v] H and
    ga[n,v] = sum_{u,k} x[n,u,k] H in one reduction kernel.  ``x`` / ``attrs`` may be None when the gradient that needs
    them is not wanted (a zero operand stands in: the kernel always forms both)."""
    if spec.in_layout != "cf":
        raise NotImplementedError("self-connection backward expects the channel-fastest input layout")
    rows, dev = gy.shape[0], gy.device
    lib = L.load()
    if x is None:
        x = torch.zeros(rows, spec.d_in, device=dev, dtype=torch.float32)
    if attrs is None:
        attrs = torch.zeros(rows, spec.v, device=dev, dtype=torch.float32)
    gx = (torch.empty if spec.in_covered else torch.zeros)(rows, spec.d_in, device=dev, dtype=torch.float32)
    ga = torch.zeros(rows, spec.v, device=dev, dtype=torch.float32)
    hmax = max(ins.dim * ins.mul_in for ins in spec.instr) * spec.v
    H = torch.empty(rows * hmax, device=dev, dtype=torch.float32)
    seen_in = set()
    for ins in spec.instr:
        a_r2, a_k = _layout_strides(spec.out_layout, ins.mul_out, ins.dim)
        uv = ins.mul_in * spec.v
        p = L.GemmProblem()
        p.A, p.A2, p.B, p.C, p.bias = _addr(gy, ins.out_off), None, _addr(weight, ins.w_off), _addr(H), None
        p.M1, p.M2, p.N, p.K, p.V = rows, ins.dim, uv, ins.mul_out, 0
        p.accumulate = 0
        p.a_r1, p.a_r2, p.a_k = spec.d_out, a_r2, a_k
        p.b_k, p.b_n = 1, ins.mul_out  # W viewed [(u,v), w] transposed
        p.c_r1, p.c_r2, p.c_n = ins.dim * uv, uv, 1
        p.alpha = ins.alpha
        _run_gemm([p])
        L.check(
            lib.e3k_fctp_reduce_bwd(_addr(H), _addr(x, ins.in_off), _addr(attrs), rows, ins.dim, ins.mul_in, spec.v,
                                    spec.d_in, ins.mul_in, spec.v, _addr(gx, ins.in_off),
                                    1 if ins.i_in in seen_in else 0, _addr(ga), L.stream_ptr()),
            "e3k_fctp_reduce_bwd",
        )
        seen_in.add(ins.i_in)
    return (gx if want_x else None), (ga if want_a else None)


class FctpFn(torch.autograd.Function):
    """out[n, w, k] = alpha * sum_{u,v} W[u,v,w] x[n,u,k] attrs[n,v]   per instruction.

    F = <g, out(x, a, W)> is linear in each of (x, a, W, g), so every derivative of every order is the same form with
    some slots open: open g = this forward; open x = ``FctpDxFn`` (g . W^T reduced with a); open a = ``FctpDaFn``
    (g . W^T reduced with x); open W = ``FctpDwFn`` (the outer-mode weight-gradient GEMM).  The backward of each is the
    other three with the cotangent put in the slot the output stood for -- the double backward of force training runs the
    same HIP kernels (round 2 differentiated a torch restatement here)."""

    @staticmethod
    def forward(ctx, x, attrs, weight, spec: FctpSpec):
        ctx.param_slots = _param_slots(x, attrs, weight)
        L.require_cuda(x, attrs, weight)
        x, attrs, weight = L.f32c(x), L.f32c(attrs), L.f32c(weight)
        rows = x.shape[0]
        assert x.shape[1] == spec.d_in and attrs.shape == (rows, spec.v)
        y = _fctp_fwd_raw(x, attrs, weight, spec)
        ctx.save_for_backward(x, attrs, weight)
        ctx.spec = spec
        ctx.in_fork = bool(IN_FORK)
        return y

    @staticmethod
    def backward(ctx, gy):
        need = _needs(ctx)
        x, attrs, weight = ctx.saved_tensors
        spec: FctpSpec = ctx.spec
        if torch.is_grad_enabled():
            gx = FctpDxFn.apply(gy, attrs, weight, spec) if need[0] else None
            ga = FctpDaFn.apply(gy, x, weight, spec) if need[1] else None
            gw = FctpDwFn.apply(x, attrs, gy, spec, weight.numel()).view_as(weight) if need[2] else None
            return gx, ga, gw, None
        gy = L.f32c(gy)
        rows = x.shape[0]
        gx = ga = gw = None
        if need[0] or need[1]:
            gx, ga = _fctp_bwd_inputs_raw(gy, x, attrs, weight, spec, bool(need[0]), bool(need[1]))
        sunk = False
        if need[2]:
            gw = _sink_for(weight)
            sunk = gw is not None
            if not sunk:
                gw = torch.zeros_like(weight)
            if (sunk and WGRAD_SIDE and ctx.in_fork and rows >= WGRAD_SIDE_MIN_ROWS
                    and not torch.cuda.is_current_stream_capturing()):
                cur = current_stream(x.device)   # off the critical path: see StridedLinearFn.backward
                st = side_stream(x.device, 2)
                st.wait_stream(cur)
                with on_stream(st, cur):
                    _fctp_wgrad_raw(x, attrs, gy, gw, spec)
                for t_ in (x, attrs, gy):
                    t_.record_stream(st)
            else:
                _fctp_wgrad_raw(x, attrs, gy, gw, spec)
        if sunk:
            gw = None
        return gx, ga, gw, None


class FctpDxFn(torch.autograd.Function):
    """gx = dF/dx (g, a, W)."""

    @staticmethod
    def forward(ctx, gy, attrs, weight, spec: FctpSpec):
        gy, attrs, weight = L.f32c(gy), L.f32c(attrs), L.f32c(weight)
        ctx.save_for_backward(gy, attrs, weight)
        ctx.spec = spec
        return _fctp_bwd_inputs_raw(gy, None, attrs, weight, spec, True, False)[0]

    @staticmethod
    def backward(ctx, hx):      # hx ~ x
        gy, attrs, weight = ctx.saved_tensors
        spec = ctx.spec
        n = ctx.needs_input_grad
        g_g = FctpFn.apply(hx, attrs, weight, spec) if n[0] else None
        g_a = FctpDaFn.apply(gy, hx, weight, spec) if n[1] else None
        g_w = FctpDwFn.apply(hx, attrs, gy, spec, weight.numel()).view_as(weight) if n[2] else None
        return g_g, g_a, g_w, None


class FctpDaFn(torch.autograd.Function):
    """ga = dF/da (g, x, W)."""

    @staticmethod
    def forward(ctx, gy, x, weight, spec: FctpSpec):
        gy, x, weight = L.f32c(gy), L.f32c(x), L.f32c(weight)
        ctx.save_for_backward(gy, x, weight)
        ctx.spec = spec
        return _fctp_bwd_inputs_raw(gy, x, None, weight, spec, False, True)[1]

    @staticmethod
    def backward(ctx, ha):      # ha ~ attrs
        gy, x, weight = ctx.saved_tensors
        spec = ctx.spec
        n = ctx.needs_input_grad
        g_g = FctpFn.apply(x, ha, weight, spec) if n[0] else None
        g_x = FctpDxFn.apply(gy, ha, weight, spec) if n[1] else None
        g_w = FctpDwFn.apply(x, ha, gy, spec, weight.numel()).view_as(weight) if n[2] else None
        return g_g, g_x, g_w, None


class FctpDwFn(torch.autograd.Function):
    """gW = dF/dW (x, a, g)  (flat, the weight's layout)."""

    @staticmethod
    def forward(ctx, x, attrs, gy, spec: FctpSpec, w_numel: int):
        x, attrs, gy = L.f32c(x), L.f32c(attrs), L.f32c(gy)
        ctx.save_for_backward(x, attrs, gy)
        ctx.spec = spec
        gw = torch.zeros(w_numel, device=x.device, dtype=torch.float32)
        _fctp_wgrad_raw(x, attrs, gy, gw, spec)
        return gw

    @staticmethod
    def backward(ctx, hw):      # hw ~ W
        x, attrs, gy = ctx.saved_tensors
        spec = ctx.spec
        hw = hw.reshape(-1)
        n = ctx.needs_input_grad
        g_x = FctpDxFn.apply(gy, attrs, hw, spec) if n[0] else None
        g_a = FctpDaFn.apply(gy, x, hw, spec) if n[1] else None
        g_g = FctpFn.apply(x, attrs, hw, spec) if n[2] else None
        return g_x, g_a, g_g, None, None


def fctp(x, attrs, weight, spec: FctpSpec):
    return FctpFn.apply(_c(x), _c(attrs), _c(weight), spec)


# --------------------------------------------------------------------------------------
# Self-connection over *keyed* node attributes: rows of node_attrs that carry the same integer key
# (structurally identical rows, e.g. attrs = Linear(one_hot(species))) share the contracted weight
# M[t] = sum_v attrs_t[v] W[:, v, :], so the self-connection becomes one small GEMM per key group
# with K = mul_in instead of K = mul_in * V  (V = 20x fewer FLOPs for config_energy).
# --------------------------------------------------------------------------------------
@dataclass
class RowGroups:
    """Nodes grouped by key, all on the device (no host sync): ``perm`` (int32 [N], node ids sorted
    by key, stable), ``bounds`` (int32 [K, 2] = {start, count} per key), ``reps`` (int64 [K], one
    representative node per key; arbitrary for empty groups), ``n_keys``."""
    perm: torch.Tensor
    bounds: torch.Tensor
    reps: torch.Tensor
    n_keys: int


def _kw_array(spec: "FctpSpec", m_off):
    def build():
        arr = (L.KwInstr * len(spec.instr))()
        for i, (ins, mo) in enumerate(zip(spec.instr, m_off)):
            arr[i].w_off, arr[i].m_off, arr[i].u, arr[i].w_out = ins.w_off, int(mo), ins.mul_in, ins.mul_out
        return arr
    return _templates(spec, ("kw", tuple(m_off)), build)


def _kw_fwd_raw(a_rep, weight, spec: "FctpSpec", m_off, ld_m: int):
    k = a_rep.shape[0]
    m = torch.empty(k, ld_m, device=a_rep.device, dtype=torch.float32)
    L.check(L.load().e3k_keyed_weights_fwd(L.ptr(a_rep), L.ptr(weight), _kw_array(spec, m_off), len(spec.instr), k, spec.v,
                                           ld_m, L.ptr(m), L.stream_ptr()), "e3k_keyed_weights_fwd")
    return m


def _kw_bwd_raw(a_rep, weight, gm, spec: "FctpSpec", m_off, ld_m: int, want_a: bool, gw, acc: int):
    """ga [K,V] = sum_c gM[t,c] W[c,v] (needs weight, gm) and / or gw (+)= sum_t a[t,v] gM[t,c] (needs a_rep, gm); the
    operand a pass does not read may be any tensor of the right dtype."""
    lib, arr = L.load(), _kw_array(spec, m_off)
    k = gm.shape[0]
    ga = work = None
    if want_a:
        ga = torch.zeros(k, spec.v, device=gm.device, dtype=torch.float32)
        work = torch.empty(lib.e3k_keyed_weights_bwd_workspace(arr, len(spec.instr), k, spec.v), device=gm.device, dtype=torch.float32)
    L.check(lib.e3k_keyed_weights_bwd(L.ptr(a_rep), L.ptr(weight), L.ptr(gm), arr, len(spec.instr), k, spec.v, ld_m, L.ptr(ga),
                                      L.ptr(gw), acc, L.ptr(work), L.stream_ptr()), "e3k_keyed_weights_bwd")
    return ga


def _kw_weight_buffer(weight, spec: "FctpSpec"):
    covered = sum(i.mul_in * spec.v * i.mul_out for i in spec.instr) == weight.numel()
    return torch.empty_like(weight) if covered else torch.zeros_like(weight)


class KeyedWeightsFn(torch.autograd.Function):
    """M[t, (j,u,w)] = sum_v a_rep[t,v] W_j[u,v,w]: the per-key contracted self-connection weights, read straight from
    the e3nn-ordered flat weight (no permuted copies), one launch; backward two launches.  M is bilinear in (a, W): with
    F = <gM, M(a, W)> every derivative of every order is one of three kernels -- forward (leave M open), bwd_a (leave a
    open), bwd_w (leave W open) -- with operands exchanged, so the double backward of force training
    (GradientOutput, create_graph=True) runs the same HIP kernels (KwBwdAFn / KwBwdWFn below)."""

    @staticmethod
    def forward(ctx, a_rep, weight, spec: "FctpSpec", m_off: Tuple[int, ...], ld_m: int):
        ctx.param_slots = _param_slots(a_rep, weight)
        L.require_cuda(a_rep, weight)
        a_rep, weight = L.f32c(a_rep), L.f32c(weight)
        ctx.save_for_backward(a_rep, weight)
        ctx.cfg = (spec, m_off, ld_m)
        return _kw_fwd_raw(a_rep, weight, spec, m_off, ld_m)

    @staticmethod
    def backward(ctx, gm):
        need = _needs(ctx)
        a_rep, weight = ctx.saved_tensors
        spec, m_off, ld_m = ctx.cfg
        need_a, need_w = need[:2]
        if torch.is_grad_enabled():
            ga = KwBwdAFn.apply(gm, weight, spec, m_off, ld_m) if need_a else None
            gw = KwBwdWFn.apply(a_rep, gm, weight, spec, m_off, ld_m) if need_w else None
            return ga, gw, None, None, None
        gm = L.f32c(gm)
        gw = ret_w = None
        acc = 0
        if need_w:
            gw = _sink_for(weight)
            if gw is not None:
                acc = 1
            else:
                gw = ret_w = _kw_weight_buffer(weight, spec)
        ga = None
        if need_a or gw is not None:
            ga = _kw_bwd_raw(a_rep, weight, gm, spec, m_off, ld_m, bool(need_a), gw, acc)
        return ga, ret_w, None, None, None


class KwBwdAFn(torch.autograd.Function):
    """ga = dF/da (gM, W): bilinear in (gM, W)."""

    @staticmethod
    def forward(ctx, gm, weight, spec: "FctpSpec", m_off: Tuple[int, ...], ld_m: int):
        gm, weight = L.f32c(gm), L.f32c(weight)
        ctx.save_for_backward(gm, weight)
        ctx.cfg = (spec, m_off, ld_m)
        return _kw_bwd_raw(gm, weight, gm, spec, m_off, ld_m, True, None, 0)

    @staticmethod
    def backward(ctx, ha):      # ha ~ a
        gm, weight = ctx.saved_tensors
        spec, m_off, ld_m = ctx.cfg
        need_g, need_w = ctx.needs_input_grad[:2]
        g_g = KeyedWeightsFn.apply(ha, weight, spec, m_off, ld_m) if need_g else None
        g_w = KwBwdWFn.apply(ha, gm, weight, spec, m_off, ld_m) if need_w else None
        return g_g, g_w, None, None, None


class KwBwdWFn(torch.autograd.Function):
    """gw = dF/dW (a, gM): bilinear in (a, gM); ``weight`` only lends its shape."""

    @staticmethod
    def forward(ctx, a_rep, gm, weight, spec: "FctpSpec", m_off: Tuple[int, ...], ld_m: int):
        a_rep, gm = L.f32c(a_rep), L.f32c(gm)
        ctx.save_for_backward(a_rep, gm)
        ctx.cfg = (spec, m_off, ld_m)
        gw = _kw_weight_buffer(weight, spec)
        _kw_bwd_raw(a_rep, gw, gm, spec, m_off, ld_m, False, gw, 0)
        return gw

    @staticmethod
    def backward(ctx, hw):      # hw ~ W
        a_rep, gm = ctx.saved_tensors
        spec, m_off, ld_m = ctx.cfg
        need_a, need_g = ctx.needs_input_grad[:2]
        g_a = KwBwdAFn.apply(gm, hw, spec, m_off, ld_m) if need_a else None
        g_g = KeyedWeightsFn.apply(a_rep, hw, spec, m_off, ld_m) if need_g else None
        return g_a, g_g, None, None, None, None


def keyed_weights(a_rep, weight, spec: "FctpSpec", m_off: Sequence[int], ld_m: int):
    m = KeyedWeightsFn.apply(_c(a_rep), _c(weight), spec, tuple(int(v) for v in m_off), int(ld_m))
    if not a_rep.requires_grad and _param_only(weight):
        m._e3k_param_only = True     # a function of Parameters alone: inputs_only_backward() skips its gradient
    return m


def _grouped_templates(spec, m_off, mode: str):
    """One template problem per instruction with byte offsets in the pointer fields (``e3k_gemm_grouped_rebased``
    expands them over the keys inside the kernel), split into rounds so that no two problems of one launch write the
    same block; cached on the spec."""
    def build():
        tmpl = []
        for j, ins in enumerate(spec.instr):
            in_r2, in_k = _layout_strides(spec.in_layout, ins.mul_in, ins.dim)
            out_r2, out_n = _layout_strides(spec.out_layout, ins.mul_out, ins.dim)
            p = L.GemmProblem()
            p.bias, p.A2, p.V = None, None, 0
            p.M1, p.M2 = 0, ins.dim
            p.alpha = ins.alpha
            if mode == "fwd":      # y = x . M
                p.A, p.B, p.C = 4 * ins.in_off, 4 * m_off[j], 4 * ins.out_off
                p.N, p.K = ins.mul_out, ins.mul_in
                p.a_r1, p.a_r2, p.a_k = spec.d_in, in_r2, in_k
                p.b_k, p.b_n = ins.mul_out, 1
                p.c_r1, p.c_r2, p.c_n = spec.d_out, out_r2, out_n
            elif mode in ("dgrad", "dgrad_acc"):  # gx (+)= gy . M^T      (A := gy, C := gx)
                p.A, p.B, p.C = 4 * ins.out_off, 4 * m_off[j], 4 * ins.in_off
                p.N, p.K = ins.mul_in, ins.mul_out
                p.a_r1, p.a_r2, p.a_k = spec.d_out, out_r2, out_n
                p.b_k, p.b_n = 1, ins.mul_out
                p.c_r1, p.c_r2, p.c_n = spec.d_in, in_r2, in_k
            else:                  # wgrad: gm += x^T . gy   (C := gy)
                p.A, p.B, p.C = 4 * ins.in_off, 4 * m_off[j], 4 * ins.out_off
                p.N, p.K = ins.mul_out, ins.mul_in
                p.a_r1, p.a_r2, p.a_k = spec.d_in, in_r2, in_k
                p.b_k, p.b_n = ins.mul_out, 1
                p.c_r1, p.c_r2, p.c_n = spec.d_out, out_r2, out_n
            tmpl.append((ins, p))
        key = {"fwd": "i_out", "dgrad": "i_in", "dgrad_acc": "i_in", "wgrad": ""}[mode]
        seen: Dict[int, int] = {}
        rounds: List[List] = []
        for ins, p in tmpl:
            k = getattr(ins, key) if key else id(p)
            r = seen.get(k, 0)
            seen[k] = r + 1
            while len(rounds) <= r:
                rounds.append([])
            p.accumulate = 1 if (r > 0 or mode in ("wgrad", "dgrad_acc")) else 0
            rounds[r].append(p)
        if mode != "wgrad":
            rounds = _chain_rounds(rounds)
        return [((L.GemmProblem * len(g))(*g), len(g)) for g in rounds]

    return _templates(spec, ("grouped", mode, tuple(m_off)), build)


def _run_grouped(rounds, a, b, c, rows: int, groups: RowGroups, ld_m: int, wgrad: bool):
    lib, st = L.load(), L.stream_ptr()
    perm, bounds = groups.perm.data_ptr(), groups.bounds.data_ptr()
    for arr, n in rounds:
        L.check(lib.e3k_gemm_grouped_rebased(arr, n, a.data_ptr(), b.data_ptr(), c.data_ptr(), rows, perm, bounds,
                                             groups.n_keys, ld_m, int(wgrad), st), "e3k_gemm_grouped_rebased")


# ---- several descriptor arrays in one call (e3k_gemm_multi): what a convolution layer issues together -------------------
def _seg(rounds, a, b, c, rows: int, groups: Optional["RowGroups"] = None, ld_m: int = 0, bias=None):
    """One segment per round of a cached template set (``_GemmTemplates.rounds`` / ``_grouped_templates``): a list, the
    rounds of one set write the same blocks and must run in order -- callers merge only single-round sets."""
    out = []
    for arr, n in rounds:
        sg = L.GemmSegment()
        sg.templates, sg.n_templates = arr, n
        sg.a_base, sg.a2_base, sg.b_base, sg.c_base = a.data_ptr(), None, b.data_ptr(), c.data_ptr()
        sg.bias_base = bias.data_ptr() if bias is not None else None
        sg.M1 = rows
        if groups is not None:
            sg.n_keys, sg.perm, sg.groups_dev, sg.b_key_stride = groups.n_keys, groups.perm.data_ptr(), groups.bounds.data_ptr(), ld_m
        out.append(sg)
    return out


def _run_segments(segment_lists, wgrad: bool = False) -> None:
    """``segment_lists``: one list per template set (see ``_seg``).  Sets that have a single round go out in ONE call (the
    library packs up to 16 problems per launch); a multi-round set runs its rounds in order, alone."""
    lib, st = L.load(), L.stream_ptr()
    merged = [sl[0] for sl in segment_lists if len(sl) == 1]
    if merged:
        arr = (L.GemmSegment * len(merged))(*merged)
        L.check(lib.e3k_gemm_multi(arr, len(merged), int(wgrad), st), "e3k_gemm_multi")
    for sl in segment_lists:
        if len(sl) > 1:
            for sg in sl:
                arr = (L.GemmSegment * 1)(sg)
                L.check(lib.e3k_gemm_multi(arr, 1, int(wgrad), st), "e3k_gemm_multi")


def _lin_fwd_segs(x, weight, y, spec: "LinearSpec", scale: float, accumulate: bool):
    t = _templates(spec, ("fwd", scale, bool(accumulate), 0, 1.0, False),
                   lambda: _lin_fwd_templates(spec, scale, bool(accumulate), 0, 1.0, False))
    return _seg(t.rounds, x, weight, y, x.shape[0])


def _lin_dgrad_segs(gy, weight, gx, spec: "LinearSpec", scale: float, accumulate: bool):
    t = _templates(spec, ("dgrad", scale, bool(accumulate)), lambda: _lin_dgrad_templates(spec, scale, bool(accumulate)))
    return _seg(t.rounds, gy, weight, gx, gy.shape[0])


def _lin_wgrad_segs(x, gy, gw, spec: "LinearSpec", scale: float):
    t = _templates(spec, ("wgrad", scale), lambda: _lin_wgrad_templates(spec, scale))
    return _seg(t.rounds, x, gw, gy, x.shape[0])


def _grp_segs(mode: str, a, m, c, groups, spec, m_off):
    """mode 'fwd': c = a . M[key];  'dgrad' / 'dgrad_acc': c (+)= a . M[key]^T;  'wgrad': m += a^T . c (c = gradient rows)."""
    return _seg(_grouped_templates(spec, m_off, mode), a, m, c, a.shape[0], groups, m.shape[1])


def _grp_fwd_raw(x, m, groups, spec, m_off):
    rows, ld_m = x.shape[0], m.shape[1]
    # rows of absent keys do not exist, every node belongs to exactly one key: full coverage
    y = (torch.empty if spec.out_covered else torch.zeros)(rows, spec.d_out, device=x.device, dtype=torch.float32)
    _run_grouped(_grouped_templates(spec, m_off, "fwd"), x, m, y, rows, groups, ld_m, False)
    return y


def _grp_dgrad_raw(gy, m, groups, spec, m_off, out=None):
    """gx = gy . M[key]^T per instruction; ``out``: accumulate into this buffer instead."""
    rows, ld_m = gy.shape[0], m.shape[1]
    if out is not None:
        _run_grouped(_grouped_templates(spec, m_off, "dgrad_acc"), gy, m, out, rows, groups, ld_m, False)
        return out
    gx = (torch.empty if spec.in_covered else torch.zeros)(rows, spec.d_in, device=gy.device, dtype=torch.float32)
    _run_grouped(_grouped_templates(spec, m_off, "dgrad"), gy, m, gx, rows, groups, ld_m, False)
    return gx


def _grp_wgrad_raw(x, gy, m_shape, groups, spec, m_off):
    gm = torch.zeros(m_shape, device=x.device, dtype=torch.float32)
    _run_grouped(_grouped_templates(spec, m_off, "wgrad"), x, gm, gy, x.shape[0], groups, m_shape[1], True)
    return gm


class GroupedLinearFn(torch.autograd.Function):
    """out[n, w, k] = alpha * sum_u M[key(n)][u, w] x[n, u, k] per instruction; M is
    [K, sum_j U_j * W_j] with instruction j's block at column offset ``m_off[j]``.  Bilinear in
    (x, M): forward / dgrad / wgrad are closed under differentiation (double backward)."""

    @staticmethod
    def forward(ctx, x, m, groups: RowGroups, spec: "FctpSpec", m_off: Tuple[int, ...]):
        ctx.param_slots = _param_slots(x, m)
        L.require_cuda(x, m)
        x, m = L.f32c(x), L.f32c(m)
        ctx.save_for_backward(x, m)
        ctx.groups, ctx.spec, ctx.m_off = groups, spec, m_off
        return _grp_fwd_raw(x, m, groups, spec, m_off)

    @staticmethod
    def backward(ctx, gy):
        need = _needs(ctx)
        x, m = ctx.saved_tensors
        a = (ctx.groups, ctx.spec, ctx.m_off)
        if torch.is_grad_enabled():
            gx = GroupedDgradFn.apply(gy, m, *a) if need[0] else None
            gm = GroupedWgradFn.apply(x, gy, tuple(m.shape), *a) if need[1] else None
            return gx, gm, None, None, None
        gy = L.f32c(gy)
        gx = _grp_dgrad_raw(gy, m, *a) if need[0] else None
        gm = _grp_wgrad_raw(x, gy, tuple(m.shape), *a) if need[1] else None
        return gx, gm, None, None, None


class GroupedDgradFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gy, m, groups, spec, m_off):
        gy, m = L.f32c(gy), L.f32c(m)
        ctx.save_for_backward(gy, m)
        ctx.a = (groups, spec, m_off)
        return _grp_dgrad_raw(gy, m, groups, spec, m_off)

    @staticmethod
    def backward(ctx, h):
        gy, m = ctx.saved_tensors
        g_gy = GroupedLinearFn.apply(h, m, *ctx.a) if ctx.needs_input_grad[0] else None
        g_m = GroupedWgradFn.apply(h, gy, tuple(m.shape), *ctx.a) if ctx.needs_input_grad[1] else None
        return g_gy, g_m, None, None, None


class GroupedWgradFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gy, m_shape, groups, spec, m_off):
        x, gy = L.f32c(x), L.f32c(gy)
        ctx.save_for_backward(x, gy)
        ctx.a = (groups, spec, m_off)
        return _grp_wgrad_raw(x, gy, m_shape, groups, spec, m_off)

    @staticmethod
    def backward(ctx, h):
        x, gy = ctx.saved_tensors
        g_x = GroupedDgradFn.apply(gy, h, *ctx.a) if ctx.needs_input_grad[0] else None
        g_gy = GroupedLinearFn.apply(x, h, *ctx.a) if ctx.needs_input_grad[1] else None
        return g_x, g_gy, None, None, None, None


def grouped_linear(x, m, groups: RowGroups, spec: "FctpSpec", m_off: Sequence[int]):
    return GroupedLinearFn.apply(_c(x), _c(m), groups, spec, tuple(int(v) for v in m_off))


# --------------------------------------------------------------------------------------
# fused uvu tensor product + destination reduce
# --------------------------------------------------------------------------------------
class TpPlan:
    """Owns an ``e3k_tp_plan`` (device copies of the group table)."""

    def __init__(self, groups: Sequence[L.TpGroup], d_in: int, d_sh: int, w_numel: int, d_mid: int):
        self.groups = list(groups)
        self.d_in, self.d_sh, self.w_numel, self.d_mid = d_in, d_sh, w_numel, d_mid
        self._handles: Dict[int, int] = {}  # device index -> plan pointer
        self._bwd_x_overwrites: Optional[bool] = None

    def handle(self, device: torch.device) -> int:
        idx = device.index if device.index is not None else torch.cuda.current_device()
        h = self._handles.get(idx)
        if h is None:
            lib = L.load()
            arr = (L.TpGroup * len(self.groups))(*self.groups)
            out = C.c_void_p()
            with torch.cuda.device(idx):
                L.check(lib.e3k_tp_plan_create(arr, len(self.groups), self.d_in, self.d_sh, self.w_numel, self.d_mid,
                                               C.byref(out)), "e3k_tp_plan_create")
            h = out.value
            self._handles[idx] = h
        return h

    def bwd_x_overwrites(self, device: torch.device) -> bool:
        """g_x needs no zero-fill: the backward w.r.t. x stores every element (see e3k_tp_bwd_x_overwrites)."""
        if self._bwd_x_overwrites is None:
            self._bwd_x_overwrites = bool(L.load().e3k_tp_bwd_x_overwrites(self.handle(device)))
        return self._bwd_x_overwrites

    def __del__(self):
        try:
            lib = L.load()
            for h in self._handles.values():
                lib.e3k_tp_plan_destroy(h)
        except Exception:
            pass


def _tp_fwd_raw(x, sh, w, topo: GraphTopo, plan: TpPlan):
    n, e = x.shape[0], sh.shape[0]
    assert x.shape[1] == plan.d_in and sh.shape[1] == plan.d_sh and w.shape == (e, plan.w_numel)
    assert topo.num_nodes == n and topo.num_edges == e
    out = torch.empty(n, plan.d_mid, device=x.device, dtype=torch.float32)
    handle = plan.handle(x.device)
    with timed_launch("tp_fwd", (n, e, plan)):
        L.check(L.load().e3k_tp_fwd(handle, L.ptr(x), L.ptr(sh), L.ptr(w), L.ptr(topo.src), L.ptr(topo.dst_ptr),
                                    L.ptr(topo.dst_perm), n, e, L.ptr(out), L.stream_ptr()), "e3k_tp_fwd")
    return out


def _tp_bwd_x_raw(sh, w, g_out, topo: GraphTopo, plan: TpPlan):
    n, e = topo.num_nodes, topo.num_edges
    gx = (torch.empty if plan.bwd_x_overwrites(sh.device) else torch.zeros)(n, plan.d_in, device=sh.device, dtype=torch.float32)
    handle = plan.handle(sh.device)
    with timed_launch("tp_bwd_x", (n, e, plan)):
        L.check(L.load().e3k_tp_bwd_x(handle, L.ptr(sh), L.ptr(w), L.ptr(g_out), L.ptr(topo.dst),
                                      L.ptr(topo.src_ptr), L.ptr(topo.src_perm), n, e, L.ptr(gx), L.stream_ptr()), "e3k_tp_bwd_x")
    return gx


def _tp_bwd_xw_raw(x, sh, w, g_out, topo: GraphTopo, plan: TpPlan):
    """(g_x [N, d_in], g_w [E, W]) in ONE walk of the source CSR (channel-complete plans: ``e3k_tp_table_supported``)"""
    n, e = topo.num_nodes, topo.num_edges
    gx = (torch.empty if plan.bwd_x_overwrites(sh.device) else torch.zeros)(n, plan.d_in, device=sh.device, dtype=torch.float32)
    gw = torch.empty(e, plan.w_numel, device=sh.device, dtype=torch.float32)
    with timed_launch("tp_bwd_x", (n, e, plan)):
        L.check(L.load().e3k_tp_bwd_xw(plan.handle(sh.device), L.ptr(x), L.ptr(sh), L.ptr(w), L.ptr(g_out), L.ptr(topo.dst),
                                       L.ptr(topo.src_ptr), L.ptr(topo.src_perm), n, e, L.ptr(gx), L.ptr(gw), L.stream_ptr()),
                "e3k_tp_bwd_xw")
    return gx, gw


def _tp_bwd_w_raw(x, sh, w, g_out, topo: GraphTopo, plan: TpPlan, want_sh: bool, want_w: bool = True):
    n, e = topo.num_nodes, topo.num_edges
    assert want_w or want_sh
    gw = torch.empty(e, plan.w_numel, device=x.device, dtype=torch.float32) if want_w else None
    gsh = torch.zeros(e, plan.d_sh, device=x.device, dtype=torch.float32) if want_sh else None
    handle = plan.handle(x.device)
    with timed_launch("tp_bwd_w_sh" if want_sh else "tp_bwd_w", (n, e, plan)):
        L.check(L.load().e3k_tp_bwd_w(handle, L.ptr(x), L.ptr(sh), L.ptr(w), L.ptr(g_out), L.ptr(topo.src),
                                      L.ptr(topo.dst_ptr), L.ptr(topo.dst_perm), n, e, L.ptr(gw), L.ptr(gsh),
                                      L.stream_ptr()), "e3k_tp_bwd_w")
    return gw, gsh


class TpFn(torch.autograd.Function):
    """out[n] = sum over edges into n of TP(x[src], sh, w): trilinear in (x, sh, w).  With
    F = <g, TP(x, sh, w)> every derivative of every order is one of three kernels — forward (contract
    nothing), bwd_x (leave x open), bwd_w (leave w and sh open) — with operands exchanged.
    ``sh_data``: sh is a function of the input data alone (no gradient under ``params_only_backward``)."""

    @staticmethod
    def forward(ctx, x, sh, w, topo: GraphTopo, plan: TpPlan, sh_data: bool = False):
        L.require_cuda(x, sh, w)
        x, sh, w = L.f32c(x), L.f32c(sh), L.f32c(w)
        ctx.save_for_backward(x, sh, w)
        ctx.topo, ctx.plan, ctx.sh_data = topo, plan, sh_data
        return _tp_fwd_raw(x, sh, w, topo, plan)

    @staticmethod
    def backward(ctx, g_out):
        x, sh, w = ctx.saved_tensors
        topo, plan = ctx.topo, ctx.plan
        need_x, need_sh, need_w = ctx.needs_input_grad[:3]
        need_sh = need_sh and not (PARAMS_ONLY and ctx.sh_data)
        if torch.is_grad_enabled():
            gx = TpBwdXFn.apply(sh, w, g_out, topo, plan, ctx.sh_data) if need_x else None
            gw = gsh = None
            if need_sh or need_w:
                gw, gsh = TpBwdWFn.apply(x, sh, w, g_out, topo, plan, bool(need_sh), bool(need_w), ctx.sh_data)
            return gx, (gsh if need_sh else None), (gw if need_w else None), None, None, None
        g_out = L.f32c(g_out)
        gx = gsh = gw = None
        if need_x:
            gx = _tp_bwd_x_raw(sh, w, g_out, topo, plan)
        if need_sh or need_w:
            gw, gsh = _tp_bwd_w_raw(x, sh, w, g_out, topo, plan, bool(need_sh), bool(need_w))
        return gx, gsh, gw, None, None, None


class TpBwdXFn(torch.autograd.Function):
    """gx = dF/dx (sh, w, g)."""

    @staticmethod
    def forward(ctx, sh, w, g_out, topo: GraphTopo, plan: TpPlan, sh_data: bool = False):
        sh, w, g_out = L.f32c(sh), L.f32c(w), L.f32c(g_out)
        ctx.save_for_backward(sh, w, g_out)
        ctx.topo, ctx.plan, ctx.sh_data = topo, plan, sh_data
        return _tp_bwd_x_raw(sh, w, g_out, topo, plan)

    @staticmethod
    def backward(ctx, h):   # h ~ x
        sh, w, g_out = ctx.saved_tensors
        topo, plan = ctx.topo, ctx.plan
        need_sh, need_w, need_g = ctx.needs_input_grad[:3]
        need_sh = need_sh and not (PARAMS_ONLY and ctx.sh_data)
        g_g = TpFn.apply(h, sh, w, topo, plan, ctx.sh_data) if need_g else None
        g_w = g_sh = None
        if need_sh or need_w:
            g_w, g_sh = TpBwdWFn.apply(h, sh, w, g_out, topo, plan, bool(need_sh), bool(need_w), ctx.sh_data)
        return (g_sh if need_sh else None), (g_w if need_w else None), g_g, None, None, None


class TpBwdWFn(torch.autograd.Function):
    """(gw, gsh) = (dF/dw, dF/dsh) (x, sh, w, g);  gw does not depend on w, gsh does not depend on sh."""

    @staticmethod
    def forward(ctx, x, sh, w, g_out, topo: GraphTopo, plan: TpPlan, want_sh: bool, want_w: bool = True,
                sh_data: bool = False):
        x, sh, w, g_out = L.f32c(x), L.f32c(sh), L.f32c(w), L.f32c(g_out)
        ctx.save_for_backward(x, sh, w, g_out)
        ctx.topo, ctx.plan, ctx.want_sh, ctx.want_w, ctx.sh_data = topo, plan, want_sh, want_w, sh_data
        ctx.set_materialize_grads(False)
        gw, gsh = _tp_bwd_w_raw(x, sh, w, g_out, topo, plan, want_sh, want_w)
        dead = []
        if gsh is None:
            gsh = torch.zeros(0, device=x.device)
            dead.append(gsh)
        if gw is None:
            gw = torch.zeros(0, device=x.device)
            dead.append(gw)
        if dead:
            ctx.mark_non_differentiable(*dead)
        return gw, gsh

    @staticmethod
    def backward(ctx, hw, hsh):   # hw ~ w, hsh ~ sh
        x, sh, w, g_out = ctx.saved_tensors
        topo, plan = ctx.topo, ctx.plan
        need_x, need_sh, need_w, need_g = ctx.needs_input_grad[:4]
        need_sh = need_sh and not (PARAMS_ONLY and ctx.sh_data)
        if not ctx.want_sh:
            hsh = None
        if not ctx.want_w:
            hw = None
        t = (topo, plan)
        g_x = g_sh = g_w = g_g = None

        def acc(a, b):
            return b if a is None else a + b
        if hw is not None:      # term <g, TP(x, sh, hw)>
            if need_x:
                g_x = acc(g_x, TpBwdXFn.apply(sh, hw, g_out, *t, ctx.sh_data))
            if need_g:
                g_g = acc(g_g, TpFn.apply(x, sh, hw, *t, ctx.sh_data))
            if need_sh:     # only the sh half of the kernel: no [E, W] gradient is written
                g_sh = acc(g_sh, TpBwdWFn.apply(x, sh, hw, g_out, *t, True, False, ctx.sh_data)[1])
        if hsh is not None:     # term <g, TP(x, hsh, w)>
            if need_x:
                g_x = acc(g_x, TpBwdXFn.apply(hsh, w, g_out, *t))
            if need_g:
                g_g = acc(g_g, TpFn.apply(x, hsh, w, *t))
            if need_w:
                g_w = acc(g_w, TpBwdWFn.apply(x, hsh, w, g_out, *t, False)[0])
        return g_x, g_sh, g_w, g_g, None, None, None, None, None


def tp_uvu_scatter(x, sh, w, topo: GraphTopo, plan: TpPlan):
    return TpFn.apply(_c(x), _c(sh), _c(w), topo, plan, is_data_only(sh))


# --------------------------------------------------------------------------------------
# elementwise / node-side ops
# --------------------------------------------------------------------------------------
class ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act_id: int, cst: float):
        L.require_cuda(x)
        x = L.f32c(x)
        y = torch.empty_like(x)
        L.check(L.load().e3k_act_fwd(L.ptr(x), x.numel(), act_id, cst, L.ptr(y), L.stream_ptr()), "e3k_act_fwd")
        ctx.save_for_backward(x)
        ctx.act_id, ctx.cst = act_id, cst
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        if torch.is_grad_enabled():
            return ActBwdFn.apply(x, gy, ctx.act_id, ctx.cst), None, None
        return _act_bwd_raw(x, L.f32c(gy), ctx.act_id, ctx.cst), None, None


def _act_bwd_raw(x, gy, act_id, cst):
    gx = torch.empty_like(x)
    L.check(L.load().e3k_act_bwd(L.ptr(x), L.ptr(gy), x.numel(), act_id, cst, L.ptr(gx), L.stream_ptr()), "e3k_act_bwd")
    return gx


class ActBwdFn(torch.autograd.Function):
    """gx = gy * cst * act'(x), differentiable once more (e3k_act_bwd2: act'')."""

    @staticmethod
    def forward(ctx, x, gy, act_id: int, cst: float):
        x, gy = L.f32c(x), L.f32c(gy)
        ctx.save_for_backward(x, gy)
        ctx.act_id, ctx.cst = act_id, cst
        return _act_bwd_raw(x, gy, act_id, cst)

    @staticmethod
    @once_differentiable
    def backward(ctx, h):
        x, gy = ctx.saved_tensors
        h = L.f32c(h)
        g_x = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        g_gy = torch.empty_like(x) if ctx.needs_input_grad[1] else None
        if g_x is not None or g_gy is not None:
            L.check(L.load().e3k_act_bwd2(L.ptr(x), L.ptr(gy), L.ptr(h), x.numel(), ctx.act_id, ctx.cst, L.ptr(g_gy),
                                          L.ptr(g_x), L.stream_ptr()), "e3k_act_bwd2")
        return g_x, g_gy, None, None


def activation(x, name: str, cst: float):
    return ActFn.apply(_c(x), ACT_IDS[name], float(cst))


_BLOCK_ARRAYS: Dict[Tuple, object] = {}


def _blocks(blocks: Sequence[Tuple[int, int, int]]):
    key = tuple(blocks)
    arr = _BLOCK_ARRAYS.get(key)
    if arr is None:
        arr = (L.Block * max(len(blocks), 1))()
        for i, (off, mul, dim) in enumerate(blocks):
            arr[i].off, arr[i].mul, arr[i].dim = off, mul, dim
        _BLOCK_ARRAYS[key] = arr
    return arr


def _relayout_raw(x, blocks, to_cf: bool):
    blocks = tuple(b for b in blocks if b[1] > 1 and b[2] > 1)
    if not blocks:
        return x
    y = torch.empty_like(x)
    L.check(L.load().e3k_relayout(L.ptr(x), x.shape[0], x.shape[1], _blocks(blocks), len(blocks), int(to_cf), L.ptr(y),
                                  L.stream_ptr()), "e3k_relayout")
    return y


class RelayoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, blocks, to_cf: bool):
        L.require_cuda(x)
        x = L.f32c(x)
        y = torch.empty_like(x)
        L.check(L.load().e3k_relayout(L.ptr(x), x.shape[0], x.shape[1], _blocks(blocks), len(blocks), int(to_cf), L.ptr(y),
                                      L.stream_ptr()), "e3k_relayout")
        ctx.blocks, ctx.to_cf = blocks, to_cf
        return y

    @staticmethod
    def backward(ctx, gy):   # a permutation: its backward is the inverse permutation (any order of derivative)
        return RelayoutFn.apply(gy, ctx.blocks, not ctx.to_cf), None, None


def relayout(x, blocks: Sequence[Tuple[int, int, int]], to_cf: bool):
    """blocks: (offset, mul, 2l+1) of the feature row; blocks with dim 1 or mul 1 are no-ops."""
    blocks = tuple(b for b in blocks if b[1] > 1 and b[2] > 1)
    if not blocks:
        return x
    return RelayoutFn.apply(_c(x), blocks, to_cf)


@dataclass
class GateSpec:
    in_dim: int
    out_dim: int
    segs: List[Tuple[int, int, int, int, int, int, int, float]]  # kind,in_off,gate_off,out_off,mul,dim,act,cst

    def c_array(self):
        arr = self.__dict__.get("_arr")
        if arr is None:
            arr = (L.GateSeg * len(self.segs))()
            for i, s in enumerate(self.segs):
                (arr[i].kind, arr[i].in_off, arr[i].gate_off, arr[i].out_off, arr[i].mul, arr[i].dim, arr[i].act, arr[i].cst) = s
            self.__dict__["_arr"] = arr
        return arr


class GateFn(torch.autograd.Function):
    """x (channel-fastest) -> gated output in the e3nn layout, or (``out_cf``) in the channel-fastest layout: consecutive
    MessagePassing layers hand their features over in cf, so neither direction pays a relayout between them."""

    @staticmethod
    def forward(ctx, x, spec: GateSpec, out_cf: bool = False):
        L.require_cuda(x)
        x = L.f32c(x)
        assert x.shape[1] == spec.in_dim
        y = torch.empty(x.shape[0], spec.out_dim, device=x.device, dtype=torch.float32)
        L.check(L.load().e3k_gate_fwd(L.ptr(x), x.shape[0], spec.in_dim, spec.out_dim, spec.c_array(), len(spec.segs),
                                      int(out_cf), L.ptr(y), L.stream_ptr()), "e3k_gate_fwd")
        ctx.save_for_backward(x)
        ctx.spec, ctx.out_cf = spec, bool(out_cf)
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        if torch.is_grad_enabled():
            return GateBwdFn.apply(x, gy, ctx.spec, ctx.out_cf), None, None
        return _gate_bwd_raw(x, L.f32c(gy), ctx.spec, ctx.out_cf), None, None


def _gate_fwd_raw(x, spec: GateSpec, out_cf: bool = False):
    y = torch.empty(x.shape[0], spec.out_dim, device=x.device, dtype=torch.float32)
    L.check(L.load().e3k_gate_fwd(L.ptr(x), x.shape[0], spec.in_dim, spec.out_dim, spec.c_array(), len(spec.segs),
                                  int(out_cf), L.ptr(y), L.stream_ptr()), "e3k_gate_fwd")
    return y


def _gate_bwd_raw(x, gy, spec: GateSpec, out_cf: bool = False, gy2=None):
    gx = torch.empty_like(x)
    L.check(L.load().e3k_gate_bwd(L.ptr(x), L.ptr(gy), L.ptr(gy2), x.shape[0], spec.in_dim, spec.out_dim, spec.c_array(),
                                  len(spec.segs), int(out_cf), L.ptr(gx), L.stream_ptr()), "e3k_gate_bwd")
    return gx


class GateBwdFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gy, spec: GateSpec, out_cf: bool = False):
        x, gy = L.f32c(x), L.f32c(gy)
        ctx.save_for_backward(x, gy)
        ctx.spec, ctx.out_cf = spec, bool(out_cf)
        return _gate_bwd_raw(x, gy, spec, out_cf)

    @staticmethod
    @once_differentiable
    def backward(ctx, h):
        x, gy = ctx.saved_tensors
        spec = ctx.spec
        h = L.f32c(h)
        g_x = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        g_gy = torch.empty_like(gy) if ctx.needs_input_grad[1] else None
        if g_x is not None or g_gy is not None:
            L.check(L.load().e3k_gate_bwd2(L.ptr(x), L.ptr(gy), L.ptr(h), x.shape[0], spec.in_dim, spec.out_dim,
                                           spec.c_array(), len(spec.segs), int(ctx.out_cf), L.ptr(g_gy), L.ptr(g_x),
                                           L.stream_ptr()), "e3k_gate_bwd2")
        return g_x, g_gy, None, None


def gate(x, spec: GateSpec, out_cf: bool = False):
    return GateFn.apply(_c(x), spec, bool(out_cf))


class NormActFn(torch.autograd.Function):
    """e3nn.nn.NormActivation: x (channel-fastest) -> x * act(|x|) / |x| per irrep channel, e3nn layout out."""

    @staticmethod
    def forward(ctx, x, blocks, act_id: int, eps: float, normalize: bool):
        L.require_cuda(x)
        x = L.f32c(x)
        y = torch.empty_like(x)
        L.check(L.load().e3k_norm_act_fwd(L.ptr(x), x.shape[0], x.shape[1], _blocks(blocks), len(blocks), act_id, eps,
                                          int(normalize), L.ptr(y), L.stream_ptr()), "e3k_norm_act_fwd")
        ctx.save_for_backward(x)
        ctx.cfg = (blocks, act_id, eps, normalize)
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        blocks, act_id, eps, normalize = ctx.cfg
        if torch.is_grad_enabled():   # double backward (force training): e3k_norm_act_bwd2
            return NormActBwdFn.apply(x, gy, blocks, act_id, eps, normalize), None, None, None, None
        return _norm_act_bwd_raw(x, gy, blocks, act_id, eps, normalize), None, None, None, None


def _norm_act_bwd_raw(x, gy, blocks, act_id, eps, normalize):
    gy = L.f32c(gy)
    gx = torch.empty_like(x)
    L.check(L.load().e3k_norm_act_bwd(L.ptr(x), L.ptr(gy), x.shape[0], x.shape[1], _blocks(blocks), len(blocks), act_id,
                                      eps, int(normalize), L.ptr(gx), L.stream_ptr()), "e3k_norm_act_bwd")
    return gx


class NormActBwdFn(torch.autograd.Function):
    """The first backward of NormActivation as a differentiable op (round 2 differentiated a torch restatement here)."""

    @staticmethod
    def forward(ctx, x, gy, blocks, act_id: int, eps: float, normalize: bool):
        x, gy = L.f32c(x), L.f32c(gy)
        ctx.save_for_backward(x, gy)
        ctx.cfg = (blocks, act_id, eps, normalize)
        return _norm_act_bwd_raw(x, gy, blocks, act_id, eps, normalize)

    @staticmethod
    @once_differentiable
    def backward(ctx, h):
        x, gy = ctx.saved_tensors
        blocks, act_id, eps, normalize = ctx.cfg
        h = L.f32c(h)
        g_x = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        g_gy = torch.empty_like(gy) if ctx.needs_input_grad[1] else None
        if g_x is not None or g_gy is not None:
            L.check(L.load().e3k_norm_act_bwd2(L.ptr(x), L.ptr(gy), L.ptr(h), x.shape[0], x.shape[1], _blocks(blocks), len(blocks),
                                               act_id, eps, int(normalize), L.ptr(g_gy), L.ptr(g_x), L.stream_ptr()),
                    "e3k_norm_act_bwd2")
        return g_x, g_gy, None, None, None, None


def norm_activation(x, blocks, act: str, eps: float, normalize: bool):
    return NormActFn.apply(_c(x), tuple(blocks), ACT_IDS[act], float(eps), bool(normalize))


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, std, blocks):
        L.require_cuda(x, std)
        x, std = L.f32c(x), L.f32c(std)
        y = torch.empty_like(x)
        inv = torch.empty(x.shape[0], len(blocks), device=x.device, dtype=torch.float32)
        L.check(L.load().e3k_layernorm_fwd(L.ptr(x), x.shape[0], x.shape[1], _blocks(blocks), len(blocks), L.ptr(std),
                                           L.ptr(y), L.ptr(inv), L.stream_ptr()), "e3k_layernorm_fwd")
        ctx.save_for_backward(x, std, inv)
        ctx.blocks = blocks
        return y

    @staticmethod
    def backward(ctx, gy):
        x, std, inv = ctx.saved_tensors
        if torch.is_grad_enabled():      # double backward (force training): e3k_layernorm_bwd2
            gx, gstd = LayerNormBwdFn.apply(x, std, inv, gy, ctx.blocks)
            return (gx if ctx.needs_input_grad[0] else None), (gstd if ctx.needs_input_grad[1] else None), None
        gy = L.f32c(gy)
        gx = torch.empty_like(x)
        gstd = torch.zeros_like(std)
        L.check(L.load().e3k_layernorm_bwd(L.ptr(x), L.ptr(gy), L.ptr(inv), x.shape[0], x.shape[1], _blocks(ctx.blocks),
                                           len(ctx.blocks), L.ptr(std), L.ptr(gx), L.ptr(gstd), L.stream_ptr()),
                "e3k_layernorm_bwd")
        return gx, gstd, None


def _layer_norm_bwd_raw(x, std, inv, gy, blocks):
    gx = torch.empty_like(x) if _blocks_cover(blocks, x.shape[1]) else torch.zeros_like(x)
    gstd = torch.zeros_like(std)
    L.check(L.load().e3k_layernorm_bwd(L.ptr(x), L.ptr(gy), L.ptr(inv), x.shape[0], x.shape[1], _blocks(blocks), len(blocks),
                                       L.ptr(std), L.ptr(gx), L.ptr(gstd), L.stream_ptr()), "e3k_layernorm_bwd")
    return gx, gstd


def _blocks_cover(blocks, row_dim: int) -> bool:
    return sum(mul * dim for _, mul, dim in blocks) == row_dim


class LayerNormBwdFn(torch.autograd.Function):
    """(g_x, g_std) of the normalisation as a differentiable op; its backward is e3k_layernorm_bwd2."""

    @staticmethod
    def forward(ctx, x, std, inv, gy, blocks):
        x, std, gy = L.f32c(x), L.f32c(std), L.f32c(gy)
        ctx.save_for_backward(x, std, inv, gy)
        ctx.blocks = blocks
        return _layer_norm_bwd_raw(x, std, inv, gy, blocks)

    @staticmethod
    @once_differentiable
    def backward(ctx, h, hs):
        x, std, inv, gy = ctx.saved_tensors
        blocks = ctx.blocks
        need_x, need_std, _, need_g = ctx.needs_input_grad[:4]
        if h is None:
            h = torch.zeros_like(x)
        h = L.f32c(h)
        hs = L.f32c(hs) if hs is not None else None
        cover = _blocks_cover(blocks, x.shape[1])
        alloc = torch.empty_like if cover else torch.zeros_like
        g_x = alloc(x) if need_x else None
        g_g = alloc(x) if need_g else None
        g_s = torch.zeros_like(std) if need_std else None
        if g_x is not None or g_g is not None or g_s is not None:
            L.check(L.load().e3k_layernorm_bwd2(L.ptr(x), L.ptr(gy), L.ptr(h), L.ptr(hs), L.ptr(inv), x.shape[0], x.shape[1],
                                                _blocks(blocks), len(blocks), L.ptr(std), L.ptr(g_g), L.ptr(g_x), L.ptr(g_s),
                                                L.stream_ptr()), "e3k_layernorm_bwd2")
        return g_x, g_s, None, g_g, None


def layer_norm(x, std, blocks):
    return LayerNormFn.apply(_c(x), _c(std), tuple(blocks))


class SegmentSumFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ptr, seg_index, mean: bool):
        L.require_cuda(x)
        x = L.f32c(x)
        n_seg = ptr.numel() - 1
        out = torch.empty(n_seg, x.shape[1], device=x.device, dtype=torch.float32)
        L.check(L.load().e3k_segment_sum(L.ptr(x), L.ptr(ptr), n_seg, x.shape[1], int(mean), L.ptr(out), L.stream_ptr()),
                "e3k_segment_sum")
        ctx.save_for_backward(ptr, seg_index)
        ctx.mean = mean
        return out

    @staticmethod
    def backward(ctx, g):   # torch ops only: differentiable to any order
        ptr, seg_index = ctx.saved_tensors
        if ctx.mean:
            cnt = (ptr[1:] - ptr[:-1]).clamp(min=1).to(g.dtype).view(-1, 1)
            g = g / cnt
        return g.index_select(0, seg_index), None, None, None


class SqErrorFn(torch.autograd.Function):
    """scale * sum_i w_i (pred_i - target_i)^2 (w = 1 / n without weights: scale * mse_loss) -- loss and gradient in one launch."""

    @staticmethod
    def forward(ctx, pred, target, weight, scale: float):
        L.require_cuda(pred, target)
        p, t = L.f32c(pred).reshape(-1), L.f32c(target).reshape(-1)
        w = None if weight is None else L.f32c(weight).reshape(-1)
        assert p.numel() == t.numel() and (w is None or (w.numel() > 0 and p.numel() % w.numel() == 0))
        loss = torch.empty((), device=p.device, dtype=torch.float32)
        grad = torch.empty_like(p)
        L.check(L.load().e3k_sq_error(L.ptr(p), L.ptr(t), L.ptr(w), 1 if w is None else p.numel() // w.numel(), p.numel(), float(scale),
                                      L.ptr(loss), L.ptr(grad), L.stream_ptr()), "e3k_sq_error")
        ctx.save_for_backward(grad)
        ctx.shape = pred.shape
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        unit = _UNIT.get(g.device.index)
        if unit is not None and g.data_ptr() == unit.data_ptr():
            return grad.view(ctx.shape), None, None, None      # (the loss's own upstream gradient, the constant 1: nothing to multiply)
        return (grad * g).view(ctx.shape), None, None, None


_UNIT: Dict[int, torch.Tensor] = {}


def unit_gradient(loss: torch.Tensor) -> torch.Tensor:
    """A persistent scalar 1.0 on ``loss``'s device for ``loss.backward(gradient=ops.unit_gradient(loss))``: autograd otherwise makes
    a ``ones_like(loss)`` per backward (a fill launch) and the loss nodes multiply by it (another): two launch-sized kernels per
    step that a replayed graph pays for every time.  ``SqErrorFn`` recognises this tensor and skips the multiplication.  Allocated
    on first use -- outside a capture (``CapturedStep``'s eager warm-up sees to that)."""
    idx = loss.device.index if loss.device.index is not None else torch.cuda.current_device()
    unit = _UNIT.get(idx)
    if unit is None:
        if torch.cuda.is_current_stream_capturing():
            return torch.ones((), device=loss.device, dtype=loss.dtype)
        unit = _UNIT[idx] = torch.ones((), device=loss.device, dtype=torch.float32)
    return unit if loss.dtype == torch.float32 and loss.dim() == 0 else torch.ones_like(loss)


def sq_error(pred, target, weight=None, scale: float = 1.0):
    """``scale * (weight * (pred - target) ** 2).sum()`` (``weight`` per entry, or per ROW of ``pred``), or
    ``scale * mse_loss(pred, target)`` without weights (first order only: a loss term is differentiated once)."""
    return SqErrorFn.apply(pred, target, weight, float(scale))


def segment_sum(x, ptr, seg_index, mean=False):
    return SegmentSumFn.apply(_c(x), ptr, seg_index, bool(mean))


# --------------------------------------------------------------------------------------
# edge geometry
# --------------------------------------------------------------------------------------
class EdgeVectorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pos, topo: GraphTopo):
        L.require_cuda(pos)
        pos = L.f32c(pos)
        e = topo.num_edges
        vec = torch.empty(e, 3, device=pos.device, dtype=torch.float32)
        length = torch.empty(e, device=pos.device, dtype=torch.float32)
        L.check(L.load().e3k_edge_vector_fwd(L.ptr(pos), L.ptr(topo.src), L.ptr(topo.dst), e, L.ptr(vec), L.ptr(length),
                                             L.stream_ptr()), "e3k_edge_vector_fwd")
        ctx.save_for_backward(vec, length)
        ctx.topo, ctx.n = topo, pos.shape[0]
        return vec, length

    @staticmethod
    def backward(ctx, g_vec, g_len):
        vec, length = ctx.saved_tensors
        topo = ctx.topo
        if torch.is_grad_enabled():
            # d len = vec/len . d vec (elementwise torch ops), then the linear scatter dst(+) / src(-)
            gv = g_vec
            if g_len is not None:
                f = torch.where(length > 0, g_len / length.clamp_min(1e-30), torch.zeros_like(length))
                gl = f.unsqueeze(1) * vec
                gv = gl if gv is None else gv + gl
            return EdgeScatterFn.apply(gv, topo, ctx.n), None
        g_vec = L.f32c(g_vec) if g_vec is not None else None
        g_len = L.f32c(g_len) if g_len is not None else None
        g_pos = torch.empty(ctx.n, 3, device=vec.device, dtype=torch.float32)
        L.check(L.load().e3k_edge_vector_bwd(L.ptr(g_vec), L.ptr(g_len), L.ptr(vec), L.ptr(length), L.ptr(topo.dst_ptr),
                                             L.ptr(topo.dst_perm), L.ptr(topo.src_ptr), L.ptr(topo.src_perm), ctx.n,
                                             L.ptr(g_pos), L.stream_ptr()), "e3k_edge_vector_bwd")
        return g_pos, None


class EdgeScatterFn(torch.autograd.Function):
    """g_pos[n] = sum_{dst(e)=n} gv[e] - sum_{src(e)=n} gv[e]  (adjoint of the edge-vector gather)."""

    @staticmethod
    def forward(ctx, gv, topo: GraphTopo, n: int):
        gv = L.f32c(gv)
        g_pos = torch.empty(n, 3, device=gv.device, dtype=torch.float32)
        L.check(L.load().e3k_edge_vector_bwd(L.ptr(gv), None, None, None, L.ptr(topo.dst_ptr), L.ptr(topo.dst_perm),
                                             L.ptr(topo.src_ptr), L.ptr(topo.src_perm), n, L.ptr(g_pos),
                                             L.stream_ptr()), "e3k_edge_vector_bwd")
        ctx.topo = topo
        return g_pos

    @staticmethod
    def backward(ctx, h):   # h ~ pos: gather h[dst] - h[src]
        return EdgeVectorFn.apply(h, ctx.topo)[0], None, None


def edge_vector(pos, topo: GraphTopo):
    return EdgeVectorFn.apply(_c(pos), topo)


class SphHarmFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vec, ls: Tuple[int, ...], normalize: bool, normalization: int):
        L.require_cuda(vec)
        vec = L.f32c(vec)
        rows = vec.numel() // 3
        dim = sum(2 * l + 1 for l in ls)
        sh = torch.empty(rows, dim, device=vec.device, dtype=torch.float32)
        arr = (C.c_int32 * len(ls))(*ls)
        L.check(L.load().e3k_sph_harm_fwd(L.ptr(vec), rows, arr, len(ls), int(normalize), normalization, L.ptr(sh),
                                          L.stream_ptr()), "e3k_sph_harm_fwd")
        ctx.save_for_backward(vec)
        ctx.cfg = (ls, normalize, normalization)
        return sh

    @staticmethod
    def backward(ctx, g_sh):
        (vec,) = ctx.saved_tensors
        if torch.is_grad_enabled():
            return SphHarmBwdFn.apply(vec, g_sh, ctx.cfg), None, None, None
        return _sph_bwd_raw(vec, L.f32c(g_sh), ctx.cfg), None, None, None


def _sph_bwd_raw(vec, g_sh, cfg):
    ls, normalize, normalization = cfg
    rows = vec.numel() // 3
    g_vec = torch.empty_like(vec)
    arr = (C.c_int32 * len(ls))(*ls)
    L.check(L.load().e3k_sph_harm_bwd(L.ptr(vec), L.ptr(g_sh), rows, arr, len(ls), int(normalize), normalization,
                                      L.ptr(g_vec), L.stream_ptr()), "e3k_sph_harm_bwd")
    return g_vec


class SphHarmBwdFn(torch.autograd.Function):
    """g_vec = J(vec)^T g_sh; its backward (e3k_sph_harm_bwd2) evaluates the same polynomials on dual numbers."""

    @staticmethod
    def forward(ctx, vec, g_sh, cfg):
        vec, g_sh = L.f32c(vec), L.f32c(g_sh)
        ctx.save_for_backward(vec, g_sh)
        ctx.cfg = cfg
        return _sph_bwd_raw(vec, g_sh, cfg)

    @staticmethod
    @once_differentiable
    def backward(ctx, h):
        vec, g_sh = ctx.saved_tensors
        ls, normalize, normalization = ctx.cfg
        h = L.f32c(h)
        rows = vec.numel() // 3
        g_vec = torch.empty_like(vec) if ctx.needs_input_grad[0] else None
        g_gsh = torch.empty_like(g_sh) if ctx.needs_input_grad[1] else None
        if g_vec is not None or g_gsh is not None:
            arr = (C.c_int32 * len(ls))(*ls)
            L.check(L.load().e3k_sph_harm_bwd2(L.ptr(vec), L.ptr(g_sh), L.ptr(h), rows, arr, len(ls), int(normalize),
                                               normalization, L.ptr(g_gsh), L.ptr(g_vec), L.stream_ptr()),
                    "e3k_sph_harm_bwd2")
        return g_vec, g_gsh, None


NORMALIZATIONS = {"component": 0, "integral": 1, "norm": 2}


def spherical_harmonics(vec, ls: Sequence[int], normalize: bool, normalization: str):
    return SphHarmFn.apply(_c(vec), tuple(int(l) for l in ls), bool(normalize), NORMALIZATIONS[normalization])


class RadialBasisFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, r, bessel_w, r_max, r_min, p, one_over_r, cutoff_kind):
        ctx.param_slots = _param_slots(r, bessel_w)
        L.require_cuda(r, bessel_w)
        r_shape = r.shape
        assert r.dim() == 1, "radial_basis() flattens r before apply (saved tensors must be the true inputs)"
        r, bessel_w = L.f32c(r), L.f32c(bessel_w)
        e, nb = r.numel(), bessel_w.numel()
        out = torch.empty(e, nb, device=r.device, dtype=torch.float32)
        L.check(L.load().e3k_radial_basis_fwd(L.ptr(r), e, L.ptr(bessel_w), nb, r_max, r_min, p, int(one_over_r),
                                              cutoff_kind, L.ptr(out), L.stream_ptr()), "e3k_radial_basis_fwd")
        ctx.save_for_backward(r, bessel_w)
        ctx.cfg = (r_max, r_min, p, int(one_over_r), cutoff_kind)
        ctx.r_shape = r_shape
        return out

    @staticmethod
    def backward(ctx, g_out):
        need = _needs(ctx)
        r, bessel_w = ctx.saved_tensors
        need_r, need_w = need[:2]
        if torch.is_grad_enabled():
            g_r, g_w = RadialBasisBwdFn.apply(r, bessel_w, g_out, ctx.cfg, bool(need_r), bool(need_w))
        else:
            g_r, g_w = _radial_bwd_raw(r, bessel_w, L.f32c(g_out), ctx.cfg, need_r, need_w)
        if g_r is not None and ctx.r_shape != g_r.shape:
            g_r = g_r.view(ctx.r_shape)
        return (g_r if need_r else None), (g_w if need_w else None), None, None, None, None, None


def _radial_bwd_raw(r, bessel_w, g_out, cfg, need_r, need_w):
    r_max, r_min, p, one_over_r, kind = cfg
    g_r = torch.empty_like(r) if need_r else None
    g_w = torch.zeros_like(bessel_w) if need_w else None
    if g_r is not None or g_w is not None:
        L.check(L.load().e3k_radial_basis_bwd(L.ptr(r), L.ptr(g_out), r.numel(), L.ptr(bessel_w), bessel_w.numel(),
                                              r_max, r_min, p, one_over_r, kind, L.ptr(g_r), L.ptr(g_w),
                                              L.stream_ptr()), "e3k_radial_basis_bwd")
    return g_r, g_w


class RadialBasisBwdFn(torch.autograd.Function):
    """(g_r, g_w) of the radial basis; its backward is e3k_radial_basis_bwd2 (dual numbers)."""

    @staticmethod
    def forward(ctx, r, bessel_w, g_out, cfg, need_r: bool, need_w: bool):
        r, bessel_w, g_out = L.f32c(r), L.f32c(bessel_w), L.f32c(g_out)
        ctx.save_for_backward(r, bessel_w, g_out)
        ctx.cfg, ctx.flags = cfg, (need_r, need_w)
        ctx.set_materialize_grads(False)
        g_r, g_w = _radial_bwd_raw(r, bessel_w, g_out, cfg, need_r, need_w)
        dead = []
        if g_r is None:
            g_r = torch.zeros(0, device=r.device)
            dead.append(g_r)
        if g_w is None:
            g_w = torch.zeros(0, device=r.device)
            dead.append(g_w)
        if dead:
            ctx.mark_non_differentiable(*dead)
        return g_r, g_w

    @staticmethod
    @once_differentiable
    def backward(ctx, hr, hw):
        r, bessel_w, g_out = ctx.saved_tensors
        r_max, r_min, p, one_over_r, kind = ctx.cfg
        if not ctx.flags[0]:
            hr = None
        if not ctx.flags[1]:
            hw = None
        if hr is None and hw is None:
            return None, None, None, None, None, None
        hr = L.f32c(hr).view(-1) if hr is not None else None
        hw = L.f32c(hw) if hw is not None else None
        g_r = torch.empty_like(r) if ctx.needs_input_grad[0] else None
        g_w = torch.zeros_like(bessel_w) if ctx.needs_input_grad[1] else None
        g_go = torch.empty_like(g_out) if ctx.needs_input_grad[2] else None
        if g_r is not None or g_w is not None or g_go is not None:
            L.check(L.load().e3k_radial_basis_bwd2(L.ptr(r), L.ptr(g_out), L.ptr(hr), L.ptr(hw), r.numel(),
                                                   L.ptr(bessel_w), bessel_w.numel(), r_max, r_min, p, one_over_r, kind,
                                                   L.ptr(g_go), L.ptr(g_r), L.ptr(g_w), L.stream_ptr()),
                    "e3k_radial_basis_bwd2")
        return g_r, g_w, g_go, None, None, None


def radial_basis(r, bessel_w, r_max, r_min, p, one_over_r, cutoff_kind):
    return RadialBasisFn.apply(_c(r).reshape(-1), _c(bessel_w), float(r_max), float(r_min), float(p), bool(one_over_r), int(cutoff_kind))
