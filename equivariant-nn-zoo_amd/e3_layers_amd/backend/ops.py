"""autograd wrappers around the libe3k C ABI.

Every op here is ``torch.autograd.Function`` glue: it allocates outputs with torch, fills the
C structs of ``include/e3k.h`` with raw device pointers and enqueues the HIP kernels on the
current stream.  Backward passes call the matching backward kernels; they are *once
differentiable* (double backward — force training, ``e3_layers/nn/output.py:42`` with
``create_graph=True`` — is not built yet and fails loudly).
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch.autograd.function import once_differentiable

from . import lib as L
from .graph import GraphTopo

# bench.py sets this to a list to collect (start_event, end_event, N, E, plan) per launch of the
# fused TP+reduce forward kernel (HIP events on the launching stream); None = no profiling.
PROFILE_TP = None

# Independent backward kernels of one op (e.g. grad-wrt-weights streams grad_w to HBM while
# grad-wrt-x gathers from L2/MALL; dgrad and wgrad of a GEMM read the same gradient) are issued on
# two HIP streams so that they overlap; the side stream is joined before the op returns.
import os as _os
OVERLAP_STREAMS = int(_os.environ.get("E3K_OVERLAP", "0"))  # 0 off (default), 1 TP backward only, 2 also GEMM dgrad/wgrad
_side_streams: Dict[int, "torch.cuda.Stream"] = {}


class _Fork:
    """with _Fork(dev) as f: f.side(lambda: ...); ...main work...   -> joins on exit."""

    def __init__(self, device):
        self.device = device
        self.used = False

    def __enter__(self):
        self.cur = torch.cuda.current_stream(self.device)
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        st = _side_streams.get(idx)
        if st is None:
            st = _side_streams[idx] = torch.cuda.Stream(device=self.device)
        self.st = st
        return self

    def side(self, fn):
        if not OVERLAP_STREAMS or torch.cuda.is_current_stream_capturing():
            return fn()
        self.st.wait_stream(self.cur)
        with torch.cuda.stream(self.st):
            out = fn()
        self.used = True
        return out

    def __exit__(self, *exc):
        if self.used:
            self.cur.wait_stream(self.st)
        return False


# Gradient sink: run/parallel.FlatGradients registers (data_ptr, numel) -> view of its flat gradient
# buffer for every parameter.  A backward that finds its weight there accumulates the weight
# gradient straight into the (pre-zeroed) flat buffer and returns None for it: no zero-fill of a
# temporary and no autograd "+=" per parameter.
GRAD_SINK: Dict[Tuple[int, int], torch.Tensor] = {}


def _sink_for(t: torch.Tensor):
    return GRAD_SINK.get((t.data_ptr(), t.numel())) if GRAD_SINK else None


ACT_IDS = {None: 0, "identity": 0, "ssp": 1, "silu": 2, "tanhlu": 3, "tanh": 4, "abs": 5}


def _addr(t: torch.Tensor, elem_off: int = 0) -> int:
    return t.data_ptr() + 4 * int(elem_off)


def _run_gemm(problems: List[L.GemmProblem], wgrad: bool = False) -> None:
    if not problems:
        return
    arr = (L.GemmProblem * len(problems))(*problems)
    lib = L.load()
    fn = lib.e3k_gemm_wgrad if wgrad else lib.e3k_gemm
    L.check(fn(arr, len(problems), L.stream_ptr()), "e3k_gemm_wgrad" if wgrad else "e3k_gemm")


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


class StridedLinearFn(torch.autograd.Function):
    """y[rows, d_out] = (base +) sum over instructions  alpha * x_block @ W_block  (+ bias)."""

    @staticmethod
    def forward(ctx, x, weight, bias, base, spec: LinearSpec, scale: float, act: int = 0, act_cst: float = 1.0):
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
        bias_at = {}
        if bias is not None:
            bias = L.f32c(bias)
            for off, mul, boff in spec.bias_blocks:
                bias_at[off] = _addr(bias, boff)
        done_bias = set()
        for r, group in enumerate(spec.rounds("i_out")):
            probs = []
            for ins in group:
                a_r2, a_k = _layout_strides(spec.in_layout, ins.mul_in, ins.dim)
                c_r2, c_n = _layout_strides(spec.out_layout, ins.mul_out, ins.dim)
                p = L.GemmProblem()
                p.A, p.A2, p.B, p.C = _addr(x, ins.in_off), None, _addr(weight, ins.w_off), _addr(y, ins.out_off)
                p.bias = None
                if r == 0 and ins.out_off in bias_at:
                    p.bias = bias_at[ins.out_off]
                    done_bias.add(ins.out_off)
                p.M1, p.M2, p.N, p.K, p.V = rows, ins.dim, ins.mul_out, ins.mul_in, 0
                p.accumulate = 1 if (r > 0 or base is not None) else 0
                p.a_r1, p.a_r2, p.a_k = spec.d_in, a_r2, a_k
                p.b_k, p.b_n = ins.mul_out, 1
                p.c_r1, p.c_r2, p.c_n = spec.d_out, c_r2, c_n
                p.alpha = ins.alpha * scale
                if act:
                    p.act, p.act_cst = act, act_cst
                probs.append(p)
            _run_gemm(probs)
        if act and len(spec.rounds("i_out")) != 1:
            raise NotImplementedError("fused activation needs single-round linears")
        for off, mul, boff in spec.bias_blocks:  # biased block without any incoming path
            if bias is not None and off not in done_bias:
                y[:, off:off + mul] += bias[boff:boff + mul]
        if act:
            ctx.save_for_backward(x, weight, y)
        else:
            ctx.save_for_backward(x, weight)
        ctx.act, ctx.act_cst = act, act_cst
        ctx.spec, ctx.scale, ctx.has_bias, ctx.has_base = spec, scale, bias is not None, base is not None
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        if ctx.act:
            x, weight, y = ctx.saved_tensors
            gy = L.f32c(gy)
            gz = torch.empty_like(gy)
            L.check(L.load().e3k_act_bwd_from_output(L.ptr(y), L.ptr(gy), gy.numel(), ctx.act, ctx.act_cst, L.ptr(gz),
                                                     L.stream_ptr()), "e3k_act_bwd_from_output")
            gy = gz
        else:
            x, weight = ctx.saved_tensors
        spec: LinearSpec = ctx.spec
        scale = ctx.scale
        gy = L.f32c(gy)
        rows = x.shape[0]
        gx = gw = gb = None
        with _Fork(x.device) as fork:
            sunk = False
            if ctx.needs_input_grad[1]:
                gw = _sink_for(weight)
                sunk = gw is not None
                if not sunk:
                    gw = torch.zeros_like(weight)

                def _wgrad():
                    probs = []
                    for ins in spec.instr:
                        a_r2, a_k = _layout_strides(spec.in_layout, ins.mul_in, ins.dim)
                        c_r2, c_n = _layout_strides(spec.out_layout, ins.mul_out, ins.dim)
                        p = L.GemmProblem()
                        p.A, p.A2, p.B, p.C, p.bias = _addr(x, ins.in_off), None, _addr(gw, ins.w_off), _addr(gy, ins.out_off), None
                        p.M1, p.M2, p.N, p.K, p.V = rows, ins.dim, ins.mul_out, ins.mul_in, 0
                        p.accumulate = 1
                        p.a_r1, p.a_r2, p.a_k = spec.d_in, a_r2, a_k
                        p.b_k, p.b_n = ins.mul_out, 1
                        p.c_r1, p.c_r2, p.c_n = spec.d_out, c_r2, c_n
                        p.alpha = ins.alpha * scale
                        probs.append(p)
                    _run_gemm(probs, wgrad=True)

                if OVERLAP_STREAMS >= 2 and ctx.needs_input_grad[0] and rows * spec.d_out >= (1 << 22):
                    fork.side(_wgrad)   # big enough for the overlap to pay for the stream join
                else:
                    _wgrad()
            if ctx.needs_input_grad[0]:
                gx = (torch.empty if spec.in_covered else torch.zeros)(rows, spec.d_in, device=x.device, dtype=torch.float32)
                for r, group in enumerate(spec.rounds("i_in")):
                    probs = []
                    for ins in group:
                        a_r2, a_k = _layout_strides(spec.out_layout, ins.mul_out, ins.dim)
                        c_r2, c_n = _layout_strides(spec.in_layout, ins.mul_in, ins.dim)
                        p = L.GemmProblem()
                        p.A, p.A2, p.B, p.C, p.bias = _addr(gy, ins.out_off), None, _addr(weight, ins.w_off), _addr(gx, ins.in_off), None
                        p.M1, p.M2, p.N, p.K, p.V = rows, ins.dim, ins.mul_in, ins.mul_out, 0
                        p.accumulate = 1 if r > 0 else 0
                        p.a_r1, p.a_r2, p.a_k = spec.d_out, a_r2, a_k
                        p.b_k, p.b_n = 1, ins.mul_out  # W^T
                        p.c_r1, p.c_r2, p.c_n = spec.d_in, c_r2, c_n
                        p.alpha = ins.alpha * scale
                        probs.append(p)
                    _run_gemm(probs)
        if sunk:
            gw = None   # already accumulated into the flat gradient buffer
        if ctx.has_bias and ctx.needs_input_grad[2]:
            nb = sum(m for _, m, _ in spec.bias_blocks)
            gb = torch.zeros(nb, device=x.device, dtype=torch.float32)
            lib = L.load()
            for off, mul, boff in spec.bias_blocks:
                L.check(lib.e3k_colsum(_addr(gy, off), rows, mul, spec.d_out, _addr(gb, boff), L.stream_ptr()), "e3k_colsum")
        gbase = gy if (ctx.has_base and ctx.needs_input_grad[3]) else None
        return gx, gw, gb, gbase, None, None, None, None


def strided_linear(x, weight, bias, spec: LinearSpec, base=None, scale: float = 1.0, act: Optional[str] = None,
                   act_cst: float = 1.0):
    """``act='ssp'`` fuses ``act_cst * ssp(.)`` into the GEMM epilogue (radial MLP layers)."""
    if act is None:
        return StridedLinearFn.apply(x, weight, bias, base, spec, float(scale))
    if act != "ssp":
        return activation(StridedLinearFn.apply(x, weight, bias, base, spec, float(scale)), act, act_cst)
    return StridedLinearFn.apply(x, weight, bias, base, spec, float(scale), 1, float(act_cst))


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


class FctpFn(torch.autograd.Function):
    """out[n, w, k] = alpha * sum_{u,v} W[u,v,w] x[n,u,k] attrs[n,v]   per instruction."""

    @staticmethod
    def forward(ctx, x, attrs, weight, spec: FctpSpec):
        L.require_cuda(x, attrs, weight)
        x, attrs, weight = L.f32c(x), L.f32c(attrs), L.f32c(weight)
        rows = x.shape[0]
        assert x.shape[1] == spec.d_in and attrs.shape == (rows, spec.v)
        y = (torch.empty if spec.out_covered else torch.zeros)(rows, spec.d_out, device=x.device, dtype=torch.float32)
        for r, group in enumerate(spec.rounds("i_out")):
            probs = []
            for ins in group:
                a_r2, a_k = _layout_strides(spec.in_layout, ins.mul_in, ins.dim)
                c_r2, c_n = _layout_strides(spec.out_layout, ins.mul_out, ins.dim)
                p = L.GemmProblem()
                p.A, p.A2, p.B, p.C, p.bias = _addr(x, ins.in_off), _addr(attrs), _addr(weight, ins.w_off), _addr(y, ins.out_off), None
                p.M1, p.M2, p.N, p.K, p.V = rows, ins.dim, ins.mul_out, ins.mul_in * spec.v, spec.v
                p.accumulate = 1 if r > 0 else 0
                p.a_r1, p.a_r2, p.a_k, p.a2_r1 = spec.d_in, a_r2, a_k, spec.v
                p.b_k, p.b_n = ins.mul_out, 1
                p.c_r1, p.c_r2, p.c_n = spec.d_out, c_r2, c_n
                p.alpha = ins.alpha
                probs.append(p)
            _run_gemm(probs)
        ctx.save_for_backward(x, attrs, weight)
        ctx.spec = spec
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, attrs, weight = ctx.saved_tensors
        spec: FctpSpec = ctx.spec
        gy = L.f32c(gy)
        rows = x.shape[0]
        lib = L.load()
        gx = ga = gw = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            if spec.in_layout != "cf":
                raise NotImplementedError("self-connection backward expects the channel-fastest input layout")
            gx = (torch.empty if spec.in_covered else torch.zeros)(rows, spec.d_in, device=x.device, dtype=torch.float32)
            ga = torch.zeros(rows, spec.v, device=x.device, dtype=torch.float32)
            hmax = max(ins.dim * ins.mul_in for ins in spec.instr) * spec.v
            H = torch.empty(rows * hmax, device=x.device, dtype=torch.float32)
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
        sunk = False
        if ctx.needs_input_grad[2]:
            gw = _sink_for(weight)
            sunk = gw is not None
            if not sunk:
                gw = torch.zeros_like(weight)
            probs = []
            for ins in spec.instr:
                a_r2, a_k = _layout_strides(spec.in_layout, ins.mul_in, ins.dim)
                c_r2, c_n = _layout_strides(spec.out_layout, ins.mul_out, ins.dim)
                p = L.GemmProblem()
                p.A, p.A2, p.B, p.C, p.bias = _addr(x, ins.in_off), _addr(attrs), _addr(gw, ins.w_off), _addr(gy, ins.out_off), None
                p.M1, p.M2, p.N, p.K, p.V = rows, ins.dim, ins.mul_out, ins.mul_in * spec.v, spec.v
                p.accumulate = 1
                p.a_r1, p.a_r2, p.a_k, p.a2_r1 = spec.d_in, a_r2, a_k, spec.v
                p.b_k, p.b_n = ins.mul_out, 1
                p.c_r1, p.c_r2, p.c_n = spec.d_out, c_r2, c_n
                p.alpha = ins.alpha
                probs.append(p)
            _run_gemm(probs, wgrad=True)
        if not ctx.needs_input_grad[0]:
            gx = None
        if not ctx.needs_input_grad[1]:
            ga = None
        if sunk:
            gw = None
        return gx, ga, gw, None


def fctp(x, attrs, weight, spec: FctpSpec):
    return FctpFn.apply(x, attrs, weight, spec)


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


def _grouped_templates(x, m_like, y, spec, m_off, ld_m, mode: str):
    """One template problem per instruction; e3k_gemm_grouped expands them over the keys."""
    rows = x.shape[0]
    out = []
    for j, ins in enumerate(spec.instr):
        in_r2, in_k = _layout_strides(spec.in_layout, ins.mul_in, ins.dim)
        out_r2, out_n = _layout_strides(spec.out_layout, ins.mul_out, ins.dim)
        p = L.GemmProblem()
        p.bias, p.A2, p.V = None, None, 0
        p.M1, p.M2 = rows, ins.dim
        p.alpha = ins.alpha
        if mode == "fwd":      # y = x . M
            p.A, p.B, p.C = _addr(x, ins.in_off), _addr(m_like, m_off[j]), _addr(y, ins.out_off)
            p.N, p.K = ins.mul_out, ins.mul_in
            p.a_r1, p.a_r2, p.a_k = spec.d_in, in_r2, in_k
            p.b_k, p.b_n = ins.mul_out, 1
            p.c_r1, p.c_r2, p.c_n = spec.d_out, out_r2, out_n
        elif mode == "dgrad":  # gx = gy . M^T      (x := gy, y := gx)
            p.A, p.B, p.C = _addr(x, ins.out_off), _addr(m_like, m_off[j]), _addr(y, ins.in_off)
            p.N, p.K = ins.mul_in, ins.mul_out
            p.a_r1, p.a_r2, p.a_k = spec.d_out, out_r2, out_n
            p.b_k, p.b_n = 1, ins.mul_out
            p.c_r1, p.c_r2, p.c_n = spec.d_in, in_r2, in_k
        else:                  # wgrad: gm += x^T . gy   (y := gy)
            p.A, p.B, p.C = _addr(x, ins.in_off), _addr(m_like, m_off[j]), _addr(y, ins.out_off)
            p.N, p.K = ins.mul_out, ins.mul_in
            p.a_r1, p.a_r2, p.a_k = spec.d_in, in_r2, in_k
            p.b_k, p.b_n = ins.mul_out, 1
            p.c_r1, p.c_r2, p.c_n = spec.d_out, out_r2, out_n
        out.append((ins, p))
    return out


def _run_grouped(templates, groups: RowGroups, ld_m: int, wgrad: bool, key: str):
    """Launch rounds so that no two problems of one launch write the same block."""
    seen: Dict[int, int] = {}
    rounds: List[List] = []
    for ins, p in templates:
        k = getattr(ins, key) if key else id(p)
        r = seen.get(k, 0)
        seen[k] = r + 1
        while len(rounds) <= r:
            rounds.append([])
        p.accumulate = 1 if (r > 0 or wgrad) else 0
        rounds[r].append(p)
    lib = L.load()
    for grp in rounds:
        arr = (L.GemmProblem * len(grp))(*grp)
        L.check(lib.e3k_gemm_grouped(arr, len(grp), L.ptr(groups.perm), L.ptr(groups.bounds), groups.n_keys, ld_m,
                                     int(wgrad), L.stream_ptr()), "e3k_gemm_grouped")


class GroupedLinearFn(torch.autograd.Function):
    """out[n, w, k] = alpha * sum_u M[key(n)][u, w] x[n, u, k] per instruction; M is
    [K, sum_j U_j * W_j] with instruction j's block at column offset ``m_off[j]``."""

    @staticmethod
    def forward(ctx, x, m, groups: RowGroups, spec: "FctpSpec", m_off: Tuple[int, ...]):
        L.require_cuda(x, m)
        x, m = L.f32c(x), L.f32c(m)
        rows, ld_m = x.shape[0], m.shape[1]
        # rows of absent keys do not exist, every node belongs to exactly one key: full coverage
        y = (torch.empty if spec.out_covered else torch.zeros)(rows, spec.d_out, device=x.device, dtype=torch.float32)
        _run_grouped(_grouped_templates(x, m, y, spec, m_off, ld_m, "fwd"), groups, ld_m, False, "i_out")
        ctx.save_for_backward(x, m)
        ctx.groups, ctx.spec, ctx.m_off = groups, spec, m_off
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, m = ctx.saved_tensors
        groups, spec, m_off = ctx.groups, ctx.spec, ctx.m_off
        gy = L.f32c(gy)
        rows, ld_m = x.shape[0], m.shape[1]
        gx = gm = None
        if ctx.needs_input_grad[0]:
            gx = (torch.empty if spec.in_covered else torch.zeros)(rows, spec.d_in, device=x.device, dtype=torch.float32)
            _run_grouped(_grouped_templates(gy, m, gx, spec, m_off, ld_m, "dgrad"), groups, ld_m, False, "i_in")
        if ctx.needs_input_grad[1]:
            gm = torch.zeros_like(m)
            _run_grouped(_grouped_templates(x, gm, gy, spec, m_off, ld_m, "wgrad"), groups, ld_m, True, "")
        return gx, gm, None, None, None


def grouped_linear(x, m, groups: RowGroups, spec: "FctpSpec", m_off: Sequence[int]):
    return GroupedLinearFn.apply(x, m, groups, spec, tuple(int(v) for v in m_off))


# --------------------------------------------------------------------------------------
# fused uvu tensor product + destination reduce
# --------------------------------------------------------------------------------------
class TpPlan:
    """Owns an ``e3k_tp_plan`` (device copies of the group table)."""

    def __init__(self, groups: Sequence[L.TpGroup], d_in: int, d_sh: int, w_numel: int, d_mid: int):
        self.groups = list(groups)
        self.d_in, self.d_sh, self.w_numel, self.d_mid = d_in, d_sh, w_numel, d_mid
        self._handles: Dict[int, int] = {}  # device index -> plan pointer

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

    def __del__(self):
        try:
            lib = L.load()
            for h in self._handles.values():
                lib.e3k_tp_plan_destroy(h)
        except Exception:
            pass


class TpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, sh, w, topo: GraphTopo, plan: TpPlan):
        L.require_cuda(x, sh, w)
        x, sh, w = L.f32c(x), L.f32c(sh), L.f32c(w)
        n, e = x.shape[0], sh.shape[0]
        assert x.shape[1] == plan.d_in and sh.shape[1] == plan.d_sh and w.shape == (e, plan.w_numel)
        assert topo.num_nodes == n and topo.num_edges == e
        out = torch.empty(n, plan.d_mid, device=x.device, dtype=torch.float32)
        handle = plan.handle(x.device)
        prof = PROFILE_TP
        if prof is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        L.check(L.load().e3k_tp_fwd(handle, L.ptr(x), L.ptr(sh), L.ptr(w), L.ptr(topo.src), L.ptr(topo.dst_ptr),
                                    L.ptr(topo.dst_perm), n, e, L.ptr(out), L.stream_ptr()), "e3k_tp_fwd")
        if prof is not None:
            ev1.record()
            prof.append((ev0, ev1, n, e, plan))
        ctx.save_for_backward(x, sh, w)
        ctx.topo, ctx.plan = topo, plan
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g_out):
        x, sh, w = ctx.saved_tensors
        topo, plan = ctx.topo, ctx.plan
        g_out = L.f32c(g_out)
        n, e = x.shape[0], sh.shape[0]
        lib = L.load()
        gx = gsh = gw = None
        handle = plan.handle(x.device)
        with _Fork(x.device) as fork:
            if ctx.needs_input_grad[0]:
                gx = torch.zeros_like(x)
                fork.side(lambda: L.check(
                    lib.e3k_tp_bwd_x(handle, L.ptr(sh), L.ptr(w), L.ptr(g_out), L.ptr(topo.dst), L.ptr(topo.src_ptr),
                                     L.ptr(topo.src_perm), n, e, L.ptr(gx), L.stream_ptr()), "e3k_tp_bwd_x"))
            if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
                gw = torch.empty_like(w)
                if ctx.needs_input_grad[1]:
                    gsh = torch.zeros_like(sh)
                L.check(lib.e3k_tp_bwd_w(handle, L.ptr(x), L.ptr(sh), L.ptr(w), L.ptr(g_out), L.ptr(topo.src),
                                         L.ptr(topo.dst_ptr), L.ptr(topo.dst_perm), n, e, L.ptr(gw), L.ptr(gsh),
                                         L.stream_ptr()), "e3k_tp_bwd_w")
        if not ctx.needs_input_grad[2]:
            gw = None
        return gx, gsh, gw, None, None


def tp_uvu_scatter(x, sh, w, topo: GraphTopo, plan: TpPlan):
    return TpFn.apply(x, sh, w, topo, plan)


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
    @once_differentiable
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        gy = L.f32c(gy)
        gx = torch.empty_like(x)
        L.check(L.load().e3k_act_bwd(L.ptr(x), L.ptr(gy), x.numel(), ctx.act_id, ctx.cst, L.ptr(gx), L.stream_ptr()), "e3k_act_bwd")
        return gx, None, None


def activation(x, name: str, cst: float):
    return ActFn.apply(x, ACT_IDS[name], float(cst))


def _blocks(blocks: Sequence[Tuple[int, int, int]]):
    arr = (L.Block * max(len(blocks), 1))()
    for i, (off, mul, dim) in enumerate(blocks):
        arr[i].off, arr[i].mul, arr[i].dim = off, mul, dim
    return arr


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
    @once_differentiable
    def backward(ctx, gy):
        gy = L.f32c(gy)
        gx = torch.empty_like(gy)
        L.check(L.load().e3k_relayout(L.ptr(gy), gy.shape[0], gy.shape[1], _blocks(ctx.blocks), len(ctx.blocks),
                                      int(not ctx.to_cf), L.ptr(gx), L.stream_ptr()), "e3k_relayout")
        return gx, None, None


def relayout(x, blocks: Sequence[Tuple[int, int, int]], to_cf: bool):
    """blocks: (offset, mul, 2l+1) of the feature row; blocks with dim 1 or mul 1 are no-ops."""
    blocks = tuple(b for b in blocks if b[1] > 1 and b[2] > 1)
    if not blocks:
        return x
    return RelayoutFn.apply(x, blocks, to_cf)


@dataclass
class GateSpec:
    in_dim: int
    out_dim: int
    segs: List[Tuple[int, int, int, int, int, int, int, float]]  # kind,in_off,gate_off,out_off,mul,dim,act,cst

    def c_array(self):
        arr = (L.GateSeg * len(self.segs))()
        for i, s in enumerate(self.segs):
            (arr[i].kind, arr[i].in_off, arr[i].gate_off, arr[i].out_off, arr[i].mul, arr[i].dim, arr[i].act, arr[i].cst) = s
        return arr


class GateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, spec: GateSpec):
        L.require_cuda(x)
        x = L.f32c(x)
        assert x.shape[1] == spec.in_dim
        y = torch.empty(x.shape[0], spec.out_dim, device=x.device, dtype=torch.float32)
        L.check(L.load().e3k_gate_fwd(L.ptr(x), x.shape[0], spec.in_dim, spec.out_dim, spec.c_array(), len(spec.segs),
                                      L.ptr(y), L.stream_ptr()), "e3k_gate_fwd")
        ctx.save_for_backward(x)
        ctx.spec = spec
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        spec = ctx.spec
        gy = L.f32c(gy)
        gx = torch.empty_like(x)
        L.check(L.load().e3k_gate_bwd(L.ptr(x), L.ptr(gy), x.shape[0], spec.in_dim, spec.out_dim, spec.c_array(),
                                      len(spec.segs), L.ptr(gx), L.stream_ptr()), "e3k_gate_bwd")
        return gx, None


def gate(x, spec: GateSpec):
    return GateFn.apply(x, spec)


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
    @once_differentiable
    def backward(ctx, gy):
        x, std, inv = ctx.saved_tensors
        gy = L.f32c(gy)
        gx = torch.empty_like(x)
        gstd = torch.zeros_like(std)
        L.check(L.load().e3k_layernorm_bwd(L.ptr(x), L.ptr(gy), L.ptr(inv), x.shape[0], x.shape[1], _blocks(ctx.blocks),
                                           len(ctx.blocks), L.ptr(std), L.ptr(gx), L.ptr(gstd), L.stream_ptr()),
                "e3k_layernorm_bwd")
        return gx, gstd, None


def layer_norm(x, std, blocks):
    return LayerNormFn.apply(x, std, tuple(blocks))


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
    @once_differentiable
    def backward(ctx, g):
        ptr, seg_index = ctx.saved_tensors
        if ctx.mean:
            cnt = (ptr[1:] - ptr[:-1]).clamp(min=1).to(g.dtype).view(-1, 1)
            g = g / cnt
        return g.index_select(0, seg_index), None, None, None


def segment_sum(x, ptr, seg_index, mean=False):
    return SegmentSumFn.apply(x, ptr, seg_index, bool(mean))


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
    @once_differentiable
    def backward(ctx, g_vec, g_len):
        vec, length = ctx.saved_tensors
        topo = ctx.topo
        g_vec = L.f32c(g_vec) if g_vec is not None else None
        g_len = L.f32c(g_len) if g_len is not None else None
        g_pos = torch.empty(ctx.n, 3, device=vec.device, dtype=torch.float32)
        L.check(L.load().e3k_edge_vector_bwd(L.ptr(g_vec), L.ptr(g_len), L.ptr(vec), L.ptr(length), L.ptr(topo.dst_ptr),
                                             L.ptr(topo.dst_perm), L.ptr(topo.src_ptr), L.ptr(topo.src_perm), ctx.n,
                                             L.ptr(g_pos), L.stream_ptr()), "e3k_edge_vector_bwd")
        return g_pos, None


def edge_vector(pos, topo: GraphTopo):
    return EdgeVectorFn.apply(pos, topo)


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
    @once_differentiable
    def backward(ctx, g_sh):
        (vec,) = ctx.saved_tensors
        ls, normalize, normalization = ctx.cfg
        g_sh = L.f32c(g_sh)
        rows = vec.numel() // 3
        g_vec = torch.empty_like(vec)
        arr = (C.c_int32 * len(ls))(*ls)
        L.check(L.load().e3k_sph_harm_bwd(L.ptr(vec), L.ptr(g_sh), rows, arr, len(ls), int(normalize), normalization,
                                          L.ptr(g_vec), L.stream_ptr()), "e3k_sph_harm_bwd")
        return g_vec, None, None, None


NORMALIZATIONS = {"component": 0, "integral": 1, "norm": 2}


def spherical_harmonics(vec, ls: Sequence[int], normalize: bool, normalization: str):
    return SphHarmFn.apply(vec, tuple(int(l) for l in ls), bool(normalize), NORMALIZATIONS[normalization])


class RadialBasisFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, r, bessel_w, r_max, r_min, p, one_over_r, cutoff_kind):
        L.require_cuda(r, bessel_w)
        r, bessel_w = L.f32c(r).view(-1), L.f32c(bessel_w)
        e, nb = r.numel(), bessel_w.numel()
        out = torch.empty(e, nb, device=r.device, dtype=torch.float32)
        L.check(L.load().e3k_radial_basis_fwd(L.ptr(r), e, L.ptr(bessel_w), nb, r_max, r_min, p, int(one_over_r),
                                              cutoff_kind, L.ptr(out), L.stream_ptr()), "e3k_radial_basis_fwd")
        ctx.save_for_backward(r, bessel_w)
        ctx.cfg = (r_max, r_min, p, int(one_over_r), cutoff_kind)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g_out):
        r, bessel_w = ctx.saved_tensors
        r_max, r_min, p, one_over_r, kind = ctx.cfg
        g_out = L.f32c(g_out)
        g_r = torch.empty_like(r) if ctx.needs_input_grad[0] else None
        g_w = torch.zeros_like(bessel_w) if ctx.needs_input_grad[1] else None
        if g_r is not None or g_w is not None:
            L.check(L.load().e3k_radial_basis_bwd(L.ptr(r), L.ptr(g_out), r.numel(), L.ptr(bessel_w), bessel_w.numel(),
                                                  r_max, r_min, p, one_over_r, kind, L.ptr(g_r), L.ptr(g_w),
                                                  L.stream_ptr()), "e3k_radial_basis_bwd")
        return g_r, g_w, None, None, None, None, None


def radial_basis(r, bessel_w, r_max, r_min, p, one_over_r, cutoff_kind):
    return RadialBasisFn.apply(r, bessel_w, float(r_max), float(r_min), float(p), bool(one_over_r), int(cutoff_kind))
