"""Operator-level building blocks: the host side of what the reference gets from e3nn.

``Linear``, ``FullyConnectedNet``, ``FullyConnectedTensorProduct`` (scalar second operand),
``Gate`` and ``UVUTensorProduct`` keep e3nn 0.4.4's parameter layout (one flat ``weight`` per
operator, instruction-major — SURVEY.md A.1/A.4/A.5/A.6) and normalisation, but they only
*describe* the arithmetic: each builds the struct tables that ``backend/ops.py`` hands to the
HIP kernels.  Reference call sites: ``e3_layers/nn/message_passing.py:58-87``,
``e3_layers/nn/pointwise.py:18,61-92``.

Layouts: "e3nn" = block [mul][2l+1] (what the data dict carries), "cf" = block [2l+1][mul]
(channel-fastest, internal to a convolution).
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from ..backend.tuning import knob as _knob
from torch import nn

from ..backend import lib as L
from ..backend import ops
from ..backend.graph import GraphTopo
from ..o3 import Irrep, Irreps
from ..utils.utils import act_output_parity, act_second_moment_const, activation_name

# degree limits of the generated Clebsch-Gordan tables (tools/gen_cg.py, csrc/e3k_cg_gen.h)
TP_L1MAX, TP_L2MAX, TP_L3MAX = 3, 2, 3


# ---- row keys -----------------------------------------------------------------------------
# A tensor produced row-wise from a categorical input carries ``_e3k_key = (index int64 [rows],
# n_keys)``: rows with equal keys are the same function of the same inputs, hence identical.
_KEY_CAP = 1 << 20


def get_row_key(t):
    return getattr(t, "_e3k_key", None)


def set_row_key(t, index, n_keys: int):
    if index is not None and 0 < n_keys <= _KEY_CAP:
        t._e3k_key = (index, int(n_keys))
    return t


def combine_row_keys(tensors):
    """Key of a row-wise function of several keyed tensors (None if any is unkeyed or too many combinations)."""
    index, n = None, 1
    for t in tensors:
        k = get_row_key(t)
        if k is None:
            return None
        index = k[0] if index is None else index * k[1] + k[0]
        n *= k[1]
        if n > _KEY_CAP:
            return None
    return (index, n) if index is not None else None


_groups_cache: Dict[int, tuple] = {}


def row_groups(index: torch.Tensor, n_keys: int) -> "ops.RowGroups":
    """Group rows by key with device-side ops only (stable sort, scatter-add counts, cumsum): no host
    synchronisation, so a forward can be captured in a HIP graph.  Memoised per key tensor so the
    layers of one forward (and their backward) share one sort."""
    import weakref

    hit = _groups_cache.get(id(index))
    if hit is not None and hit[0]() is index and hit[1] == n_keys:
        return hit[2]
    idx = index.reshape(-1)
    if idx.is_cuda and n_keys <= 256 and idx.dtype == torch.int64:
        # csrc/e3k_graph.hip: one single-workgroup launch (a radix sort + scatter_add + cumsum + gather otherwise)
        from ..backend import lib as L

        idx = idx.contiguous()
        buf = torch.empty(idx.numel() + 2 * n_keys + 1, dtype=torch.int32, device=idx.device)
        perm, bounds, flag = buf[:idx.numel()], buf[idx.numel():idx.numel() + 2 * n_keys].view(n_keys, 2), buf[-1:]
        reps = torch.empty(n_keys, dtype=torch.int64, device=idx.device)
        with torch.cuda.device(idx.device):
            L.check(L.load().e3k_group_rows(L.ptr(idx), idx.numel(), n_keys, L.ptr(perm), L.ptr(bounds), L.ptr(reps), L.ptr(flag),
                                            L.stream_ptr()), "e3k_group_rows")
            from ..backend.graph import defer_flag

            # rows whose key is outside [0, n_keys) are left out of perm / bounds: reported like bad edge endpoints
            defer_flag(flag, f"a row key outside [0, {n_keys}) reached the keyed self-connection (species index beyond num_types?)")
        groups = ops.RowGroups(perm, bounds, reps, n_keys)
    else:
        perm = torch.argsort(idx, stable=True)
        counts = torch.zeros(n_keys, dtype=torch.long, device=idx.device).scatter_add_(0, idx, torch.ones_like(idx))
        starts = torch.cumsum(counts, 0) - counts
        reps = perm[starts.clamp(max=max(idx.numel() - 1, 0))]
        bounds = torch.stack([starts, counts], dim=1).to(torch.int32).contiguous()
        groups = ops.RowGroups(perm.to(torch.int32), bounds, reps, n_keys)
    if len(_groups_cache) > 16:
        _groups_cache.clear()
    _groups_cache[id(index)] = (weakref.ref(index), n_keys, groups)
    return groups


def irreps_blocks(irreps: Irreps) -> List[Tuple[int, int, int]]:
    """(offset, mul, 2l+1) per entry."""
    out, pos = [], 0
    for mi in irreps:
        out.append((pos, mi.mul, mi.ir.dim))
        pos += mi.dim
    return out


class Linear(nn.Module):
    """``o3.Linear``: per-irrep channel mixing, ``1/sqrt(fan_in)`` path weights, biases on 0e."""

    def __init__(self, irreps_in, irreps_out, biases: bool = False, **_unused):
        super().__init__()
        self.irreps_in, self.irreps_out = Irreps(irreps_in), Irreps(irreps_out)
        in_off, out_off = self.irreps_in.offsets(), self.irreps_out.offsets()
        pairs = [(i, o) for i, a in enumerate(self.irreps_in) for o, b in enumerate(self.irreps_out)
                 if a.ir == b.ir and a.mul > 0 and b.mul > 0]
        fan = [0] * len(self.irreps_out)
        for i, o in pairs:
            fan[o] += self.irreps_in[i].mul
        instr, w_off = [], 0
        for i, o in pairs:
            a, b = self.irreps_in[i], self.irreps_out[o]
            instr.append(ops.LinInstr(in_off[i], out_off[o], a.mul, b.mul, a.ir.dim, w_off, 1.0 / math.sqrt(fan[o]), i, o))
            w_off += a.mul * b.mul
        self.weight_numel = w_off
        self.weight = nn.Parameter(torch.randn(w_off))
        bias_blocks, b_off = [], 0
        if biases:
            for o, b in enumerate(self.irreps_out):
                if b.ir.is_scalar() and b.mul > 0:
                    bias_blocks.append((out_off[o], b.mul, b_off))
                    b_off += b.mul
        if b_off:
            self.bias = nn.Parameter(torch.zeros(b_off))
        else:
            self.register_parameter("bias", None)
        covered_out = {o for _, o in pairs}
        covered_in = {i for i, _ in pairs}
        self._instr, self._bias_blocks = instr, bias_blocks
        self._out_cov = all(o in covered_out or b.dim == 0 for o, b in enumerate(self.irreps_out))
        self._in_cov = all(i in covered_in or a.dim == 0 for i, a in enumerate(self.irreps_in))
        self._specs: Dict[Tuple[str, str], ops.LinearSpec] = {}

    def spec(self, in_layout: str, out_layout: str) -> ops.LinearSpec:
        key = (in_layout, out_layout)
        if key not in self._specs:
            self._specs[key] = ops.LinearSpec(self.irreps_in.dim, self.irreps_out.dim, self._instr, in_layout, out_layout,
                                              self._bias_blocks, self._out_cov, self._in_cov, self.weight_numel)
        return self._specs[key]

    def forward(self, x, in_layout: str = "e3nn", out_layout: str = "e3nn", base=None, scale: float = 1.0):
        return ops.strided_linear(x, self.weight, self.bias, self.spec(in_layout, out_layout), base=base, scale=scale)

    def extra_repr(self):
        return f"{self.irreps_in} -> {self.irreps_out} | {self.weight_numel} weights"


class _FCLayer(nn.Module):
    def __init__(self, h_in: int, h_out: int, act: Optional[str]):
        super().__init__()
        self.h_in, self.h_out, self.act = h_in, h_out, act
        self.cst = act_second_moment_const(act) if act else 1.0
        self.weight = nn.Parameter(torch.randn(h_in, h_out))
        self._spec = ops.LinearSpec(h_in, h_out, [ops.LinInstr(0, 0, h_in, h_out, 1, 0, 1.0 / math.sqrt(h_in))],
                                    "e3nn", "e3nn", [], True, True, h_in * h_out)

    def forward(self, x):
        # (a fused activation epilogue exists -- strided_linear(..., act="ssp") -- but measured slower here: the
        # transcendental lands in the store phase that already bounds the small-K GEMM)
        y = ops.strided_linear(x, self.weight.view(-1), None, self._spec)
        return ops.activation(y, self.act, self.cst) if self.act else y


class FullyConnectedNet(nn.Sequential):
    """``e3nn.nn.FullyConnectedNet``: x @ (W / sqrt(h_in)) -> normalised activation, last layer linear."""

    def __init__(self, hs: Sequence[int], act=None):
        name = activation_name(act) if act is not None else None
        layers = {}
        hs = [int(h) for h in hs]
        for i, (a, b) in enumerate(zip(hs[:-1], hs[1:])):
            layers[f"layer{i}"] = _FCLayer(a, b, name if i < len(hs) - 2 else None)
        super().__init__()
        for k, v in layers.items():
            self.add_module(k, v)
        self.hs = hs
        self.act_name = name
        # hidden chain in one launch (csrc/e3k_mlp.hip) when the widths fit; E3K_FUSED_MLP=0 keeps the per-layer ops
        self.fused_hidden = (len(hs) >= 3 and name is not None and _knob("E3K_FUSED_MLP") != 0
                             and ops.mlp_hidden_supported(hs[0], hs[1:-1], name))

    def forward(self, x):
        if not self.fused_hidden or not x.is_cuda:
            return super().forward(x)
        layers = list(self.children())
        hidden, last = layers[:-1], layers[-1]
        h = ops.mlp_hidden(x, [m.weight for m in hidden], [1.0 / math.sqrt(m.h_in) for m in hidden], self.act_name,
                           hidden[0].cst)
        with ops.timed_launch("radial_last_fwd", (h.shape[0], last.h_in, last.h_out)):
            return last(h)


class FullyConnectedTensorProduct(nn.Module):
    """``o3.FullyConnectedTensorProduct(in1, in2, out)`` for an all-scalar ``in2`` (the node
    attributes of the self-connection, ``e3_layers/nn/message_passing.py:81-87``):
    out[n,w,k] = (sum_paths mul1*mul2)^(-1/2) * sum_{u,v} W[u,v,w] x[n,u,k] a[n,v]."""

    def __init__(self, irreps_in1, irreps_in2, irreps_out):
        super().__init__()
        self.irreps_in1, self.irreps_in2, self.irreps_out = Irreps(irreps_in1), Irreps(irreps_in2), Irreps(irreps_out)
        in2 = self.irreps_in2
        if len(in2) != 1 or not in2[0].ir.is_scalar():
            raise NotImplementedError(
                f"the HIP self-connection handles a single block of even scalars as second operand, got {in2}")
        v = in2[0].mul
        in_off, out_off = self.irreps_in1.offsets(), self.irreps_out.offsets()
        pairs = [(i, o) for i, a in enumerate(self.irreps_in1) for o, b in enumerate(self.irreps_out) if a.ir == b.ir]
        fan = [0] * len(self.irreps_out)
        for i, o in pairs:
            fan[o] += self.irreps_in1[i].mul * v
        instr, w_off = [], 0
        for i, o in pairs:
            a, b = self.irreps_in1[i], self.irreps_out[o]
            instr.append(ops.FctpInstr(in_off[i], out_off[o], a.mul, b.mul, a.ir.dim, w_off, 1.0 / math.sqrt(fan[o]), i, o))
            w_off += a.mul * v * b.mul
        self.weight_numel = w_off
        self.weight = nn.Parameter(torch.randn(w_off))
        cov_o, cov_i = {o for _, o in pairs}, {i for i, _ in pairs}
        self._spec = ops.FctpSpec(self.irreps_in1.dim, self.irreps_out.dim, v, instr, "cf", "cf",
                                  all(o in cov_o for o in range(len(self.irreps_out))),
                                  all(i in cov_i for i in range(len(self.irreps_in1))))

    # keyed attrs: use the per-key contracted weights when there are few keys and many rows
    KEY_MAX = _knob("E3K_KEY_MAX")
    KEY_MIN_ROWS = 256
    KEY_MIN_ROWS_PER_KEY = 8

    @classmethod
    def keyed_pays(cls, key, rows: int) -> bool:
        """Few keys, many rows per key: QM9 species (5-10 keys); residue type x protein in the protein score net
        (20 x 4 = 80 keys over 1 536 residues: 12.9 -> 12.4 ms per step); not atom type x molecule (2 304 keys)."""
        return (key is not None and key[1] <= cls.KEY_MAX and rows >= cls.KEY_MIN_ROWS
                and rows >= cls.KEY_MIN_ROWS_PER_KEY * key[1])

    def forward(self, x_cf, attrs):
        """x in the channel-fastest layout -> output in the channel-fastest layout."""
        key = get_row_key(attrs)
        if self.keyed_pays(key, x_cf.shape[0]):
            return self._forward_keyed(x_cf, attrs, key)
        return ops.fctp(x_cf, attrs, self.weight, self._spec)

    # (Contracting general attributes into one weight matrix per node first -- M[n] = sum_v a[n,v] W[:,v,:], then a
    # (2l+1) x U x W product per node -- halves the multiply-adds of the protein configuration but moves a 250 MB matrix
    # through HBM six times per layer: measured 18.0 vs 13.4 ms per step, removed in round 2.)

    def _forward_keyed(self, x_cf, attrs, key):
        groups = row_groups(key[0], key[1])
        v = self._spec.v
        a_rep = attrs.index_select(0, groups.reps)                     # [K, V] one row per key
        m_off, pos = [], 0
        for ins in self._spec.instr:
            m_off.append(pos)
            pos += ins.mul_in * ins.mul_out
        # M[t] = sum_v attrs_t[v] W[:, v, :], one launch reading the e3nn-ordered ('uvw') flat weight in place
        m = ops.keyed_weights(a_rep, self.weight, self._spec, m_off, pos)
        return ops.grouped_linear(x_cf, m, groups, self._spec, m_off)


class Gate(nn.Module):
    """``e3nn.nn.Gate``; consumes the conv output in cf layout, emits the e3nn layout."""

    def __init__(self, irreps_scalars, act_scalars, irreps_gates, act_gates, irreps_gated):
        super().__init__()
        sc, gt, gd = Irreps(irreps_scalars), Irreps(irreps_gates), Irreps(irreps_gated)
        if gt.num_irreps != gd.num_irreps:
            raise ValueError(f"{gt.num_irreps} gates for {gd.num_irreps} gated irreps")
        if any(mi.ir.l != 0 for mi in sc) or any(mi.ir.l != 0 for mi in gt):
            raise ValueError("scalars and gates must be l=0")
        a_sc = [activation_name(a) for a in act_scalars]
        a_gt = [activation_name(a) for a in act_gates]
        self.irreps_in = (sc + gt + gd).simplify()
        out_sc = Irreps([(mi.mul, Irrep(0, act_output_parity(a, mi.ir.p))) for mi, a in zip(sc, a_sc)])
        for mi, a in zip(gt, a_gt):
            if act_output_parity(a, mi.ir.p) != 1:
                raise ValueError("gate activations must produce even scalars")
        self.irreps_out = out_sc + gd
        # flat list of gate channels, one activation per gates entry
        gate_chan = []  # (offset in input row, act, cst) per gates entry
        pos = 0
        segs = []
        out_pos = 0
        for mi, a in zip(sc, a_sc):
            segs.append((0, pos, 0, out_pos, mi.mul, 1, ops.ACT_IDS[a], act_second_moment_const(a)))
            pos += mi.mul
            out_pos += mi.mul
        gates_start = pos
        gate_entries = []
        for mi, a in zip(gt, a_gt):
            gate_entries.append((pos, mi.mul, a))
            pos += mi.mul
        # walk gated entries, consuming gate channels in order (entries must line up one to one)
        if [g[1] for g in gate_entries] != [mi.mul for mi in gd]:
            raise NotImplementedError("gates entries must match the gated entries one to one")
        for (g_off, g_mul, a), mi in zip(gate_entries, gd):
            segs.append((1, pos, g_off, out_pos, mi.mul, mi.ir.dim, ops.ACT_IDS[a], act_second_moment_const(a)))
            pos += mi.dim
            out_pos += mi.dim
        # the simplified input irreps must not merge gated blocks (cf block stride == mul)
        simp_gd = gd.simplify()
        if len(simp_gd) != len(gd):
            raise NotImplementedError("adjacent gated entries with the same irrep are not supported by the gate kernel")
        self._spec = ops.GateSpec(self.irreps_in.dim, self.irreps_out.dim, segs)
        assert pos == self.irreps_in.dim and out_pos == self.irreps_out.dim

    def forward(self, x_cf, out_cf: bool = False):
        return ops.gate(x_cf, self._spec, out_cf)


class NormActivation(nn.Module):
    """``e3nn.nn.NormActivation(irreps_in, scalar_nonlinearity, normalize=True, epsilon=None, bias=False)`` — the
    ``nonlinearity_type="norm"`` branch of MessagePassing (``e3_layers/nn/message_passing.py:212-219``).  Like ``Gate``
    it consumes the convolution output in cf layout and emits the e3nn layout."""

    def __init__(self, irreps_in, scalar_nonlinearity, normalize: bool = True, epsilon: Optional[float] = None,
                 bias: bool = False):
        super().__init__()
        if bias:
            raise NotImplementedError("NormActivation(bias=True) is not built (the reference passes bias=False)")
        self.irreps_in = Irreps(irreps_in)
        self.irreps_out = Irreps(irreps_in)
        if epsilon is None and normalize:
            epsilon = 1e-8
        elif epsilon is not None and not normalize:
            raise ValueError("epsilon and normalize = False don't make sense together")
        elif not normalize:
            epsilon = 0.0
        self.epsilon, self.normalize = float(epsilon), bool(normalize)
        self.act = activation_name(scalar_nonlinearity)
        self._blocks = tuple((off, mi.mul, mi.ir.dim) for off, mi in zip(self.irreps_in.offsets(), self.irreps_in))

    def forward(self, x_cf):
        return ops.norm_activation(x_cf, self._blocks, self.act, self.epsilon, self.normalize)


def tp_slots(l1: int) -> List[Tuple[int, int]]:
    """Valid (l2, l3) pairs for an input of degree l1, in the slot order of e3k_cg_gen.h."""
    return [(l2, l3) for l2 in range(TP_L2MAX + 1) for l3 in range(abs(l1 - l2), min(l1 + l2, TP_L3MAX) + 1)]


class UVUTensorProduct(nn.Module):
    """Weighted 'uvu' product ``left (x) right`` with one output slot per path — the
    ``o3.TensorProduct`` that ``TensorProductExpansion`` builds (``e3_layers/nn/pointwise.py:61-85``):
    paths (i, j, ir_out in ir_i*ir_j if ir_out in output) in i-major order, outputs sorted by
    irrep, per-sample weights consumed in path order (mul_i * mul_j each).

    The kernel writes the *merged* cf layout: all paths ending in the same irrep are adjacent
    channels of one block ``[2l+1][K_ir]`` — exactly the input the trailing ``Linear(mid.simplify()
    -> output)`` expects."""

    def __init__(self, irreps_left, irreps_right, irreps_output_filter):
        super().__init__()
        left, right, flt = Irreps(irreps_left), Irreps(irreps_right), Irreps(irreps_output_filter)
        paths = []  # (i, j, ir_out, mul)
        for i, a in enumerate(left):
            for j, b in enumerate(right):
                for ir in a.ir * b.ir:
                    if ir in flt:
                        if b.mul != 1:
                            raise NotImplementedError("the fused uvu kernel needs multiplicity-1 right operands (edge SH)")
                        paths.append((i, j, ir, a.mul))
        mid = Irreps([(mul, ir) for _, _, ir, mul in paths])
        self.irreps_mid, perm, _ = mid.sort()
        self.irreps_mid_simplified = self.irreps_mid.simplify()
        self.irreps_in1, self.irreps_in2 = left, right
        self.weight_numel = sum(mul for *_, mul in paths)
        self.paths = paths
        # channel offset of each path inside its merged block
        simp = self.irreps_mid_simplified
        simp_off = {mi.ir: off for mi, off in zip(simp, simp.offsets())}
        simp_mul = {mi.ir: mi.mul for mi in simp}
        if len(simp_off) != len(simp):
            raise AssertionError("sorted mid irreps did not merge into unique blocks")
        order = sorted(range(len(paths)), key=lambda k: perm[k])
        chan_off = {}
        running: Dict[Irrep, int] = {}
        for k in order:
            ir = paths[k][2]
            chan_off[k] = running.get(ir, 0)
            running[ir] = chan_off[k] + paths[k][3]
        # kernel groups
        x_off, y_off = left.offsets(), right.offsets()
        groups: List[L.TpGroup] = []
        w_off = 0
        open_group: Dict[int, L.TpGroup] = {}
        for k, (i, j, ir, mul) in enumerate(paths):
            l1, l2, l3 = left[i].ir.l, right[j].ir.l, ir.l
            if l1 > TP_L1MAX or l2 > TP_L2MAX or l3 > TP_L3MAX:
                raise NotImplementedError(
                    f"path {left[i].ir} x {right[j].ir} -> {ir} exceeds the compiled CG tables "
                    f"(l1<={TP_L1MAX}, l2<={TP_L2MAX}, l3<={TP_L3MAX}); regenerate with tools/gen_cg.py")
            q = tp_slots(l1).index((l2, l3))
            g = open_group.get(i)
            if g is None or (g.mask >> q) & 1 or (g.y_off[l2] >= 0 and g.y_off[l2] != y_off[j]):
                g = L.TpGroup()
                g.l1, g.x_off, g.mul, g.mask = l1, x_off[i], mul, 0
                for t in range(4):
                    g.y_off[t] = -1
                groups.append(g)
                open_group[i] = g
            g.mask |= 1 << q
            g.y_off[l2] = y_off[j]
            g.w_off[q] = w_off
            g.out_off[q] = simp_off[ir] + chan_off[k]
            g.out_stride[q] = simp_mul[ir]
            g.coeff[q] = math.sqrt(ir.dim / right[j].mul)
            w_off += mul
        self.plan = ops.TpPlan(groups, left.dim, right.dim, self.weight_numel, simp.dim) if groups else None
        self.d_mid = simp.dim

    def fused(self, x_cf, sh, weight, topo: GraphTopo):
        """x_cf [N, left.dim] (cf), sh [E, right.dim], weight [E, weight_numel] -> mid [N, d_mid] (merged cf),
        summed over the in-edges of every node."""
        if self.plan is None:
            return x_cf.new_zeros(x_cf.shape[0], 0)
        return ops.tp_uvu_scatter(x_cf, sh, weight, topo, self.plan)
