"""ORACLE — CPU restatement of the reference's tensor-product message-passing path.

*** TEST INFRASTRUCTURE ONLY ***  Nothing under ``oracle/`` is imported by the product
package (``equivariant-nn-zoo_amd/``).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it — as the checker / the timed CPU baseline,
never as the thing shipped.

*** PARITY UNPINNED ***  The reference (20171130/Equivariant-NN-Zoo) computes this path
through ``e3nn==0.4.4`` and ``torch-runstats==0.2.0`` (``requirements.txt:27,145``), neither of
which is vendored in ``/root/reference`` nor installable here, and the reference ships no
tests or golden vectors.  This file therefore restates the
published e3nn-0.4.4 operator semantics in plain PyTorch, op for op
in the reference's *unfused* structure (materialised gather, per-path einsum, per-edge
Linear, ``index_add_`` scatter).  It is pinned by closed-form known answers and group
theoretic invariants in ``tests/test_oracle_*.py``, not by reference outputs.

Each function cites the reference call site it stands in for (paths relative to
``/root/reference``).  dtype follows the inputs: float64 for checking, float32 for the
timed CPU baseline.
"""
from __future__ import annotations

import math
import re
from collections import OrderedDict
from fractions import Fraction
from functools import lru_cache, partial
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
from torch import Tensor, nn

# --------------------------------------------------------------------------------------
# Irreps (just enough; e3nn.o3.Irreps stand-in — SURVEY.md A.6)
# --------------------------------------------------------------------------------------


def parse_irreps(spec) -> List[Tuple[int, int, int]]:
    """``"64x0e+64x1o"`` -> ``[(64, 0, +1), (64, 1, -1)]``.  Accepts str, list of
    ``(mul, (l, p))`` / ``(mul, "1o")`` pairs or anything whose ``str()`` is an irreps string."""
    if isinstance(spec, (list, tuple)):
        out = []
        for item in spec:
            if isinstance(item, (list, tuple)) and len(item) == 3 and all(isinstance(v, int) for v in item):
                out.append(tuple(item))
            else:
                mul, ir = item
                if isinstance(ir, str):
                    m = re.match(r"^(\d+)([eo])$", ir.strip())
                    out.append((int(mul), int(m.group(1)), 1 if m.group(2) == "e" else -1))
                else:
                    out.append((int(mul), int(ir[0]), int(ir[1])))
        return out
    text = str(spec).strip()
    out = []
    if not text:
        return out
    for chunk in text.split("+"):
        chunk = chunk.strip()
        mul, ir = chunk.split("x") if "x" in chunk else ("1", chunk)
        m = re.match(r"^(\d+)([eo])$", ir.strip())
        if m is None:
            raise ValueError(f"bad irreps chunk {chunk!r}")
        out.append((int(mul), int(m.group(1)), 1 if m.group(2) == "e" else -1))
    return out


def irreps_str(irreps) -> str:
    return "+".join(f"{mul}x{l}{'e' if p == 1 else 'o'}" for mul, l, p in irreps)


def irreps_dim(irreps) -> int:
    return sum(mul * (2 * l + 1) for mul, l, _ in parse_irreps(irreps))


def irreps_slices(irreps) -> List[Tuple[int, int]]:
    out, pos = [], 0
    for mul, l, _ in irreps:
        out.append((pos, pos + mul * (2 * l + 1)))
        pos += mul * (2 * l + 1)
    return out


def irreps_simplify(irreps):
    out = []
    for mul, l, p in irreps:
        if mul == 0:
            continue
        if out and out[-1][1:] == (l, p):
            out[-1] = (out[-1][0] + mul, l, p)
        else:
            out.append((mul, l, p))
    return out


def irreps_sort(irreps):
    """Stable sort by (l, p): the e3nn-0.4.4 ``Irreps.sort``.  Returns
    (sorted, p) with p[old] = new."""
    order = sorted(range(len(irreps)), key=lambda i: (irreps[i][1], irreps[i][2], i))
    p = [0] * len(irreps)
    for new, old in enumerate(order):
        p[old] = new
    return [irreps[i] for i in order], p


def ir_product(l1, p1, l2, p2):
    return [(l, p1 * p2) for l in range(abs(l1 - l2), l1 + l2 + 1)]


def tp_path_exists(irreps_in1, irreps_in2, ir_out) -> bool:
    """``e3_layers/utils/utils.py:87-96``."""
    a = irreps_simplify(parse_irreps(irreps_in1))
    b = irreps_simplify(parse_irreps(irreps_in2))
    if isinstance(ir_out, str):
        _, lo, po = parse_irreps("1x" + ir_out)[0]
    else:
        lo, po = ir_out
    for _, l1, p1 in a:
        for _, l2, p2 in b:
            if (lo, po) in ir_product(l1, p1, l2, p2):
                return True
    return False


# --------------------------------------------------------------------------------------
# Wigner 3j (e3nn.o3.wigner_3j — SURVEY.md A.2; reference call site nn/output.py:172)
# --------------------------------------------------------------------------------------


def _fact(n: int) -> int:
    return math.factorial(int(round(n)))


def _su2_cg_entry(j1, m1, j2, m2, j3, m3) -> float:
    if m3 != m1 + m2:
        return 0.0
    lo = int(max(-j1 + j2 + m3, -j1 + m1, 0))
    hi = int(min(j2 + j3 + m1, j3 - j1 + j2, j3 + m3))
    c2 = (2.0 * j3 + 1.0) * Fraction(
        _fact(j3 + j1 - j2) * _fact(j3 - j1 + j2) * _fact(j1 + j2 - j3) * _fact(j3 + m3) * _fact(j3 - m3),
        _fact(j1 + j2 + j3 + 1) * _fact(j1 - m1) * _fact(j1 + m1) * _fact(j2 - m2) * _fact(j2 + m2),
    )
    s = Fraction(0)
    for v in range(lo, hi + 1):
        s += (-1) ** int(v + j2 + m2) * Fraction(
            _fact(j2 + j3 + m1 - v) * _fact(j1 - m1 + v),
            _fact(v) * _fact(j3 - j1 + j2 - v) * _fact(j3 + m3 - v) * _fact(v + j1 - j2 - m3),
        )
    return float(c2) ** 0.5 * float(s)


def _q_real_to_complex(l: int) -> Tensor:
    q = torch.zeros(2 * l + 1, 2 * l + 1, dtype=torch.complex128)
    r = 2 ** -0.5
    for m in range(-l, l + 1):
        if m < 0:
            q[l + m, l - m] = r
            q[l + m, l + m] = -1j * r
        elif m == 0:
            q[l, l] = 1.0
        else:
            q[l + m, l + m] = (-1) ** m * r
            q[l + m, l - m] = 1j * (-1) ** m * r
    return (-1j) ** l * q


@lru_cache(maxsize=None)
def wigner_3j(l1: int, l2: int, l3: int) -> Tensor:
    """Real 3j tensor, float64, Frobenius norm 1."""
    assert abs(l1 - l2) <= l3 <= l1 + l2
    c = torch.zeros(2 * l1 + 1, 2 * l2 + 1, 2 * l3 + 1, dtype=torch.complex128)
    for m1 in range(-l1, l1 + 1):
        for m2 in range(-l2, l2 + 1):
            if abs(m1 + m2) <= l3:
                c[l1 + m1, l2 + m2, l3 + m1 + m2] = _su2_cg_entry(l1, m1, l2, m2, l3, m1 + m2)
    q1, q2, q3 = _q_real_to_complex(l1), _q_real_to_complex(l2), _q_real_to_complex(l3)
    c = torch.einsum("ij,kl,mn,ikn->jlm", q1, q2, torch.conj(q3.T), c)
    assert c.imag.abs().max() < 1e-9
    c = c.real.contiguous()
    return c / c.norm()


# --------------------------------------------------------------------------------------
# Spherical harmonics (o3.SphericalHarmonics — SURVEY.md A.3; call site nn/embedding.py:163-178)
# --------------------------------------------------------------------------------------


def spherical_harmonics(ls: Sequence[int], vec: Tensor, normalize: bool = True, normalization: str = "component") -> Tensor:
    """Real SH of ``vec[..., 3]`` for the listed degrees (l <= 3), concatenated on the last dim."""
    if normalize:
        vec = torch.nn.functional.normalize(vec, dim=-1)  # v / max(|v|, 1e-12)
    x, y, z = vec[..., 0], vec[..., 1], vec[..., 2]
    blocks = []
    y2 = None
    for l in ls:
        if l == 0:
            sh = torch.ones_like(x).unsqueeze(-1)
        elif l == 1:
            sh = math.sqrt(3.0) * torch.stack([x, y, z], dim=-1)
        elif l in (2, 3):
            x2, yy, z2 = x * x, y * y, z * z
            s15, s5 = math.sqrt(15.0), math.sqrt(5.0)
            y2 = [s15 * x * z, s15 * x * y, s5 * (yy - 0.5 * (x2 + z2)), s15 * y * z, 0.5 * s15 * (z2 - x2)]
            if l == 2:
                sh = torch.stack(y2, dim=-1)
            else:
                x2z2 = x2 + z2
                a, b, s7 = math.sqrt(42.0) / 6.0, math.sqrt(168.0) / 8.0, math.sqrt(7.0)
                sh = torch.stack(
                    [
                        a * (y2[0] * z + y2[4] * x),
                        s7 * y2[0] * y,
                        b * (4.0 * yy - x2z2) * x,
                        0.5 * s7 * y * (2.0 * yy - 3.0 * x2z2),
                        b * z * (4.0 * yy - x2z2),
                        s7 * y2[4] * y,
                        a * (y2[4] * z - y2[0] * x),
                    ],
                    dim=-1,
                )
        else:
            raise NotImplementedError("oracle SH restated for l <= 3 only")
        if normalization == "integral":
            sh = sh / math.sqrt(4.0 * math.pi)
        elif normalization == "norm":
            sh = sh / math.sqrt(2 * l + 1)
        elif normalization != "component":
            raise ValueError(normalization)
        blocks.append(sh)
    return torch.cat(blocks, dim=-1)


# --------------------------------------------------------------------------------------
# Activations + second-moment normalisation (e3nn.math.normalize2mom — SURVEY.md A.5;
# table e3_layers/utils/utils.py:64-84)
# --------------------------------------------------------------------------------------


def _ssp(x):
    return torch.nn.functional.softplus(x) - math.log(2.0)


def _tanhlu(x):
    return torch.tanh(x) * torch.abs(x)


ACTIVATIONS: Dict[str, Callable] = {
    "abs": torch.abs,
    "tanh": torch.tanh,
    "ssp": _ssp,
    "silu": torch.nn.functional.silu,
    "tanhlu": _tanhlu,
}


@lru_cache(maxsize=None)
def act_norm_const(name: str) -> float:
    """``(E_{z~N(0,1)} act(z)^2)^(-1/2)`` estimated exactly as e3nn does: 1e6 float64 normal
    samples from a CPU generator seeded with 0."""
    gen = torch.Generator(device="cpu").manual_seed(0)
    z = torch.randn(1_000_000, generator=gen, dtype=torch.float64)
    c = ACTIVATIONS[name](z).pow(2).mean().pow(-0.5).item()
    return 1.0 if abs(c - 1.0) < 1e-4 else c


def normalized_act(name: str) -> Callable:
    c = act_norm_const(name)
    f = ACTIVATIONS[name]
    return lambda x: f(x) * c


def act_parity(name: str, p_in: int) -> int:
    """Parity of act(x) for an input scalar of parity p_in (e3nn.nn.Activation rule)."""
    if p_in == 1:
        return 1
    x = torch.linspace(0.0, 10.0, 256, dtype=torch.float64)
    f = ACTIVATIONS[name]
    a, b = f(x), f(-x)
    if (a - b).abs().max() < 1e-10:
        return 1
    if (a + b).abs().max() < 1e-10:
        return -1
    raise ValueError(f"activation {name} is neither even nor odd: cannot act on an odd scalar")


# --------------------------------------------------------------------------------------
# scatter (torch_runstats.scatter.scatter — SURVEY.md A.7; call sites
# nn/message_passing.py:109, nn/output.py:69)
# --------------------------------------------------------------------------------------


def scatter(src: Tensor, index: Tensor, dim_size: Optional[int] = None, reduce: str = "sum") -> Tensor:
    if dim_size is None:
        dim_size = int(index.max().item()) + 1 if index.numel() else 0
    out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    out.index_add_(0, index, src)
    if reduce == "mean":
        cnt = torch.zeros(dim_size, dtype=src.dtype, device=src.device)
        cnt.index_add_(0, index, torch.ones_like(index, dtype=src.dtype))
        out = out / cnt.clamp(min=1).view((-1,) + (1,) * (src.dim() - 1))
    elif reduce != "sum":
        raise ValueError(reduce)
    return out


# --------------------------------------------------------------------------------------
# o3.Linear (call sites nn/message_passing.py:58, nn/pointwise.py:18,87,142)
# --------------------------------------------------------------------------------------


class Linear(nn.Module):
    def __init__(self, irreps_in, irreps_out, biases: bool = False):
        super().__init__()
        self.irreps_in = parse_irreps(irreps_in)
        self.irreps_out = parse_irreps(irreps_out)
        self.instr = [
            (i, o)
            for i, (_, li, pi) in enumerate(self.irreps_in)
            for o, (_, lo, po) in enumerate(self.irreps_out)
            if (li, pi) == (lo, po)
        ]
        fan = [0] * len(self.irreps_out)
        for i, o in self.instr:
            fan[o] += self.irreps_in[i][0]
        self.pw = [1.0 / math.sqrt(f) if f > 0 else 0.0 for f in fan]
        numel = sum(self.irreps_in[i][0] * self.irreps_out[o][0] for i, o in self.instr)
        self.weight = nn.Parameter(torch.randn(numel))
        self.bias_blocks = [o for o, (_, l, p) in enumerate(self.irreps_out) if biases and l == 0 and p == 1]
        nb = sum(self.irreps_out[o][0] for o in self.bias_blocks)
        if nb > 0:
            self.bias = nn.Parameter(torch.zeros(nb))
        else:
            self.register_parameter("bias", None)

    def forward(self, x: Tensor) -> Tensor:
        z = x.shape[0]
        sl_in, sl_out = irreps_slices(self.irreps_in), irreps_slices(self.irreps_out)
        # e3nn's generated forward reshapes to (-1, irreps_in.dim): a narrower input is an error, not a silent mis-slice
        assert x.shape[-1] == (sl_in[-1][1] if sl_in else 0), (tuple(x.shape), sl_in[-1][1] if sl_in else 0)
        outs = [None] * len(self.irreps_out)
        pos = 0
        for i, o in self.instr:
            mi, l, _ = self.irreps_in[i]
            mo = self.irreps_out[o][0]
            w = self.weight[pos : pos + mi * mo].view(mi, mo)
            pos += mi * mo
            xi = x[:, sl_in[i][0] : sl_in[i][1]].reshape(z, mi, 2 * l + 1)
            y = torch.einsum("uw,zum->zwm", w, xi) * self.pw[o]
            outs[o] = y if outs[o] is None else outs[o] + y
        bpos = 0
        for o in self.bias_blocks:
            mo = self.irreps_out[o][0]
            b = self.bias[bpos : bpos + mo].view(1, mo, 1)
            bpos += mo
            outs[o] = b.expand(z, mo, 1) if outs[o] is None else outs[o] + b
        cols = []
        for o, (mo, l, _) in enumerate(self.irreps_out):
            if outs[o] is None:
                cols.append(x.new_zeros(z, mo * (2 * l + 1)))
            else:
                cols.append(outs[o].reshape(z, mo * (2 * l + 1)))
        return torch.cat(cols, dim=1) if cols else x.new_zeros(z, 0)


# --------------------------------------------------------------------------------------
# o3.TensorProduct / FullyConnectedTensorProduct (call sites
# nn/pointwise.py:78-85, nn/message_passing.py:83-87)
# --------------------------------------------------------------------------------------


class TensorProduct(nn.Module):
    """instructions: (i_in1, i_in2, i_out, mode) with mode in {"uvu", "uvw"}; all weighted.
    ``internal_weights=True`` -> one shared parameter vector; else per-sample weights are
    passed to forward.  Normalisation: component / element (e3nn defaults)."""

    def __init__(self, irreps_in1, irreps_in2, irreps_out, instructions, internal_weights: bool):
        super().__init__()
        self.in1, self.in2, self.out = parse_irreps(irreps_in1), parse_irreps(irreps_in2), parse_irreps(irreps_out)
        self.instr = [tuple(ins[:4]) for ins in instructions]
        self.internal_weights = internal_weights
        shapes = []
        for i1, i2, io, mode in self.instr:
            m1, m2, mo = self.in1[i1][0], self.in2[i2][0], self.out[io][0]
            if mode == "uvu":
                assert mo == m1
                shapes.append((m1, m2))
            elif mode == "uvw":
                shapes.append((m1, m2, mo))
            else:
                raise NotImplementedError(mode)
        self.shapes = shapes
        self.weight_numel = sum(math.prod(s) for s in shapes)

        def n_el(ins):
            i1, i2, _, mode = ins
            return self.in2[i2][0] if mode == "uvu" else self.in1[i1][0] * self.in2[i2][0]

        self.coeff = []
        for ins in self.instr:
            lo = self.out[ins[2]][1]
            denom = sum(n_el(other) for other in self.instr if other[2] == ins[2])
            self.coeff.append(math.sqrt((2 * lo + 1) / denom))
        if internal_weights:
            self.weight = nn.Parameter(torch.randn(self.weight_numel))

    def forward(self, x1: Tensor, x2: Tensor, weight: Optional[Tensor] = None) -> Tensor:
        z = x1.shape[0]
        s1, s2 = irreps_slices(self.in1), irreps_slices(self.in2)
        outs = [None] * len(self.out)
        if self.internal_weights:
            weight = self.weight
        pos = 0
        for ins, shape, coeff in zip(self.instr, self.shapes, self.coeff):
            i1, i2, io, mode = ins
            m1, l1, _ = self.in1[i1]
            m2, l2, _ = self.in2[i2]
            mo, lo, _ = self.out[io]
            n = math.prod(shape)
            if self.internal_weights:
                w = weight[pos : pos + n].view(shape)
            else:
                w = weight[:, pos : pos + n].reshape((z,) + shape)
            pos += n
            a = x1[:, s1[i1][0] : s1[i1][1]].reshape(z, m1, 2 * l1 + 1)
            b = x2[:, s2[i2][0] : s2[i2][1]].reshape(z, m2, 2 * l2 + 1)
            c = wigner_3j(l1, l2, lo).to(x1.dtype)
            # materialised outer products x1 (x) x2 -> contraction with the 3j tensor as one GEMM
            # (the contraction order opt_einsum_fx picks for e3nn's "zuv,ijk,zuvij->zuk")
            outer = torch.einsum("zui,zvj->zuvij", a, b)
            d1, d2, d3 = 2 * l1 + 1, 2 * l2 + 1, 2 * lo + 1
            t = (outer.reshape(z * m1 * m2, d1 * d2) @ c.reshape(d1 * d2, d3)).reshape(z, m1, m2, d3)
            if mode == "uvu":
                if self.internal_weights:
                    r = torch.einsum("uv,zuvk->zuk", w, t)
                else:
                    r = torch.einsum("zuv,zuvk->zuk", w, t)
            else:
                if self.internal_weights:
                    r = torch.einsum("uvw,zuvk->zwk", w, t)
                else:
                    r = torch.einsum("zuvw,zuvk->zwk", w, t)
            r = r * coeff
            outs[io] = r if outs[io] is None else outs[io] + r
        cols = []
        for io, (mo, lo, _) in enumerate(self.out):
            cols.append(x1.new_zeros(z, mo * (2 * lo + 1)) if outs[io] is None else outs[io].reshape(z, mo * (2 * lo + 1)))
        return torch.cat(cols, dim=1)


class FullyConnectedTensorProduct(TensorProduct):
    def __init__(self, irreps_in1, irreps_in2, irreps_out):
        in1, in2, out = parse_irreps(irreps_in1), parse_irreps(irreps_in2), parse_irreps(irreps_out)
        instr = [
            (i1, i2, io, "uvw")
            for i1, (_, l1, p1) in enumerate(in1)
            for i2, (_, l2, p2) in enumerate(in2)
            for io, (_, lo, po) in enumerate(out)
            if (lo, po) in ir_product(l1, p1, l2, p2)
        ]
        super().__init__(in1, in2, out, instr, internal_weights=True)


# --------------------------------------------------------------------------------------
# e3nn.nn.FullyConnectedNet / Gate (call sites nn/message_passing.py:74,195)
# --------------------------------------------------------------------------------------


class _FCLayer(nn.Module):
    def __init__(self, h_in, h_out, act_name):
        super().__init__()
        self.h_in = h_in
        self.act = normalized_act(act_name) if act_name else None
        self.weight = nn.Parameter(torch.randn(h_in, h_out))

    def forward(self, x):
        x = x @ (self.weight / math.sqrt(self.h_in))
        return self.act(x) if self.act is not None else x


class FullyConnectedNet(nn.Sequential):
    def __init__(self, hs: Sequence[int], act_name: str):
        layers = OrderedDict()
        for i, (a, b) in enumerate(zip(hs[:-1], hs[1:])):
            layers[f"layer{i}"] = _FCLayer(a, b, act_name if i < len(hs) - 2 else None)
        super().__init__(layers)
        self.hs = list(hs)


class Gate(nn.Module):
    """input ``[scalars | gates | gated]`` -> ``[act(scalars) | gated * act(gates)]``."""

    def __init__(self, irreps_scalars, act_scalars, irreps_gates, act_gates, irreps_gated):
        super().__init__()
        self.sc, self.gt, self.gd = parse_irreps(irreps_scalars), parse_irreps(irreps_gates), parse_irreps(irreps_gated)
        assert sum(m for m, _, _ in self.gt) == sum(m for m, _, _ in self.gd)
        assert all(l == 0 for _, l, _ in self.sc + self.gt)
        self.act_sc = [normalized_act(a) for a in act_scalars]
        self.act_gt = [normalized_act(a) for a in act_gates]
        self.irreps_in = irreps_simplify(self.sc + self.gt + self.gd)
        sc_out = [(m, 0, act_parity(a, p)) for (m, _, p), a in zip(self.sc, act_scalars)]
        for (m, _, p), a in zip(self.gt, act_gates):
            assert act_parity(a, p) == 1, "gates must come out even"
        self.irreps_out = sc_out + list(self.gd)

    def forward(self, x: Tensor) -> Tensor:
        z = x.shape[0]
        pos, cols = 0, []
        for (m, _, _), act in zip(self.sc, self.act_sc):
            cols.append(act(x[:, pos : pos + m]))
            pos += m
        gates = []
        for (m, _, _), act in zip(self.gt, self.act_gt):
            gates.append(act(x[:, pos : pos + m]))
            pos += m
        gates = torch.cat(gates, dim=1) if gates else x.new_zeros(z, 0)
        gpos = 0
        for m, l, _ in self.gd:
            blk = x[:, pos : pos + m * (2 * l + 1)].reshape(z, m, 2 * l + 1)
            cols.append((blk * gates[:, gpos : gpos + m].unsqueeze(-1)).reshape(z, m * (2 * l + 1)))
            pos += m * (2 * l + 1)
            gpos += m
        assert pos == x.shape[1]
        return torch.cat(cols, dim=1)


# --------------------------------------------------------------------------------------
# Layer modules (dict-in / dict-out protocol of nn/sequential.py:12-39,70-88)
# --------------------------------------------------------------------------------------


def _split_spec(value):
    """``irreps`` or ``(irreps, custom_key)`` -> (irreps_str, custom_key|None)."""
    if isinstance(value, (list, tuple)) and len(value) == 2 and not isinstance(value[0], (list, tuple, int)):
        return str(value[0]), value[1]
    return str(value), None


class OModule(nn.Module):
    """Key-mapping protocol (``Module.init_irreps``, nn/sequential.py:13-39)."""

    def init_irreps(self, output_keys=(), **kw):
        if isinstance(output_keys, str):
            output_keys = [output_keys]
        self.irreps_in, self.irreps_out = {}, {}
        self.in_map, self.out_map = {}, {}
        for key, value in kw.items():
            if value is None:
                continue
            irreps, custom = _split_spec(value)
            custom = custom if custom is not None else key
            if key in output_keys:
                self.irreps_out[key] = irreps
                self.out_map[key] = custom
            else:
                self.irreps_in[key] = irreps
                self.in_map[custom] = key


def _remap(d: dict, mapping: dict) -> dict:
    out = {}
    for k, v in d.items():
        if k in mapping:
            out[mapping[k]] = v
        else:
            out[k] = v
    return out


def compute_edge_vector(data: dict, attrs: dict, key: str = "pos"):
    """``computeEdgeVector`` (data/compute_edge.py:13-36): edge_vec = pos[dst] - pos[src]."""
    attrs["edge_vector"] = ("edge", "1x1o")
    attrs["edge_length"] = ("edge", "1x0e")
    if "edge_vector" not in data:
        pos, ei = data[key], data["edge_index"]
        data["edge_vector"] = pos[ei[1]] - pos[ei[0]]
    if "edge_length" not in data:
        data["edge_length"] = torch.linalg.norm(data["edge_vector"], dim=-1)
    return data, attrs


def compute_edge_index(data: dict, attrs: dict, r_max: float, key: str = "pos", criteria=None):
    """``computeEdgeIndex`` (data/compute_edge.py:38-113), intent of SURVEY.md appendix C:
    per graph all ordered pairs, src-major / dst-minor, keep ``|pos_src - pos_dst| < r_max``
    (strict, in the dtype of pos) or ``criteria``; drop self loops; pre-existing edges are kept
    and their edge attributes carried over (zero rows for new edges).  Returns the new
    ``edge_index`` and writes ``_n_edges`` into data."""
    pos = data[key]
    n_nodes = [int(v) for v in data["_n_nodes"].view(-1).tolist()]
    src_l, dst_l, start = [], [], 0
    for n in n_nodes:
        ids = torch.arange(start, start + n, dtype=torch.long)
        src_l.append(ids.repeat_interleave(n))
        dst_l.append(ids.repeat(n))
        start += n
    cand = torch.stack([torch.cat(src_l), torch.cat(dst_l)]).to(pos.device)
    dist = torch.linalg.norm(pos[cand[0]] - pos[cand[1]], dim=-1)
    keep = dist < r_max
    if criteria is not None:
        keep = keep | criteria(data, cand)
    keep = keep & (cand[0] != cand[1])
    total = start
    if "edge_index" in data:
        old = data["edge_index"]
        old_flat = old[0] * total + old[1]
        cand_flat = cand[0] * total + cand[1]
        pos_in_cand = torch.searchsorted(cand_flat, old_flat)
        assert bool((cand_flat[pos_in_cand] == old_flat).all()), "existing edge crosses graphs"
        keep[pos_in_cand] = True
    new = cand[:, keep]
    if "edge_index" in data:
        new_flat = new[0] * total + new[1]
        where = torch.searchsorted(new_flat, old_flat)
        for k in list(attrs):
            if attrs[k][0] == "edge" and k in data:
                old_val = data[k]
                fresh = torch.zeros((new.shape[1],) + tuple(old_val.shape[1:]), dtype=old_val.dtype, device=pos.device)
                fresh[where] = old_val
                data[k] = fresh
    seg = torch.repeat_interleave(torch.arange(len(n_nodes)), torch.tensor(n_nodes))
    n_edges = torch.bincount(seg[new[0].cpu()], minlength=len(n_nodes)).view(-1, 1)
    attrs["_n_edges"] = ("graph", "1x0e")
    data["_n_edges"] = n_edges
    return {"edge_index": new}, attrs


class OneHotEncoding(OModule):
    """nn/embedding.py:258-281."""

    def __init__(self, num_types, irreps_out, irreps_in="0x0e"):
        super().__init__()
        self.num_types = num_types
        self.init_irreps(input=irreps_in, one_hot=irreps_out, output_keys="one_hot")

    def forward(self, data, attrs):
        idx = data["input"].squeeze(-1)
        ref = next((v for v in data.values() if torch.is_floating_point(v)), None)
        dtype = ref.dtype if ref is not None else torch.get_default_dtype()
        oh = torch.nn.functional.one_hot(idx, num_classes=self.num_types).to(dtype)
        return {"one_hot": oh}, {"one_hot": (attrs["input"][0], self.irreps_out["one_hot"])}


class PointwiseLinear(OModule):
    """nn/pointwise.py:14-30."""

    def __init__(self, irreps_in, irreps_out, biases=True):
        super().__init__()
        self.init_irreps(input=irreps_in, output=irreps_out, output_keys=["output"])
        self.linear = Linear(self.irreps_in["input"], self.irreps_out["output"], biases=biases)

    def forward(self, data, attrs):
        return {"output": self.linear(data["input"])}, {"output": (attrs["input"][0], self.irreps_out["output"])}


class Concat(OModule):
    """nn/pointwise.py:134-152."""

    def __init__(self, irreps_out, **irreps_in):
        super().__init__()
        self.init_irreps(**irreps_in, output=irreps_out, output_keys=["output"])
        cat = []
        for v in self.irreps_in.values():
            cat += parse_irreps(v)
        self.linear = Linear(cat, self.irreps_out["output"], biases=True)

    def forward(self, data, attrs):
        x = torch.cat([data[k] for k in self.irreps_in], dim=1)
        first = next(iter(self.irreps_in))
        return {"output": self.linear(x)}, {"output": (attrs[first][0], self.irreps_out["output"])}


class LayerNormalization(OModule):
    """nn/pointwise.py:32-51: per-irreps-block RMS normalisation."""

    def __init__(self, irreps_in, irreps_out):
        super().__init__()
        self.init_irreps(input=irreps_in, output=irreps_out, output_keys=["output"])
        self.blocks = parse_irreps(self.irreps_in["input"])
        self.std = nn.Parameter(torch.ones(len(self.blocks)))

    def forward(self, data, attrs):
        x = data["input"]
        cols = []
        for i, ((a, b), (mul, _, _)) in enumerate(zip(irreps_slices(self.blocks), self.blocks)):
            t = x[:, a:b]
            nrm = ((t * t).sum(dim=-1, keepdim=True) / mul + 1e-6) ** 0.5
            cols.append(t / nrm * self.std[i])
        return {"output": torch.cat(cols, dim=1)}, attrs


class SphericalEncoding(OModule):
    """nn/embedding.py:131-178."""

    def __init__(self, irreps_out, edge_sh_normalization="component", edge_sh_normalize=True, irreps_in="1x1o"):
        super().__init__()
        self.init_irreps(vectors=irreps_in, spherical_harmonics=irreps_out, output_keys=["spherical_harmonics"])
        self.mul = parse_irreps(self.irreps_in["vectors"])[0][0]
        self.ls = []
        for mul, l, p in parse_irreps(self.irreps_out["spherical_harmonics"]):
            assert mul == self.mul
            self.ls.append(l)
        self.normalize, self.normalization = edge_sh_normalize, edge_sh_normalization

    def forward(self, data, attrs):
        v = data["vectors"]
        n = v.shape[0]
        width = self.mul * sum(2 * l + 1 for l in self.ls)
        sh = spherical_harmonics(self.ls, v.view(n, self.mul, 3), self.normalize, self.normalization).reshape(n, width)
        return {"spherical_harmonics": sh}, {"spherical_harmonics": ("edge", self.irreps_out["spherical_harmonics"])}


def poly_cutoff(x: Tensor, factor: float, p: float = 6.0) -> Tensor:
    """``_poly_cutoff`` (nn/embedding.py:31-40)."""
    x = x * factor
    out = 1.0 - ((p + 1.0) * (p + 2.0) / 2.0) * torch.pow(x, p)
    out = out + p * (p + 2.0) * torch.pow(x, p + 1.0)
    out = out - (p * (p + 1.0) / 2.0) * torch.pow(x, p + 2.0)
    return out * (x < 1.0)


def symmetric_cutoff(x: Tensor, factor: float, p: float = 6.0) -> Tensor:
    """``symmetricCutoff`` (nn/embedding.py:26-29)."""
    x = x * factor
    return (x - 1) ** 2 * (x + 1) ** 2 * (x.abs() < 1.0).to(x.dtype)


CUTOFFS = {"_poly_cutoff": poly_cutoff, "symmetricCutoff": symmetric_cutoff, "poly": poly_cutoff, "symmetric": symmetric_cutoff}


class BesselBasis(nn.Module):
    """nn/embedding.py:74-127."""

    def __init__(self, r_max, r_min=0, num_basis=8, trainable=True, one_over_r=True):
        super().__init__()
        self.r_max, self.r_min = float(r_max), float(r_min)
        self.prefactor = 2.0 / (self.r_max - self.r_min)
        self.one_over_r = one_over_r
        w = torch.linspace(1.0, num_basis, num_basis) * math.pi
        if trainable:
            self.bessel_weights = nn.Parameter(w)
        else:
            self.register_buffer("bessel_weights", w)

    def forward(self, x):
        y = self.prefactor * torch.sin(self.bessel_weights * x.unsqueeze(-1) / (self.r_max - self.r_min))
        return y / x.unsqueeze(-1) if self.one_over_r else y


class RadialBasisEncoding(OModule):
    """nn/embedding.py:182-219."""

    def __init__(self, r_max, trainable, irreps_out, r_min=0, polynomial_degree=6, basis=None, cutoff=None,
                 irreps_in="1x0e", one_over_r=True):
        super().__init__()
        self.init_irreps(input=irreps_in, radial_embedding=irreps_out, output_keys=["radial_embedding"])
        nb = parse_irreps(self.irreps_out["radial_embedding"])[0][0]
        self.basis = BesselBasis(r_max, r_min, nb, trainable, one_over_r=one_over_r)
        name = getattr(cutoff, "__name__", cutoff) if cutoff is not None else "_poly_cutoff"
        name = getattr(cutoff, "name", name)  # scripted functions
        self.cutoff = CUTOFFS[name if name in CUTOFFS else "_poly_cutoff"]
        self.factor, self.p = 1.0 / float(r_max), float(polynomial_degree)

    def forward(self, data, attrs):
        x = data["input"]
        emb = self.basis(x) * self.cutoff(x, self.factor, self.p)[:, None]
        emb = emb.reshape(x.shape[0], emb.shape[-1])
        return {"radial_embedding": emb}, {"radial_embedding": (attrs["input"][0], self.irreps_out["radial_embedding"])}


class Broadcast(OModule):
    """nn/embedding.py:223-254."""

    def __init__(self, irreps_in, irreps_out, to):
        super().__init__()
        self.init_irreps(input=irreps_in, output=irreps_out, output_keys=["output"])
        self.to_ = to

    def forward(self, data, attrs):
        assert attrs["input"][0] == "graph"
        seg = data["_node_segment"] if self.to_ == "node" else data["_edge_segment"]
        return {"output": data["input"][seg]}, {"output": (self.to_, self.irreps_out["output"])}


class RelativePositionEncoding(OModule):
    """nn/embedding.py:284-312."""

    def __init__(self, radial_encoding, segment, irreps_out, id=None):
        super().__init__()
        self.init_irreps(input=segment, output=irreps_out, id=id, output_keys=["output"])
        cfg = dict(radial_encoding)
        cfg["irreps_in"] = "1x0e"
        cfg["irreps_out"] = self.irreps_out["output"]
        self.radial = build(cfg)

    def forward(self, data, attrs):
        seg, ei = data["input"], data["edge_index"]
        if "id" in self.irreps_in:
            rel = data["id"][ei[0]] - data["id"][ei[1]]
        else:
            rel = ei[0] - ei[1]
        ref = next(v for v in data.values() if torch.is_floating_point(v))
        mask = (seg[ei[0]] == seg[ei[1]]).to(ref.dtype).view(-1, 1)
        rel = mask * rel.view(-1, 1).to(ref.dtype) + (1 - mask) * 1e5
        out, _ = self.radial({"input": rel.view(-1)}, {"input": ("edge", "1x0e")})
        return {"output": out["radial_embedding"]}, {"output": ("edge", self.irreps_out["output"])}


class TensorProductExpansion(OModule):
    """nn/pointwise.py:54-100: weighted 'uvu' product with one output slot per path, then a
    Linear from the (sorted, simplified) concatenation to the requested output irreps."""

    def __init__(self, left, right, output, instruction="uvu", internal_weight=True):
        super().__init__()
        self.init_irreps(left=left, right=right, output=output, output_keys=["output"])
        in1, in2 = parse_irreps(self.irreps_in["left"]), parse_irreps(self.irreps_in["right"])
        out = parse_irreps(self.irreps_out["output"])
        out_set = {(l, p) for _, l, p in out}
        mid, instr = [], []
        for i, (mul, l1, p1) in enumerate(in1):
            for j, (_, l2, p2) in enumerate(in2):
                for lo, po in ir_product(l1, p1, l2, p2):
                    if (lo, po) in out_set:
                        instr.append((i, j, len(mid), instruction))
                        mid.append((mul, lo, po))
        mid_sorted, perm = irreps_sort(mid)
        instr = [(i, j, perm[k], mode) for i, j, k, mode in instr]
        self.tp = TensorProduct(in1, in2, mid_sorted, instr, internal_weights=internal_weight)
        self.internal_weight = internal_weight
        self.linear = Linear(irreps_simplify(mid_sorted), out, biases=False)

    def forward(self, left=None, right=None, weight=None):
        y = self.tp(left, right) if self.internal_weight else self.tp(left, right, weight)
        return self.linear(y)


class FactorizedConvolution(OModule):
    """nn/message_passing.py:21-124 — the hot loop, unfused."""

    def __init__(self, input_features, output_features, node_attrs, edge_radial, edge_spherical,
                 invariant_layers=1, invariant_neurons=8, avg_num_neighbors=None, use_sc=True,
                 nonlinearity_scalars=None, reduce=True):
        super().__init__()
        self.init_irreps(input_features=input_features, output_features=output_features, node_attrs=node_attrs,
                         edge_radial=edge_radial, edge_spherical=edge_spherical, output_keys=["output_features"])
        self.avg_num_neighbors, self.use_sc, self.reduce = avg_num_neighbors, use_sc, reduce
        f_in, f_out = self.irreps_in["input_features"], self.irreps_out["output_features"]
        self.linear_1 = Linear(f_in, f_in)
        self.tp = TensorProductExpansion(f_in, (self.irreps_in["edge_spherical"], "edge_spherical"),
                                         (f_out, "edge_features"), "uvu", internal_weight=False)
        n_rad = sum(m for m, _, _ in parse_irreps(self.irreps_in["edge_radial"]))
        self.fc = FullyConnectedNet([n_rad] + invariant_layers * [invariant_neurons] + [self.tp.tp.weight_numel], "ssp")
        self.sc = FullyConnectedTensorProduct(f_in, self.irreps_in["node_attrs"], f_out) if use_sc else None

    def forward(self, data, attrs):
        weight = self.fc(data["edge_radial"])
        x = data["input_features"]
        src, dst = data["edge_index"][0], data["edge_index"][1]
        sc = self.sc(x, data["node_attrs"]) if self.sc is not None else None
        x = self.linear_1(x)
        ef = self.tp(left=x[src], right=data["edge_spherical"], weight=weight)
        if self.reduce:
            x = scatter(ef, dst, dim_size=x.shape[0])
            if self.avg_num_neighbors is not None:
                x = x / self.avg_num_neighbors ** 0.5
            if sc is not None:
                x = x + sc
        else:
            x = ef
        return {"output_features": x}, {"output_features": (attrs["input_features"][0], self.irreps_out["output_features"])}


class NormActivation(nn.Module):
    """``e3nn.nn.NormActivation(irreps_in, scalar_nonlinearity, normalize=True, epsilon=1e-8, bias=False)`` as built
    at nn/message_passing.py:212-219 (e3nn 0.4.4 ``nn/_normact.py``): per irrep channel the squared norm
    ``n2 = sum_m x_m^2`` (``o3.Norm(squared=True)``) is clamped from below at epsilon^2, ``n = sqrt(n2)``,
    ``scaling = act(n) / n`` (normalize) and the channel is multiplied by it.  The activation is the raw function of
    ``utils.activations`` (no second-moment normalisation: that is ``e3nn.nn.Activation``'s job, not this module's)."""

    def __init__(self, irreps_in, scalar_nonlinearity, normalize=True, epsilon=None, bias=False):
        super().__init__()
        if bias:
            raise NotImplementedError("bias=True is not used by the reference")
        self.irreps_in = parse_irreps(irreps_in) if not isinstance(irreps_in, list) else irreps_in
        self.irreps_out = self.irreps_in
        if epsilon is None and normalize:
            epsilon = 1e-8
        elif epsilon is not None and not normalize:
            raise ValueError("epsilon and normalize = False don't make sense together")
        elif not normalize:
            epsilon = 0.0
        self.eps2 = float(epsilon) ** 2
        self.normalize = normalize
        self.act = ACTIVATIONS[scalar_nonlinearity] if isinstance(scalar_nonlinearity, str) else scalar_nonlinearity

    def forward(self, x):
        z, pos, cols = x.shape[0], 0, []
        for m, l, _ in self.irreps_in:
            d = 2 * l + 1
            blk = x[:, pos:pos + m * d].reshape(z, m, d)
            n2 = blk.pow(2).sum(-1)
            if self.eps2 > 0:
                n2 = torch.where(n2 < self.eps2, torch.full_like(n2, self.eps2), n2)
                n = n2.sqrt()
            else:
                n = n2
            s = self.act(n)
            if self.normalize:
                s = s / n
            cols.append((blk * s.unsqueeze(-1)).reshape(z, m * d))
            pos += m * d
        return torch.cat(cols, dim=1)


class MessagePassing(OModule):
    """nn/message_passing.py:127-262 (gate and norm nonlinearity branches)."""

    def __init__(self, input_features, output_features, node_attrs, edge_radial, edge_spherical, convolution,
                 resnet=False, nonlinearity_type="gate", nonlinearity_scalars=None, nonlinearity_gates=None,
                 normalize=False):
        super().__init__()
        nonlinearity_scalars = nonlinearity_scalars or {"e": "ssp", "o": "tanh"}
        nonlinearity_gates = nonlinearity_gates or {"e": "ssp", "o": "abs"}
        self.init_irreps(input_features=input_features, output_features=output_features, node_attrs=node_attrs,
                         edge_radial=edge_radial, edge_spherical=edge_spherical, output_keys=["output_features"])
        assert nonlinearity_type in ("gate", "norm")
        a_sc = {1: nonlinearity_scalars["e"], -1: nonlinearity_scalars["o"]}
        a_gt = {1: nonlinearity_gates["e"], -1: nonlinearity_gates["o"]}
        prev = parse_irreps(self.irreps_in["input_features"])
        sh = parse_irreps(self.irreps_in["edge_spherical"])
        hidden = parse_irreps(self.irreps_out["output_features"])
        scalars = [(m, l, p) for m, l, p in hidden if l == 0 and tp_path_exists(prev, sh, (l, p))]
        gated = [(m, l, p) for m, l, p in hidden if l > 0 and tp_path_exists(prev, sh, (l, p))]
        layer_out = irreps_simplify(scalars + gated)
        gates = [(m, 0, 1) for m, _, _ in gated]
        if nonlinearity_type == "gate":
            self.gate = Gate(scalars, [a_sc[p] for _, _, p in scalars], gates, [a_gt[p] for _, _, p in gates], gated)
            conv_out = irreps_simplify(self.gate.irreps_in)
        else:   # :207-219: the norm is an even scalar, so the 'e' scalar nonlinearity is used
            conv_out = layer_out
            self.gate = NormActivation(conv_out, nonlinearity_scalars["e"], normalize=True, epsilon=1e-8, bias=False)
        self.resnet = bool(resnet) and layer_out == prev
        conv_cfg = dict(convolution)
        self.conv = build(conv_cfg, input_features=input_features, output_features=irreps_str(conv_out),
                          node_attrs=node_attrs, edge_radial=edge_radial, edge_spherical=edge_spherical)
        self.normalize = normalize
        if normalize:
            self.norm = LayerNormalization(self.irreps_out["output_features"], self.irreps_out["output_features"])

    def forward(self, data, attrs):
        old = data["input_features"]
        out, _ = self.conv(data, attrs)  # called directly on the already key-mapped dict (:247)
        y = self.gate(out["output_features"])
        if self.resnet:
            y = old + y
        if self.normalize:
            y = self.norm({"input": y}, attrs)[0]["output"]
        return {"output_features": y}, {"output_features": (attrs["input_features"][0], self.irreps_out["output_features"])}


class PerTypeScaleShift(OModule):
    """nn/scaling.py:9-67."""

    def __init__(self, num_types, shifts, scales, scales_trainable=False, shifts_trainable=False,
                 irreps_in="1x0e", irreps_out="1x0e", species="1x0e"):
        super().__init__()
        self.init_irreps(input=irreps_in, output=irreps_out, species=species, output_keys=["output"])
        for name, val in (("shifts", shifts), ("scales", scales)):
            if val is None:
                setattr(self, "has_" + name, False)
                continue
            setattr(self, "has_" + name, True)
            t = torch.as_tensor(val, dtype=torch.get_default_dtype()).reshape(-1)
            if t.numel() == 1:
                t = t.expand(num_types).clone()
            assert t.shape == (num_types,)
            self.register_buffer(name, t)

    def forward(self, data, attrs):
        x, sp = data["input"], data["species"].view(-1)
        if self.has_scales:
            x = self.scales.to(x.dtype)[sp].view(-1, 1) * x
        if self.has_shifts:
            x = self.shifts.to(x.dtype)[sp].view(-1, 1) + x
        return {"output": x}, {"output": (attrs["input"][0], self.irreps_out["output"])}


class Pooling(OModule):
    """nn/output.py:56-74."""

    def __init__(self, irreps_in, irreps_out, reduce):
        super().__init__()
        self.init_irreps(input=irreps_in, output=irreps_out, output_keys=["output"])
        assert reduce in ("sum", "mean")
        self.reduce = reduce

    def forward(self, data, attrs):
        n_graphs = data["_n_nodes"].shape[0]
        y = scatter(data["input"], data["_node_segment"], dim_size=n_graphs, reduce=self.reduce)
        return {"output": y}, {"output": ("graph", self.irreps_out["output"])}


class SequentialGraphNetwork(nn.Module):
    """nn/sequential.py:42-88, operating on a plain ``(data, attrs)`` pair of dicts."""

    def __init__(self, layers, **_ignored):
        super().__init__()
        self.steps = []
        mods = OrderedDict()
        for key, value in layers:
            if hasattr(value, "keys"):
                m = build(value)
                mods[key] = m
                self.steps.append((key, m))
            elif callable(value):
                self.steps.append((key, resolve_callable(value)))
            else:
                raise TypeError("invalid config node")
        self.mods = nn.ModuleDict(mods)

    def forward(self, data: dict, attrs: dict):
        data, attrs = dict(data), dict(attrs)
        add_segments(data)
        for key, step in self.steps:
            if isinstance(step, OModule):
                d, a = step(_remap(data, step.in_map), _remap(attrs, step.in_map))
                d, a = _remap(d, step.out_map), _remap(a, step.out_map)
            else:
                d, a = step(data, attrs)
            data.update(d)
            attrs.update(a)
        return data, attrs


class GradientOutput(nn.Module):
    """nn/output.py:19-53: gradients = sign * d(sum y)/dx."""

    def __init__(self, func, x, y, gradients, sign=1.0, **kw):
        super().__init__()
        self.sign = float(sign)
        self.x_key = _split_spec(x)[1] or "x"
        self.y_key = _split_spec(y)[1] or "y"
        self.g_key = _split_spec(gradients)[1] or "gradients"
        self.g_irreps = _split_spec(gradients)[0]
        self.func = build(func, **kw) if hasattr(func, "keys") else func

    def forward(self, data: dict, attrs: dict):
        data = dict(data)
        x = data[self.x_key]
        if not x.requires_grad:      # (nn/output.py:33-35 flips requires_grad on the caller's tensor and back; a private leaf does the
            x = x.detach().clone().requires_grad_(True)      # same without touching it.  A caller's tensor that already requires
        data[self.x_key] = x                                 # grad is used as it is: gradients of the output flow back to it)
        out, oattrs = self.func(data, attrs)
        (g,) = torch.autograd.grad(out[self.y_key].sum(), x, create_graph=self.training)
        out[self.g_key] = self.sign * g
        oattrs[self.g_key] = (attrs[self.x_key][0], self.g_irreps)
        return out, oattrs


def add_segments(data: dict) -> None:
    """``Batch.nodeSegment/edgeSegment`` (data/batch.py:164-178)."""
    if "_n_nodes" in data and "_node_segment" not in data:
        n = data["_n_nodes"].view(-1)
        data["_node_segment"] = torch.repeat_interleave(torch.arange(n.numel(), device=n.device), n)
    if "_n_edges" in data and "_edge_segment" not in data:
        n = data["_n_edges"].view(-1)
        data["_edge_segment"] = torch.repeat_interleave(torch.arange(n.numel(), device=n.device), n)


# --------------------------------------------------------------------------------------
# build() over a config tree whose "module" entries are classes of *another* package with
# the same names (utils/utils.py:99-136)
# --------------------------------------------------------------------------------------

_REGISTRY: Dict[str, Callable] = {}


def _register():
    for cls in (OneHotEncoding, PointwiseLinear, Concat, LayerNormalization, SphericalEncoding, RadialBasisEncoding,
                Broadcast, RelativePositionEncoding, TensorProductExpansion, FactorizedConvolution, MessagePassing,
                PerTypeScaleShift, Pooling, SequentialGraphNetwork, GradientOutput):
        _REGISTRY[cls.__name__] = cls


def resolve_callable(fn):
    base = fn.func if isinstance(fn, partial) else fn
    name = getattr(base, "__name__", "")
    table = {"computeEdgeVector": compute_edge_vector, "computeEdgeIndex": compute_edge_index,
             "compute_edge_vector": compute_edge_vector, "compute_edge_index": compute_edge_index}
    if name not in table:
        raise KeyError(f"oracle has no restatement of callable layer {name!r}")
    mine = table[name]
    if isinstance(fn, partial):
        return partial(mine, *fn.args, **fn.keywords)
    return mine


def build(node, **kwargs):
    import inspect

    node = {k: node[k] for k in node.keys()}
    target = node.pop("module")
    name = target if isinstance(target, str) else target.__name__
    cls = _REGISTRY[name]
    kwargs.update(node)
    sig = inspect.signature(cls.__init__)
    if not any(p.kind == p.VAR_KEYWORD for p in sig.parameters.values()):
        kwargs = {k: v for k, v in kwargs.items() if k in sig.parameters}
    return cls(**kwargs)


_register()
