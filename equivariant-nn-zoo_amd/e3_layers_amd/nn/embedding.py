"""Edge / node embeddings — interfaces of ``e3_layers/nn/embedding.py``:
``symmetricCutoff`` (:26-29), ``_poly_cutoff`` (:31-40), ``PolynomialCutoff`` (:43-71),
``BesselBasis`` (:74-127), ``SphericalEncoding`` (:131-178), ``RadialBasisEncoding`` (:182-219),
``Broadcast`` (:223-254), ``OneHotEncoding`` (:258-281), ``RelativePositionEncoding`` (:284-312).
The arithmetic of the spherical and radial encodings runs in ``csrc/e3k_edge.hip``.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch
from torch import Tensor, nn

from ..backend import ops, radial_table
from ..o3 import Irreps
from ..utils.utils import build
from .core import set_row_key
from .sequential import Module


class _Cutoff:
    """Envelope selector; the function itself is evaluated inside the radial-basis kernel."""

    def __init__(self, name: str, kind: int):
        self.__name__, self.kind = name, kind

    def __repr__(self):
        return f"<cutoff {self.__name__}>"


_poly_cutoff = _Cutoff("_poly_cutoff", 0)        # 1 - (p+1)(p+2)/2 x^p + p(p+2) x^(p+1) - p(p+1)/2 x^(p+2), x < 1
symmetricCutoff = _Cutoff("symmetricCutoff", 1)  # (x-1)^2 (x+1)^2, |x| < 1


class PolynomialCutoff(nn.Module):
    def __init__(self, r_max: float, p: float = 6, cutoff=_poly_cutoff):
        super().__init__()
        assert p >= 2.0
        self.p, self._factor, self.cutoff = float(p), 1.0 / float(r_max), cutoff


class BesselBasis(nn.Module):
    """Holds the (trainable) frequencies n*pi; evaluated fused with the cutoff."""

    def __init__(self, r_max, r_min=0, num_basis=8, trainable=True, one_over_r=True):
        super().__init__()
        self.trainable, self.num_basis = trainable, num_basis
        self.r_max, self.r_min = float(r_max), float(r_min)
        self.prefactor = 2.0 / (self.r_max - self.r_min)
        self.one_over_r = one_over_r
        w = torch.linspace(start=1.0, end=num_basis, steps=num_basis) * math.pi
        if trainable:
            self.bessel_weights = nn.Parameter(w)
        else:
            self.register_buffer("bessel_weights", w)


class SphericalEncoding(Module):
    data_only = True      # parameter-free: ``SequentialGraphNetwork.prepare_data`` may run it ahead of the step

    def __init__(self, irreps_out, edge_sh_normalization: str = "component", edge_sh_normalize: bool = True,
                 irreps_in="1x1o"):
        super().__init__()
        self.init_irreps(vectors=irreps_in, spherical_harmonics=irreps_out, output_keys=["spherical_harmonics"])
        self.mul = Irreps(self.irreps_in["vectors"])[0].mul
        self.ls = []
        for mi in Irreps(self.irreps_out["spherical_harmonics"]):
            assert mi.mul == self.mul
            self.ls.append(mi.ir.l)
        self.normalize, self.normalization = bool(edge_sh_normalize), edge_sh_normalization

    def forward(self, data: Dict[str, Tensor], attrs: Dict[str, Tuple[str, str]]):
        vec = data["vectors"]
        rows = vec.shape[0]
        sh = ops.spherical_harmonics(vec.reshape(rows * self.mul, 3), self.ls, self.normalize, self.normalization)
        width = self.mul * sum(2 * l + 1 for l in self.ls)      # explicit: rows may be 0 (a batch without edges)
        sh = ops.mark_data_only(sh.view(rows, width), ops.is_data_only(vec))
        return ({"spherical_harmonics": sh},
                {"spherical_harmonics": ("edge", self.irreps_out["spherical_harmonics"])})


class RadialBasisEncoding(Module):
    def __init__(self, r_max, trainable, irreps_out, r_min=0, polynomial_degree=6, basis=BesselBasis,
                 cutoff=_poly_cutoff, irreps_in="1x0e", one_over_r=True):
        super().__init__()
        self.init_irreps(input=irreps_in, radial_embedding=irreps_out, output_keys=["radial_embedding"])
        num_basis = Irreps(self.irreps_out["radial_embedding"])[0].mul
        self.basis = basis(r_max, r_min, num_basis, trainable, one_over_r=one_over_r)
        self.cutoff = PolynomialCutoff(r_max, p=polynomial_degree, cutoff=cutoff)
        self.r_max = r_max

    def forward(self, data, attrs):
        x = data["input"]
        b, c = self.basis, self.cutoff
        out = ops.radial_basis(x.reshape(-1), b.bessel_weights, b.r_max, b.r_min, c.p, b.one_over_r, c.cutoff.kind)
        per_row = x.numel() // x.shape[0] if x.shape[0] else (x.shape[1] if x.dim() > 1 else 1)
        out = out.view(x.shape[0], per_row * out.shape[-1])
        if per_row == 1 and c.cutoff.kind == 0 and b.r_min == 0.0:
            # a pure function of one radius per row, constant beyond r_max: the convolutions may evaluate their radial MLPs
            # on a knot table instead of per edge (backend/radial_table.py); any op that builds a new tensor drops the tag
            out._e3k_radial_src = radial_table.RadialSource(self, x.reshape(-1), out._version, prepared=radial_table.prepared_bins(x))
        return ({"radial_embedding": out},
                {"radial_embedding": (attrs["input"][0], self.irreps_out["radial_embedding"])})

    def prepare_batch(self, network, data, avail) -> None:
        """``SequentialGraphNetwork.prepare_data`` hook: the radii are batch data, so the knot bins of the radial table (which edge
        interpolates between which rows, the edges grouped by knot: ``e3k_rtable_bins``, three launches) can be built before the
        step; ``forward`` hands them to the ``RadialSource`` it tags its output with."""
        key = next((g for g, loc in self.input_key_mapping.items() if loc == "input"), None)
        x = data.get(key) if key in avail else None
        b, c = self.basis, self.cutoff
        if (x is None or not x.is_cuda or not radial_table.ENABLED or x.dim() > 1 and x.shape[1] != 1 or c.cutoff.kind != 0
                or b.r_min != 0.0):
            return
        knots = radial_table.KNOTS
        rows = radial_table.layout(float(b.r_max), knots)[0] + 1
        if x.shape[0] >= radial_table.MIN_EDGES_PER_KNOT * rows:
            radial_table.prepare_bins(x, float(b.r_max), knots)


class Broadcast(Module):
    """graph -> node / edge broadcast by segment ids (index plumbing)."""

    def __init__(self, irreps_in, irreps_out, to):
        super().__init__()
        self.init_irreps(input=irreps_in, output=irreps_out, output_keys=["output"])
        if to not in ("node", "edge"):
            raise ValueError(to)
        self.to_kind = to

    def forward(self, data, attrs):
        assert attrs["input"][0] == "graph"
        seg = data["_node_segment"] if self.to_kind == "node" else data["_edge_segment"]
        out = data["input"][seg]
        set_row_key(out, seg, data["input"].shape[0])   # rows of one graph are identical
        return {"output": out}, {"output": (self.to_kind, self.irreps_out["output"])}


class OneHotEncoding(Module):
    num_types: int
    data_only = True      # parameter-free: ``SequentialGraphNetwork.prepare_data`` may run it ahead of the step

    def __init__(self, num_types: int, irreps_out, irreps_in="0x0e"):
        super().__init__()
        self.num_types = num_types
        self.init_irreps(input=irreps_in, one_hot=irreps_out, output_keys="one_hot")

    def forward(self, data, attrs):
        src = data["input"]
        # the same index VIEW object for the same source tensor (and version): the key groups of the keyed
        # self-connection are memoised on the identity of this tensor (nn/core.py: row_groups), so a batch that is
        # stepped repeatedly (batch.view()) sorts its species once, not once per step
        memo = getattr(src, "_e3k_flat_index", None)
        if memo is not None and memo[0] == src._version:
            idx = memo[1]
        else:
            # a COPY, not a view: a view keeps a C++ reference to its base, and base -> attribute -> view -> base is a
            # cycle through the C++ reference counts that nothing ever collects
            idx = src.squeeze(-1).clone()
            src._e3k_flat_index = (src._version, idx)
        if idx.is_cuda and idx.dtype == torch.int64 and idx.dim() == 1:      # one launch instead of zeros + scatter + conversion
            from ..backend import lib as L

            from ..backend import graph as _graph

            one_hot = torch.empty(idx.shape[0], self.num_types, device=idx.device, dtype=torch.float32)
            # an index outside [0, num_types) leaves a zero row and a bit in the device's persistent error flag: raised by the next
            # topology build / optimizer step / check_indices(), as a bad edge endpoint is (the reference's one_hot raises at once)
            capturing = torch.cuda.is_current_stream_capturing()
            flag = _graph._capture_flags.get(idx.device.index) if capturing else _graph.persistent_flag(idx.device)
            L.check(L.load().e3k_onehot(L.ptr(idx), idx.shape[0], self.num_types, L.ptr(one_hot), L.ptr(flag), L.stream_ptr()), "e3k_onehot")
            if flag is not None:
                _graph.report_persistent(idx.device)
        else:
            one_hot = torch.nn.functional.one_hot(idx, num_classes=self.num_types).to(dtype=torch.float)
        set_row_key(one_hot, idx, self.num_types)   # rows are a function of the type index only
        return {"one_hot": one_hot}, {"one_hot": (attrs["input"][0], self.irreps_out["one_hot"])}

    def prepare_batch(self, network, data, avail) -> None:
        """``prepare_data`` hook: the rows grouped by type (what the keyed self-connections of the convolutions sort their nodes
        by; memoised on the index tensor, ``nn/core.row_groups``) -- only when some layer of the network has a self-connection."""
        key = self.output_key_mapping.get("one_hot")
        t = data.get(key) if key in avail else None
        rk = getattr(t, "_e3k_key", None) if t is not None else None
        if rk is not None and t.is_cuda and any(getattr(getattr(m, "conv", None), "sc", None) is not None for _, m in network.layers):
            from .core import row_groups

            row_groups(rk[0], rk[1])


class RelativePositionEncoding(Module):
    def __init__(self, radial_encoding, segment, irreps_out, id=None):
        super().__init__()
        self.init_irreps(input=segment, output=irreps_out, id=id, output_keys=["output"])
        cfg = {k: radial_encoding[k] for k in radial_encoding.keys()}
        cfg["irreps_in"] = "1x0e"
        cfg["irreps_out"] = self.irreps_out["output"]
        self.radial = build(cfg)

    def forward(self, data, attrs):
        seg, ei = data["input"], data["edge_index"]
        if "id" in self.irreps_in:
            rel = data["id"][ei[0]] - data["id"][ei[1]]
        else:
            rel = ei[0] - ei[1]
        mask = (seg[ei[0]] == seg[ei[1]]).float().view(-1, 1)
        rel = mask * rel.view(-1, 1).float() + (1 - mask) * 1e5
        out, _ = self.radial({"input": rel}, {"input": ("edge", "1x0e")})
        return {"output": out["radial_embedding"]}, {"output": ("edge", self.irreps_out["output"])}
