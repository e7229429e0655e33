"""Pointwise layers: ``PointwiseLinear``, ``LayerNormalization``, ``TensorProductExpansion``,
``Concat`` — interfaces of ``e3_layers/nn/pointwise.py:14-30,32-51,54-100,134-152``.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
from torch import Tensor

from ..backend import ops
from ..backend.graph import build_topology
from ..o3 import Irreps
from .core import Linear, UVUTensorProduct, combine_row_keys, get_row_key, irreps_blocks, set_row_key
from .sequential import Module


class PointwiseLinear(Module):
    def __init__(self, irreps_in, irreps_out, biases=True, **kwargs):
        super().__init__()
        self.init_irreps(input=irreps_in, output=irreps_out, output_keys=["output"])
        self.linear = Linear(self.irreps_in["input"], self.irreps_out["output"], biases=biases)

    def forward(self, data: Dict[str, Tensor], attrs: Dict[str, Tuple[str, str]]):
        out = self.linear(data["input"])
        key = get_row_key(data["input"])
        if key is not None:
            set_row_key(out, *key)      # a row-wise map keeps rows with equal keys equal
        return {"output": out}, {"output": (attrs["input"][0], self.irreps_out["output"])}


class LayerNormalization(Module):
    """x / sqrt(sum x^2 / mul + 1e-6) * std_i per irreps entry."""

    def __init__(self, irreps_in, irreps_out, **kwargs):
        super().__init__()
        self.init_irreps(input=irreps_in, output=irreps_out, output_keys=["output"])
        assert Irreps(self.irreps_in["input"]) == Irreps(self.irreps_out["output"])
        self._blocks = tuple(irreps_blocks(Irreps(self.irreps_in["input"])))
        self.std = torch.nn.Parameter(torch.ones(len(self._blocks)))

    def forward(self, data, attrs):
        return {"output": ops.layer_norm(data["input"], self.std, self._blocks)}, attrs


class TensorProductExpansion(Module):
    """'uvu' product of ``left`` with ``right`` restricted to the irreps of ``output``, one slot
    per path, followed by a Linear to ``output``.  ``forward(left, right, weight)`` is the
    per-sample (per-edge) API of the reference; ``FactorizedConvolution`` uses ``fused`` +
    ``linear`` instead so that the Linear runs on nodes, after the sum."""

    def __init__(self, left, right, output, instruction="uvu", internal_weight=True, **kwargs):
        super().__init__()
        if instruction != "uvu":
            raise NotImplementedError("only the 'uvu' instruction is built (the one every shipped config uses)")
        self.init_irreps(left=left, right=right, output=output, output_keys=["output"])
        l_ir, r_ir, o_ir = (Irreps(self.irreps_in["left"]), Irreps(self.irreps_in["right"]), Irreps(self.irreps_out["output"]))
        self.tp = UVUTensorProduct(l_ir, r_ir, o_ir)
        self.internal_weight = internal_weight
        if internal_weight:
            self.weight = torch.nn.Parameter(torch.randn(self.tp.weight_numel))
        self.linear = Linear(self.tp.irreps_mid_simplified, o_ir)
        self._left_blocks = tuple(irreps_blocks(l_ir))

    def forward(self, left=None, right=None, weight=None):
        rows = left.shape[0]
        if self.internal_weight:
            weight = self.weight.unsqueeze(0).expand(rows, -1)
        # every sample is its own "node" with exactly one in-edge
        ids = torch.arange(rows, device=left.device)
        topo = build_topology(torch.stack([ids, ids]), rows)
        mid = self.tp.fused(ops.relayout(left, self._left_blocks, True), right, weight.contiguous(), topo)
        return self.linear(mid, in_layout="cf", out_layout="e3nn")


class Concat(Module):
    def __init__(self, irreps_out, **irreps_in):
        super().__init__()
        self.init_irreps(**irreps_in, output=irreps_out, output_keys=["output"])
        total = Irreps()
        for value in self.irreps_in.values():
            total = total + Irreps(value)
        self.linear = Linear(total, Irreps(self.irreps_out["output"]), biases=True)

    def forward(self, data, attrs):
        keys = list(self.irreps_in.keys())
        x = torch.cat([data[k] for k in keys], dim=1)
        out = self.linear(x)
        key = combine_row_keys([data[k] for k in keys])
        if key is not None:
            set_row_key(out, *key)
        self._tag_keyed_radial(data, keys, out)
        return {"output": out}, {"output": (attrs[keys[0]][0], self.irreps_out["output"])}

    KEYED_RADIAL_MAX_KEYS = 8

    def _tag_keyed_radial(self, data, keys, out) -> None:
        """``Concat(one_hot(small key), RadialBasisEncoding(edge_length))`` (``e3_layers/configs/config_diffusion.py:73-82``: the 4-way
        bond type beside the Bessel basis) is a row-wise function of (radius, key): the convolution layers' radial MLPs can then run
        on one knot table per key instead of per edge (``backend/radial_table.KeyedRadialSource``).  Applies when exactly one input
        carries a radial source that still describes it, and every other input is a one-hot of ONE small key."""
        from ..backend import radial_table

        if not out.is_cuda:
            return
        radial = [k for k in keys if radial_table.source_of(data[k]) is not None]
        others = [k for k in keys if k not in radial]
        if len(radial) != 1 or not others:
            return
        src = radial_table.source_of(data[radial[0]])
        if not isinstance(src, radial_table.RadialSource) or data[radial[0]]._version != src.version:
            return
        okey = combine_row_keys([data[k] for k in others])
        if okey is None or not (1 < okey[1] <= self.KEYED_RADIAL_MAX_KEYS):
            return
        if any(data[k].shape[1] != get_row_key(data[k])[1] for k in others):      # (plain one-hot rows: rebuilt from the key below)
            return
        order = list(keys)
        widths = {k: data[k].shape[1] for k in keys}
        nkeys = {k: get_row_key(data[k])[1] for k in others}
        linear = self.linear

        def rows_fn(kidx, basis_rows):
            # the Concat's own arithmetic on synthetic rows: for every (key, knot) the one-hot blocks of the key and the basis row
            cols, rest = [], kidx
            parts = {}
            for k in reversed(others):                 # (combine_row_keys: index = ((k0) n1 + k1) n2 + k2 ...)
                parts[k] = rest % nkeys[k]
                rest = rest // nkeys[k]
            for k in order:
                if k in parts:
                    cols.append(torch.nn.functional.one_hot(parts[k], num_classes=widths[k]).to(basis_rows.dtype))
                else:
                    cols.append(basis_rows)
            return linear(torch.cat(cols, dim=1))

        out._e3k_radial_src = radial_table.KeyedRadialSource(src, okey[0], okey[1], rows_fn, out._version)
