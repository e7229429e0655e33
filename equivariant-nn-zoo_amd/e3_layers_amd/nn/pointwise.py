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
        return {"output": out}, {"output": (attrs[keys[0]][0], self.irreps_out["output"])}
