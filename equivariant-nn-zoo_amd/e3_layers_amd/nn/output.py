"""Output heads: ``GradientOutput`` (forces / scores by autograd through the HIP kernels) and
``Pooling`` — interfaces of ``e3_layers/nn/output.py:19-53,56-74``.  ``Pairwise`` and
``TensorProductContraction`` (legacy Hamiltonian head) are out of scope.
"""
from __future__ import annotations

import torch

from ..backend import ops
from ..o3 import Irreps
from ..utils.utils import _is_mapping, build
from .sequential import Module


class GradientOutput(Module):
    def __init__(self, func, x, y, gradients, sign: float = 1.0, **kwargs):
        super().__init__()
        sign = float(sign)
        assert sign in (1.0, -1.0)
        self.sign = sign
        self.init_irreps(x=x, y=y, gradients=gradients, output_keys=["gradients"])
        assert Irreps(self.irreps_in["y"]).lmax == 0
        if _is_mapping(func):
            func = build(func, **kwargs)
        self.func = func

    def prepare_data(self, batch) -> dict:
        """``SequentialGraphNetwork.prepare_data`` of the wrapped network, minus everything that reads ``x``: the forward
        differentiates w.r.t. it, so edge vectors, lengths and spherical harmonics belong to the autograd graph of the step (what is
        left: topology, one-hot species, key groups)."""
        inner = getattr(self.func, "prepare_data", None)
        if inner is None:
            return {}
        x_key = next((g for g, loc in self.input_key_mapping.items() if loc == "x"), "x")
        return inner(batch, exclude=(x_key,))

    def forward(self, data):
        wrt = self.inputKeyMap(data)["x"]
        old = wrt.requires_grad
        wrt.requires_grad_(True)
        #  edge vectors the CALLER supplied and that require grad hang on some other leaf (a cell, a strain): the force
        # block's second pass hands back parameter gradients only, so the force-loss share of that leaf's gradient would be
        # missing -- such a call takes the composed path too
        pre = data.get("edge_vector") if "edge_vector" in data else None
        foreign = pre is not None and bool(pre.requires_grad)
        if (old and self.training) or foreign:
            # the caller differentiates w.r.t. ``x`` itself (it required grad before this call) and the graph of the gradient is
            # kept: d loss / d x of a loss on the gradients needs third derivatives of the layers -- the force block
            # (backend/conv_force.py) does not form them, the composed per-kernel path does
            from ..backend import conv_force

            with conv_force.declined():
                output = self.func(data)
        else:
            output = self.func(data)
        # only d y / d x is asked for: the backward functions skip every Parameter gradient of this pass
        with ops.inputs_only_backward():
            (grad,) = torch.autograd.grad(self.inputKeyMap(output)["y"].sum(), wrt, create_graph=self.training)
        wrt.requires_grad_(old)
        is_per = self.inputKeyMap(data.attrs)["x"][0]
        output.attrs.update(self.outputKeyMap({"gradients": (is_per, self.irreps_out["gradients"])}))
        output.update(self.outputKeyMap({"gradients": self.sign * grad}))
        return output


class Pooling(Module):
    """node -> graph reduction over the (sorted) node segments."""

    def __init__(self, irreps_in, irreps_out, reduce):
        super().__init__()
        self.init_irreps(input=irreps_in, output=irreps_out, output_keys=["output"])
        assert reduce in ("sum", "mean")
        self.reduce = reduce

    @staticmethod
    def _row_pointers(counts, device):
        """int32 [G + 1] row pointers of the per-graph node counts; memoised on the counts tensor (``prepare_batch`` builds them
        ahead of the step: they are batch data)."""
        hit = getattr(counts, "_e3k_ptr", None)
        if hit is not None and hit[0] == counts._version and hit[1].device == device:
            return hit[1]
        n = counts.view(-1).to(device)
        ptr = torch.empty(n.numel() + 1, dtype=torch.int32, device=device)
        if n.is_cuda and n.dtype == torch.int64 and n.is_contiguous():
            from ..backend import lib as L

            L.check(L.load().e3k_counts_to_ptr(L.ptr(n), n.numel(), L.ptr(ptr), L.stream_ptr()), "e3k_counts_to_ptr")
        else:
            ptr[0] = 0
            ptr[1:] = torch.cumsum(n, 0).to(torch.int32)
        counts._e3k_ptr = (counts._version, ptr)
        return ptr

    def prepare_batch(self, network, data, avail) -> None:
        """``SequentialGraphNetwork.prepare_data`` hook: the segment pointers depend on ``_n_nodes`` alone."""
        if "_n_nodes" in avail and "_n_nodes" in data and data["_n_nodes"].is_cuda:
            self._row_pointers(data["_n_nodes"], data["_n_nodes"].device)

    def forward(self, data, attrs):
        x = data["input"]
        ptr = self._row_pointers(data["_n_nodes"], x.device)
        out = ops.segment_sum(x, ptr, data["_node_segment"].to(x.device), self.reduce == "mean")
        return {"output": out}, {"output": ("graph", self.irreps_out["output"])}
