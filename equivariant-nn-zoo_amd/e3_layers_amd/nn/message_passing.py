"""``FactorizedConvolution`` and ``MessagePassing`` — the hot loop.

Interfaces of ``e3_layers/nn/message_passing.py:21-124`` and ``:127-262``; the data flow is
re-planned for MI355X:

    reference (per layer, all [E, .] tensors materialised)      here
    ------------------------------------------------------      -------------------------------------------
    weight = fc(edge_radial)                     [E, W]          same, f32-MFMA GEMMs (e3k_gemm)
    sc = FCTP(x, node_attrs)                     [N, out]        outer-mode GEMM, x (x) attrs never materialised
    x  = linear_1(x)                             [N, in]         GEMM, output in channel-fastest (cf) layout
    ef = tp(x[src], sh, weight) (+ Linear/edge)  [E, mid]->[E,out]   fused gather + CG product + per-destination
    x  = scatter(ef, dst) / sqrt(avg) + sc       [N, out]        reduce in registers (e3k_tp_fwd) -> [N, mid];
                                                                 the Linear commutes with the sum and runs on
                                                                 nodes, accumulating into sc with the 1/sqrt(avg)
                                                                 folded into alpha
    Gate                                                         one elementwise kernel, cf -> e3nn layout
"""
from __future__ import annotations

import math
import os
import weakref

from typing import Callable, Dict, Optional, Tuple

import torch

from ..backend.tuning import knob as _knob
from torch import Tensor

from ..backend import conv_block, ops, radial_table
from ..backend.graph import get_topology
from ..o3 import Irrep, Irreps
from ..utils.utils import _is_mapping, activations, build, tp_path_exists
from .core import (FullyConnectedNet, FullyConnectedTensorProduct, Gate, Linear, NormActivation, get_row_key, irreps_blocks,
                   row_groups)
from .pointwise import LayerNormalization, TensorProductExpansion
from .sequential import Module


# 1 (default): the edge-side branch of a convolution (radial MLP) runs on a side stream next to the node-side branch
# (relayout, self-connection, linear_1); autograd replays the same split in the backward.  Measured +6 % on the bench.
FWD_FORK = _knob("E3K_FWD_FORK")
FWD_FORK_SC = _knob("E3K_FWD_FORK_SC")   # 1: the self-connection runs on a third stream (+7 % on the bench)
# The fork pays when the branches are long enough to hide the extra stream switches (≈ 0.3 ms of host time per
# step): the yardstick is the size of the per-edge weight tensor, edges x weight_numel, counted in edges of a
# 1920-weight layer (n_dim 64, l_max 2).  Measured: config_energy 256 molecules (69 k x 1920) +12 %, protein
# (29 k x 1920) +10 %, config_diffusion (42 k x 960) -15 %, config_energy_force 64 molecules (20 k x 1920) -10 %.
# Round 3, with the radial MLP on the knot table (its branch is then a few launches over 2 049 rows, not per-edge GEMMs):
# config_energy 96 / 128 / 160 molecules run 17 / 6 / 1 % FASTER on one stream, 192 / 256 molecules 8 % faster forked --
# layers on the table use the larger yardstick.
FORK_MIN_EDGES = _knob("E3K_FORK_MIN_EDGES")
# Last third of round 3, with the stacks on one stream at every size: 192 molecules run faster on ONE stream (4.38 vs 4.52 ms),
# 384 / 512 molecules 4-5 % faster forked (6.73 vs 7.03, 8.25 vs 8.68); 256 molecules depend on the HOST: forked 5.00 vs 5.26 ms
# on an idle one, 5.23 vs 5.24 on a loaded one, where the fork's extra host work (4.0 instead of 2.6 ms per step) leaves no
# slack.  Layers on the table fork from 60 000 edges; bench.py times both layouts and keeps the faster (E3K_FORK_MIN_EDGES_TABLE).
FORK_MIN_EDGES_TABLE = _knob("E3K_FORK_MIN_EDGES_TABLE")
_FORK_REF_WIDTH = 1920
# 1: consecutive MessagePassing layers pass their node features in the channel-fastest layout (MessagePassing._emit_cf)
CF_CHAIN = _knob("E3K_CF_CHAIN")
# 1: the radial MLPs of all the layers that read one edge embedding run as one batch on the knot table (first layer of
# the chain), see MessagePassing._stack_rows
RADIAL_STACK = _knob("E3K_RADIAL_STACK")
# ... while the step is launch-bound: with more edges than this every layer runs its own MLP (its backward then overlaps
# the earlier layers' backward instead of forming one tail behind the first layer's; measured cross-over)
STACK_MAX_EDGES = _knob("E3K_STACK_MAX_EDGES")
# 1: likewise the per-key self-connection weights M_l of all the layers that read one node_attrs tensor
# (MessagePassing._kw_stack_rows); not while the all-reduce is overlapped with the backward (run/parallel.py): the
# self-connection weights are most of a layer's parameters and their gradients would then only exist at the very end
# 1: a layer whose self-connection has general (un-keyed) node attributes still runs as a fused block: the self-connection is
# computed by ops.fctp outside and handed in as an addend (MessagePassing._forward_block_addend)
CF_CHAIN_NORM = _knob("E3K_CF_CHAIN_NORM")      # cf hand-over through LayerNormalization
BLOCK_ADDEND = _knob("E3K_BLOCK_ADDEND")
ADDEND_FORK = _knob("E3K_ADDEND_FORK")
KW_STACK = _knob("E3K_KW_STACK")
KW_STACK_MAX_EDGES = _knob("E3K_KW_STACK_MAX_EDGES")      # (192 / 256 molecules: -0.10 / -0.03 ms: no limit)


def _stream_alias(t: Tensor, stream) -> Tensor:
    """A view of ``t`` whose autograd node lives on ``stream`` (call it with that stream current).  Every convolution
    reads the edge embedding on the radial side stream and the node attributes on the self-connection stream; without
    the alias their gradient contributions are summed at the producer's node, on the MAIN stream, which then waits
    for each layer's side branch in turn (≈ 250 us per layer in the backward).  Summed at the alias they stay on the
    side stream and reach the main stream once."""
    if not (t.requires_grad and torch.is_grad_enabled()):
        return t
    cache = getattr(t, "_e3k_alias", None)
    if cache is None:
        cache = t._e3k_alias = {}
    key = stream.cuda_stream
    ref = cache.get(key)
    alias = ref() if ref is not None else None
    if alias is None:
        alias = t.view_as(t)
        # weak: the view already points at ``t`` (``_base``); a strong reference back would be a cycle that only the
        # cyclic collector frees -- 0.3 MB of device memory per step piled up between its runs.  The consumers' autograd
        # graphs keep the alias alive for as long as it matters.
        cache[key] = weakref.ref(alias)
        for attr in ("_e3k_key", "_e3k_data_only", "_e3k_param_only", "_e3k_blocks"):      # row keys / provenance marks / the block count
            # of a keyed source's stacked knot basis (without it the guard differenced across the tables' seams) ride along
            if hasattr(t, attr):
                setattr(alias, attr, getattr(t, attr))
    return alias


class FactorizedConvolution(Module):
    avg_num_neighbors: Optional[float]
    use_sc: bool

    def __init__(self, input_features, output_features, node_attrs, edge_radial, edge_spherical,
                 invariant_layers=1, invariant_neurons=8, avg_num_neighbors=None, use_sc=True,
                 nonlinearity_scalars: Dict[int, Callable] = {"e": "ssp"}, reduce=True) -> None:
        super().__init__()
        self.init_irreps(input_features=input_features, output_features=output_features, node_attrs=node_attrs,
                         edge_radial=edge_radial, edge_spherical=edge_spherical, output_keys=["output_features"])
        self.avg_num_neighbors = avg_num_neighbors
        self.use_sc = use_sc
        self.reduce = reduce
        f_in = Irreps(self.irreps_in["input_features"])
        f_out = Irreps(self.irreps_out["output_features"])
        sh = Irreps(self.irreps_in["edge_spherical"])

        self.linear_1 = Linear(f_in, f_in)
        self.tp = TensorProductExpansion(f_in, (sh, "edge_spherical"), (f_out, "edge_features"), "uvu",
                                         internal_weight=False)
        n_radial = Irreps(self.irreps_in["edge_radial"]).num_irreps
        self._weight_numel = int(self.tp.tp.weight_numel)
        self.fc = FullyConnectedNet([n_radial] + invariant_layers * [invariant_neurons] + [self.tp.tp.weight_numel],
                                    activations["ssp"])
        self.sc = None
        if self.use_sc:
            self.sc = FullyConnectedTensorProduct(f_in, Irreps(self.irreps_in["node_attrs"]), f_out)
        self._in_blocks = tuple(irreps_blocks(f_in))
        self._out_blocks = tuple(irreps_blocks(f_out))

    _next_conv = None      # set by SequentialGraphNetwork

    def _fork_pays(self, n_edges: int, table: bool = False) -> bool:
        # enough per-edge weights in this layer, or so many edges that even the narrow first layer is worth it
        if table:       # (every layer of the network the same way: a mix of forked and one-stream layers was the slowest)
            return n_edges >= FORK_MIN_EDGES_TABLE
        ref = FORK_MIN_EDGES
        return (n_edges * self._weight_numel >= ref * _FORK_REF_WIDTH) or n_edges >= 2 * ref

    def forward_cf(self, data: Dict[str, Tensor]) -> Tensor:
        """Convolution output [N, out.dim] in the channel-fastest layout (reduce=True path)."""
        x = data["input_features"]
        topo = get_topology(data, x.shape[0])
        in_cf = bool(getattr(x, "_e3k_cf", False))     # the previous MessagePassing handed its features over in cf
        table = radial_table.applicable(data["edge_radial"], radial_table.last_weight(self.fc))
        if table:      # knot bins and the basis on the knots: once per batch / forward, on this stream (the branches wait for it)
            src = radial_table.source_of(data["edge_radial"])
            src.bins()
            src.knot_basis()
        if (FWD_FORK and x.is_cuda and self._fork_pays(data["edge_radial"].shape[0], bool(table))
                and (ops.FORK_IN_CAPTURE or not torch.cuda.is_current_stream_capturing())):
            # the radial MLP (edge side: one big GEMM) and the node side (relayout, self-connection, linear_1: small
            # launches that leave most CUs idle) are independent until the tensor product: run them on two streams
            with ops.IN_FORK:
                main = ops.current_stream(x.device)
                side = ops.side_stream(x.device)
                radial = data["edge_radial"]
                # (issuing the NEXT convolution's radial MLP here, one layer early -- rounds 1-4's E3K_RADIAL_AHEAD -- measured no gain
                #  on this composed path and was removed in round 5; the fused block has its own look-ahead)
                side.wait_stream(main)
                with ops.on_stream(side, main):
                    weight = (radial_table.table_weights(self.fc, radial) if table
                              else self.fc(_stream_alias(radial, side)))
                    ready = torch.cuda.Event()
                    ready.record(side)
                x_cf = x if in_cf else ops.relayout(x, self._in_blocks, True)
                sc = None
                if self.sc is not None and FWD_FORK_SC:
                    # third branch: the self-connection only meets the others at the trailing Linear
                    side2 = ops.side_stream(x.device, 1)
                    side2.wait_stream(main)
                    with ops.on_stream(side2, main):
                        sc = self.sc(x_cf, _stream_alias(data["node_attrs"], side2))
                elif self.sc is not None:
                    sc = self.sc(x_cf, data["node_attrs"])
                x1 = self.linear_1(x_cf, in_layout="cf", out_layout="cf")
                main.wait_event(ready)
                weight.record_stream(main)
                if sc is not None and FWD_FORK_SC:
                    mid = self.tp.tp.fused(x1, data["edge_spherical"], weight, topo)
                    main.wait_stream(side2)
                    sc.record_stream(main)
                    scale = 1.0 if self.avg_num_neighbors is None else float(self.avg_num_neighbors) ** -0.5
                    return self.tp.linear(mid, in_layout="cf", out_layout="cf", base=sc, scale=scale)
        else:
            weight = radial_table.table_weights(self.fc, data["edge_radial"]) if table else self.fc(data["edge_radial"])
            x_cf = x if in_cf else ops.relayout(x, self._in_blocks, True)
            sc = self.sc(x_cf, data["node_attrs"]) if self.sc is not None else None
            x1 = self.linear_1(x_cf, in_layout="cf", out_layout="cf")
        mid = self.tp.tp.fused(x1, data["edge_spherical"], weight, topo)
        scale = 1.0 if self.avg_num_neighbors is None else float(self.avg_num_neighbors) ** -0.5
        return self.tp.linear(mid, in_layout="cf", out_layout="cf", base=sc, scale=scale)

    def forward(self, data: Dict[str, Tensor], attrs: Dict[str, Tuple[str, str]]):
        if self.reduce:
            out = ops.relayout(self.forward_cf(data), self._out_blocks, False)
        else:
            # per-edge messages, no reduction: the reference's unfused module API
            x = data["input_features"]
            weight = self.fc(data["edge_radial"])
            x1 = self.linear_1(x)
            out = self.tp(left=x1[data["edge_index"][0]], right=data["edge_spherical"], weight=weight)
        return ({"output_features": out},
                {"output_features": (attrs["input_features"][0], self.irreps_out["output_features"])})


class MessagePassing(Module):
    """Convolution -> gate nonlinearity -> optional residual -> optional LayerNormalization."""

    def __init__(self, input_features, output_features, node_attrs, edge_radial, edge_spherical, convolution,
                 resnet: bool = False, nonlinearity_type: str = "gate",
                 nonlinearity_scalars: Dict[int, Callable] = {"e": "ssp", "o": "tanh"},
                 nonlinearity_gates: Dict[int, Callable] = {"e": "ssp", "o": "abs"}, normalize=False):
        super().__init__()
        self.init_irreps(input_features=input_features, output_features=output_features, node_attrs=node_attrs,
                         edge_radial=edge_radial, edge_spherical=edge_spherical, output_keys=["output_features"])
        assert nonlinearity_type in ("gate", "norm")
        acts_s = {1: nonlinearity_scalars["e"], -1: nonlinearity_scalars["o"]}
        acts_g = {1: nonlinearity_gates["e"], -1: nonlinearity_gates["o"]}
        sh = Irreps(self.irreps_in["edge_spherical"])
        prev = Irreps(self.irreps_in["input_features"])
        self.feature_irreps_hidden = Irreps(self.irreps_out["output_features"])

        reachable = [mi for mi in self.feature_irreps_hidden if tp_path_exists(prev, sh, mi.ir)]
        irreps_scalars = Irreps([mi for mi in reachable if mi.ir.l == 0])
        irreps_gated = Irreps([mi for mi in reachable if mi.ir.l > 0])
        irreps_layer_out = (irreps_scalars + irreps_gated).simplify()
        irreps_gates = Irreps([(mi.mul, "0e") for mi in irreps_gated])
        if nonlinearity_type == "gate":
            self.equivariant_nonlin = Gate(
                irreps_scalars=irreps_scalars, act_scalars=[acts_s[mi.ir.p] for mi in irreps_scalars],
                irreps_gates=irreps_gates, act_gates=[acts_g[mi.ir.p] for mi in irreps_gates],
                irreps_gated=irreps_gated)
            conv_irreps_out = self.equivariant_nonlin.irreps_in.simplify()
        else:   # :207-219: the norm is an even scalar, so the 'e' scalar nonlinearity acts on it
            conv_irreps_out = irreps_layer_out.simplify()
            self.equivariant_nonlin = NormActivation(conv_irreps_out, nonlinearity_scalars["e"], normalize=True,
                                                     epsilon=1e-8, bias=False)
        self.resnet = bool(resnet) and irreps_layer_out == prev
        self.conv = build(convolution, input_features=input_features, output_features=conv_irreps_out,
                          node_attrs=node_attrs, edge_radial=edge_radial, edge_spherical=edge_spherical)
        self.normalize = normalize
        if self.normalize:
            self.norm = LayerNormalization(self.irreps_out["output_features"], self.irreps_out["output_features"])

    # Set by SequentialGraphNetwork when the NEXT layer is a MessagePassing that reads (and overwrites) this layer's
    # output key and nothing else can see it: the features are then handed over in the channel-fastest layout the
    # convolution kernels work in -- no cf -> e3nn relayout here, no e3nn -> cf relayout there (nor their backward passes).
    _emit_cf = False

    def cf_chain_ok(self, nxt) -> bool:
        """This layer may hand ``nxt`` its output in cf layout (see ``_emit_cf``)."""
        # (LayerNormalization is a per-block reduction over ALL elements of an irrep block: the same numbers in either layout,
        #  so a normalised layer hands over in cf too -- the protein score net's eight layers)
        return (CF_CHAIN and isinstance(nxt, MessagePassing) and isinstance(self.equivariant_nonlin, Gate)
                and not self.resnet and (not self.normalize or CF_CHAIN_NORM) and not nxt.resnet
                and isinstance(nxt.conv, FactorizedConvolution) and nxt.conv.reduce
                and tuple(irreps_blocks(self.equivariant_nonlin.irreps_out)) == nxt.conv._in_blocks)

    # ---- the layer as one autograd node (backend/conv_block.py) --------------------------------------------------
    def _block_plan(self, addend: bool = False):
        """Static part of the fused block, or None when this layer's structure is not served by it.  ``addend``: the variant
        whose self-connection is computed OUTSIDE the block and handed in (general node attributes, ``ConvBlockPlan.addend``)."""
        slot = "_cb_plan_add" if addend else "_cb_plan"
        plan = self.__dict__.get(slot, False)
        if plan is not False:
            return plan
        plan = None
        conv = self.conv
        ok = (isinstance(conv, FactorizedConvolution) and conv.reduce and isinstance(self.equivariant_nonlin, Gate)
              and conv.fc.fused_hidden and conv.tp.tp.plan is not None and conv.linear_1.bias is None
              and conv.tp.linear.bias is None)
        if ok:
            fc = list(conv.fc.children())
            hidden, last = fc[:-1], fc[-1]
            sc_spec = sc_m_off = sc_ld = None
            if conv.sc is not None and not addend:
                sc_spec = conv.sc._spec
                sc_m_off, pos = [], 0
                for ins in sc_spec.instr:
                    sc_m_off.append(pos)
                    pos += ins.mul_in * ins.mul_out
                sc_ld = pos
            plan = conv_block.ConvBlockPlan(
                in_blocks=conv._in_blocks, lin1_spec=conv.linear_1.spec("cf", "cf"),
                mlp_alphas=[1.0 / math.sqrt(m.h_in) for m in hidden], mlp_act=conv.fc.act_name, mlp_cst=hidden[0].cst,
                mlp_k0=hidden[0].h_in,
                last_spec=last._spec, tp_plan=conv.tp.tp.plan, post_spec=conv.tp.linear.spec("cf", "cf"),
                scale=1.0 if conv.avg_num_neighbors is None else float(conv.avg_num_neighbors) ** -0.5,
                sc_spec=sc_spec, sc_m_off=sc_m_off, sc_ld_m=sc_ld, gate_spec=self.equivariant_nonlin._spec,
                addend=bool(addend and conv.sc is not None))
            plan.guard_key = last.weight
        self.__dict__[slot] = plan
        return plan

    def _stack_rows(self, cache: dict, rows, edge_radial, plan, fork: bool, use_table: bool, slope=None, bessel=None):
        """This layer's radial-MLP output rows (on the knots of the table, or per edge), from one batched evaluation of the
        MLPs of every layer that follows on the same edge embedding (``_next_mp`` chain) -- or None when the stack does not
        apply.  The first layer of a chain computes it (on the radial stream when forked) and leaves the others' rows in
        ``cache`` (which lives as long as this forward pass's edge embedding); every layer takes its own entry out.
        ``slope`` (force training; see ``conv_native.RadialStackFn``): the entry is then the pair (table T, slope table D)."""
        from ..backend import conv_native

        if not conv_native.ENABLED or conv_native.native_layer(plan) is None:
            return None
        grad = torch.is_grad_enabled()
        hit = cache.pop(id(self), None)
        if hit is not None and hit[1] == (grad, fork, slope is not None):
            return hit[0]
        sig = (plan.mlp_k0, plan.last_spec.d_in, tuple(plan.mlp_alphas), plan.mlp_act, plan.mlp_cst)
        chain, m = [], self
        n_edges = edge_radial.shape[0]
        while m is not None and len(chain) < 16:
            pl = m._block_plan() if m is not self else plan
            if pl is None or conv_native.native_layer(pl) is None:
                break
            fc = list(m.conv.fc.children())
            if (pl.mlp_k0, pl.last_spec.d_in, tuple(pl.mlp_alphas), pl.mlp_act, pl.mlp_cst) != sig:
                break
            if m is not self and slope is not None:
                if not radial_table.knots_for(edge_radial, fc[-1].weight, allow_grad=True):
                    break
            elif m is not self and (radial_table.applicable(edge_radial, fc[-1].weight) != use_table
                                    or bool(FWD_FORK and m.conv._fork_pays(n_edges, use_table))
                                    != bool(FWD_FORK and self.conv._fork_pays(n_edges, use_table))):
                break
            chain.append((m, pl, fc))
            m = m.__dict__.get("_next_mp")
        weights = []
        for _, _, fc in chain:
            weights.append(fc[-1].weight)
            weights.extend(mod.weight for mod in fc[:-1])
        plans = [pl for _, pl, _ in chain]
        if fork:
            main = ops.current_stream(rows.device)
            side = ops.side_stream(rows.device)
            side.wait_stream(main)              # (the knot basis was evaluated on this stream)
            with ops.on_stream(side, main):     # forward AND backward of the stack live on the radial stream
                outs = conv_native.RadialStackFn.apply(rows, plans, use_table, main, slope, bessel, *weights)
        else:
            outs = conv_native.RadialStackFn.apply(rows, plans, use_table, None, slope, bessel, *weights)
        if slope is not None:
            n = len(chain)
            outs = [(outs[i], outs[n + i]) for i in range(n)]
        for (m, _, _), out in zip(chain[1:], outs[1:]):
            cache[id(m)] = (out, (grad, fork, slope is not None))
        return outs[0]

    def _kw_stack_rows(self, node_attrs, attrs, groups, plan, fork: bool, n_edges: int, use_table: bool):
        """This layer's per-key self-connection weights M [n_keys, ld_m] from one batched evaluation for every layer of the
        ``_next_mp`` chain that has a keyed self-connection on the same attributes (``conv_native.KwStackFn``), or None.
        ``node_attrs``: the tensor the layers share (the rows of the others wait on it); ``attrs``: its alias on the
        self-connection stream when forked."""
        from ..backend import conv_native

        if not conv_native.ENABLED or conv_native.native_layer(plan) is None:
            return None
        grad = torch.is_grad_enabled()
        cache = getattr(node_attrs, "_e3k_kw_stack", None)
        if cache is None:
            cache = node_attrs._e3k_kw_stack = {}
        hit = cache.pop(id(self), None)
        if hit is not None and hit[1] == (grad, fork, id(groups)):
            return hit[0]
        chain, m = [], self
        while m is not None and len(chain) < 8:
            pl = m._block_plan() if m is not self else plan
            if pl is None or pl.sc_spec is None or conv_native.native_layer(pl) is None or pl.sc_spec.v != plan.sc_spec.v:
                break
            if m is not self and (bool(FWD_FORK and m.conv._fork_pays(n_edges, use_table))
                                  != bool(FWD_FORK and self.conv._fork_pays(n_edges, use_table))):
                break
            chain.append((m, pl))
            m = m.__dict__.get("_next_mp")
        plans = [pl for _, pl in chain]
        weights = [m.conv.sc.weight for m, _ in chain]
        if fork:
            main = ops.current_stream(attrs.device)
            side2 = ops.side_stream(attrs.device, 1)
            side2.wait_stream(main)             # (the attributes were produced on this stream)
            with ops.on_stream(side2, main):    # forward AND backward of the stack live on the self-connection stream
                outs = conv_native.KwStackFn.apply(attrs, groups, plans, main, *weights)
        else:
            outs = conv_native.KwStackFn.apply(attrs, groups, plans, None, *weights)
        for (m, _), out in zip(chain[1:], outs[1:]):
            cache[id(m)] = (out, (grad, fork, id(groups)))
        return outs[0]

    def _forward_force(self, data, out_cf: bool):
        """The layer through the force block (``backend/conv_force.py``: differentiable twice, on the value + slope knot tables)
        when the spherical harmonics and the radii require grad and the layer's structure is served, else None."""
        from ..backend import conv_force

        x, sh, radial = data["input_features"], data["edge_spherical"], data["edge_radial"]
        plan = self._block_plan()
        if plan is None or not conv_force.supported(plan, x.device):
            return None
        conv = self.conv
        src = radial_table.source_of(radial)
        if conv.sc is None or src is None or not src.r.requires_grad:
            return None
        attrs = data["node_attrs"]
        key = get_row_key(attrs)
        if not conv.sc.keyed_pays(key, x.shape[0]):
            return None
        fc = list(conv.fc.children())
        knots = radial_table.knots_for(radial, fc[-1].weight, allow_grad=True)
        if not knots:
            return None
        groups = row_groups(key[0], key[1])
        topo = get_topology(data, x.shape[0])
        bins = src.bins(knots)
        mod = src.module()
        b, c = mod.basis, mod.cutoff
        slope = (radial_table.knot_radii(float(b.r_max), knots, x.device), float(b.r_max), float(b.r_min), float(c.p),
                 int(b.one_over_r), int(c.cutoff.kind))
        tables = self._stack_rows(src._stack, src.knot_basis(knots), radial, plan, False, True, slope=slope, bessel=b.bessel_weights)
        if tables is None:
            return None
        m = None
        if KW_STACK and ops.GRAD_READY is None:
            m = self._kw_stack_rows(attrs, attrs, groups, plan, False, radial.shape[0], True)
        if m is None:
            m = ops.keyed_weights(attrs.index_select(0, groups.reps), conv.sc.weight, plan.sc_spec, plan.sc_m_off, plan.sc_ld_m)
        return conv_force.force_block(x, m, tables[0], tables[1], sh, src.r, plan, topo, groups, bins,
                                      bool(getattr(x, "_e3k_cf", False)), out_cf, conv.linear_1.weight, conv.tp.linear.weight)

    def _forward_block(self, data, out_cf: bool):
        """The layer through ``conv_block`` when it applies to this call, else None (composed path)."""
        if not conv_block.ENABLED:
            return None
        x, sh, radial = data["input_features"], data["edge_spherical"], data["edge_radial"]
        if x.is_cuda and sh.requires_grad:             # forces / double backward: the force block, else the composed ops
            return self._forward_force(data, out_cf)
        if not x.is_cuda:
            return None
        plan = self._block_plan()
        if plan is None:
            return None
        conv = self.conv
        groups = attrs = None
        if conv.sc is not None:
            attrs = data["node_attrs"]
            key = get_row_key(attrs)
            if not conv.sc.keyed_pays(key, x.shape[0]):
                # general (un-keyed) attributes: the self-connection is the outer-product form (ops.fctp) -- computed here, its
                # output handed to the block, which adds the convolution to it in front of the gate (config_diffusion: atom
                # type x the molecule's time embedding, 2 304 distinct rows over 2 335 nodes)
                return self._forward_block_addend(data, x, sh, radial, out_cf)
            groups = row_groups(key[0], key[1])
        topo = get_topology(data, x.shape[0])
        table = None
        fc = list(conv.fc.children())
        n_edges = radial.shape[0]
        if radial_table.applicable(radial, fc[-1].weight):      # the radial MLP on a knot table; every edge interpolates (backend/radial_table.py)
            src = radial_table.source_of(radial)
            table = src.bins()
            radial = src.knot_basis()            # [knots + 1, n_basis]: the block's MLP runs on these rows
        fork = bool(FWD_FORK and conv._fork_pays(n_edges, table is not None)
                    and (ops.FORK_IN_CAPTURE or not torch.cuda.is_current_stream_capturing()))
        if fork:      # gradient contributions of the shared inputs are summed on the streams that produce them
            main = ops.current_stream(x.device)
            side = ops.side_stream(x.device)
            with ops.on_stream(side, main):          # (the alias's autograd node lives on the stream current NOW)
                radial = _stream_alias(radial, side)
            if attrs is not None:
                side2 = ops.side_stream(x.device, 1)
                with ops.on_stream(side2, main):
                    attrs = _stream_alias(attrs, side2)
        m_pre = None
        if KW_STACK and groups is not None and ops.GRAD_READY is None and n_edges <= KW_STACK_MAX_EDGES:
            m_pre = self._kw_stack_rows(data["node_attrs"], attrs, groups, plan, fork, n_edges, table is not None)
        nxt = None
        nmp = self.__dict__.get("_next_mp")          # set by SequentialGraphNetwork: the next layer reads the same edge embedding
        pre = None
        # (on ONE stream -- small batches, a captured step -- there is no backward overlap to lose: the stack at every size;
        #  graph-replayed 256 / 512 molecules 5.57 -> 5.33 / 9.14 -> 8.83 ms)
        if RADIAL_STACK and (n_edges <= STACK_MAX_EDGES or not fork):
            if table is not None:
                cache = src._stack
            else:      # per-edge MLPs: the rows of the layers wait on the edge embedding of this forward pass
                cache = getattr(data["edge_radial"], "_e3k_stack", None)
                if cache is None:
                    cache = data["edge_radial"]._e3k_stack = {}
            pre = self._stack_rows(cache, radial, data["edge_radial"], plan, fork, table is not None)
        if pre is not None:      # the MLPs of all the layers on this edge embedding ran as one batch (conv_native.RadialStackFn)
            if nmp is not None and fork and table is not None:
                pre_n = cache.get(id(nmp))
                if pre_n is not None:
                    nxt = (nmp._block_plan(), pre_n[0])
            return conv_block.conv_block(x, attrs, None, sh, plan, topo, groups, bool(getattr(x, "_e3k_cf", False)), out_cf, fork,
                                         conv.linear_1.weight, conv.tp.linear.weight,
                                         conv.sc.weight if conv.sc is not None else None, None, (), table=table, nxt=nxt, pre=pre,
                                         m_pre=m_pre)
        if nmp is not None and fork and nmp.conv._fork_pays(n_edges, table is not None):
            plan_n = nmp._block_plan()
            if plan_n is not None and (nmp.conv.sc is None) == (conv.sc is None):
                fc_n = list(nmp.conv.fc.children())
                # (the next layer must take the same radial path as this one: its guard may have vetoed the table)
                same_path = (table is not None) == radial_table.applicable(data["edge_radial"], fc_n[-1].weight)
                if fc_n[0].weight.shape[0] == radial.shape[1] and same_path:
                    nxt = (plan_n, fc_n[-1].weight, [m.weight for m in fc_n[:-1]])
        y = conv_block.conv_block(x, attrs, radial, sh, plan, topo, groups, bool(getattr(x, "_e3k_cf", False)), out_cf, fork,
                                  conv.linear_1.weight, conv.tp.linear.weight, conv.sc.weight if conv.sc is not None else None,
                                  fc[-1].weight, [m.weight for m in fc[:-1]], table=table, nxt=nxt, m_pre=m_pre)
        return y

    def _forward_block_addend(self, data, x, sh, radial, out_cf: bool):
        from ..backend import conv_native

        plan = self._block_plan(addend=True) if BLOCK_ADDEND else None
        if plan is None or not conv_native.ENABLED or conv_native.native_layer(plan) is None:
            return None
        conv = self.conv
        x_cf = x if getattr(x, "_e3k_cf", False) else ops.relayout(x, conv._in_blocks, True)
        sc_out = conv.sc(x_cf, data["node_attrs"])           # [N, d_conv], cf
        topo = get_topology(data, x.shape[0])
        fc = list(conv.fc.children())
        table = None
        if radial_table.applicable(radial, fc[-1].weight):
            src = radial_table.source_of(radial)
            table = src.bins()
            radial = src.knot_basis()
        # no stacks, no look-ahead (the shapes that reach this are the score nets); the radial branch and the weight
        # gradients fork onto their streams under the same rule as the keyed block
        fork = bool(ADDEND_FORK and FWD_FORK and conv._fork_pays(data["edge_radial"].shape[0], table is not None)
                    and (ops.FORK_IN_CAPTURE or not torch.cuda.is_current_stream_capturing()))
        if fork:
            main = ops.current_stream(x.device)
            side = ops.side_stream(x.device)
            with ops.on_stream(side, main):
                radial = _stream_alias(radial, side)
        return conv_block.conv_block(x_cf, None, radial, sh, plan, topo, None, True, out_cf, fork,
                                     conv.linear_1.weight, conv.tp.linear.weight, None,
                                     fc[-1].weight, [m.weight for m in fc[:-1]], table=table, m_pre=sc_out)

    def prepare_batch(self, network, data, avail) -> None:
        """``SequentialGraphNetwork.prepare_data`` hook: the EDGE RECORDS the packed-table tensor-product kernels walk (one 64-byte
        block per edge and CSR direction: ``KnotBins.records``) depend on the batch alone -- topology, knot bins, spherical
        harmonics -- and are shared by all layers: built here when all three are known before the step."""
        from ..backend import conv_native
        from ..backend.graph import GraphTopo

        if not (conv_native.ENABLED and conv_native.TP_TABLE and conv_native.TP_TABLE_PACKED):
            return
        inv = {loc: g for g, loc in self.input_key_mapping.items()}
        sh_key, rad_key = inv.get("edge_spherical"), inv.get("edge_radial")
        sh = data.get(sh_key) if sh_key in avail else None
        topo = GraphTopo.from_dict(data)
        if sh is None or topo is None or not sh.is_cuda or sh.dim() != 2 or sh.shape[1] > 9 or sh.shape[0] != topo.num_edges:
            return
        for _, layer in network.layers:      # the layer that produces this convolution's edge embedding: its radii carry the bins
            if rad_key in getattr(layer, "output_key_mapping", {}).values() and hasattr(layer, "basis"):
                r_key = next((g for g, loc in layer.input_key_mapping.items() if loc == "input"), None)
                bins = radial_table.prepared_bins(data[r_key]) if (r_key in avail and r_key in data) else None
                for b in (bins or {}).values():
                    b.records(topo, sh, "dst")
                    if torch.is_grad_enabled() or self.training:
                        b.records(topo, sh, "src")

    def forward(self, data: Dict[str, Tensor], attrs: Dict[str, Tuple[str, str]]):
        old_x = data["input_features"]
        blk = self._forward_block(data, bool(self._emit_cf))
        if blk is not None:
            if self._emit_cf:
                if self.normalize:
                    blk = self.norm({"input": blk}, attrs)[0]["output"]
                blk._e3k_cf = True
                return ({"output_features": blk},
                        {"output_features": (attrs["input_features"][0], self.irreps_out["output_features"])})
            output = blk
            if self.resnet:
                output = old_x + output
            if self.normalize:
                output = self.norm({"input": output}, attrs)[0]["output"]
            return ({"output_features": output},
                    {"output_features": (attrs["input_features"][0], self.irreps_out["output_features"])})
        if isinstance(self.conv, FactorizedConvolution) and self.conv.reduce:
            conv_cf = self.conv.forward_cf(data)
        else:
            if getattr(old_x, "_e3k_cf", False):
                raise RuntimeError("cf hand-over reached a convolution that cannot take it (container bug)")
            out, _ = self.conv(data, attrs)
            conv_cf = ops.relayout(out["output_features"], tuple(irreps_blocks(self.equivariant_nonlin.irreps_in)), True)
        if self._emit_cf:
            output = self.equivariant_nonlin(conv_cf, out_cf=True)
            if self.normalize:
                output = self.norm({"input": output}, attrs)[0]["output"]
            output._e3k_cf = True
            return ({"output_features": output},
                    {"output_features": (attrs["input_features"][0], self.irreps_out["output_features"])})
        output = self.equivariant_nonlin(conv_cf)
        if self.resnet:
            output = old_x + output
        if self.normalize:
            output = self.norm({"input": output}, attrs)[0]["output"]
        return ({"output_features": output},
                {"output_features": (attrs["input_features"][0], self.irreps_out["output_features"])})
