"""Edge geometry entry points of the layer graph.

``computeEdgeVector`` mirrors ``e3_layers/data/compute_edge.py:13-36`` (edge_vec = pos[dst] -
pos[src], edge_length) and runs the HIP kernel of ``csrc/e3k_edge.hip``; as the first layer of
every model (``e3_layers/configs/layer_configs.py:43``) it also attaches the per-batch CSR
topology the fused convolution needs (``backend/graph.py``).

``computeEdgeIndex`` mirrors ``e3_layers/data/compute_edge.py:38-113`` with the intent recorded
in SURVEY.md appendix C: per graph all ordered pairs in (src slow, dst fast) order, keep
``|pos_src - pos_dst| < r_max`` (strict, fp32) or ``criteria``, drop self loops, keep
pre-existing edges and carry their edge attributes (zero rows for new edges).  On CPU tensors
(dataset preprocessing, as in the reference) it is vectorised torch integer plumbing; on device
tensors without a ``criteria`` callback it runs the two-pass HIP radius-graph kernels
(``e3k_radius_graph_count/fill``: one wave per source node, ballot compaction keeps the
reference's edge order without a sort) — the per-step edge rebuild of the sampling loop
(``e3_layers/run/sde_sampling.py:237-242``) then never leaves the GPU.  ``criteria`` callbacks are
arbitrary Python over the candidate list and keep the torch path on whichever device holds the data.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
from torch import Tensor

from ..backend import lib as L
from ..backend import ops
from ..backend.graph import TOPO_KEYS, get_topology
from .data import segment_ids


def computeEdgeVector(data: Dict[str, Tensor], attrs: Dict[str, Tuple[str, str]], key: str = "pos",
                      with_lengths: bool = True):
    attrs["edge_vector"] = ("edge", "1x1o")
    attrs["edge_length"] = ("edge", "1x0e")
    pos = data[key]
    if "edge_vector" in data:
        if with_lengths and "edge_length" not in data:
            data["edge_length"] = torch.linalg.norm(data["edge_vector"], dim=-1)
        return data, attrs
    topo = get_topology(data, pos.shape[0])
    data.update(topo.as_dict())
    vec, length = ops.edge_vector(pos, topo)
    # functions of the input positions alone: no gradient under ops.params_only_backward()
    from_input = pos.grad_fn is None and not isinstance(pos, torch.nn.Parameter)
    ops.mark_data_only(vec, from_input)
    ops.mark_data_only(length, from_input)
    data["edge_vector"] = vec
    if with_lengths:
        data["edge_length"] = length
    return data, attrs


computeEdgeVector.data_only_inputs = ("pos", "edge_index")      # (SequentialGraphNetwork.prepare_data: parameter-free, reads these keys)


def _all_pairs(n_nodes: Tensor, device) -> Tensor:
    """[2, sum n_g^2] candidate edges, graphs concatenated, (i, j) lexicographic inside a graph; built on ``device``
    (one host sync for the candidate count when the counts live on the GPU)."""
    dev = torch.device(device)
    n = n_nodes.reshape(-1).to(dev, torch.long)
    if n.numel() == 0:
        return torch.zeros(2, 0, dtype=torch.long, device=dev)
    sq = n * n
    total_sq = int(sq.sum())
    if total_sq == 0:
        return torch.zeros(2, 0, dtype=torch.long, device=dev)
    start = torch.cumsum(n, 0) - n
    graph = segment_ids(sq, total_sq)
    local = torch.arange(total_sq, device=dev) - (torch.cumsum(sq, 0) - sq)[graph]
    ng = n[graph]
    src = torch.div(local, ng, rounding_mode="floor") + start[graph]
    dst = local % ng + start[graph]
    return torch.stack([src, dst])


_STALE = ("_edge_segment",) + TOPO_KEYS + ("edge_vector", "edge_length")


def _radius_graph_device(data, attrs, pos: Tensor, r_max: float):
    """HIP path: edge_index [2,E] int64 in the reference's order, `_n_edges`, carried-over edge attributes."""
    dev = pos.device
    pos = L.f32c(pos.detach())
    n = data["_n_nodes"].reshape(-1).to(dev)
    total = pos.shape[0]
    n_graphs = n.numel()
    gid = segment_ids(n, total)
    ends = torch.cumsum(n, 0)
    g_end = ends[gid].to(torch.int32)
    g_start = (ends - n)[gid].to(torch.int32)
    old_ptr = old_dst = old_id = None
    if "edge_index" in data:
        old = data["edge_index"].to(dev)
        if old.numel() and not bool((gid[old[0]] == gid[old[1]]).all()):
            raise ValueError("an existing edge connects two different graphs")
        old_id = old[0] * total + old[1]
        order = torch.argsort(old_id)
        old_dst = old[1][order].to(torch.int32).contiguous()
        old_ptr = torch.zeros(total + 1, dtype=torch.int32, device=dev)
        old_ptr[1:] = torch.cumsum(torch.bincount(old[0], minlength=total), 0).to(torch.int32)
    lib = L.load()
    counts = torch.empty(total, dtype=torch.int32, device=dev)
    L.check(lib.e3k_radius_graph_count(L.ptr(pos), L.ptr(g_start), L.ptr(g_end), total, float(r_max), L.ptr(old_ptr),
                                       L.ptr(old_dst), L.ptr(counts), L.stream_ptr()), "e3k_radius_graph_count")
    incl = torch.cumsum(counts, 0, dtype=torch.int64)
    offsets = (incl - counts).contiguous()
    n_edge = int(incl[-1].item()) if total else 0          # the one host sync: the size of the output
    edge_index = torch.empty(2, n_edge, dtype=torch.int64, device=dev)
    L.check(lib.e3k_radius_graph_fill(L.ptr(pos), L.ptr(g_start), L.ptr(g_end), total, float(r_max), L.ptr(old_ptr),
                                      L.ptr(old_dst), L.ptr(offsets), n_edge, L.ptr(edge_index), L.stream_ptr()),
            "e3k_radius_graph_fill")
    if old_id is not None:
        where = torch.searchsorted(edge_index[0] * total + edge_index[1], old_id)
        for k in list(attrs.keys()):
            if attrs[k][0] == "edge" and k in data:
                prev = data[k].to(dev)
                fresh = torch.zeros((n_edge,) + tuple(prev.shape[1:]), dtype=prev.dtype, device=dev)
                fresh[where] = prev
                data[k] = fresh
    per_graph = torch.zeros(n_graphs, dtype=torch.int64, device=dev).index_add_(0, gid, counts.to(torch.int64))
    attrs["_n_edges"] = ("graph", "1x0e")
    data["_n_edges"] = per_graph.view(-1, 1)
    for k in _STALE:
        data.pop(k, None)
    return {"edge_index": edge_index}, attrs


def computeEdgeIndex(data, attrs, r_max: float = None, key: str = "pos", criteria=None):
    pos = torch.as_tensor(data[key], dtype=torch.get_default_dtype())
    if pos.is_cuda and criteria is None and r_max is not None:
        return _radius_graph_device(data, attrs, pos, r_max)
    n_nodes = data["_n_nodes"]
    total = pos.shape[0]
    cand = _all_pairs(n_nodes, pos.device)
    dist = torch.linalg.norm(pos[cand[0]] - pos[cand[1]], dim=-1)
    keep = dist < r_max
    if criteria is not None:
        keep = torch.logical_or(keep, criteria(data, cand))
    keep = torch.logical_and(keep, cand[0] != cand[1])
    had_edges = "edge_index" in data
    if had_edges:
        old = data["edge_index"]
        old_id = old[0] * total + old[1]
        cand_id = cand[0] * total + cand[1]          # strictly increasing
        at = torch.searchsorted(cand_id, old_id)
        if not bool((cand_id[at.clamp(max=cand_id.numel() - 1)] == old_id).all()):
            raise ValueError("an existing edge connects two different graphs")
        keep[at] = True
    edge_index = cand[:, keep]
    if had_edges:
        where = torch.searchsorted(edge_index[0] * total + edge_index[1], old_id)
        for k in list(attrs.keys()):
            if attrs[k][0] == "edge" and k in data:
                prev = data[k]
                fresh = torch.zeros((edge_index.shape[1],) + tuple(prev.shape[1:]), dtype=prev.dtype, device=pos.device)
                fresh[where] = prev
                data[k] = fresh
    seg = segment_ids(n_nodes.reshape(-1).to(pos.device))
    n_edges = torch.bincount(seg[edge_index[0]], minlength=n_nodes.numel()).view(-1, 1)
    attrs["_n_edges"] = ("graph", "1x0e")
    data["_n_edges"] = n_edges
    # stale per-edge caches belong to the old edge set
    for k in _STALE:
        data.pop(k, None)
    return {"edge_index": edge_index}, attrs
