"""``Data`` / ``Batch``: dict-of-tensors graph containers (host-side plumbing).

Interface contract taken from the reference (``e3_layers/data/data.py:13-96,170-202`` and
``e3_layers/data/batch.py:10-201``; SURVEY.md §8b "Batch contract"):

* ``attrs[key] = (is_per, irreps)`` with ``is_per`` in {'node','edge','graph'}; a described
  tensor is stored as ``[count, irreps.dim]``;
* keys containing ``index`` or ``face`` concatenate on the last dimension, everything else on
  dim 0; ``edge_index`` is offset by the running node count when graphs are batched;
* ``_n_nodes`` / ``_n_edges`` are ``[G, 1]`` int64, ``_node_segment`` / ``_edge_segment`` are the
  per-node / per-edge graph ids;
* floats are stored as fp32 and integers as int64 by ``Batch.from_data_list``;
* a ``Batch`` is dict-like by str key and list-like by int / slice / index array.

Written vectorised: segments come from ``repeat_interleave`` on the tensor's own device (the
reference builds Python lists per graph and syncs per element).
"""
from __future__ import annotations

import copy
from collections.abc import Sequence
from typing import Dict, Iterable, Optional

import numpy as np
import torch

from ..o3 import Irreps

_INT_DTYPES = (torch.int64, torch.int32, torch.int16, torch.int8, torch.uint8, torch.bool)


def feature_dim(spec) -> Optional[int]:
    """Width implied by an attrs entry: an int, a digit string or an irreps string."""
    if spec is None:
        return None
    if isinstance(spec, int):
        return spec
    if isinstance(spec, str) and spec.isdigit():
        return int(spec)
    try:
        return Irreps(spec).dim
    except (ValueError, TypeError):
        return None


def cat_dim(key: str) -> int:
    return -1 if ("index" in key or "face" in key) else 0


class Data:
    def __init__(self, attrs=None, **tensors):
        self.attrs = {} if attrs is None else attrs
        self.data: Dict[str, torch.Tensor] = {}
        self.device = None
        for key, value in tensors.items():
            self[key] = value

    # ---- dict protocol ------------------------------------------------------
    def keys(self):
        return self.data.keys()

    def items(self):
        return list(self.data.items())

    def values(self):
        return self.data.values()

    def __contains__(self, key) -> bool:
        return key in self.data

    def __iter__(self):
        return iter(self.data)

    def __len__(self) -> int:
        return len(self.data)

    def __getitem__(self, key):
        return self.data[key]

    def __setitem__(self, key, value):
        if not isinstance(value, torch.Tensor):
            value = torch.as_tensor(value)
        width = feature_dim(self.attrs[key][1]) if key in self.attrs else None
        if width is not None and not (value.dim() == 2 and value.shape[-1] == width):
            value = value.reshape(-1, width)
        self.data[key] = value

    def update(self, other):
        for key, value in (other.items() if hasattr(other, "items") else other):
            self[key] = value
        return self

    def pop(self, key):
        self.attrs.pop(key, None)
        return self.data.pop(key, None)

    def get(self, key, default=None):
        return self.data.get(key, default)

    # ---- counts -------------------------------------------------------------
    def _count(self, kind: str) -> Optional[int]:
        for key, value in self.data.items():
            if key in self.attrs and self.attrs[key][0] == kind:
                return value.shape[cat_dim(key)]
        return None

    @property
    def n_nodes(self):
        if "_n_nodes" in self.data:
            return int(self.data["_n_nodes"].sum())
        return self._count("node")

    @property
    def n_edges(self):
        if "edge_index" in self.data:
            return self.data["edge_index"].shape[-1]
        return self._count("edge")

    @property
    def num_edges(self):
        return self.data["edge_index"].shape[-1]

    # ---- tensor plumbing ------------------------------------------------------
    def apply(self, func, *keys):
        for key in (keys or list(self.data.keys())):
            self.data[key] = func(self.data[key])
        return self

    def to(self, device, **kwargs):
        self.device = device
        return self.apply(lambda t: t.to(device, **kwargs))

    def cpu(self):
        return self.to("cpu")

    def cuda(self, device=None, non_blocking=False):
        return self.to("cuda" if device is None else device, non_blocking=non_blocking)

    def contiguous(self):
        return self.apply(lambda t: t.contiguous())

    def pin_memory(self):
        return self.apply(lambda t: t.pin_memory())

    def clone(self):
        out = self.__class__.__new__(self.__class__)
        out.attrs = copy.deepcopy(self.attrs)
        out.data = {k: v.clone() for k, v in self.data.items()}
        out.device = self.device
        return out

    def view(self):
        """A new container over the SAME tensors (new key dict, new attrs dict).  The network only ever adds keys
        to the batch it is given and never writes into an input tensor, so a training step that wants to keep its
        input batch pristine needs this, not ``clone()``'s ~50 device copies (``out = self.model(data.clone())``,
        e3_layers/run/trainer.py:365, copies every tensor each step)."""
        out = self.__class__.__new__(self.__class__)
        out.attrs = copy.deepcopy(self.attrs)
        out.data = dict(self.data)
        out.device = self.device
        done = getattr(self, "_e3k_done", None)      # (SequentialGraphNetwork.prepare_data: validated against the tensors at forward time)
        if done:
            out._e3k_done = done
        return out

    def __repr__(self):
        shapes = {k: (tuple(v.shape), str(v.dtype).replace("torch.", "")) for k, v in self.data.items()}
        return f"{self.__class__.__name__}(attrs={self.attrs}, tensors={shapes})"


def segment_ids(counts: torch.Tensor, output_size: Optional[int] = None) -> torch.Tensor:
    """[0]*counts[0] + [1]*counts[1] + ...  on the device of ``counts``.  On the host this goes through ``np.repeat``:
    ``torch.repeat_interleave`` runs its few thousand elements through the intra-op thread pool, which costs tens of
    milliseconds per call when the pool is oversubscribed (64 ms measured in an 8-CPU container, 0.03 ms single-threaded
    — it was 3/4 of ``Batch.from_data_list``)."""
    counts = counts.reshape(-1)
    if counts.device.type == "cpu":
        c = counts.detach().numpy()
        return torch.from_numpy(np.repeat(np.arange(c.shape[0], dtype=np.int64), c))
    return torch.repeat_interleave(torch.arange(counts.numel(), device=counts.device), counts, output_size=output_size)


class Batch(Data):
    """Padding-free concatenation of graphs."""

    def __init__(self, attrs=None, **tensors):
        super().__init__(attrs, **tensors)
        self._refresh_segments()

    def _refresh_segments(self):
        if "_n_nodes" in self.data:
            self.nodeSegment()
        if "_n_edges" in self.data:
            self.edgeSegment()

    def nodeSegment(self):
        seg = self.data.get("_node_segment")
        n = self.data["_n_nodes"]
        if seg is None or seg.device != n.device:  # trusted when present: no device sync
            seg = segment_ids(n)
            self.data["_node_segment"] = seg
        return seg

    def edgeSegment(self):
        seg = self.data.get("_edge_segment")
        n = self.data["_n_edges"]
        if seg is None or seg.device != n.device:
            seg = segment_ids(n)
            self.data["_edge_segment"] = seg
        return seg

    @staticmethod
    def _total(counts: torch.Tensor) -> int:
        return int(counts.sum()) if counts.numel() else 0

    @property
    def n_graphs(self) -> int:
        return self.data["_n_nodes"].shape[0]

    num_graphs = n_graphs

    def __len__(self) -> int:
        return self.n_graphs

    # ---- construction ----------------------------------------------------------
    @classmethod
    def from_data_list(cls, lst: Sequence, attrs=None):
        """Padding-free concatenation of samples (``e3_layers/data/batch.py:39-111``).  Samples may live on the host or on
        the device: counts are taken from shapes, the per-graph ``edge_index`` offsets are applied to the concatenated
        tensor by ONE gather-add (not one add per sample), and nothing is read back from the device -- a list of
        device-resident samples is collated without a host sync."""
        attrs = {} if attrs is None else attrs
        attrs["_n_nodes"] = ("graph", "1x0e")
        attrs["_n_edges"] = ("graph", "1x0e")
        items = [dict(x.items()) if not isinstance(x, dict) else dict(x) for x in lst]
        dev = next((v.device for v in items[0].values() if isinstance(v, torch.Tensor)), torch.device("cpu"))
        node_key = next((k for k in items[0] if k in attrs and attrs[k][0] == "node"), None)
        node_tot = []         # nodes per SAMPLE as host ints when shapes tell (a sample may itself hold several graphs)
        for it in items:
            if "_n_nodes" not in it:
                if node_key is None:
                    raise ValueError("cannot infer the number of nodes: no tensor is described as per-node")
                width = feature_dim(attrs[node_key][1])
                t = torch.as_tensor(it[node_key])
                n = t.reshape(-1, width).shape[0] if width else t.shape[0]
                it["_n_nodes"] = torch.full((1, 1), n, dtype=torch.long)
                node_tot.append(n)
            else:
                it["_n_nodes"] = torch.as_tensor(it["_n_nodes"], dtype=torch.long).reshape(-1, 1)
                if node_key is not None and node_key in it:
                    width = feature_dim(attrs[node_key][1])
                    t = torch.as_tensor(it[node_key])
                    node_tot.append(t.reshape(-1, width).shape[0] if width else t.shape[0])
                elif not it["_n_nodes"].is_cuda:
                    node_tot.append(int(it["_n_nodes"].sum()))
                else:
                    node_tot.append(None)
            if "edge_index" in it:
                ei = torch.as_tensor(it["edge_index"]).long()
                it["edge_index"] = ei
                if "_n_edges" not in it:
                    it["_n_edges"] = torch.full((1, 1), ei.shape[-1], dtype=torch.long)
            if "_n_edges" in it:
                it["_n_edges"] = torch.as_tensor(it["_n_edges"], dtype=torch.long).reshape(-1, 1)

        merged: Dict[str, torch.Tensor] = {}
        merged["_n_nodes"] = torch.cat([it["_n_nodes"].to(dev) for it in items])
        if "_n_edges" in items[0]:
            merged["_n_edges"] = torch.cat([it["_n_edges"].to(dev) for it in items])
        for key in items[0]:
            if key in merged or key in ("_node_segment", "_edge_segment"):
                continue
            if key == "edge_index":
                cat = torch.cat([it[key].to(dev) for it in items], dim=-1)
                if any(n is None for n in node_tot):          # counts only known on the device: one sync, as a last resort
                    node_tot = [int(it["_n_nodes"].sum()) for it in items]
                starts, run = [], 0
                for n in node_tot:
                    starts.append(run)
                    run += n
                sizes = [it[key].shape[-1] for it in items]
                offs = torch.repeat_interleave(torch.tensor(starts, dtype=torch.long).to(dev, non_blocking=True),
                                               torch.tensor(sizes, dtype=torch.long).to(dev, non_blocking=True),
                                               output_size=sum(sizes))
                merged[key] = cat + offs
                continue
            width = feature_dim(attrs[key][1]) if key in attrs else None
            parts = []
            for it in items:
                t = torch.as_tensor(it[key])
                parts.append(t.reshape(-1, width) if width is not None else t)
            t = torch.cat(parts, dim=cat_dim(key))
            merged[key] = t.long() if t.dtype in _INT_DTYPES else t.float()
        if dev.type != "cpu":        # segment ids with their lengths known from shapes: the constructor would sync for them
            if all(n is not None for n in node_tot):
                merged["_node_segment"] = segment_ids(merged["_n_nodes"], output_size=sum(node_tot))
            if "edge_index" in merged and "_n_edges" in merged:
                merged["_edge_segment"] = segment_ids(merged["_n_edges"], output_size=merged["edge_index"].shape[-1])
        return cls(attrs, **merged)

    # ---- list-like access --------------------------------------------------------
    def _offsets(self):
        """Exclusive prefix sums of the per-graph node / edge counts on the host, cached per counts tensor (``batch[i]``
        in a loop asked for them once per graph)."""
        nn, ne = self.data["_n_nodes"], self.data.get("_n_edges")
        cached = self.__dict__.get("_offsets_cache")
        if cached is not None and cached[0] is nn and cached[1] is ne:     # (the cache keeps both tensors alive)
            return cached[2], cached[3]
        node_off, edge_off = self._offsets_uncached()
        self.__dict__["_offsets_cache"] = (nn, ne, node_off, edge_off)
        return node_off, edge_off

    def _offsets_uncached(self):
        nn = self.data["_n_nodes"].reshape(-1).cpu()
        node_off = torch.zeros(nn.numel() + 1, dtype=torch.long)
        node_off[1:] = torch.cumsum(nn, 0)
        edge_off = None
        if "_n_edges" in self.data:
            ne = self.data["_n_edges"].reshape(-1).cpu()
            edge_off = torch.zeros(ne.numel() + 1, dtype=torch.long)
            edge_off[1:] = torch.cumsum(ne, 0)
        return node_off, edge_off

    def get(self, idx, default=None):
        if isinstance(idx, str):
            return self.data.get(idx, default)
        node_off, edge_off = self._offsets()
        idx = int(idx)
        if idx < 0:
            idx += self.n_graphs
        out = {}
        for key, value in self.data.items():
            if key in ("_node_segment", "_edge_segment") or key.startswith("_e3k_"):
                continue
            if key == "edge_index":
                a, b = int(edge_off[idx]), int(edge_off[idx + 1])
                out[key] = value[:, a:b] - int(node_off[idx])
                continue
            if key not in self.attrs:
                continue
            kind = self.attrs[key][0]
            if kind == "graph":
                a, b = idx, idx + 1
            elif kind == "node":
                a, b = int(node_off[idx]), int(node_off[idx + 1])
            elif kind == "edge":
                a, b = int(edge_off[idx]), int(edge_off[idx + 1])
            else:
                continue
            out[key] = value[a:b]
        return Data(self.attrs, **out)

    def index_select(self, idx):
        if isinstance(idx, slice):
            idx = list(range(self.n_graphs))[idx]
        elif isinstance(idx, torch.Tensor):
            idx = idx.flatten().nonzero().flatten().tolist() if idx.dtype == torch.bool else idx.flatten().tolist()
        elif isinstance(idx, np.ndarray):
            idx = idx.flatten().nonzero()[0].tolist() if idx.dtype == np.bool_ else idx.flatten().tolist()
        elif isinstance(idx, Sequence) and not isinstance(idx, str):
            idx = list(idx)
        else:
            raise IndexError(f"unsupported batch index of type {type(idx).__name__}")
        return self._gather_graphs(idx)

    def _gather_graphs(self, idx):
        """Sub-batch of the graphs ``idx`` (any order, repeats allowed) without a per-graph loop: element ranges of the
        selected graphs are built with segment arithmetic on the batch's own device, every tensor is one gather
        (the reference's ``Batch.index_select``, e3_layers/data/batch.py:133-162, rebuilds Data objects one by one)."""
        n_nodes = self.data["_n_nodes"].reshape(-1)
        dev = n_nodes.device
        if dev.type == "cpu":
            ids_np = np.asarray(idx, dtype=np.int64).reshape(-1)
            if ids_np.size and (int(ids_np.min()) < -self.n_graphs or int(ids_np.max()) >= self.n_graphs):
                raise IndexError("graph index out of range")
            ids = torch.from_numpy(np.where(ids_np < 0, ids_np + self.n_graphs, ids_np))
        else:
            ids = torch.as_tensor(idx, dtype=torch.long, device=dev).reshape(-1)
            if ids.numel() and (int(ids.min()) < -self.n_graphs or int(ids.max()) >= self.n_graphs):
                raise IndexError("graph index out of range")
            ids = torch.where(ids < 0, ids + self.n_graphs, ids)

        host = dev.type == "cpu"     # host batches: numpy index arithmetic (no intra-op thread pool, see segment_ids)

        def ranges(counts_all):
            if host:
                c_all, sel = counts_all.numpy(), ids.numpy()
                off = np.cumsum(c_all) - c_all
                cnt = c_all[sel]
                total = int(cnt.sum())
                new_off = np.cumsum(cnt) - cnt
                seg = np.repeat(np.arange(sel.shape[0], dtype=np.int64), cnt)
                index = off[sel][seg] + (np.arange(total, dtype=np.int64) - new_off[seg])
                return index, seg, off, new_off
            off = torch.cumsum(counts_all, 0) - counts_all
            cnt = counts_all[ids]
            total = int(cnt.sum())
            new_off = torch.cumsum(cnt, 0) - cnt
            seg = segment_ids(cnt, total)
            index = off[ids][seg] + (torch.arange(total, device=dev) - new_off[seg])
            return index, seg, off, new_off

        def take(value, index, dim=0):
            if host and value.dtype != torch.bfloat16:
                return torch.from_numpy(np.take(value.detach().numpy(), index, axis=dim))
            return value.index_select(dim, torch.as_tensor(index, device=value.device))

        node_index, _, node_off, new_node_off = ranges(n_nodes)
        edge_index_sel = edge_seg = None
        if "_n_edges" in self.data:
            edge_index_sel, edge_seg, _, _ = ranges(self.data["_n_edges"].reshape(-1).to(dev))
        sel = ids.numpy() if host else ids
        out = {}
        for key, value in self.data.items():
            if key in ("_node_segment", "_edge_segment") or key.startswith("_e3k_"):
                continue
            if key == "edge_index":
                if edge_index_sel is None:
                    raise KeyError("edge_index without _n_edges")
                shift = (new_node_off - node_off[sel])[edge_seg]
                if host:      # numpy add: a torch CPU op of this size spins up the intra-op thread pool (see segment_ids)
                    out[key] = torch.from_numpy(np.take(value.detach().numpy(), edge_index_sel, axis=1) + shift)
                else:
                    out[key] = take(value, edge_index_sel, 1) + shift
                continue
            if key not in self.attrs:
                continue
            kind = self.attrs[key][0]
            if kind == "graph":
                out[key] = take(value, sel)
            elif kind == "node":
                out[key] = take(value, node_index)
            elif kind == "edge":
                out[key] = take(value, edge_index_sel)
        attrs = {k: v for k, v in self.attrs.items() if k not in ("_node_segment", "_edge_segment")}
        return Batch(attrs, **out)

    def __getitem__(self, idx):
        if isinstance(idx, str):
            return self.data[idx]
        if isinstance(idx, (int, np.integer)):
            return self.get(idx)
        return self.index_select(idx)

    def __setitem__(self, key, value):
        if not isinstance(key, str):
            raise NotImplementedError("assigning a graph by integer index is not supported")
        super().__setitem__(key, value)

    def to(self, device, **kwargs):
        super().to(device, **kwargs)
        return self

    def clone(self):
        out = super().clone()
        return out
