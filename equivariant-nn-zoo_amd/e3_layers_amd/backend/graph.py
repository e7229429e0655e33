"""Per-batch graph topology for the HIP kernels: int32 endpoints plus CSR by destination and
by source.

The reference scatters messages with ``scatter(edge_features, edge_dst, dim=0, dim_size=N)``
(``e3_layers/nn/message_passing.py:109``), i.e. an ``index_add_`` whose CPU summation order is
ascending edge id.  The fused kernel instead walks, for every destination node, the list of its
in-edges; building that list with a *stable* sort keeps the ascending-edge-id order inside every
segment, so the sum order (and hence fp32 rounding) follows the reference's CPU path, with no
atomics.  The by-source CSR serves the backward pass (grad wrt node features, grad wrt pos).

On the device the lists are built by ``csrc/e3k_graph.hip`` (count -> scan -> fill -> per-row rank sort: four small
launches and one fill, no host sync); the torch construction below (stable argsort / bincount / cumsum) serves CPU tensors (host
logic tests, gloo runs) and is the reference the device build is tested bit-exact against.
"""
from __future__ import annotations

import weakref
from dataclasses import dataclass
from typing import Dict, Optional

import torch

TOPO_KEYS = ("_e3k_src", "_e3k_dst", "_e3k_dst_ptr", "_e3k_dst_perm", "_e3k_src_ptr", "_e3k_src_perm")


@dataclass
class GraphTopo:
    src: torch.Tensor       # int32 [E]
    dst: torch.Tensor       # int32 [E]
    dst_ptr: torch.Tensor   # int32 [N+1]
    dst_perm: torch.Tensor  # int32 [E]
    src_ptr: torch.Tensor   # int32 [N+1]
    src_perm: torch.Tensor  # int32 [E]

    @property
    def num_nodes(self) -> int:
        return self.dst_ptr.numel() - 1

    @property
    def num_edges(self) -> int:
        return self.src.numel()

    def as_dict(self) -> Dict[str, torch.Tensor]:
        return {k: getattr(self, k[len("_e3k_"):]) for k in TOPO_KEYS}

    @staticmethod
    def from_dict(data) -> Optional["GraphTopo"]:
        if all(k in data for k in TOPO_KEYS):
            return GraphTopo(*(data[k] for k in TOPO_KEYS))
        return None


def _csr(index: torch.Tensor, num_nodes: int):
    perm = torch.argsort(index, stable=True)
    counts = torch.bincount(index, minlength=num_nodes)
    ptr = torch.zeros(num_nodes + 1, dtype=torch.int32, device=index.device)
    ptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
    return ptr, perm.to(torch.int32)


_pending_flags: list = []      # (event, pinned host flag) of device builds whose index check has not been read yet
_FLAG_RING = 256               # slots of ONE pinned buffer, allocated once: a pinned allocation per build stalled the host
_flag_ring = None              # until the device had drained (hipHostMalloc synchronises) -- 3 ms per forward when it
_flag_next = 0                 # happened mid-step


_EDGE_MSG = ("an earlier batch's edge_index held node ids outside [0, num_nodes): its topology is invalid (such edges were "
             "attached to node 0)")


def _check_pending_flags() -> None:
    """Out-of-range indices are detected on the device without a host sync: the flag travels to pinned memory
    asynchronously and is looked at once its copy has landed -- by the next build, by the optimizer step
    (``FusedAdamEMA.step`` polls), or by ``check_indices()`` (blocking: evaluation, before a graph capture, tests)."""
    while _pending_flags and _pending_flags[0][0].query():
        _, host, msg, sticky = _pending_flags.pop(0)
        if int(host[0]) != 0:
            # (a persistent flag -- ``sticky`` -- was cleared on the device by the same launch that sent this value home
            #  (``e3k_flag_fetch_clear``, in stream order): every flagged batch is reported exactly once, nothing that was flagged
            #  between this copy and a later host-side clear can be lost, and a caller that catches the error and skips
            #  the batch does not see it again on later, valid batches)
            raise ValueError(msg)


def poll_indices() -> None:
    """Non-blocking look at the flags that have arrived (called once per optimizer step)."""
    if _pending_flags:
        _check_pending_flags()


def check_indices() -> None:
    """Block until every device-side index check so far (edge endpoints, row keys) has been read."""
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    _check_pending_flags()


_capture_flags: dict = {}      # device index -> persistent int32 [1] flag that captured builds OR their own flags into


def capture_flag(device) -> torch.Tensor:
    """The persistent error flag of ``device`` for builds recorded into a HIP graph (allocated OUTSIDE any capture: call this
    before recording starts -- ``run/graph_step.CapturedStep`` does).  A flag made inside the graph lives in the graph's private
    pool and is overwritten by every replay; this one survives, ``poll_capture_flags()`` reads it back after a replay."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    f = _capture_flags.get(idx)
    if f is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("capture_flag() must be called before the capture starts (CapturedStep does)")
        f = _capture_flags[idx] = torch.zeros(1, dtype=torch.int32, device=torch.device("cuda", idx))
    return f


def poll_capture_flags(device=None) -> None:
    """After a graph replay on ``device`` (default: the current one): send that device's persistent flag to the host (no sync; it
    is looked at with the other deferred flags, and cleared once it has been reported)."""
    idx = torch.cuda.current_device() if device is None else (device.index if device.index is not None else torch.cuda.current_device())
    f = _capture_flags.get(idx)
    if f is not None:
        with torch.cuda.device(idx):
            defer_flag(f, _PERSISTENT_MSG, sticky=f)


_PERSISTENT_MSG = ("a batch held node ids outside [0, num_nodes) in its edge_index (such edges were attached to node 0), row keys "
                   "outside [0, n_keys) (such rows were dropped from the keyed self-connection), a type index outside [0, num_types) "
                   "(OneHotEncoding: an all-zero row where the reference's one_hot raises) or an edge key outside the keyed radial tables")


def persistent_flag(device) -> torch.Tensor:
    """The device's persistent error flag for kernels that only ever OR bits into it (``e3k_onehot``, ``e3k_rtable_bins_keyed``,
    captured index checks); allocated on first use -- OUTSIDE a capture (``CapturedStep`` does that before it records)."""
    return capture_flag(device)


def report_persistent(device) -> None:
    """Eager callers of the OR-only kernels: send the device's persistent flag home behind what was just enqueued (one launch, no
    sync; read by the next build / optimizer step / ``check_indices()``).  While a stream captures nothing is recorded here:
    ``CapturedStep`` fetches the flag after every replay."""
    if torch.cuda.is_current_stream_capturing():
        return
    poll_capture_flags(device)


def defer_flag(flag: torch.Tensor, message: str, sticky: Optional[torch.Tensor] = None) -> None:
    """Registers a device int32 [1] error flag (non-zero = bad input) to be read back without a sync.  ``sticky``: the persistent
    flag this copy was taken from (zeroed when a non-zero copy is reported)."""
    global _flag_ring, _flag_next
    if torch.cuda.is_current_stream_capturing():
        # recorded into the graph: every replay folds this build's flag into the device's persistent one (sticky: a bad batch
        # stays flagged until somebody reads it -- CapturedStep polls after each replay)
        idx = flag.device.index if flag.device.index is not None else torch.cuda.current_device()
        keep = _capture_flags.get(idx)
        if keep is not None:
            keep.bitwise_or_(flag.reshape(1).to(torch.int32))
        return
    _check_pending_flags()
    if len(_pending_flags) >= _FLAG_RING - 1:
        check_indices()
    if _flag_ring is None:
        _flag_ring = torch.zeros(_FLAG_RING, dtype=torch.int32).pin_memory()
    host = _flag_ring[_flag_next:_flag_next + 1]
    _flag_next = (_flag_next + 1) % _FLAG_RING
    if sticky is not None and sticky is flag:
        from . import lib as L

        L.check(L.load().e3k_flag_fetch_clear(L.ptr(flag), host.data_ptr(), L.stream_ptr()), "e3k_flag_fetch_clear")
    else:
        host.copy_(flag, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    _pending_flags.append((ev, host, message, sticky))


def _build_topology_device(edge_index: torch.Tensor, num_nodes: int) -> GraphTopo:
    """csrc/e3k_graph.hip: four small launches + one fill, no host sync, bit-identical to the stable-sort construction."""
    from . import lib as L

    lib = L.load()
    dev = edge_index.device
    e = edge_index.shape[1]
    ei = edge_index if (edge_index.dtype == torch.int64 and edge_index.is_contiguous()) else edge_index.long().contiguous()
    i32 = dict(dtype=torch.int32, device=dev)
    # one allocation for every output and the workspace; [flag | dst_ptr | src_ptr] first and adjacent: the library
    # zero-fills them with one memset
    sizes = [1, num_nodes + 1, num_nodes + 1, e, e, e, e, int(lib.e3k_csr_workspace_ints(num_nodes, e))]
    buf = torch.empty(sum(sizes), **i32)
    flag, dst_ptr, src_ptr, src, dst, dst_perm, src_perm, work = torch.split(buf, sizes)
    with torch.cuda.device(dev):
        L.check(lib.e3k_csr_build(L.ptr(ei), num_nodes, e, L.ptr(src), L.ptr(dst), L.ptr(dst_ptr), L.ptr(dst_perm),
                                  L.ptr(src_ptr), L.ptr(src_perm), L.ptr(work), L.ptr(flag), L.stream_ptr()), "e3k_csr_build")
        defer_flag(flag, _EDGE_MSG)
    return GraphTopo(src, dst, dst_ptr, dst_perm, src_ptr, src_perm)


def build_topology(edge_index: torch.Tensor, num_nodes: int) -> GraphTopo:
    """edge_index: int64 [2, E] (row 0 = source, row 1 = destination)."""
    if edge_index.dim() != 2 or edge_index.shape[0] != 2:
        raise ValueError(f"edge_index must be [2, E], got {tuple(edge_index.shape)}")
    if int(num_nodes) >= 2 ** 31 - 1 or edge_index.shape[1] >= 2 ** 31 - 1:
        raise ValueError("node and edge ids must fit in int32")
    if edge_index.is_cuda:
        return _build_topology_device(edge_index, int(num_nodes))
    src64, dst64 = edge_index[0].contiguous(), edge_index[1].contiguous()
    dst_ptr, dst_perm = _csr(dst64, num_nodes)
    src_ptr, src_perm = _csr(src64, num_nodes)
    return GraphTopo(src64.to(torch.int32), dst64.to(torch.int32), dst_ptr, dst_perm, src_ptr, src_perm)


_cache: "Dict[int, tuple]" = {}


def get_topology(data, num_nodes: int) -> GraphTopo:
    """Topology carried by the batch if present, else built from ``data['edge_index']`` and
    memoised on the identity of that tensor."""
    topo = GraphTopo.from_dict(data)
    if topo is not None and topo.num_nodes == num_nodes and topo.num_edges == data["edge_index"].shape[1]:
        return topo
    ei = data["edge_index"]
    key = id(ei)
    hit = _cache.get(key)
    if hit is not None:
        ref, version, n, topo = hit
        if ref() is ei and version == ei._version and n == num_nodes:
            return topo
    topo = build_topology(ei, num_nodes)
    if len(_cache) > 64:
        _cache.clear()
    _cache[key] = (weakref.ref(ei), ei._version, num_nodes, topo)
    return topo
