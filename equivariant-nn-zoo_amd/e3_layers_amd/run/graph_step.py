"""Whole-step HIP-graph replay (the latency-bound callers of the hot path).

Small batches are host-bound when launched eagerly: a config_energy_force step on 64 molecules issues ≈ 700
kernels of 5–100 µs from Python autograd functions.  Every op of the path is capture-safe — no host
synchronisation, step-dependent optimizer scalars live on the device (``run/optim.py``), device RNG is
graph-aware — so a training step (forward, loss, backward or double backward, clip + Adam + EMA, the flat
all-reduce) or a sampler step can be recorded once and replayed as ONE graph launch.

    step = CapturedStep(lambda: train_one())      # train_one reads its batch from fixed device tensors
    for _ in range(n):
        loss = step()                             # graph replay; `loss` is the captured (static) output tensor

Inputs must live at fixed addresses (copy the next batch *into* the captured tensors); shapes are frozen, so a
batch with a different number of nodes or edges needs its own capture (``e3_layers/run/sde_sampling.py``'s
static-edge mode does exactly that for the reverse-diffusion loop).
"""
from __future__ import annotations

import os
from typing import Any, Callable

import torch


class CapturedStep:
    def __init__(self, fn: Callable[[], Any], warmup: int = 3, device=None, generators=()):
        """``generators``: device ``torch.Generator`` objects ``fn`` draws from (other than the default one): they are registered
        with the graph, so that every replay advances them."""
        if not torch.cuda.is_available():
            raise RuntimeError("CapturedStep needs a HIP device: there is no CPU fallback for graph replay")
        self.fn = fn
        self.generators = tuple(generators)
        self.dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.recaptures = 0      # how often a knot-table veto made this step record itself again (tests, logs)
        self._capture(max(1, warmup))

    def _eager(self, n: int):
        """``n`` eager runs of ``fn`` on a side stream: lazy initialisation (plans, side streams, allocator pools, the guards' device
        state) must not be captured."""
        dev = self.dev
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        out = None
        with torch.cuda.stream(side):
            for _ in range(n):
                out = self.fn()
        torch.cuda.current_stream(dev).wait_stream(side)
        return out

    def _capture(self, warmup: int, keep_last: bool = False):
        dev = self.dev
        from ..backend.graph import capture_flag, check_indices
        from ..backend import radial_table

        for _ in range(4):
            level = radial_table.REFINEMENTS
            out = self._eager(warmup)
            torch.cuda.synchronize(dev)
            check_indices()      # the warm-up batches' deferred index checks (none are recorded while capturing)
            radial_table.drain_guards()      # ... and the knot-table guards' pending read-backs (no Event.query() inside a capture)
            if radial_table.REFINEMENTS == level:
                break            # (a bound read just now doubled the knot counts: warm up again, on the tables the graph will hold)
            warmup = 1
        if not keep_last:
            out = None
        capture_flag(dev)                # the persistent flag the captured index checks fold into (a bad batch fed
                                         # through a replay must raise, as it does on the eager path)
        self.graph = torch.cuda.CUDAGraph()
        for g in self.generators:
            self.graph.register_generator_state(g)
        radial_table.CAPTURE_LOG = []
        try:
            with torch.cuda.graph(self.graph):
                self.out = self.fn()
            self._guards = tuple(radial_table.CAPTURE_LOG)      # the knot-table guards this graph evaluates with every replay
        finally:
            radial_table.CAPTURE_LOG = None
        self._replays = 0
        self._vetoed = False
        self._level = radial_table.REFINEMENTS
        return out

    @property
    def stale(self) -> bool:
        """The next call records the step again instead of replaying it."""
        from ..backend import radial_table

        return bool(self._vetoed or (self._guards and self._level != radial_table.REFINEMENTS))

    def __call__(self):
        from ..backend import radial_table
        from ..backend.graph import poll_capture_flags

        if self.stale:
            # A knot table this graph interpolates from has been refined or switched off (its error bound, which the captured step
            # evaluates on the device with every replay, passed the tolerance: the weights moved under the optimizer).  The graph still
            # holds the coarse table's kernels: this call runs the step EAGERLY once (on the finer tables, or with the vetoed layer on
            # its per-edge path -- also the warm-up of those kernels) and records the step again; later calls replay the new graph.
            self.recaptures += 1
            out = self._capture(1, keep_last=True)
            return out
        self.graph.replay()
        self._replays += 1
        poll_capture_flags(self.dev)     # (one 4-byte async copy: read by the next build / optimizer step / check_indices())
        if self._guards:                 # every N-th replay: the guards' running maxima travel home (no sync)
            self._vetoed = radial_table.poll_replay(self._guards, self._replays) or self._level != radial_table.REFINEMENTS
        return self.out


# ---------------------------------------------------------------------------------------------------------------------
# Graph replay with a NEW batch every step: pad every batch to one size bucket with a ghost graph, copy it into the captured
# tensors, replay.  
#
# A HIP graph freezes shapes and addresses.  The batches of a training run differ in N and E, so each is padded to the
# bucket's (n_cap, e_cap) with ONE extra graph that holds the surplus nodes and edges: ghost nodes on a line 1.0-3.5 A apart
# (valid geometry: finite spherical harmonics and radial basis), ghost edges between neighbouring ghost nodes (repeated as
# often as needed: edges are independent), a valid species.  The ghost graph's energy is excluded from the loss (its weight in
# the masked mean is 0), so the gradient that flows into the ghost's nodes and edges is exactly zero and every weight gradient
# equals the un-padded batch's up to the order of the sums.  All host-side decisions of the model (knot table, stacks,
# streams, keyed self-connection) depend on sizes only, which the bucket fixes.
# ---------------------------------------------------------------------------------------------------------------------
def ghost_sample(like, n_nodes: int, n_edges: int):
    """A sample with ``n_nodes`` (>= 2 when it has edges) collinear nodes and ``n_edges`` nearest-neighbour edges, with the
    keys of ``like`` (a ``Data`` sample on the host: ``pos``, ``species``, ``edge_index``, per-graph targets)."""
    from ..data.data import Data

    if n_nodes < (2 if n_edges else 1):
        raise ValueError(f"a ghost graph with {n_edges} edges needs at least two nodes (got {n_nodes}): raise the node capacity")
    # neighbour distances spread over [1.0, 3.5) A (golden-ratio sequence): ghost edges must not pile up in ONE knot bin of
    # the radial table (a bin's edges are walked by one wave, its CSR row is ranked in O(len^2 / 64))
    gaps = 1.0 + 2.5 * torch.frac(0.6180339887 * torch.arange(n_nodes, dtype=torch.float64))
    pos = torch.zeros(n_nodes, 3, dtype=like["pos"].dtype)
    pos[:, 0] = (torch.cumsum(gaps, 0) - gaps[0]).to(like["pos"].dtype)
    k = torch.arange(n_edges, dtype=torch.int64)
    a = k % max(n_nodes - 1, 1)
    flip = (k // max(n_nodes - 1, 1)) % 2 == 1
    src, dst = torch.where(flip, a + 1, a), torch.where(flip, a, a + 1)
    tensors = {}
    for key in like.keys():
        v = like[key]
        if key == "pos":
            tensors[key] = pos
        elif key == "edge_index":
            tensors[key] = torch.stack([src, dst]).to(v.dtype)
        elif key == "_n_nodes":
            tensors[key] = torch.full_like(v, n_nodes)
        elif key == "_n_edges":
            tensors[key] = torch.full_like(v, n_edges)
        elif torch.is_tensor(v) and (like.attrs.get(key, ("",))[0] == "node"
                                     or (key not in like.attrs and v.dim() >= 1 and v.shape[0] == like["pos"].shape[0])):
            tensors[key] = v[:1].expand(n_nodes, *v.shape[1:]).clone()      # node-wise (species): the first node's value
        elif torch.is_tensor(v) and like.attrs.get(key, ("",))[0] == "edge":
            tensors[key] = (v[:1].expand(n_edges, *v.shape[1:]).clone() if v.shape[0] else v.new_zeros((n_edges,) + tuple(v.shape[1:])))
        elif torch.is_tensor(v):
            tensors[key] = torch.zeros_like(v)                               # graph-wise targets
        else:
            tensors[key] = v
    return Data(attrs=dict(like.attrs), **tensors)


def pad_batch(batch, n_cap: int, e_cap: int):
    """``batch`` (host or device) + one ghost graph so that it has exactly ``n_cap`` nodes and ``e_cap`` edges; returns the
    padded Batch (on the batch's device) with ``_graph_weight`` [G + 1, 1] = 1 / G for the real graphs, 0 for the ghost, and
    ``_node_weight`` [n_cap, 1] = 1 / N for the real nodes, 0 for the ghost's."""
    from ..data.data import Batch
    from ..data.loader import samples_of

    dev = batch["pos"].device
    samples = samples_of(batch)
    n, e = int(batch["pos"].shape[0]), int(batch["edge_index"].shape[1])
    if n_cap < n + 2 or e_cap < e:
        raise ValueError(f"batch with {n} nodes / {e} edges does not fit the bucket ({n_cap}, {e_cap}; two ghost nodes are the minimum)")
    samples.append(ghost_sample(samples[0], n_cap - n, e_cap - e))
    out = Batch.from_data_list(samples, attrs=dict(samples[0].attrs))
    w = torch.full((len(samples), 1), 1.0 / (len(samples) - 1), dtype=torch.float32)
    w[-1] = 0.0
    out["_graph_weight"] = w
    out.attrs["_graph_weight"] = ("graph", "1x0e")
    wn = torch.zeros(n_cap, 1, dtype=torch.float32)      # node-wise losses (forces): 1 / N_real on the real nodes
    wn[:n] = 1.0 / n
    out["_node_weight"] = wn
    out.attrs["_node_weight"] = ("node", "1x0e")
    return out.to(dev)


GHOST_DEGREE = 16      # ghost edges per ghost node the capacities below aim for (the edge kernels walk a node's edges in one wave)


def bucket_capacity(sizes, node_multiple: int = 32, edge_multiple: int = 1024):
    """(n_cap, e_cap) for batches of ``sizes`` = [(n_nodes, n_edges), ...]: every batch fits with at least two ghost nodes, and
    the ghost graph that absorbs a batch's missing edges has about ``GHOST_DEGREE`` edges per node -- a ghost node with
    hundreds of edges is a wave that walks hundreds of edges on its own while the chip waits (measured: 40 ms instead of
    3.5 ms per step at 128 molecules with 21 ghost nodes for 3 300 ghost edges)."""
    e_cap = -(-max(e for _, e in sizes) // edge_multiple) * edge_multiple
    n_cap = max(n + max(2, -(-(e_cap - e) // GHOST_DEGREE)) for n, e in sizes)
    return -(-n_cap // node_multiple) * node_multiple, e_cap


def forget_batch_memos(batch=None) -> None:
    """Drops what the framework remembers about a batch by the identity of its tensors: CSR views (``backend/graph.py``),
    species groups (``nn/core.py``) and the memos attached to the tensors themselves (``_e3k_*`` attributes: the flat species
    index of ``OneHotEncoding``, stream aliases)."""
    from ..backend import graph as _graph
    from ..nn import core as _core

    _graph._cache.clear()
    _core._groups_cache.clear()
    if batch is not None:
        for k in batch.keys():
            v = batch[k]
            if torch.is_tensor(v):
                for name in [a for a in vars(v) if a.startswith("_e3k_")] if hasattr(v, "__dict__") else []:
                    delattr(v, name)


class BucketedStep:
    """``step = BucketedStep(train_on, example)``: ``train_on(batch)`` is captured once on a copy of the padded batch
    ``example``; ``step(padded)`` copies the next padded batch of the same bucket into the captured tensors and replays.

    ``tail``: what follows every replay eagerly, on the current stream.  Several ranks: ``fn`` = forward + loss + backward into
    the flat gradient buffer (it must end with ``ops.join_side_streams()``), ``tail`` = the flat all-reduce and the optimizer
    step -- three launches and the collectives behind one graph launch, so the step needs no host time to speak of and the
    RCCL calls stay outside the capture.  The warm-up and the capture run ``fn`` alone (no collective is issued while a rank
    may still fail to capture; the caller agrees on the outcome over the ranks before the first ``step()``)."""

    def __init__(self, fn: Callable[[Any], Any], example, warmup: int = 3, generators=(), tail: Callable[[], Any] = None):
        self.tail = tail
        self.static = example.clone()
        self.keys = [k for k in self.static.keys() if torch.is_tensor(self.static[k])]

        def captured_fn():
            # The per-batch memos (CSR views keyed on the identity of ``edge_index``, species groups keyed on the key tensor, the
            # flat species index OneHotEncoding keeps on its input tensor)
            # would hit on the static tensors and leave the CSR build / grouping kernels OUT of the graph: the replay would then
            # walk the warm-up batch's topology.  Forget them, so that these kernels are part of what is captured.
            forget_batch_memos(self.static)
            return fn(self.static.view())

        self.captured = CapturedStep(captured_fn, warmup=warmup, generators=generators)

    @property
    def recaptures(self) -> int:
        return self.captured.recaptures

    def __call__(self, padded):
        by_dtype = {}
        for k in self.keys:
            dst, src = self.static[k], padded[k]
            if dst.shape != src.shape:
                raise ValueError(f"{k}: {tuple(src.shape)} does not fit the captured {tuple(dst.shape)} (another bucket?)")
            if src.device == dst.device and src.dtype == dst.dtype and src.is_contiguous() and dst.is_contiguous():
                pair = by_dtype.setdefault(dst.dtype, ([], []))
                pair[0].append(dst)
                pair[1].append(src)
            else:
                dst.copy_(src, non_blocking=True)
        for dsts, srcs in by_dtype.values():      # one multi-tensor launch per dtype instead of one copy per field (eleven fields:
            torch._foreach_copy_(dsts, srcs)      # 53 us in front of every replay)
        out = self.captured()
        if self.tail is not None:
            self.tail()
        return out


def stream_beside(main, candidates: int = 6):
    """A stream whose work demonstrably runs BESIDE ``main``'s.  HIP multiplexes the streams of a process over a few hardware queues
    (four by default) in creation order, and two streams that share a queue execute in submission order: a preparation stream that
    lands on the main stream's queue runs in front of the step instead of beside it (seen in the first trace of this class: every
    kernel of both graphs on one queue, no gain).  A stream of another PRIORITY has a queue of its own, but mixing priority levels
    makes the queues starve each other on this part (8.4 instead of 4.2 ms per step; DESIGN.md section 5, round 5 saw the same with a
    high-priority main stream).  So: a few fresh streams are tried in turn -- a millisecond of kernels on ``main``, one tiny kernel
    on the candidate -- and the first whose kernel finishes while ``main`` is still busy is kept."""
    dev = main.device
    busy = torch.empty(16 << 20, device=dev)
    tiny = torch.zeros(64, device=dev)
    pool = [torch.cuda.Stream(dev) for _ in range(candidates)]
    for st in pool:
        torch.cuda.synchronize(dev)
        with torch.cuda.stream(main):
            for _ in range(40):
                busy.mul_(1.0)
            main_done = torch.cuda.Event()
            main_done.record(main)
        with torch.cuda.stream(st):
            tiny.add_(1.0)
            mine = torch.cuda.Event()
            mine.record(st)
        mine.synchronize()
        beside = not main_done.query()
        torch.cuda.synchronize(dev)
        if beside:
            return st
    return pool[0]


class PipelinedBucketedStep:
    """``BucketedStep`` with the batch-only part of the step taken OFF the step's critical path (round 6).

    A replayed step is one in-order queue: whatever the forward derives from the batch alone -- the device copy of the batch, both
    CSR views, edge vectors, spherical harmonics, one-hot species, species groups, knot bins, edge records: about 25 small dependent
    launches, 0.2 ms of 4.2 at 256 molecules -- runs in front of the first convolution with the chip idle.  In the reference that
    work belongs to the data loader's worker processes (``e3_layers/data/dataloader.py:30-118``: collate; ``computeEdgeIndex`` as a
    ``preprocess`` step) and overlaps the previous step.  Here it is a second HIP graph per buffer:

        P[b]  = copy of the padded batch into the static tensors of buffer b + ``model.prepare_data(static[b])``   (prep stream)
        M[b]  = ``fn(static[b].view())``: the step itself, captured with P[b]'s results in place                    (main stream)

    and two buffers, so that P of batch t + 1 runs BESIDE M of batch t:

        step(batch_t, nxt=batch_t+1):   main: wait P(batch_t) -> M[b];   prep: wait M[1 - b] of step t - 1 -> P[1 - b](batch_t+1)

    Both graphs are single-stream chains (the HIP graph executor launches those from pre-built packets: no host work to speak of);
    the only cross-stream edges are two events per step.  A batch that was not announced with ``nxt`` is prepared on the main
    stream in front of its step (the first call, or a caller that does not look ahead): same result, no overlap.  Results are
    bit-identical to ``BucketedStep`` (same kernels on the same inputs; ``tests/test_gpu_model.py``)."""

    def __init__(self, prepare: Callable[[Any], Any], fn: Callable[[Any], Any], example, warmup: int = 3, generators=(),
                 tail: Callable[[], Any] = None):
        from ..backend.graph import capture_flag

        self.tail = tail
        self.prepare = prepare
        self.dev = example["pos"].device
        self.prep_stream = stream_beside(torch.cuda.current_stream(self.dev))
        self.static, self.prep_graphs, self.steps = [], [], []
        self.ev_prep = [torch.cuda.Event() for _ in range(2)]
        self.ev_step = [torch.cuda.Event() for _ in range(2)]
        self._holds = [None, None]          # which padded batch (by identity) buffer b has been prepared for
        self._step_ran = [False, False]
        self.keys = None
        capture_flag(self.dev)              # (the persistent index-check flag captured builds fold into: before any recording)
        warm = example.clone()              # one eager preparation first: nothing a capture records may be a kernel's first launch
        prepare(warm)                       # in the process (lazy initialisation inside a capture is not something to rely on)
        torch.cuda.synchronize(self.dev)
        del warm
        for b in range(2):
            static = example.clone()        # (fresh tensors: no per-batch memo of the framework knows them yet)
            if self.keys is None:
                self.keys = [k for k in static.keys() if torch.is_tensor(static[k])]
                self._given = list(static.keys())      # (what the caller's batches carry: everything else is the preparation's)
            torch.cuda.synchronize(self.dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                prepare(static)
            g.replay()                      # (a capture records, it does not run: the step's warm-up below reads these results)
            torch.cuda.synchronize(self.dev)
            self.static.append(static)
            self.prep_graphs.append(g)
            # warm-up steps only for the first buffer's capture (lazy initialisation is process-wide); they take optimizer steps
            # like BucketedStep's do
            self.steps.append(CapturedStep((lambda s=static: fn(s.view())), warmup=warmup if b == 0 else 1, generators=generators))
        self.captured = self.steps[0]       # (bench.py / tests: .graph, .recaptures of the first buffer's step)

    @property
    def recaptures(self) -> int:
        return sum(s.recaptures for s in self.steps)

    def _record_again(self, b: int) -> None:
        """Buffer b's step has to record itself again (a knot table was refined or switched off: ``CapturedStep.stale``).  Its
        preparation graph is recorded again FIRST, on new tensors holding the buffer's present contents: the step's recording must
        find the batch as the first one did -- prepared by a graph, inputs untouched since (the copies of ``_enqueue_prepare`` bump
        the inputs' version counters: the record of what was prepared from what, and every per-tensor memo, would count as stale,
        the eager warm-up would rebuild them outside any graph and the new recording would replay the batch it was recorded on) --
        and after a refinement the preparation builds the finer resolution's bins and edge records."""
        torch.cuda.synchronize(self.dev)
        static = self.static[b]
        fresh = {k: (static.data[k].clone() if torch.is_tensor(static.data[k]) else static.data[k]) for k in self._given}
        static.data.clear()                 # (same container: the step's closure reads it when it records; the preparation's own keys are
        static.data.update(fresh)           # gone -- a layer that finds its output present keeps it)
        static._e3k_done = None
        forget_batch_memos()
        self.prep_graphs[b] = None          # (its pool holds the old bins)
        warm = static.clone()
        self.prepare(warm)                  # (eagerly first: see __init__)
        torch.cuda.synchronize(self.dev)
        del warm
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.prepare(static)
        self.prep_graphs[b] = g
        g.replay()
        torch.cuda.synchronize(self.dev)

    def _enqueue_prepare(self, b: int, padded) -> None:
        """On the CURRENT stream: the padded batch into buffer b's static tensors, then its preparation graph."""
        static = self.static[b]
        by_dtype = {}
        for k in self.keys:
            dst, src = static[k], padded[k]
            if dst.shape != src.shape:
                raise ValueError(f"{k}: {tuple(src.shape)} does not fit the captured {tuple(dst.shape)} (another bucket?)")
            if src.device == dst.device and src.dtype == dst.dtype and src.is_contiguous() and dst.is_contiguous():
                pair = by_dtype.setdefault(dst.dtype, ([], []))
                pair[0].append(dst)
                pair[1].append(src)
            else:
                dst.copy_(src, non_blocking=True)
        for dsts, srcs in by_dtype.values():
            torch._foreach_copy_(dsts, srcs)
        self.prep_graphs[b].replay()
        self._holds[b] = padded

    def __call__(self, padded, nxt=None):
        main = torch.cuda.current_stream(self.dev)
        b = next((i for i in range(2) if self._holds[i] is padded), None)
        if b is None:                       # not announced: prepared here, in front of its step
            b = 0 if self._holds[0] is None else (1 if self._holds[1] is None else 0)
            if self._holds[b] is not None:  # (an announced batch that is not coming after all: its preparation may still be running)
                main.wait_event(self.ev_prep[b])
            self._enqueue_prepare(b, padded)
        else:
            main.wait_event(self.ev_prep[b])
        if self.steps[b].stale:
            self._record_again(b)
        out = self.steps[b]()
        self.ev_step[b].record(main)
        self._step_ran[b] = True
        self._holds[b] = None
        if nxt is not None:
            o = 1 - b
            if self._step_ran[o]:
                self.prep_stream.wait_event(self.ev_step[o])      # the last step that read buffer o
            else:
                self.prep_stream.wait_stream(main)
            with torch.cuda.stream(self.prep_stream):
                self._enqueue_prepare(o, nxt)
                self.ev_prep[o].record(self.prep_stream)
        if self.tail is not None:
            self.tail()
        return out
