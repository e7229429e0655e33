"""Graph-parallel data parallelism: molecules of a Batch shard across ranks, one process per
GPU, gradients all-reduced over RCCL/xGMI (``torch.distributed`` backend "nccl" on ROCm).

What it stands in for: the reference's only multi-GPU strategy, DDP over per-rank batches
(``train.py:99,272``, ``e3_layers/run/trainer.py:138-139``).  A Batch is a
disjoint union of graphs, every op on the path is per node / edge / graph, so the data path
needs no collective: only the flat gradient is summed (and divided by the world size, DDP's
averaging) once per step.  Gradients live in ONE flat fp32 buffer whose slices are the
``.grad`` of each parameter, so the all-reduce is a single call on ~4-25 MB with no
flatten/unflatten copies.
"""
from __future__ import annotations

import weakref
from typing import Iterable, List, Sequence

import torch
import torch.distributed as dist


def partition_by_edges(n_edges: Sequence[int], world_size: int) -> List[List[int]]:
    """Contiguous split of graph ids into ``world_size`` groups with balanced edge counts
    (greedy prefix cut at multiples of total/world_size).  Every rank gets >= 1 graph when
    there are enough graphs."""
    n = len(n_edges)
    if world_size <= 1:
        return [list(range(n))]
    total = float(sum(n_edges)) or 1.0
    bounds, acc, nxt = [0], 0.0, 1
    for i, e in enumerate(n_edges):
        acc += e
        remaining_graphs = n - (i + 1)
        remaining_cuts = world_size - nxt
        if nxt < world_size and (acc >= total * nxt / world_size or remaining_graphs <= remaining_cuts) and remaining_graphs >= remaining_cuts:
            bounds.append(i + 1)
            nxt += 1
    while len(bounds) < world_size:
        bounds.append(n)
    bounds.append(n)
    return [list(range(bounds[r], bounds[r + 1])) for r in range(world_size)]


def shard_batch(batch, rank: int, world_size: int):
    """The sub-Batch of this rank (``Batch.index_select`` semantics, e3_layers/data/batch.py:133-162)."""
    if world_size <= 1:
        return batch
    counts = batch["_n_edges"].view(-1).tolist() if "_n_edges" in batch else batch["_n_nodes"].view(-1).tolist()
    mine = partition_by_edges(counts, world_size)[rank]
    if not mine:
        raise ValueError(f"rank {rank} received no graphs ({len(counts)} graphs over {world_size} ranks)")
    return batch[mine]


FLAT_ALIGN = 64   # floats: every parameter's slice starts on a 256-byte boundary (16-byte vector loads stay legal)


def flat_layout(params: Sequence[torch.Tensor], align: int = FLAT_ALIGN):
    """Offsets of each tensor in a flat buffer and the buffer length (padding slots stay zero)."""
    offs, off = [], 0
    for p in params:
        offs.append(off)
        off += -(-p.numel() // align) * align
    return offs, off


def backward_parameters(loss: torch.Tensor, params: Iterable[torch.nn.Parameter]) -> None:
    """``loss.backward()`` of a training step: gradients of the Parameters only.  In a force-training step ``pos``
    still requires grad when the loss is differentiated; plain ``loss.backward()`` would also compute d loss / d pos
    (three more grad_sh edge passes per convolution that nothing reads).  ``inputs=`` lets the engine drop the nodes
    that lead to ``pos`` alone, ``ops.params_only_backward`` tells the e3k backward functions the same."""
    from ..backend import ops

    with ops.params_only_backward():
        loss.backward(gradient=ops.unit_gradient(loss) if loss.is_cuda else None, inputs=[p for p in params if p.requires_grad])


def _radial_mlp_params(mp) -> list:
    conv = getattr(mp, "conv", None)
    fc = getattr(conv, "fc", None)
    return list(fc.parameters()) if fc is not None else []


def flat_param_order(model: torch.nn.Module) -> list:
    """``model.parameters()`` with the radial MLPs of the message-passing layers moved to the tail.  With this order
    in ``FlatGradients`` / ``FusedAdamEMA`` every layer's node-side weights (linear_1, tp.linear, sc) are one contiguous
    slice whose all-reduce starts behind that layer's backward, and the radial MLPs -- whose gradients are written by
    one batched backward at the end (``backend/conv_native.RadialStackFn``) -- are one slice of the remainder."""
    from ..nn.message_passing import MessagePassing

    tail, tail_ids = [], set()
    for mod in model.modules():
        if isinstance(mod, MessagePassing):
            for p in _radial_mlp_params(mod):
                if id(p) not in tail_ids:
                    tail_ids.add(id(p))
                    tail.append(p)
    return [p for p in model.parameters() if id(p) not in tail_ids] + tail


def param_names(model: torch.nn.Module, params: Sequence[torch.nn.Parameter]) -> List[str]:
    """``model.named_parameters()``'s name of every entry of ``params`` (``FusedAdamEMA(params, names=...)``: a checkpoint then
    survives a change of the flat parameter order)."""
    by_id = {id(p): n for n, p in model.named_parameters()}
    return [by_id[id(p)] for p in params]


class FlatGradients:
    """Points every ``p.grad`` at a slice of one contiguous buffer."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params = [p for p in params if p.requires_grad]
        self.offsets, total = flat_layout(self.params)
        ref = self.params[0]
        self.buffer = torch.zeros(total, dtype=ref.dtype, device=ref.device)
        self._sink = {}
        for p, off in zip(self.params, self.offsets):
            p.grad = self.buffer[off:off + p.numel()].view_as(p)
            self._sink[(p.data_ptr(), p.numel())] = (self.buffer[off:off + p.numel()], weakref.ref(p))

    def gather(self) -> torch.Tensor:
        """The gradients without the alignment padding, concatenated in parameter order."""
        return torch.cat([self.buffer[o:o + p.numel()] for p, o in zip(self.params, self.offsets)])

    def enable_direct_accumulation(self) -> None:
        """Let the HIP weight-gradient kernels add straight into this buffer (it must be zeroed with
        ``zero()`` before every backward; each parameter must feed exactly the op that owns it).

        Sunk gradients are written by side-stream kernels that autograd does not know about (there is no AccumulateGrad
        node for them): ``p.grad`` is only complete after ``ops.join_side_streams()``, which ``all_reduce_mean()`` and
        ``FusedAdamEMA.step()`` call.  A training loop that reads ``p.grad`` itself after ``backward()`` (a stock torch
        optimizer, ``clip_grad_norm_``) must call ``ops.join_side_streams()`` first -- or not enable this mode."""
        if self.buffer.is_cuda:
            from ..backend import ops

            ops.GRAD_SINK.update(self._sink)

    def disable_direct_accumulation(self) -> None:
        from ..backend import ops

        for k in self._sink:
            ops.GRAD_SINK.pop(k, None)

    def zero(self) -> None:
        self._drain()
        self.buffer.zero_()

    # ------------------------------------------------------------------ all-reduce, overlapped with the backward
    # The collective sequence is STATIC: ``enable_overlapped_all_reduce(model)`` fixes, from the model's structure alone, a
    # schedule of contiguous slices of the flat buffer -- one per message-passing layer in reverse layer order (the order
    # their backward passes finish), then whatever lies between and around them in ascending order -- and every step every
    # rank issues exactly these all-reduces in exactly this order.  A layer that reports through ``ops.GRAD_READY`` only
    # moves the START of its slice's collective forward (to behind the kernels that write it, on the communication stream);
    # a slice whose layer never reports on this rank (its batch took the composed path: too few rows per key, forces,
    # graph capture) is issued by ``all_reduce_mean()``, at its place in the sequence.  Which collectives run, their sizes
    # and their order never depend on the rank's data -- ranks whose shards fall on either side of a path threshold stay in
    # step (an RCCL sequence that differs between ranks hangs or pairs the wrong buffers).
    _comm = None
    _schedule = ()         # [(lo, hi)] in issue order
    _layer_of = None       # (data_ptr, numel) -> (schedule position, weakref to the Parameter)
    _layer_size = ()       # schedule position -> number of parameters in that layer's slice
    _issued = 0            # schedule entries issued this step
    _ready = None          # schedule positions whose layer has reported this step
    _works = ()
    early_start = True     # False: layers' reports are ignored, the whole sequence is issued by all_reduce_mean() (a step whose
                           # backward is a graph replay reports nothing anyway; this also keeps the capture's eager warm-up steps
                           # free of collectives, so a rank whose capture fails cannot leave the others inside one)
    overlapped_slices = 0  # slices whose collective started ahead of all_reduce_mean() since construction (tests / logs)

    def enable_overlapped_all_reduce(self, model: "torch.nn.Module" = None, layer_params=None) -> None:
        """``model``: its ``MessagePassing`` sub-modules define the layer slices (``layer_params``: an explicit list of
        parameter lists instead).  Needs the gradient sink (``enable_direct_accumulation``).  xGMI rings are per-link
        bound: a layer's slice is 4-5 MB here (1.15 M self-connection weights), large enough for the links."""
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return
        if layer_params is None:
            layer_params = []
            if model is not None:
                from ..nn.message_passing import MessagePassing

                # per layer: its node-side weights alone (contiguous when the buffer was laid out by ``flat_param_order``:
                # the radial MLPs' gradients arrive together at the END of the backward when the MLPs run as one stack, so
                # they belong to the remainder), else all its parameters
                for mod in model.modules():
                    if isinstance(mod, MessagePassing):
                        radial = {id(p) for p in _radial_mlp_params(mod)}
                        layer_params.append(([p for p in mod.parameters() if id(p) not in radial], list(mod.parameters())))
        slot = {id(p): (off, -(-p.numel() // FLAT_ALIGN) * FLAT_ALIGN) for p, off in zip(self.params, self.offsets)}
        layers = []
        for candidates in layer_params:
            if not isinstance(candidates, tuple):
                candidates = (candidates,)
            for group in candidates:
                sl = [slot.get(id(p)) for p in group if p.requires_grad]
                if not sl or any(x is None for x in sl):
                    continue
                lo, hi = min(o for o, _ in sl), max(o + n for o, n in sl)
                if sum(n for _, n in sl) != hi - lo:
                    continue                      # not one contiguous run of the flat buffer: stays in the remainder
                layers.append((lo, hi, [p for p in group if p.requires_grad]))
                break
        layers.sort(key=lambda t: -t[0])      # reverse layer order = the order the backward reaches them
        schedule, self._layer_of, self._layer_size = [], {}, []
        for pos, (lo, hi, group) in enumerate(layers):
            if any(lo < h and l < hi for l, h in schedule):
                continue                      # nested MessagePassing modules: the outer one already covers it
            schedule.append((lo, hi))
            self._layer_size.append(len(group))
            for p in group:
                self._layer_of[(p.data_ptr(), p.numel())] = (len(schedule) - 1, weakref.ref(p))
        covered, pos = sorted(schedule), 0
        for lo, hi in covered:
            if lo > pos:
                schedule.append((pos, lo))
            pos = max(pos, hi)
        if pos < self.buffer.numel():
            schedule.append((pos, self.buffer.numel()))
        self._schedule = schedule
        self._issued, self._ready, self._works = 0, set(), []
        if self.buffer.is_cuda:
            from ..backend import ops

            self._comm = torch.cuda.Stream(device=self.buffer.device)
            ops.GRAD_READY = self._on_ready      # (CPU buffers -- the gloo tests -- call _on_ready themselves)

    def disable_overlapped_all_reduce(self) -> None:
        self._drain()
        if self.buffer.is_cuda:
            from ..backend import ops

            if getattr(ops.GRAD_READY, "__self__", None) is self:
                ops.GRAD_READY = None
        self._schedule, self._layer_of = (), None

    def _issue(self, upto: int, early: bool) -> None:
        """Issues schedule entries [issued, upto) in order on the communication stream, behind everything enqueued so far
        on the current stream and the side streams (the weight-gradient kernels of the gradient sink)."""
        if upto <= self._issued:
            return
        if self._comm is not None:
            from ..backend import ops

            comm = self._comm
            comm.wait_stream(torch.cuda.current_stream())
            for st in ops.side_streams_of(self.buffer.device.index):
                comm.wait_stream(st)
            with torch.cuda.stream(comm):
                for lo, hi in self._schedule[self._issued:upto]:
                    self._works.append(dist.all_reduce(self.buffer[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
        else:
            for lo, hi in self._schedule[self._issued:upto]:
                self._works.append(dist.all_reduce(self.buffer[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
        if early:
            self.overlapped_slices += upto - self._issued
        self._issued = upto

    def _on_ready(self, weights) -> None:
        if self._layer_of is None or not self.early_start or (self.buffer.is_cuda and torch.cuda.is_current_stream_capturing()):
            return
        pos, seen = None, set()
        for w in weights:
            key = (w.data_ptr(), w.numel())
            hit = self._layer_of.get(key)
            if hit is None:
                continue                      # not part of a layer slice (the remainder is issued by all_reduce_mean())
            p = hit[1]()
            if p is None or p.data_ptr() != w.data_ptr():
                return                        # the parameter moved after enable_overlapped_all_reduce(): no early start
            if pos is not None and hit[0] != pos:
                return
            pos = hit[0]
            seen.add(key)
        if pos is None or len(seen) != self._layer_size[pos]:
            return                            # the slice holds parameters this report does not cover: their gradients may
                                              # still be on their way (e.g. a radial MLP evaluated in the stack): no early start
        self._ready.add(pos)
        upto = self._issued
        while upto in self._ready:            # only ever the next entries of the fixed sequence
            upto += 1
        self._issue(upto, early=True)

    def _drain(self):
        for work in self._works:
            work.wait()                 # the current stream waits for the collective (RCCL: no host block)
        if self._comm is not None and self._works:
            torch.cuda.current_stream().wait_stream(self._comm)
        self._works = []
        self._issued = 0
        self._ready = set()

    def all_reduce_mean(self) -> None:
        if self.buffer.is_cuda:
            from ..backend import ops

            ops.join_side_streams()   # side-stream weight-gradient kernels (gradient sink) must land first
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            if self._schedule:
                self._issue(len(self._schedule), early=False)      # the rest of the fixed sequence, in order
                self._drain()
            else:
                dist.all_reduce(self.buffer, op=dist.ReduceOp.SUM)
            self.buffer.div_(dist.get_world_size())


def broadcast_parameters(module: torch.nn.Module, src: int = 0) -> None:
    """Rank-0 parameters to all ranks (what DDP does at construction)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src)
