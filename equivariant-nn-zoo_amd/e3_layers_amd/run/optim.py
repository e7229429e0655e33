"""Fused training-step plumbing on the flat parameter vector.

Stands in for the reference's per-step sequence ``clip_grad_norm_`` → ``optim.step()`` →
``ema.update()`` (``e3_layers/run/trainer.py:374-386``; diffusion variant with the
non-finite-gradient skip ``e3_layers/run/sde_utils.py:233-248``).  Parameters, gradients, both
Adam moments and the EMA shadow live in five flat fp32 buffers; one ``e3k_adam_ema_step`` call
(a gradient-norm reduction, a one-thread tick that derives the step-dependent scalars on the
device, and one streaming update kernel) replaces ≈10 multi-tensor launches and touches each
parameter's 28-36 bytes exactly once.  Step count and bias corrections are device state, so the
whole step can be captured in a HIP graph.

Semantics follow ``torch.optim.Adam`` (amsgrad off, decoupled=False weight decay) and
``torch_ema.ExponentialMovingAverage`` (``decay=min(decay, (1+k)/(10+k))`` with ``use_num_updates``).
"""
from __future__ import annotations

from contextlib import contextmanager
from typing import Iterable, Optional, Tuple

import torch

from ..backend import lib as L
from .parallel import FlatGradients, flat_layout


def layout_moves(mine, theirs, assume_same_order: bool = False):
    """How to copy a checkpoint's flat vectors (layout ``theirs``) into an optimizer's (layout ``mine``): None = identical
    layouts (plain copies), else [(dst offset, src offset, numel)] matched by name; raises when the two cannot be matched.
    ``assume_same_order``: the caller vouches that the checkpoint was written for the same parameters in the same order -- the
    way in for checkpoints without a layout (written before round 4) or without names."""
    if theirs is None:
        if assume_same_order:
            return None
        raise ValueError("checkpoint without a parameter layout (written before round 4): it cannot be told whether its "
                         "flat vectors follow this optimizer's parameter order -- load it with assume_same_order=True if it was "
                         "written by an optimizer built over the same parameters in the same order, or re-save it with the current code")
    mine = [(int(o), int(n), tuple(sh), nm) for o, n, sh, nm in mine]
    theirs = [(int(o), int(n), tuple(sh), nm) for o, n, sh, nm in theirs]
    same_shapes = [(o, n, sh) for o, n, sh, _ in mine] == [(o, n, sh) for o, n, sh, _ in theirs]
    named_m, named_t = all(nm is not None for *_, nm in mine), all(nm is not None for *_, nm in theirs)
    if same_shapes and named_m and named_t and [nm for *_, nm in mine] == [nm for *_, nm in theirs]:
        return None
    if same_shapes and not (named_m and named_t):
        # equal shape sequences do not prove equal ORDER (several identical layers: flat_param_order(model) and
        # model.parameters() hold same-shaped tensors in different places) -- without names on both sides only the caller can tell
        if assume_same_order or len({(n, sh) for _, n, sh, _ in mine}) == len(mine):      # (all shapes distinct: the order IS determined)
            return None
        raise ValueError("checkpoint and optimizer have the same sequence of parameter shapes but names are missing on one side and "
                         "several parameters share a shape: the order cannot be verified -- build both FusedAdamEMA objects with "
                         "names=, or load with assume_same_order=True")
    if named_m and named_t:
        src = {nm: (o, n, sh) for o, n, sh, nm in theirs}
        if set(src) != {nm for *_, nm in mine}:
            raise ValueError("checkpoint and optimizer hold different parameter names: "
                             f"{sorted(set(src) ^ {nm for *_, nm in mine})[:6]} ...")
        moves = []
        for o, n, sh, nm in mine:
            so, sn, ssh = src[nm]
            if (sn, ssh) != (n, sh):
                raise ValueError(f"parameter {nm}: checkpoint shape {ssh} != {sh}")
            moves.append((o, so, n))
        return moves
    raise ValueError("the checkpoint's parameter layout differs from this optimizer's and names are missing on one side: "
                     "build both FusedAdamEMA objects with names= (or with the same parameter order)")


class FusedAdamEMA:
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999),
                 eps: float = 1e-8, weight_decay: float = 0.0, ema_decay: Optional[float] = None,
                 ema_use_num_updates: bool = True, max_grad_norm: Optional[float] = None, skip_nonfinite: bool = False,
                 max_steps_ahead: int = 2, names: Optional[Iterable[str]] = None):
        """``names``: one name per entry of ``params`` (e.g. from ``model.named_parameters()``; ``named(model, order)`` builds
        them): stored with ``state_dict()`` so that a checkpoint written under another parameter ORDER (``flat_param_order``
        moves the radial MLPs to the tail) is re-mapped by name instead of silently permuting weights and moments."""
        params = list(params)
        if names is not None:
            names = [n for n, p in zip(list(names), params) if p.requires_grad]
        self.params = [p for p in params if p.requires_grad]
        self.names = names
        if names is not None and (len(names) != len(self.params) or len(set(names)) != len(names)):
            raise ValueError("names must be unique and match params one to one")
        if not self.params:
            raise ValueError("no trainable parameters")
        L.require_cuda(*self.params)
        dev = self.params[0].device
        self.offsets, total = flat_layout(self.params)
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):
                self.flat[off:off + p.numel()].copy_(p.detach().reshape(-1))
                p.data = self.flat[off:off + p.numel()].view_as(p)      # parameters become views of the flat vector
        self.grads = FlatGradients(self.params)                      # after the re-pointing: sink keys use the new addresses
        assert self.grads.offsets == self.offsets
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.ema = self.flat.clone() if ema_decay is not None else None
        self.state = torch.zeros(16, dtype=torch.float32, device=dev)
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.ema_decay = float(ema_decay) if ema_decay is not None else 0.0
        self.ema_use_num_updates = bool(ema_use_num_updates)
        self.max_grad_norm = float(max_grad_norm) if (max_grad_norm is not None and max_grad_norm < float("inf")) else 0.0
        self.skip_nonfinite = bool(skip_nonfinite)
        # The host enqueues a step in less time than the GPU runs it, so without a bound it drifts many steps ahead.
        # With several HIP streams that is not harmless: a block that was used on a side stream (record_stream) only
        # returns to the caching allocator when the GPU has passed that point, so every step the host is ahead needs
        # its own set of activations -- 156 GB reserved for a 14 GB working set at 1024 molecules, grown by
        # multi-second hipMalloc storms in the middle of a run.  step() therefore waits for the step issued
        # ``max_steps_ahead`` steps ago (0 = unbounded).
        self.max_steps_ahead = int(max_steps_ahead)
        self._step_events = []
        self.waited_seconds = 0.0      # host time spent in that wait since construction

    # ------------------------------------------------------------------ step
    def zero_grad(self) -> None:
        self.grads.zero()

    @torch.no_grad()
    def step(self) -> None:
        """(all-reduce of the flat gradient is the caller's: ``self.grads.all_reduce_mean()``)"""
        from ..backend import ops

        from ..backend.graph import poll_indices

        poll_indices()               # device-side index checks of the batches so far (edge endpoints, row keys): no sync
        ops.join_side_streams()      # weight gradients written by side-stream kernels (gradient sink) are complete
        L.check(L.load().e3k_adam_ema_step(
            L.ptr(self.flat), L.ptr(self.grads.buffer), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq), L.ptr(self.ema),
            self.flat.numel(), self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, self.ema_decay,
            int(self.ema_use_num_updates), self.max_grad_norm, int(self.skip_nonfinite), L.ptr(self.state),
            L.stream_ptr()), "e3k_adam_ema_step")
        if self.max_steps_ahead > 0 and not torch.cuda.is_current_stream_capturing():
            done = torch.cuda.Event()
            done.record()
            self._step_events.append(done)
            if len(self._step_events) > self.max_steps_ahead:
                ev = self._step_events.pop(0)
                if not ev.query():          # the host is ahead of the GPU: wait, and account for it (bench.py: host busy time)
                    import time

                    t0 = time.perf_counter()
                    ev.synchronize()
                    self.waited_seconds += time.perf_counter() - t0

    # ------------------------------------------------------------------ introspection (each is a device->host sync)
    @property
    def steps_taken(self) -> int:
        return int(self.state[0].item())

    @property
    def last_grad_norm(self) -> float:
        """Total gradient norm seen by the last step (only computed when clipping / skipping is on)."""
        return float(self.state[7].item())

    # ------------------------------------------------------------------ EMA weights (validation / checkpoints)
    @contextmanager
    def average_parameters(self):
        """``with opt.average_parameters(): validate(model)`` — torch_ema's context manager
        (``e3_layers/run/trainer.py:438-439``): parameters hold the EMA inside, are restored after."""
        from ..backend.graph import check_indices

        check_indices()              # a natural sync point: every batch so far had valid indices
        if self.ema is None:
            yield
            return
        saved = self.flat.clone()
        self.flat.copy_(self.ema)
        try:
            yield
        finally:
            self.flat.copy_(saved)

    def layout(self) -> list:
        """[(offset, numel, shape, name or None)] of every parameter in the flat vectors -- the meaning of ``state_dict()``'s tensors."""
        names = self.names if self.names is not None else [None] * len(self.params)
        return [(int(o), int(p.numel()), tuple(p.shape), n) for p, o, n in zip(self.params, self.offsets, names)]

    def state_dict(self) -> dict:
        return {"layout": self.layout(), "flat": self.flat.clone(), "exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(),
                "ema": None if self.ema is None else self.ema.clone(), "state": self.state.clone(),
                "hyper": dict(lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=self.weight_decay,
                              ema_decay=self.ema_decay, ema_use_num_updates=self.ema_use_num_updates,
                              max_grad_norm=self.max_grad_norm, skip_nonfinite=self.skip_nonfinite)}

    def load_state_dict(self, sd: dict, assume_same_order: bool = False) -> None:
        """The flat vectors carry no structure of their own: the checkpoint's ``layout`` must be this optimizer's (same
        parameters in the same order), or both sides must carry names, in which case every slice is copied to where its
        parameter lives here.  Anything else raises -- equal lengths do not make two layouts the same
        (``flat_param_order(model)`` and ``model.parameters()`` hold the same tensors in different orders) -- unless the caller
        passes ``assume_same_order=True`` (layout-less checkpoints of earlier rounds; checkpoints without names)."""
        moves = layout_moves(self.layout(), sd.get("layout"), assume_same_order)
        if moves is None and sd["flat"].numel() != self.flat.numel():
            raise ValueError(f"checkpoint holds {sd['flat'].numel()} parameters, this optimizer {self.flat.numel()}")

        def put(dst, src_t):
            if moves is None:
                dst.copy_(src_t)
            else:
                for o, so, n in moves:
                    dst[o:o + n].copy_(src_t[so:so + n])

        with torch.no_grad():
            put(self.flat, sd["flat"])
            put(self.exp_avg, sd["exp_avg"])
            put(self.exp_avg_sq, sd["exp_avg_sq"])
            if self.ema is not None and sd.get("ema") is not None:
                put(self.ema, sd["ema"])
            self.state.copy_(sd["state"])
        for k, v in sd.get("hyper", {}).items():
            setattr(self, k, v)
