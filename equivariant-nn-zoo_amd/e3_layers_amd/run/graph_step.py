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

from typing import Any, Callable

import torch


class CapturedStep:
    def __init__(self, fn: Callable[[], Any], warmup: int = 3, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("CapturedStep needs a HIP device: there is no CPU fallback for graph replay")
        self.fn = fn
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        # warm-up on a side stream: lazy initialisation (plans, side streams, allocator pools) must not be captured
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                out = fn()
        torch.cuda.current_stream(dev).wait_stream(side)
        del out
        torch.cuda.synchronize(dev)
        from ..backend.graph import check_indices

        check_indices()      # the warm-up batches' deferred index checks (none are recorded while capturing)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = fn()

    def __call__(self):
        self.graph.replay()
        return self.out
