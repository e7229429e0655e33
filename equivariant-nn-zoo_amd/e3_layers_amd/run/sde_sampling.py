"""Predictor-corrector sampling of the reverse VP-SDE — the loop of
``e3_layers/run/sde_sampling.py:185-244``, kept on the device.

What the reference does per reverse step (``pc_sampler`` :229-242): set the time, run the corrector
(Langevin, :117-143), drop ``edge_index`` / ``edge_vector``, run the predictor (Euler-Maruyama on the reverse
SDE, :98-105 + ``sde_utils.py:104-119``), drop the edges again — the model's own ``edge_index`` layer (or the
dataset's preprocess functions) rebuilds the neighbour list from the moved positions: 2·N network
evaluations, each latency-bound.

MI355X-first changes, same arithmetic:
* the edge rebuild runs the device radius-graph kernels (``data/compute_edge.py``), no host round trip of
  positions; the CSR topology the fused convolution needs is rebuilt with it;
* when the edge set cannot change (fully connected molecules: ``config_diffusion`` preprocesses with
  ``r_max=9999``) the edges and topology are built once and one whole corrector+predictor step — two
  network evaluations, noise draws and updates — is captured in a HIP graph and replayed N times
  (``graph=True``): the step time falls from launch-bound to GPU-bound;
* the time ``t`` lives in a device tensor that is updated in place, so the captured graph sees it.

Noise comes from ``generator`` (or an injected ``noise_fn(shape)`` — the parity tests feed the oracle the
same draws).  The reference's corrector evaluates the score ``n_steps`` times on the *unchanged* batch
(:131-142 never writes ``x`` back inside the loop); here every inner step sees the updated positions, which
is identical for the default ``n_steps=1``.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Sequence

import torch

from ..backend.graph import TOPO_KEYS
from .sde_utils import VPSDE, _node_t, _randn, get_score_fn, prior_sampling, reverse_step

class _Registry(dict):
    """name -> class table with a decorator: ``@table.register(name="langevin")`` (bare ``@table.register`` uses the
    class name); a name is registered once."""

    def __init__(self, kind: str):
        super().__init__()
        self.kind = kind

    def register(self, cls=None, *, name=None):
        def add(c):
            key = name or c.__name__
            if key in self:
                raise ValueError(f"{self.kind} {key!r} is already registered")
            self[key] = c
            return c
        return add if cls is None else add(cls)


_PREDICTORS, _CORRECTORS = _Registry("predictor"), _Registry("corrector")
# the reference's public names (e3_layers/run/sde_sampling.py:14-60)
register_predictor, register_corrector = _PREDICTORS.register, _CORRECTORS.register
get_predictor, get_corrector = _PREDICTORS.__getitem__, _CORRECTORS.__getitem__


class Predictor:
    def __init__(self, sde: VPSDE, score_fn):
        self.sde, self.score_fn = sde, score_fn

    def update_fn(self, batch, generator=None, noise_fn=None):
        raise NotImplementedError


class Corrector:
    def __init__(self, sde: VPSDE, score_fn, snr: float, n_steps: int):
        self.sde, self.score_fn, self.snr, self.n_steps = sde, score_fn, snr, n_steps

    def update_fn(self, batch, generator=None, noise_fn=None):
        raise NotImplementedError


@register_predictor(name="euler_maruyama")
class EulerMaruyamaPredictor(Predictor):
    def update_fn(self, batch, generator=None, noise_fn=None):
        return reverse_step(self.sde, self.score_fn, batch, generator, noise_fn)


@register_predictor(name="none")
class NonePredictor(Predictor):
    def update_fn(self, batch, generator=None, noise_fn=None):
        return batch


@register_corrector(name="langevin")
class LangevinCorrector(Corrector):
    def __init__(self, sde, score_fn, snr, n_steps):
        super().__init__(sde, score_fn, snr, n_steps)
        if not isinstance(sde, VPSDE):
            raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")

    def update_fn(self, batch, generator=None, noise_fn=None):
        sde = self.sde
        t = _node_t(batch)
        timestep = (t * (sde.N - 1) / sde.T).long()
        alpha = sde.alphas.to(t.device)[timestep]
        for _ in range(self.n_steps):
            scores = self.score_fn(batch)
            for key in sde.irreps:
                x, grad = batch[key], scores[f"score_{key}"]
                noise = _randn(x, generator, noise_fn)
                grad_norm = torch.norm(grad.reshape(grad.shape[0], -1), dim=-1).mean()
                noise_norm = torch.norm(noise.reshape(noise.shape[0], -1), dim=-1).mean()
                step_size = (self.snr * noise_norm / grad_norm) ** 2 * 2 * alpha
                batch[key] = x + step_size * grad + torch.sqrt(step_size * 2) * noise
        return batch


@register_corrector(name="none")
class NoneCorrector(Corrector):
    def update_fn(self, batch, generator=None, noise_fn=None):
        return batch


_EDGE_KEYS = ("edge_index", "edge_vector", "edge_length", "_n_edges", "_edge_segment") + TOPO_KEYS


def get_pc_sampler(sde: VPSDE, predictor, corrector, inverse_scaler: Callable = None, snr: float = 0.16,
                   n_steps: int = 1, continuous: bool = True, eps: float = 1e-3,
                   preprocess: Sequence[Callable] = (), static_edges: bool = False, graph: bool = False,
                   n_iter: Optional[int] = None):
    """``pc_sampler(model, batch, generator=None, noise_fn=None) -> (batch, n_function_evaluations)``.

    preprocess: the dataset's ``(data, attrs) -> (data, attrs)`` functions (``data_config.preprocess``) that
        rebuild ``edge_index`` when the model tree has no ``edge_index`` layer of its own.
    static_edges: the edge set does not depend on the positions (fully connected graphs): build it once.
    graph: with ``static_edges``, capture one corrector+predictor step in a HIP graph and replay it.
    n_iter: stop after this many of the ``sde.N`` reverse steps (harness addition: benchmarks and parity tests
        time / check the first steps of the N=1000 schedule instead of shrinking N, which changes dt and betas).
    """
    inverse_scaler = inverse_scaler or (lambda b: b)
    predictor = predictor or NonePredictor
    corrector = corrector or NoneCorrector
    if graph and not static_edges:
        raise ValueError("graph capture needs static_edges=True (a changing edge count changes every launch)")

    def rebuild_edges(batch):
        for k in _EDGE_KEYS:
            batch.pop(k)
        for fn in preprocess:
            new, attrs = fn(batch.data, batch.attrs)
            batch.attrs.update(attrs)
            batch.update(new)
        return batch

    steps = sde.N if n_iter is None else min(int(n_iter), sde.N)

    def pc_sampler(model, batch, generator=None, noise_fn=None):
        batch = batch.clone()
        dev = batch["_n_nodes"].device
        batch.attrs["t"] = ("graph", "1x0e")
        batch = prior_sampling(sde, batch, generator, noise_fn)
        sde.alphas = sde.alphas.to(dev)      # resident before any capture (no pageable host copy inside a graph)
        timesteps = torch.linspace(sde.T, eps, sde.N, device=dev)
        score_fn = get_score_fn(sde, model, train=False)
        pred, corr = predictor(sde, score_fn), corrector(sde, score_fn, snr, n_steps)
        batch["t"] = torch.empty(len(batch), 1, device=dev)
        if static_edges:
            from ..backend.graph import build_topology

            if "edge_index" not in batch:
                rebuild_edges(batch)
            if "_e3k_dst_ptr" not in batch and batch["edge_index"].is_cuda:
                batch.update(build_topology(batch["edge_index"], batch.n_nodes).as_dict())
        t_dev = batch["t"]
        keys = list(sde.irreps)

        def moved(b):
            # positions changed: the geometry cached in the batch by the model's first layer is stale
            # (the reference pops edge_index / edge_vector here, :236-241)
            if static_edges:
                b.pop("edge_vector")
                b.pop("edge_length")
                return b
            return rebuild_edges(b)

        def one_step(b):
            b = moved(corr.update_fn(b, generator, noise_fn))
            return moved(pred.update_fn(b, generator, noise_fn))

        with torch.no_grad():
            if graph:
                # static buffers: the diffused tensors and t; the step writes its result back into them
                state = {k: batch[k].clone() for k in keys}

                def captured():
                    work = batch.clone()           # model layers add keys: work on a copy, keep `batch` pristine
                    for k in keys:
                        work[k] = state[k]
                    work["t"] = t_dev
                    work = one_step(work)
                    for k in keys:
                        state[k].copy_(work[k])

                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    saved = {k: v.clone() for k, v in state.items()}
                    t_dev.fill_(float(timesteps[0]))
                    captured()                     # warm-up (allocations, plan creation) outside the capture
                    for k in keys:
                        state[k].copy_(saved[k])
                torch.cuda.current_stream().wait_stream(side)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    captured()
                for k in keys:                     # the capture pass itself does not execute; restore anyway
                    state[k].copy_(saved[k])
                for i in range(steps):
                    t_dev.copy_(timesteps[i].expand_as(t_dev))
                    g.replay()
                for k in keys:
                    batch[k] = state[k]
            else:
                for i in range(steps):
                    t_dev.copy_(timesteps[i].expand_as(t_dev))
                    batch["t"] = t_dev
                    batch = one_step(batch)
        return inverse_scaler(batch), steps * (n_steps + 1)

    return pc_sampler


def get_sampling_fn(config, sde: VPSDE, inverse_scaler, eps: float, **kwargs):
    """``config.sampling.{method, predictor, corrector, snr, n_steps_each}`` → sampler (:247-286; only 'pc')."""
    name = config.sampling.method.lower()
    if name != "pc":
        raise ValueError(f"Sampler name {config.sampling.method} unknown.")
    return get_pc_sampler(sde=sde, predictor=get_predictor(config.sampling.predictor.lower()),
                          corrector=get_corrector(config.sampling.corrector.lower()), inverse_scaler=inverse_scaler,
                          snr=config.sampling.snr, n_steps=config.sampling.n_steps_each,
                          continuous=getattr(getattr(config, "training", None), "continuous", True), eps=eps, **kwargs)
