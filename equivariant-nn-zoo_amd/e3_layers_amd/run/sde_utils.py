"""The VP-SDE arithmetic that drives the diffusion score networks (host-side torch plumbing on
device tensors).  Same formulas as ``e3_layers/run/sde_utils.py``:
``VPSDE.marginal`` (:54-66), ``VPSDE.sde`` (:68-81), ``prior_sampling`` (:83-86), ``reverse`` (:88-123),
``get_score_fn`` (:176-187), the loss of ``get_sde_loss_fn`` (:143-171).  The head key is ``score_{key}`` as those functions expect; a model that emits the
shipped config's plain ``score`` key is mapped onto it.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch


class VPSDE:
    def __init__(self, diffusion_keys: Dict[str, int], beta_min: float = 0.1, beta_max: float = 20.0, N: int = 1000):
        self.beta_0, self.beta_1, self.N = beta_min, beta_max, N
        self.irreps = dict(diffusion_keys)
        self.discrete_betas = torch.linspace(beta_min / N, beta_max / N, N)
        self.alphas = 1.0 - self.discrete_betas

    @property
    def T(self) -> float:
        return 1.0

    def log_mean_coeff(self, t: torch.Tensor) -> torch.Tensor:
        return -0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0

    def std(self, batch) -> torch.Tensor:
        t = batch["t"][batch.nodeSegment()]
        return torch.sqrt(1.0 - torch.exp(2.0 * self.log_mean_coeff(t)))

    def marginal(self, batch, return_std: bool = False, generator=None):
        """x_t = exp(log_mean) x_0 + std z per diffused key, with the per-graph time broadcast to nodes."""
        if return_std:
            return self.std(batch)
        t = batch["t"][batch.nodeSegment()]
        lm = self.log_mean_coeff(t)
        std = torch.sqrt(1.0 - torch.exp(2.0 * lm))
        zs = {}
        for key in self.irreps:
            x = batch[key]
            z = torch.randn(x.shape, device=x.device, dtype=x.dtype, generator=generator)
            batch[key] = torch.exp(lm) * x + std * z
            zs[key] = z
        return batch, {"zs": zs, "std": std}


def _randn(like: torch.Tensor, generator=None, noise_fn=None) -> torch.Tensor:
    if noise_fn is not None:
        return noise_fn(like.shape).to(like.device, like.dtype)
    return torch.randn(like.shape, device=like.device, dtype=like.dtype, generator=generator)


def _node_t(batch) -> torch.Tensor:
    return batch["t"].reshape(-1, 1)[batch.nodeSegment()]


def vpsde_sde(sde: VPSDE, batch, dt=None, generator=None, noise_fn=None):
    """One Euler-Maruyama step of the FORWARD SDE dx = -beta/2 x dt + sqrt(beta) dw (``VPSDE.sde`` :68-81);
    called with ``dt = -1/N`` by the reverse sampler."""
    if dt is None:
        dt = 1.0 / sde.N
    t = _node_t(batch)
    beta_t = sde.beta_0 + t * (sde.beta_1 - sde.beta_0)
    diffusion = torch.sqrt(beta_t)
    for key in sde.irreps:
        x = batch[key]
        x_mean = x + (-0.5 * beta_t * x) * dt
        batch[key] = x_mean + diffusion * (abs(dt) ** 0.5) * _randn(x, generator, noise_fn)
    return batch


def prior_sampling(sde: VPSDE, batch, generator=None, noise_fn=None):
    """x_T ~ N(0, 1) per diffused key (:83-86), drawn on the batch's device."""
    dev = batch["_n_nodes"].device
    n = batch["_node_segment"].shape[0] if "_node_segment" in batch else int(batch["_n_nodes"].sum())
    for key, dim in sde.irreps.items():
        batch[key] = _randn(torch.empty(n, dim, device=dev), generator, noise_fn)
    return batch


def reverse_step(sde: VPSDE, score_fn, batch, generator=None, noise_fn=None):
    """One step of the reverse-time SDE (``RSDE.sde`` :104-119): forward-SDE Euler step with ``dt = -1/N``,
    then ``x -= dt * beta_t * score``."""
    scores = score_fn(batch)
    t = _node_t(batch)
    beta_t = sde.beta_0 + t * (sde.beta_1 - sde.beta_0)
    dt = -1.0 / sde.N
    batch = vpsde_sde(sde, batch, dt, generator, noise_fn)
    for key in sde.irreps:
        batch[key] = batch[key] - dt * beta_t * scores[f"score_{key}"]
    return batch


def get_score_fn(sde: VPSDE, model, train: bool = False):
    def score_fn(batch):
        model.train(train)
        result = model(batch)
        std = sde.std(batch)
        for key in sde.irreps:
            name = f"score_{key}"
            raw = result[name] if name in result else result["score"]
            result[name] = -raw / std - batch[key]
        return result

    return score_fn


def sde_loss(sde: VPSDE, model, batch, eps: float = 1e-5, train: bool = True, generator=None,
             node_weight=None) -> Tuple[torch.Tensor, dict]:
    """mean over graphs/nodes of (score * std + z)^2, t ~ U(eps, 1) per graph.  ``node_weight`` [N, 1] (sums to 1 over the
    nodes that count): a weighted mean instead -- a batch padded with a ghost graph (run/graph_step.py) gives it weight 0."""
    pert, misc = sde_perturb(sde, batch, eps, generator)
    return sde_loss_of(sde, model, pert, misc, train, node_weight)


def sde_perturb(sde: VPSDE, batch, eps: float = 1e-5, generator=None):
    """First half of ``sde_loss``: t ~ U(eps, 1) per graph and the noised copy of the batch (no model involved) -> (pert, misc).
    A loop that wants the next batch's data-only layers early calls ``model.prepare(pert)`` on the result."""
    dev = batch["_n_nodes"].device
    t = torch.rand(len(batch), device=dev, generator=generator) * (sde.T - eps) + eps
    pert = batch.clone()
    pert.attrs["t"] = ("graph", "1x0e")
    pert["t"] = t
    return sde.marginal(pert, generator=generator)


def sde_loss_of(sde: VPSDE, model, pert, misc, train: bool = True, node_weight=None) -> Tuple[torch.Tensor, dict]:
    """Second half of ``sde_loss``: the score network on the noised batch and the denoising loss."""
    scores = get_score_fn(sde, model, train)(pert)
    losses = {}
    for key in sde.irreps:
        err = torch.square(scores[f"score_{key}"] * misc["std"] + misc["zs"][key])
        per_node = err.reshape(err.shape[0], -1).mean(dim=-1)
        losses[key] = per_node.mean() if node_weight is None else (per_node * node_weight.reshape(-1)).sum()
    total = sum(losses.values())
    losses["total"] = total
    return total, losses
