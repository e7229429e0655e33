"""Seeded synthetic stand-ins for the datasets the BASELINE configs are quoted on (no dataset
files exist offline).

``synth_qm9(seed, B)``: QM9-like molecules — n = clamp(round(18 + 4.5 randn), 3, 29) atoms,
species from {H .51, C .35, N .06, O .075, F .005}, positions by random tree growth (new atom
1.0-1.55 A from a random existing atom, rejected within 0.95 A of any atom), target
``total_energy = sum shift[Z] + 0.1 randn``.  Species are *type indices* into the first
``num_types`` chemical symbols (X, H, He, Li, Be, B, C, N, O, F -> H=1, C=6, N=7, O=8, F=9),
as the reference's ``type_names`` list implies (``e3_layers/configs/config_energy.py:47``).
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch

from .compute_edge import computeEdgeIndex
from .data import Batch, Data

_SPECIES = torch.tensor([1, 6, 7, 8, 9])
_PROBS = torch.tensor([0.51, 0.35, 0.06, 0.075, 0.005])


def _grow_molecule(n: int, gen: torch.Generator) -> torch.Tensor:
    pos = torch.zeros(n, 3)
    placed = 1
    while placed < n:
        anchor = int(torch.randint(placed, (1,), generator=gen))
        direction = torch.randn(3, generator=gen)
        direction = direction / direction.norm().clamp(min=1e-9)
        dist = 1.0 + 0.55 * float(torch.rand(1, generator=gen))
        cand = pos[anchor] + dist * direction
        if float((pos[:placed] - cand).norm(dim=1).min()) >= 0.95:
            pos[placed] = cand
            placed += 1
    return pos


def synth_qm9_list(seed: int, n_mol: int, shifts: Optional[List[float]] = None, r_max: Optional[float] = 4.0):
    gen = torch.Generator(device="cpu").manual_seed(seed)
    attrs = {"pos": ("node", "1x1o"), "species": ("node", "1x0e"), "total_energy": ("graph", "1x0e")}
    out = []
    for _ in range(n_mol):
        n = int(min(29, max(3, round(18 + 4.5 * float(torch.randn(1, generator=gen))))))
        z = _SPECIES[torch.multinomial(_PROBS, n, replacement=True, generator=gen)]
        pos = _grow_molecule(n, gen).float()
        e = 0.1 * float(torch.randn(1, generator=gen))
        if shifts is not None:
            e += float(sum(shifts[int(t)] for t in z))
        sample = {"pos": pos, "species": z.view(-1, 1).long(), "total_energy": torch.tensor([[e]]),
                  "_n_nodes": torch.tensor([[n]])}
        if r_max is not None:
            sample_attrs = dict(attrs)
            new, _ = computeEdgeIndex(sample, sample_attrs, r_max=r_max)
            sample["edge_index"] = new["edge_index"]
        out.append(sample)
    return out, attrs


def synth_qm9(seed: int, n_mol: int, shifts: Optional[List[float]] = None, r_max: float = 4.0) -> Batch:
    """A Batch with pos, species, total_energy, edge_index (cutoff r_max), _n_nodes, _n_edges."""
    lst, attrs = synth_qm9_list(seed, n_mol, shifts, r_max)
    return Batch.from_data_list(lst, dict(attrs))


def synth_qm9_diffusion(seed: int, n_mol: int, std: float = 1.4) -> Batch:
    """Inputs of the small-molecule score network (``e3_layers/configs/config_diffusion.py``):
    positions scaled by 1/std (:43-44), fully connected graphs (``r_max=9999`` preprocess, :50),
    a bond type in {0..3} per edge (0 = no bond beyond 1.7 A) and a diffusion time per graph."""
    lst, attrs = synth_qm9_list(seed, n_mol, None, r_max=None)
    gen = torch.Generator(device="cpu").manual_seed(seed + 7919)
    attrs = dict(attrs)
    attrs["bond_type"] = ("edge", "1x0e")
    attrs["t"] = ("graph", "1x0e")
    for s in lst:
        s["pos"] = s["pos"] / std
        new, _ = computeEdgeIndex(s, dict(attrs), r_max=9999.0)
        ei = new["edge_index"]
        s["edge_index"] = ei
        d = (s["pos"][ei[0]] - s["pos"][ei[1]]).norm(dim=1) * std
        bonded = d < 1.7
        kind = torch.randint(1, 4, (ei.shape[1],), generator=gen)
        s["bond_type"] = torch.where(bonded, kind, torch.zeros_like(kind)).view(-1, 1).long()
        s["t"] = torch.rand(1, 1, generator=gen) * (1.0 - 1e-5) + 1e-5
        s.pop("total_energy")
    attrs.pop("total_energy")
    return Batch.from_data_list(lst, attrs)


def synth_protein(seed: int, n_prot: int, n_res: int = 384, std: float = 25.83, n_chains: int = 2,
                  backbone: bool = False) -> Batch:
    """Inputs of the residue-level score network (``config_diffusion_CA``): per protein a C-alpha
    random walk with 3.8 A steps split into ``n_chains`` chains, residue types in [0, 21), residue
    index ``id``, ``chain_id``, a diffusion time per graph; CA centred and scaled by 1/std as the
    config's scaler does.  No edges: the model's first layer (computeEdgeIndex) builds them.
    ``backbone``: also N, C, O as ``config_diffusion_backbone``'s scaler leaves them — C and N relative to CA, O relative
    to C (bond lengths 1.52 / 1.46 / 1.23 A in random directions), scaled by 1/std."""
    gen = torch.Generator(device="cpu").manual_seed(seed)
    attrs = {"CA": ("node", "1x1o"), "species": ("node", "1x0e"), "chain_id": ("node", "1x0e"), "id": ("node", "1x0e"),
             "t": ("graph", "1x0e")}
    if backbone:
        attrs.update({"N": ("node", "1x1o"), "C": ("node", "1x1o"), "O": ("node", "1x1o")})
    lst = []
    for _ in range(n_prot):
        steps = torch.randn(n_res, 3, generator=gen)
        steps = 3.8 * steps / steps.norm(dim=1, keepdim=True)
        ca = torch.cumsum(steps, 0)
        ca = (ca - ca.mean(0, keepdim=True)) / std
        chain = (torch.arange(n_res) * n_chains // n_res).view(-1, 1)
        lst.append({"CA": ca.float(), "species": torch.randint(0, 21, (n_res, 1), generator=gen),
                    "chain_id": chain.long(), "id": torch.arange(n_res).view(-1, 1),
                    "t": torch.rand(1, 1, generator=gen) * (1.0 - 1e-5) + 1e-5, "_n_nodes": torch.tensor([[n_res]])})
        if backbone:
            for atom, bond in (("C", 1.52), ("N", 1.46), ("O", 1.23)):
                d = torch.randn(n_res, 3, generator=gen)
                lst[-1][atom] = (bond * d / d.norm(dim=1, keepdim=True) / std).float()
    b = Batch.from_data_list(lst, attrs)
    b.attrs.pop("_n_edges", None)
    return b
