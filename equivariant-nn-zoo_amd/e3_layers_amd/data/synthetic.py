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


# ---- bonds="clustered": chemistry-like geometry ---------------------------------------------------------------------------
# Real QM9 does not spread its bond lengths over 1.0-1.55 A: C-H sits at 1.09, C-C at 1.52, C-O at 1.43, C=O at 1.21 A (+- 0.01),
# and with tetrahedral angles the second neighbours cluster too (H-C-H 1.78, C-C-C 2.5 A).  Thousands of edges of a batch then
# share a handful of 2-8 mA knot bins of the radial table -- the case VERDICT r3 found untested (its rank sort and one-wave-per-bin
# transpose degenerated there).  This generator reproduces that distribution; nothing else about it is meant to be chemistry.
_BOND = {(6, 6): 1.52, (6, 7): 1.47, (6, 8): 1.43, (6, 9): 1.35, (7, 7): 1.45, (7, 8): 1.40, (8, 8): 1.48, (7, 9): 1.36, (8, 9): 1.42,
         (9, 9): 1.42, (1, 6): 1.09, (1, 7): 1.01, (1, 8): 0.96, (1, 9): 0.92, (1, 1): 0.74}
_VALENCE = {1: 1, 6: 4, 7: 3, 8: 2, 9: 1}
_TETRA = math.acos(-1.0 / 3.0)


def _grow_molecule_clustered(z: torch.Tensor, gen: torch.Generator):
    """(positions, species reordered heavy atoms first): a bonded tree with element-pair bond lengths (+- 0.01 A; a fifth of the C-O
    bonds are the 1.21 A double bond) and tetrahedral angles at every atom that already has a bond."""
    z = torch.cat([z[z != 1], z[z == 1]])
    n = z.numel()
    pos = torch.zeros(n, 3)
    free = [_VALENCE[int(t)] for t in z]
    first_bond = [None] * n                      # direction of each atom's first bond (the reference the angles are measured from)
    placed = 1
    while placed < n:
        zi = int(z[placed])
        cands = [a for a in range(placed) if free[a] > 0 and (int(z[a]) != 1 or placed == 1)]
        if not cands:
            cands = [a for a in range(placed) if int(z[a]) != 1] or list(range(placed))      # valences exhausted: over-bond a heavy atom
        ok = False
        for attempt in range(40):
            a = cands[int(torch.randint(len(cands), (1,), generator=gen))]
            za = int(z[a])
            length = _BOND[(min(za, zi), max(za, zi))]
            if {za, zi} == {6, 8} and float(torch.rand(1, generator=gen)) < 0.2:
                length = 1.21
            length += 0.01 * float(torch.randn(1, generator=gen))
            rnd = torch.randn(3, generator=gen)
            if first_bond[a] is None or attempt >= 30:
                d = rnd / rnd.norm().clamp(min=1e-9)
            else:                                # tetrahedral angle to the atom's first bond, random azimuth
                u = first_bond[a]
                perp = rnd - (rnd @ u) * u
                perp = perp / perp.norm().clamp(min=1e-9)
                d = math.cos(_TETRA) * u + math.sin(_TETRA) * perp
            cand = pos[a] + length * d
            dist = (pos[:placed] - cand).norm(dim=1)
            dist[a] = 10.0
            if float(dist.min()) >= (1.5 if attempt < 20 else 0.95):      # non-bonded contacts: >= 1.5 A (geminal H-H is 1.78)
                ok = True
                break
        if not ok:                               # crowded: anywhere 1.0-1.55 A from some atom, as the uniform generator does
            while True:
                a = int(torch.randint(placed, (1,), generator=gen))
                d = torch.randn(3, generator=gen)
                d = d / d.norm().clamp(min=1e-9)
                cand = pos[a] + (1.0 + 0.55 * float(torch.rand(1, generator=gen))) * d
                if float((pos[:placed] - cand).norm(dim=1).min()) >= 0.95:
                    break
        pos[placed] = cand
        free[a] -= 1
        free[placed] -= 1
        if first_bond[a] is None:
            first_bond[a] = d.clone()
        first_bond[placed] = -d
        placed += 1
    return pos, z


def synth_qm9_list(seed: int, n_mol: int, shifts: Optional[List[float]] = None, r_max: Optional[float] = 4.0, bonds: str = "uniform"):
    """``bonds``: "uniform" (neighbour distances U(1.0, 1.55) A in random directions) or "clustered" (element-pair
    bond lengths +- 0.01 A and tetrahedral angles: the distance distribution of real molecules, see above)."""
    if bonds not in ("uniform", "clustered"):
        raise ValueError(bonds)
    gen = torch.Generator(device="cpu").manual_seed(seed)
    attrs = {"pos": ("node", "1x1o"), "species": ("node", "1x0e"), "total_energy": ("graph", "1x0e")}
    out = []
    for _ in range(n_mol):
        n = int(min(29, max(3, round(18 + 4.5 * float(torch.randn(1, generator=gen))))))
        z = _SPECIES[torch.multinomial(_PROBS, n, replacement=True, generator=gen)]
        if bonds == "clustered":
            pos, z = _grow_molecule_clustered(z, gen)
            pos = pos.float()
        else:
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


def synth_qm9(seed: int, n_mol: int, shifts: Optional[List[float]] = None, r_max: float = 4.0, bonds: str = "uniform") -> Batch:
    """A Batch with pos, species, total_energy, edge_index (cutoff r_max), _n_nodes, _n_edges."""
    lst, attrs = synth_qm9_list(seed, n_mol, shifts, r_max, bonds)
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
