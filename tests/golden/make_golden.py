#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ (run from the repo root).

The reference cannot be executed here (e3nn 0.4.4 / torch_runstats are not installable, SURVEY.md
§0 F3) and ships no vectors of its own, so these fixtures are produced by the ORACLE
(oracle/e3ref.py, float64) — they pin the oracle against silent drift and give the HIP path fixed
vectors to hit; they are *not* outputs of the reference ("parity unpinned").  If an
environment with e3nn 0.4.4 ever becomes available, regenerate them there to pin the signs.

Fixtures:
  edge_index_8mol.npz   positions of an 8-molecule synthetic batch and the integer edge list the
                        reference's computeEdgeIndex ordering contract implies (r_max = 4.0)
  energy_small.npz      3-molecule batch, config_energy-shaped model (n_dim 8, l_max 2, 3 layers):
                        parameters, inputs, float64 outputs (per-layer node features, energies),
                        and the gradient of 1e3*MSE wrt three parameter tensors
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "equivariant-nn-zoo_amd")):
    sys.path.insert(0, p)

from e3_layers_amd.configs.layer_configs import addEnergyOutput, featureModel  # noqa: E402
from e3_layers_amd.data.synthetic import synth_qm9_list  # noqa: E402
from oracle import e3ref  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def small_tree():
    return addEnergyOutput(featureModel(n_dim=8, l_max=2, edge_spherical="1x0e+1x1o+1x2e", node_attrs="6x0e",
                                        edge_radial="8x0e", num_types=10, num_layers=3, r_max=4.0), None)


def main():
    # ---- edge index ---------------------------------------------------------------------------
    lst, _ = synth_qm9_list(42, 8, None, r_max=None)
    pos = torch.cat([s["pos"] for s in lst])
    n_nodes = torch.cat([s["_n_nodes"] for s in lst])
    data = {"pos": pos, "_n_nodes": n_nodes}
    out, _ = e3ref.compute_edge_index(data, {}, r_max=4.0)
    np.savez_compressed(os.path.join(HERE, "edge_index_8mol.npz"), pos=pos.numpy(), n_nodes=n_nodes.numpy(),
                        edge_index=out["edge_index"].numpy(), n_edges=data["_n_edges"].numpy(), r_max=4.0)

    # ---- small energy model -------------------------------------------------------------------
    torch.manual_seed(1234)
    net = e3ref.build(small_tree()).double()
    lst, attrs = synth_qm9_list(7, 3, None, r_max=4.0)
    pos = torch.cat([s["pos"] for s in lst]).double()
    species = torch.cat([s["species"] for s in lst])
    n_nodes = torch.cat([s["_n_nodes"] for s in lst])
    target = torch.tensor([[0.3], [-0.2], [0.1]], dtype=torch.float64)
    ei, off = [], 0
    for s in lst:
        ei.append(s["edge_index"] + off)
        off += s["pos"].shape[0]
    edge_index = torch.cat(ei, dim=1)
    n_edges = torch.tensor([[s["edge_index"].shape[1]] for s in lst])
    data = {"pos": pos, "species": species, "_n_nodes": n_nodes, "_n_edges": n_edges, "edge_index": edge_index}
    attrs = {"pos": ("node", "1x1o"), "species": ("node", "1x0e"), "_n_nodes": ("graph", "1x0e"), "_n_edges": ("graph", "1x0e")}
    # per-layer outputs
    d, a = dict(data), dict(attrs)
    e3ref.add_segments(d)
    per_layer = {}
    for key, step in net.steps:
        if isinstance(step, e3ref.OModule):
            o, oa = step(e3ref._remap(d, step.in_map), e3ref._remap(a, step.in_map))
            o, oa = e3ref._remap(o, step.out_map), e3ref._remap(oa, step.out_map)
        else:
            o, oa = step(d, a)
        d.update(o)
        a.update(oa)
        if key.startswith("layer"):
            per_layer[key] = d["node_features"].detach().numpy().copy()
    loss = 1e3 * torch.nn.functional.mse_loss(d["total_energy"], target)
    names = ["mods.layer1.conv.fc.layer3.weight", "mods.layer2.conv.sc.weight", "mods.radial_basis.basis.bessel_weights"]
    params = dict(net.named_parameters())
    grads = torch.autograd.grad(loss, [params[n] for n in names])
    arrays = {"pos": pos.numpy(), "species": species.numpy(), "n_nodes": n_nodes.numpy(), "n_edges": n_edges.numpy(),
              "edge_index": edge_index.numpy(), "target": target.numpy(),
              "out_edge_spherical": d["edge_spherical"].detach().numpy(), "out_edge_radial": d["edge_radial"].detach().numpy(),
              "out_energy": d["energy"].detach().numpy(), "out_total_energy": d["total_energy"].detach().numpy(),
              "loss": np.array(float(loss.detach()))}
    for k, v in per_layer.items():
        arrays["out_" + k] = v
    for n, g in zip(names, grads):
        arrays["grad::" + n] = g.numpy()
    for k, v in net.state_dict().items():
        arrays["param::" + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "energy_small.npz"), **arrays)
    print("wrote", os.listdir(HERE))


if __name__ == "__main__":
    main()
