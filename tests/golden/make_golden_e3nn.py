#!/usr/bin/env python3
"""Pins the oracle to the REFERENCE ITSELF -- the day an environment can import it.

    python tests/golden/make_golden_e3nn.py          (in the build container, from the repo root)

Needs ``e3nn`` (the reference pins 0.4.4: /root/reference/requirements.txt:27), ``torch_runstats`` (:145) and
``ml_collections`` importable, and the reference tree at /root/reference (or $E3K_REFERENCE).  None of them exists in this
image, so the script cannot run here and ``tests/golden/energy_small_e3nn.npz`` is absent:
``tests/test_golden_e3nn.py`` SKIPS, and DESIGN.md section 3 says "parity unpinned".  The moment the imports work, this
script removes the last manual step: it builds the reduced config_energy network from the reference's OWN
``featureModel`` / ``addEnergyOutput`` / ``build`` (/root/reference/e3_layers/configs/layer_configs.py:10-147,
utils/utils.py:99-116), runs it in float64 on a fixed 3-molecule batch and writes

  inputs                   pos, species, edge_index, _n_nodes, _n_edges, target
  param::<name>            the reference network's state_dict (its own names)
  out_<key>                edge_spherical, edge_radial, per-layer node features (layer0..2), energy, total_energy
  loss, grad::<name>       1e3 * MSE and its gradient w.r.t. every parameter
  w3j::l1_l2_l3            e3nn's stored Wigner-3j tables for l1 <= 3, l2 <= 2, l3 <= 3   } where a mismatch would come from:
  sh::vectors / sh::values e3nn's spherical harmonics (component, normalized) on 16 vectors } per-path signs, SH basis / phase,
  irreps::<layer>          str() of every layer's irreps_in / irreps_out (Irreps.sort tie order)

The file is DATA (arrays and strings); nothing of the reference's source travels.  It is read by tests/test_golden_e3nn.py
only -- never by the product path -- and it is generated HERE: /root/reference does not exist on the GPU box.

What the test then checks, in this order, so that a failure names its cause ("the only place a faithful
restatement could legitimately differ"): (1) oracle.wigner_3j == w3j::* entry by entry (a per-(l1, l2, l3) overall sign is the
expected failure mode: it flips the sign of that path's weights and nothing else); (2) oracle SH == sh::values; (3) the
irreps strings; (4) outputs and gradients of the oracle loaded with param::* at 1e-10 (float64); (5, GPU) the HIP path at
1e-5 / 5e-5.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = os.environ.get("E3K_REFERENCE", "/root/reference")
OUT = os.path.join(HERE, "energy_small_e3nn.npz")

SMALL = dict(n_dim=8, l_max=2, edge_spherical="1x0e+1x1o+1x2e", node_attrs="6x0e", edge_radial="8x0e", num_types=10,
             num_layers=3, r_max=4.0)      # == make_golden.small_tree()


def main() -> int:
    try:
        import e3nn
        from e3nn import o3
    except ImportError as exc:
        print(f"make_golden_e3nn: {exc} -- nothing written (this image carries no e3nn; see the module docstring)")
        return 2
    if not os.path.isdir(os.path.join(REFERENCE, "e3_layers")):
        print(f"make_golden_e3nn: no reference tree at {REFERENCE} -- nothing written")
        return 2
    sys.path.insert(0, REFERENCE)
    for p in (ROOT, os.path.join(ROOT, "equivariant-nn-zoo_amd")):
        sys.path.insert(0, p)
    try:
        from e3_layers.configs.layer_configs import addEnergyOutput, featureModel      # the REFERENCE's
        from e3_layers.data import Batch
        from e3_layers.utils import build
    except ImportError as exc:
        print(f"make_golden_e3nn: the reference package does not import ({exc}) -- nothing written")
        return 2
    from e3_layers_amd.data.synthetic import synth_qm9_list      # only the INPUT generator is ours (plain tensors)

    torch.set_default_dtype(torch.float64)
    torch.manual_seed(1234)
    tree = addEnergyOutput(featureModel(**SMALL), None)
    net = build(tree).double()

    lst, _ = synth_qm9_list(7, 3, None, r_max=SMALL["r_max"])
    pos = torch.cat([s["pos"] for s in lst]).double()
    species = torch.cat([s["species"] for s in lst])
    n_nodes = torch.cat([s["_n_nodes"] for s in lst])
    ei, off = [], 0
    for s in lst:
        ei.append(s["edge_index"] + off)
        off += s["pos"].shape[0]
    edge_index = torch.cat(ei, dim=1)
    n_edges = torch.tensor([[s["edge_index"].shape[1]] for s in lst])
    target = torch.tensor([[0.3], [-0.2], [0.1]], dtype=torch.float64)
    attrs = {"pos": ("node", "1x1o"), "species": ("node", "1x0e"), "_n_nodes": ("graph", "1x0e"), "_n_edges": ("graph", "1x0e"),
             "edge_index": ("edge", "2x0e")}
    batch = Batch(attrs, pos=pos, species=species, _n_nodes=n_nodes, _n_edges=n_edges, edge_index=edge_index)

    per_layer = {}
    hooks = []
    for name, module in net.named_modules():
        if name.split(".")[-1] in ("layer0", "layer1", "layer2"):
            key = name.split(".")[-1]
            hooks.append(module.register_forward_hook(
                lambda m, i, o, key=key: per_layer.__setitem__(key, o[0]["output_features"].detach().numpy().copy())))
    out = net(batch)
    for h in hooks:
        h.remove()
    loss = 1e3 * torch.nn.functional.mse_loss(out["total_energy"], target)
    names = [n for n, p in net.named_parameters() if p.requires_grad]
    grads = torch.autograd.grad(loss, [dict(net.named_parameters())[n] for n in names], allow_unused=True)

    arrays = {"pos": pos.numpy(), "species": species.numpy(), "n_nodes": n_nodes.numpy(), "n_edges": n_edges.numpy(),
              "edge_index": edge_index.numpy(), "target": target.numpy(), "loss": np.array(float(loss.detach())),
              "e3nn_version": np.array(e3nn.__version__), "torch_version": np.array(torch.__version__)}
    for key in ("edge_spherical", "edge_radial", "energy", "total_energy"):
        arrays["out_" + key] = out[key].detach().numpy()
    for k, v in per_layer.items():
        arrays["out_" + k] = v
    for n, g in zip(names, grads):
        if g is not None:
            arrays["grad::" + n] = g.numpy()
    for k, v in net.state_dict().items():
        arrays["param::" + k] = v.detach().numpy()
    # where a discrepancy would come from
    for l1 in range(4):
        for l2 in range(3):
            for l3 in range(4):
                if abs(l1 - l2) <= l3 <= l1 + l2:
                    arrays[f"w3j::{l1}_{l2}_{l3}"] = o3.wigner_3j(l1, l2, l3).double().numpy()
    vecs = torch.randn(16, 3, generator=torch.Generator().manual_seed(5), dtype=torch.float64)
    arrays["sh::vectors"] = vecs.numpy()
    arrays["sh::values"] = o3.spherical_harmonics([0, 1, 2, 3], vecs, normalize=True, normalization="component").numpy()
    for name, module in net.named_modules():
        if hasattr(module, "irreps_in") and hasattr(module, "irreps_out") and name:
            arrays["irreps::" + name] = np.array(repr(({k: str(v) for k, v in dict(module.irreps_in).items()},
                                                       {k: str(v) for k, v in dict(module.irreps_out).items()})))
    np.savez_compressed(OUT, **arrays)
    print("wrote", OUT, f"({len(arrays)} arrays; e3nn {e3nn.__version__})")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
