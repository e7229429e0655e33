#!/usr/bin/env python3
"""How many edges config_diffusion_CA's in-model radius graph yields per step as a function of the diffusion time the batch was
noised to (4 x 384 synthetic residues): the spread that a fixed edge capacity for HIP-graph replay would have to cover.
python tools/ca_edge_counts.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "equivariant-nn-zoo_amd")):
    sys.path.insert(0, p)
import torch
from e3_layers_amd.configs import config_diffusion_CA
from e3_layers_amd.data.synthetic import synth_protein
from e3_layers_amd.run.sde_utils import VPSDE, sde_perturb
from e3_layers_amd.utils import build

dev = torch.device("cuda", 0)
cfg = config_diffusion_CA.get_config()
model = build(cfg.model_config).to(dev)
sde = VPSDE({"CA": 3})
batch = synth_protein(1, 4, n_res=384).to(dev)
torch.manual_seed(0)
counts = []
for step in range(40):
    pert, misc = sde_perturb(sde, batch)
    model.prepare(pert)            # runs the leading data-only layers (computeEdgeIndex) in place
    e = int(pert["edge_index"].shape[1]) if "edge_index" in pert else None
    t = float(pert["t"].reshape(-1).float().mean()) if "t" in pert else float("nan")
    counts.append((t, e))
counts.sort()
print("mean t of the batch, edges:")
for t, e in counts[::4]:
    print(f"  t {t:.3f}  edges {e}")
es = [e for _, e in counts if e is not None]
print(f"edges over 40 steps: min {min(es)} max {max(es)} mean {sum(es) / len(es):.0f} (max / min = {max(es) / min(es):.2f})")
