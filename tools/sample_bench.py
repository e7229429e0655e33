#!/usr/bin/env python3
"""Reverse-diffusion sampling throughput: config_diffusion score network, B molecules, fully
connected graphs; first `n_iter` steps of the N=1000 predictor-corrector schedule, eager launches vs one HIP graph
per step.  Usage: python tools/sample_bench.py [B] [n_iter]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "equivariant-nn-zoo_amd")):
    sys.path.insert(0, p)
import torch
from e3_layers_amd.configs import config_diffusion
from e3_layers_amd.data.synthetic import synth_qm9_diffusion
from e3_layers_amd.run.sde_sampling import EulerMaruyamaPredictor, LangevinCorrector, get_pc_sampler
from e3_layers_amd.run.sde_utils import VPSDE
from e3_layers_amd.utils import build

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n_iter = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = build(config_diffusion.get_config().model_config).to(dev).eval()
batch = synth_qm9_diffusion(1, B).to(dev)
print(f"B={B} N={batch['pos'].shape[0]} E={batch['edge_index'].shape[1]}")
for graph in (False, True):
    sde = VPSDE({"pos": 3}, N=1000)
    sampler = get_pc_sampler(sde, EulerMaruyamaPredictor, LangevinCorrector, snr=0.16, static_edges=True, graph=graph,
                             n_iter=n_iter)
    sampler(model, batch)          # warm-up (plans, allocator)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out, nfe = sampler(model, batch)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"graph={graph}: {dt / n_iter * 1e3:.2f} ms per reverse step ({nfe} network evaluations, incl. capture when graph), "
          f"{B * n_iter / dt:.0f} molecule-steps/s; finite={bool(torch.isfinite(out['pos']).all())}")
