#!/usr/bin/env python3
"""Micro-benchmark of the fused TP kernels on config_energy layer 3 (l_max from argv, B=256)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "equivariant-nn-zoo_amd")):
    sys.path.insert(0, p)
import torch
from e3_layers_amd.backend import ops
from e3_layers_amd.backend.graph import build_topology
from e3_layers_amd.configs import config_energy
from e3_layers_amd.data.synthetic import synth_qm9
from e3_layers_amd.utils import build
dev = torch.device("cuda:0")
lmax = int(sys.argv[1]) if len(sys.argv) > 1 else 2
torch.manual_seed(0)
model = build(config_energy.get_config(l_max=lmax).model_config).to(dev)
batch = synth_qm9(1000, 256).to(dev)
n, e = batch["pos"].shape[0], batch["edge_index"].shape[1]
topo = build_topology(batch["edge_index"], n)
plan = model.layer3.conv.tp.tp.plan
x = torch.randn(n, plan.d_in, device=dev, requires_grad=True)
sh = torch.randn(e, plan.d_sh, device=dev)
w = torch.randn(e, plan.w_numel, device=dev, requires_grad=True)
alg = e * (4 * plan.d_in + 4 * plan.d_sh + 4 * plan.w_numel + 16) + n * 4 * plan.d_mid
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    t.record(); torch.cuda.synchronize()
    return s.elapsed_time(t) * 1e3 / reps
with torch.no_grad():
    us = timeit(lambda: ops.tp_uvu_scatter(x, sh, w, topo, plan))
print(f"lib={os.environ.get('E3K_LIB','default').split('/')[-1]:24s} l_max={lmax} tp_fwd {us:7.1f} us  {alg/us/1e3:7.1f} GB/s algorithmic  frac {alg/us/1e3/8000:.3f}")
y = ops.tp_uvu_scatter(x, sh, w, topo, plan)
g = torch.randn_like(y)
us_b = timeit(lambda: torch.autograd.grad(y, [x, w], g, retain_graph=True))
print(f"   bwd (bwd_w + bwd_x) {us_b:7.1f} us")
sh2 = sh.clone().requires_grad_(True)
y2 = ops.tp_uvu_scatter(x, sh2, w, topo, plan)
us_c = timeit(lambda: torch.autograd.grad(y2, [x, sh2, w], g, retain_graph=True))
print(f"   bwd with grad_sh (bwd_w<sh> + bwd_x) {us_c:7.1f} us")
