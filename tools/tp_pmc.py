#!/usr/bin/env python3
"""Runs the three TP kernels of config_energy layer 3 (l_max 2, 256 molecules) a few times each, for rocprofv3 --pmc passes
(VALUBusy, SQ_INSTS_VALU, ...):   rocprofv3 --pmc VALUBusy --kernel-trace --output-format csv -d out -- python3 tools/tp_pmc.py"""
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
torch.manual_seed(0)
model = build(config_energy.get_config(l_max=2).model_config).to(dev)
batch = synth_qm9(1000, 256).to(dev)
n, e = batch["pos"].shape[0], batch["edge_index"].shape[1]
topo = build_topology(batch["edge_index"], n)
plan = model.layer3.conv.tp.tp.plan
x = torch.randn(n, plan.d_in, device=dev)
sh = torch.randn(e, plan.d_sh, device=dev)
w = torch.randn(e, plan.w_numel, device=dev)
g = torch.randn(n, plan.d_mid, device=dev)
for _ in range(5):
    ops._tp_fwd_raw(x, sh, w, topo, plan)
    ops._tp_bwd_x_raw(sh, w, g, topo, plan)
    ops._tp_bwd_w_raw(x, sh, w, g, topo, plan, False, True)
torch.cuda.synchronize()
