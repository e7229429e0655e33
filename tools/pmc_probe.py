#!/usr/bin/env python3
"""PMC probe for the fused TP+reduce forward kernel (run under ``rocprofv3 --pmc ...``).

Launches, in order:
  1. a calibration stream with the SAME access shape as the TP kernel's weight stream (one dword
     per lane, 256 contiguous bytes per wave-instruction): e3k_act_fwd(identity) over 256 Mi floats
     = 1 GiB read + 1 GiB written — a known byte count to calibrate FETCH_SIZE / WRITE_SIZE with
     (MI355X_MICROARCH.md §HBM: access widths other than 16 B/lane are uncalibrated);
  2. the layer-3 convolution's e3k_tp_fwd of config_energy (l_max from argv, 256 molecules),
     5 launches, each preceded by a 1 GiB memset so that the weight stream comes from HBM.
Prints the algorithmic bytes of (2) for comparison with the counters.
"""
import json
import os
import sys

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
x = torch.randn(256 * 1024 * 1024, device=dev)
for _ in range(3):
    y = ops.activation(x, "identity", 1.0)
torch.cuda.synchronize()
del x, y

torch.manual_seed(0)
model = build(config_energy.get_config(l_max=lmax).model_config).to(dev)
batch = synth_qm9(1000, 256, config_energy.QM9_SHIFTS).to(dev)
n, e = batch["pos"].shape[0], batch["edge_index"].shape[1]
topo = build_topology(batch["edge_index"], n)
plan = model.layer3.conv.tp.tp.plan
xin = torch.randn(n, plan.d_in, device=dev)
sh = torch.randn(e, plan.d_sh, device=dev)
w = torch.randn(e, plan.w_numel, device=dev)
flush = torch.empty(256 * 1024 * 1024, device=dev)
with torch.no_grad():
    for _ in range(5):
        flush.zero_()
        ops.tp_uvu_scatter(xin, sh, w, topo, plan)
torch.cuda.synchronize()
alg = e * (4 * plan.d_in + 4 * plan.d_sh + 4 * plan.w_numel + 16) + n * 4 * plan.d_mid
print(json.dumps({"l_max": lmax, "N": n, "E": e, "d_in": plan.d_in, "W": plan.w_numel, "d_mid": plan.d_mid,
                  "algorithmic_bytes_variant_A": alg, "w_bytes": e * 4 * plan.w_numel,
                  "out_bytes": n * 4 * plan.d_mid, "x_unique_bytes": n * 4 * plan.d_in,
                  "calibration_read_bytes": 2 ** 30, "calibration_write_bytes": 2 ** 30}))
