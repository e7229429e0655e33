#!/usr/bin/env python3
"""Host-side cost of one config_energy training step split into forward / backward / optimizer enqueue time
(no device sync inside the loop: the GPU runs behind).  python tools/host_split.py [B]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "equivariant-nn-zoo_amd")):
    sys.path.insert(0, p)
import torch
from e3_layers_amd.backend.graph import build_topology
from e3_layers_amd.configs import config_energy
from e3_layers_amd.data.synthetic import synth_qm9
from e3_layers_amd.run.optim import FusedAdamEMA
from e3_layers_amd.utils import build
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = build(config_energy.get_config(l_max=2).model_config).to(dev)
opt = FusedAdamEMA(model.parameters(), lr=1e-2)
opt.grads.enable_direct_accumulation()
batch = synth_qm9(1000, B, config_energy.QM9_SHIFTS).to(dev)
batch.update(build_topology(batch["edge_index"], batch["pos"].shape[0]).as_dict())
target = batch["total_energy"]
acc = [0.0, 0.0, 0.0]
def step(rec):
    t0 = time.perf_counter()
    out = model(batch.view())
    loss = 1e3 * torch.nn.functional.mse_loss(out["total_energy"], target)
    t1 = time.perf_counter()
    opt.zero_grad(); loss.backward()
    t2 = time.perf_counter()
    opt.step()
    t3 = time.perf_counter()
    if rec:
        acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2
for _ in range(5): step(False)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n): step(True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"B={B}: forward {1e3*acc[0]/n:.2f} ms, backward {1e3*acc[1]/n:.2f} ms, optimizer {1e3*acc[2]/n:.2f} ms host per step; "
      f"enqueue {1e3*(t1-t0)/n:.2f}, wall {1e3*(t2-t0)/n:.2f} ms/step")
