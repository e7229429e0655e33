#!/usr/bin/env python3
"""cProfile of the host side of one training step (config_energy l_max=2, batch from argv)."""
import cProfile, pstats, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "equivariant-nn-zoo_amd")):
    sys.path.insert(0, p)
import torch
from e3_layers_amd.backend.graph import build_topology
from e3_layers_amd.configs import config_energy
from e3_layers_amd.data.synthetic import synth_qm9
from e3_layers_amd.run.optim import FusedAdamEMA
from e3_layers_amd.utils import build
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
cfg = config_energy.get_config(l_max=2)
torch.manual_seed(0)
model = build(cfg.model_config).to(dev)
opt = FusedAdamEMA(model.parameters(), lr=1e-2)
flat = opt.grads
flat.enable_direct_accumulation()
batch = synth_qm9(1000, B, config_energy.QM9_SHIFTS).to(dev)
batch.update(build_topology(batch["edge_index"], batch["pos"].shape[0]).as_dict())
target = batch["total_energy"]
def step():
    out = model(batch.view())
    loss = 1e3 * torch.nn.functional.mse_loss(out["total_energy"], target)
    flat.zero(); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
t1 = time.perf_counter()   # host time to enqueue (no sync)
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"B={B}: host enqueue {1e3*(t1-t0)/10:.2f} ms/step, wall {1e3*(t2-t0)/10:.2f} ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
