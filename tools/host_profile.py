#!/usr/bin/env python3
"""cProfile of the host side of training steps (config_energy l_max=2, fresh batch per step, batch size from argv):
which Python / torch calls the per-step host time goes to.   python tools/host_profile.py [B]"""
import cProfile, pstats, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "equivariant-nn-zoo_amd")):
    sys.path.insert(0, p)
import torch
from e3_layers_amd.configs import config_energy
from e3_layers_amd.data.synthetic import synth_qm9
from e3_layers_amd.run.optim import FusedAdamEMA
from e3_layers_amd.utils import build
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
cfg = config_energy.get_config(l_max=2)
torch.manual_seed(0)
model = build(cfg.model_config).to(dev)
opt = FusedAdamEMA(model.parameters(), lr=1e-2, ema_decay=0.99)
flat = opt.grads
flat.enable_direct_accumulation()
resident = [synth_qm9(1000 + k, B, config_energy.QM9_SHIFTS).to(dev) for k in range(4)]
count = [0]
def step():
    batch = resident[count[0] % 4].clone()
    count[0] += 1
    target = batch["total_energy"]
    out = model(batch)
    loss = 1e3 * torch.nn.functional.mse_loss(out["total_energy"], target)
    flat.zero(); loss.backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
t1 = time.perf_counter()   # host time to enqueue (no sync)
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"B={B}: host enqueue {1e3*(t1-t0)/10:.2f} ms/step, wall {1e3*(t2-t0)/10:.2f} ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
