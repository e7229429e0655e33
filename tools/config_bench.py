#!/usr/bin/env python3
"""Training-step time of the other BASELINE.json configurations on one MI355X (synthetic inputs, random-init weights):
  3. config_energy_force, 64 molecules, loss on energies AND forces (the double backward through every kernel)
  4. config_diffusion score net, 128 fully connected molecules, VP-SDE denoising loss
  5. config_diffusion_CA protein score net, 4 x 384 residues (edges rebuilt by the model's own edge_index layer)
python tools/config_bench.py [steps] [--graph]     (--graph: also replay configs 3 and 4 as one HIP graph per step)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "equivariant-nn-zoo_amd")):
    sys.path.insert(0, p)
import torch
from e3_layers_amd.backend.graph import build_topology
from e3_layers_amd.configs import config_diffusion, config_diffusion_CA, config_energy_force
from e3_layers_amd.data.synthetic import synth_protein, synth_qm9, synth_qm9_diffusion
from e3_layers_amd.run.optim import FusedAdamEMA
from e3_layers_amd.run.parallel import backward_parameters
from e3_layers_amd.run.sde_utils import VPSDE, sde_loss
from e3_layers_amd.utils import build, countParameters

dev = torch.device("cuda:0")
GRAPH = "--graph" in sys.argv
_pos = [a for a in sys.argv[1:] if not a.startswith("--")]
steps = int(_pos[0]) if _pos else 10
ONLY = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--only=")]     # e.g. --only=3


def run(name, model, step_fn, n_units, unit, graph=False):
    if ONLY and int(name.split()[0].split("#")[-1]) not in ONLY:
        return
    opt = FusedAdamEMA(model.parameters(), lr=1e-3, max_grad_norm=1.0)
    if os.environ.get("E3K_CB_SINK", "1") != "0":
        opt.grads.enable_direct_accumulation()      # weight-gradient kernels add straight into the flat gradient buffer
    def one():
        opt.zero_grad()
        loss = step_fn()
        if os.environ.get("E3K_CB_PARAMS_ONLY", "1") != "0":
            backward_parameters(loss, opt.params)
        else:
            loss.backward()
        opt.step()
        return loss
    for _ in range(3):
        loss = one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = one()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{name:34s} {dt * 1e3:8.2f} ms/step  {n_units / dt:9.0f} {unit}/s  params {countParameters(model)}  loss {float(loss):.4g}", flush=True)
    if not (graph and GRAPH):
        opt.grads.disable_direct_accumulation()
    if graph and GRAPH:
        # whole step (forward, loss, backward or double backward, clip + Adam) as one HIP graph
        del loss
        from e3_layers_amd.run.graph_step import CapturedStep
        g = CapturedStep(one, warmup=3)
        loss = g.out
        for _ in range(3):
            g()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            g()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        opt.grads.disable_direct_accumulation()
        print(f"{'  ... replayed as one HIP graph':34s} {dt * 1e3:8.2f} ms/step  {n_units / dt:9.0f} {unit}/s  loss {float(loss):.4g}", flush=True)


torch.manual_seed(0)
# --- 3. energy + force training
cfg = config_energy_force.get_config()
model = build(cfg.model_config).to(dev).train()
batch = synth_qm9(2000, 64, r_max=5.0).to(dev)
batch.update(build_topology(batch["edge_index"], batch["pos"].shape[0]).as_dict())
f_t = torch.randn_like(batch["pos"]); e_t = torch.randn(64, 1, device=dev)
def ef_step():
    out = model(batch.view())
    return 1e3 * ((out["total_energy"] - e_t) ** 2).mean() + 3e4 * ((out["forces"] - f_t) ** 2).mean()
run(f"#3 config_energy_force B=64 (N={batch['pos'].shape[0]}, E={batch['edge_index'].shape[1]})", model, ef_step, 64, "molecules", graph=True)

# --- 4. small-molecule diffusion
cfg = config_diffusion.get_config()
model = build(cfg.model_config).to(dev).train()
batch4 = synth_qm9_diffusion(1, 128).to(dev)
batch4.update(build_topology(batch4["edge_index"], batch4["pos"].shape[0]).as_dict())
sde = VPSDE({"pos": 3})
run(f"#4 config_diffusion B=128 (N={batch4['pos'].shape[0]}, E={batch4['edge_index'].shape[1]})", model,
    lambda: sde_loss(sde, model, batch4)[0], 128, "molecules", graph=True)

# --- 5. protein C-alpha diffusion
cfg = config_diffusion_CA.get_config()
model = build(cfg.model_config).to(dev).train()
batch5 = synth_protein(1, 4, n_res=384).to(dev)
sde5 = VPSDE({"CA": 3})
run(f"#5 config_diffusion_CA 4x384 residues", model, lambda: sde_loss(sde5, model, batch5)[0], 4, "proteins")
