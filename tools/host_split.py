#!/usr/bin/env python3
"""Host-side cost of one config_energy training step, split into batch prep (fresh device copy of a resident batch) /
forward / backward / optimizer enqueue time.  With a small batch (python tools/host_split.py 16) the GPU work is
negligible and wall = host: the per-step Python + launch cost, which does not depend on the batch size.
    python tools/host_split.py [B] [--same]      (--same: re-step one batch object, topology prebuilt)"""
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
args = [a for a in sys.argv[1:] if not a.startswith("--")]
B = int(args[0]) if args else 256
same = "--same" in sys.argv
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = build(config_energy.get_config(l_max=2).model_config).to(dev)
opt = FusedAdamEMA(model.parameters(), lr=1e-2, ema_decay=0.99)
opt.grads.enable_direct_accumulation()
resident = [synth_qm9(1000 + k, B, config_energy.QM9_SHIFTS).to(dev) for k in range(4)]
if same:
    resident[0].update(build_topology(resident[0]["edge_index"], resident[0]["pos"].shape[0]).as_dict())
if "--layer-timing" in sys.argv:     # (run with E3K_HOST_TIMING=1) host time inside the layer functions: total vs the C call
    from e3_layers_amd.backend import conv_native as _cn
    _f, _b = _cn.NativeConvBlockFn.forward, _cn.NativeConvBlockFn.backward
    def _fw(ctx, *a):
        t = time.perf_counter(); r = _f(ctx, *a); _cn.HOST_TIMING[0] += time.perf_counter() - t; _cn.HOST_TIMING[4] += 1; return r
    def _bw(ctx, *a):
        t = time.perf_counter(); r = _b(ctx, *a); _cn.HOST_TIMING[2] += time.perf_counter() - t; return r
    _cn.NativeConvBlockFn.forward, _cn.NativeConvBlockFn.backward = staticmethod(_fw), staticmethod(_bw)
acc = [0.0, 0.0, 0.0, 0.0]
count = [0]
def step(rec):
    t0 = time.perf_counter()
    batch = resident[0].view() if same else resident[count[0] % 4].clone()
    count[0] += 1
    target = batch["total_energy"]
    ta = time.perf_counter()
    out = model(batch)
    loss = 1e3 * torch.nn.functional.mse_loss(out["total_energy"], target)
    t1 = time.perf_counter()
    opt.zero_grad(); loss.backward()
    t2 = time.perf_counter()
    opt.step()
    t3 = time.perf_counter()
    if rec:
        acc[0] += ta - t0; acc[1] += t1 - ta; acc[2] += t2 - t1; acc[3] += t3 - t2
for _ in range(5): step(False)
torch.cuda.synchronize()
if "--layer-timing" in sys.argv:
    _cn.HOST_TIMING[:] = [0.0, 0.0, 0.0, 0.0, 0]
n = 20
t0 = time.perf_counter()
for _ in range(n): step(True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"B={B} {'same batch' if same else 'fresh batch'}: prep {1e3*acc[0]/n:.2f} ms, forward {1e3*acc[1]/n:.2f} ms, backward {1e3*acc[2]/n:.2f} ms, "
      f"optimizer {1e3*acc[3]/n:.2f} ms host per step; enqueue {1e3*(t1-t0)/n:.2f}, wall {1e3*(t2-t0)/n:.2f} ms/step")
if "--profile" in sys.argv:      # where the host time goes: cProfile over 30 steps (adds ~2x overhead; read the ranking, not the totals)
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(30): step(False)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(40)
    st.sort_stats("cumtime").print_stats(45)
if "--torch-profile" in sys.argv:      # host time per op / autograd node (forward and backward threads): torch.profiler, CPU side only
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        for _ in range(20): step(False)
    torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=45, max_name_column_width=60))

if "--layer-timing" in sys.argv:
    from e3_layers_amd.backend import conv_native as _cn
    ht = _cn.HOST_TIMING
    c = max(ht[4], 1)
    print(f"layer functions, host us per call over {ht[4]} calls: forward {1e6*ht[0]/c:.1f} (C call {1e6*ht[1]/c:.1f}), "
          f"backward {1e6*ht[2]/c:.1f} (C call {1e6*ht[3]/c:.1f})")
