#!/usr/bin/env python3
"""Gate forward / backward of layer 3 (config_energy, l_max 2) in isolation, HIP-event timed: python tools/gate_bench.py [molecules]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "equivariant-nn-zoo_amd"))
import torch
from e3_layers_amd.backend import ops
from e3_layers_amd.configs import config_energy
from e3_layers_amd.utils import build
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = 18 * B
model = build(config_energy.get_config(l_max=2).model_config).to(dev)
gate = model.layer3.equivariant_nonlin
spec = gate._spec
x = torch.randn(n, spec.in_dim, device=dev, requires_grad=True)
def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps
y = gate(x, out_cf=True)
gy = torch.randn_like(y)
xd = x.detach()
fwd = timeit(lambda: ops._gate_fwd_raw(xd, spec, True))
both = fwd + timeit(lambda: ops._gate_bwd_raw(xd, gy, spec, True))
mb = 4e-6 * n * (spec.in_dim + spec.out_dim)   # MB
print(f"gate rows {n} in {spec.in_dim} out {spec.out_dim}: fwd {fwd:.1f} us ({mb / fwd:.2f} TB/s), fwd+bwd {both:.1f} us, bwd ~{both - fwd:.1f} us "
      f"({4e-6 * n * (2 * spec.in_dim + spec.out_dim) / (both - fwd):.2f} TB/s)")
