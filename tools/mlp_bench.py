#!/usr/bin/env python3
"""Fused radial-MLP hidden chain: time of the forward and backward launches (E edges, 8 -> 64 -> 64 -> 64)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "equivariant-nn-zoo_amd")):
    sys.path.insert(0, p)
import torch
from e3_layers_amd.backend import ops
dev = torch.device("cuda:0")
E = int(sys.argv[1]) if len(sys.argv) > 1 else 69484
ACT = sys.argv[2] if len(sys.argv) > 2 else "ssp"
torch.manual_seed(0)
x = torch.randn(E, 8, device=dev)
ws = [torch.randn(8, 64, device=dev, requires_grad=True)] + [torch.randn(64, 64, device=dev, requires_grad=True) for _ in range(2)]
al = [8 ** -0.5, 0.125, 0.125]
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps
with torch.no_grad():
    t_inf = timeit(lambda: ops.mlp_hidden(x, ws, al, ACT, 1.8782))
t_fwd = timeit(lambda: ops.mlp_hidden(x, ws, al, ACT, 1.8782))
y = ops.mlp_hidden(x, ws, al, ACT, 1.8782)
g = torch.randn_like(y)
t_bwd = timeit(lambda: torch.autograd.grad(y, ws, g, retain_graph=True))
print(f"E={E} env={ {k: v for k, v in os.environ.items() if k.startswith('E3K_')} }: inference fwd {t_inf:.1f} us, training fwd {t_fwd:.1f} us, bwd {t_bwd:.1f} us")
