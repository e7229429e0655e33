#!/usr/bin/env python3
"""Accuracy and speed of the GEMM entry point on one shape (run once per E3K_GEMM_X3 setting):
python tools/x3_check.py K N [M]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "equivariant-nn-zoo_amd")):
    sys.path.insert(0, p)
import torch
from e3_layers_amd.backend import ops
dev = torch.device("cuda:0")
K, N = int(sys.argv[1]), int(sys.argv[2])
M = int(sys.argv[3]) if len(sys.argv) > 3 else 69484
torch.manual_seed(0)
spec = ops.LinearSpec(K, N, [ops.LinInstr(0, 0, K, N, 1, 0, 1.0)], "e3nn", "e3nn", [], True, True, K * N)
h = (torch.randn(M, K, device=dev) * torch.exp(torch.randn(M, 1, device=dev))).requires_grad_(True)
w = torch.randn(K * N, device=dev, requires_grad=True)
y = ops.strided_linear(h, w, None, spec)
ref = h.detach().double() @ w.detach().double().view(K, N)
err = float((y.detach().double() - ref).norm() / ref.norm())
g = torch.randn_like(y)
gh, gw = torch.autograd.grad(y, [h, w], g)
rgh = g.double() @ w.detach().double().view(K, N).t()
rgw = h.detach().double().t() @ g.double()
e2 = float((gh.double() - rgh).norm() / rgh.norm()); e3 = float((gw.double().view(K, N) - rgw).norm() / rgw.norm())
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps
hd, wd = h.detach(), w.detach()
us = timeit(lambda: ops.strided_linear(hd, wd, None, spec))
fl = 2.0 * M * K * N
print(f"X3={os.environ.get('E3K_GEMM_X3','0')} M={M} K={K} N={N}: fwd err {err:.2e} dgrad err {e2:.2e} wgrad err {e3:.2e} | fwd {us:8.1f} us {fl/us/1e6:6.1f} TF/s")
