#!/usr/bin/env python3
"""Micro-benchmark of the e3k GEMM entry points on the shapes of config_energy (l_max=2, layer 3,
B=256): HIP-event time per call and effective TFLOP/s.  Usage: python tools/gemm_bench.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "equivariant-nn-zoo_amd")):
    sys.path.insert(0, p)
import torch

from e3_layers_amd.backend import ops
from e3_layers_amd.nn import FullyConnectedTensorProduct, Linear

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
E, N = 69484, 4623


def timeit(fn, flops, name, bytes_=0.0):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / reps
    print(f"{name:46s} {us:9.1f} us  {flops / us / 1e6:7.1f} TF/s  {bytes_ / us / 1e3:7.2f} TB/s", flush=True)


def lin_spec(k, n):
    return ops.LinearSpec(k, n, [ops.LinInstr(0, 0, k, n, 1, 0, 1.0)], "e3nn", "e3nn", [], True, True, k * n)


torch.manual_seed(0)
# --- radial MLP last layer: [E,64] x [64,1920]
h = torch.randn(E, 64, device=dev, requires_grad=True)
w4 = torch.randn(64 * 1920, device=dev, requires_grad=True)
spec = lin_spec(64, 1920)
timeit(lambda: ops.strided_linear(h.detach(), w4.detach(), None, spec), 2 * E * 64 * 1920, "radial fwd  [E,64]x[64,1920]", E * 1920 * 4)
y = ops.strided_linear(h, w4, None, spec)
g = torch.randn_like(y)
timeit(lambda: torch.autograd.grad(y, h, g, retain_graph=True), 2 * E * 64 * 1920, "radial dgrad [E,1920]x[1920,64]", E * 1920 * 4)
timeit(lambda: torch.autograd.grad(y, w4, g, retain_graph=True), 2 * E * 64 * 1920, "radial wgrad [64,E]x[E,1920]", E * 1920 * 4)
del y, g
# --- radial hidden layer [E,64]x[64,64]
w2 = torch.randn(64 * 64, device=dev)
spec2 = lin_spec(64, 64)
timeit(lambda: ops.strided_linear(h.detach(), w2, None, spec2), 2 * E * 64 * 64, "radial hidden [E,64]x[64,64]", E * 128 * 4)
# --- node linear (cf layout), post-linear like: mid -> conv_out
ir = "64x0e+64x0o+64x1e+64x1o+64x2e+64x2o"
lin = Linear(ir, ir).to(dev)
x = torch.randn(N, lin.irreps_in.dim, device=dev, requires_grad=True)
fl = 2 * N * 18 * 64 * 64
timeit(lambda: lin(x.detach(), "cf", "cf"), fl, "linear_1 fwd (6 blocks 64x64)")
mid_ir = "320x0e+192x0o+448x1e+512x1o+384x2e+448x2o"   # merged mid blocks of an l_max=2 layer (approx.)
out_ir = "64x0e+64x0o+256x0e+64x1e+64x1o+64x2e+64x2o"
post = Linear(mid_ir, out_ir).to(dev)
xm = torch.randn(N, post.irreps_in.dim, device=dev, requires_grad=True)
fl = 2 * N * (320 * 320 + 192 * 64 + 3 * 448 * 64 + 3 * 512 * 64 + 5 * 384 * 64 + 5 * 448 * 64)
timeit(lambda: post(xm.detach(), "cf", "cf"), fl, "post-linear fwd")
ym = post(xm, "cf", "cf")
gm = torch.randn_like(ym)
timeit(lambda: torch.autograd.grad(ym, xm, gm, retain_graph=True), fl, "post-linear dgrad")
timeit(lambda: torch.autograd.grad(ym, post.weight, gm, retain_graph=True), fl, "post-linear wgrad")
del ym, gm
# --- self-connection
sc = FullyConnectedTensorProduct(ir, "20x0e", out_ir).to(dev)
a = torch.randn(N, 20, device=dev, requires_grad=True)
fl = 2 * N * 20 * 64 * (320 + 64 + 3 * 64 * 2 + 5 * 64 * 2)
timeit(lambda: sc(x.detach(), a.detach()), fl, "self-connection fwd (outer)")
ys = sc(x, a)
gs = torch.randn_like(ys)
timeit(lambda: torch.autograd.grad(ys, [x, a], gs, retain_graph=True), fl, "self-connection dgrad (H gemm + reduce)")
timeit(lambda: torch.autograd.grad(ys, sc.weight, gs, retain_graph=True), fl, "self-connection wgrad (outer)")
