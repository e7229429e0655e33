#!/usr/bin/env python3
"""The node-side Linear after the tensor product (layer 3 of config_energy, l_max 2) in isolation: forward, dgrad and wgrad
launch groups timed with HIP events, plus a K scan of one 128-row-tile problem (time = a + b K: fixed cost vs per-K cost).
python tools/postlin_bench.py [molecules]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "equivariant-nn-zoo_amd")):
    sys.path.insert(0, p)
import torch
from e3_layers_amd.backend import ops
from e3_layers_amd.configs import config_energy
from e3_layers_amd.utils import build

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = 18 * B
torch.manual_seed(0)
model = build(config_energy.get_config(l_max=2).model_config).to(dev)
lin = model.layer3.conv.tp.linear
spec = lin.spec("cf", "cf")
w = lin.weight.detach()
mid = torch.randn(n, spec.d_in, device=dev)
gy = torch.randn(n, spec.d_out, device=dev)
y = torch.empty(n, spec.d_out, device=dev)
gw = torch.zeros_like(w)


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


flops = sum(2.0 * n * ins.dim * ins.mul_in * ins.mul_out for ins in spec.instr)
for name, fn in (("fwd", lambda: ops._lin_fwd_raw(mid, w, None, y, spec, 0.3, False)),
                 ("dgrad", lambda: ops._lin_dgrad_raw(gy, w, spec, 0.3)),
                 ("wgrad", lambda: ops._lin_wgrad_raw(mid, gy, gw.view(-1), spec, 0.3))):
    us = timeit(fn)
    print(f"post-linear {name:6s} rows {n} d_in {spec.d_in} d_out {spec.d_out}: {us:7.1f} us  {flops / us / 1e6:6.1f} TF/s")

for K in (64, 128, 256, 384, 512, 768, 1024):
    for rows in (n * 3,):
        sp = ops.LinearSpec(K, 64, [ops.LinInstr(0, 0, K, 64, 1, 0, 1.0)], "e3nn", "e3nn", [], True, True, K * 64)
        x = torch.randn(rows, K, device=dev)
        ww = torch.randn(K * 64, device=dev)
        out = torch.empty(rows, 64, device=dev)
        us = timeit(lambda: ops._lin_fwd_raw(x, ww, None, out, sp, 1.0, False))
        print(f"one problem rows {rows} N 64 K {K:5d}: {us:7.1f} us  {2.0 * rows * 64 * K / us / 1e6:6.1f} TF/s")

# the same K = 384, N = 64 problem (dim 3: rows = 3 n) read out of row strides from compact to the real mid width:
# is the time of these GEMMs a function of how far apart the 1.5 KB runs of A lie in memory?
K = 384
for d_in in (K * 3, 2 * K * 3, 6528, 8192):
    ins = ops.LinInstr(0, 0, K, 64, 3, 0, 1.0, 0, 0)
    sp = ops.LinearSpec(d_in, 64 * 3, [ins], "cf", "cf", [], True, d_in == K * 3, K * 64)
    x = torch.randn(n, d_in, device=dev)
    ww = torch.randn(K * 64, device=dev)
    out = torch.empty(n, 64 * 3, device=dev)
    g = torch.randn(n, 64 * 3, device=dev)
    gww = torch.zeros(K * 64, device=dev)
    fl = 2.0 * n * 3 * K * 64
    for name, fn in (("fwd", lambda: ops._lin_fwd_raw(x, ww, None, out, sp, 1.0, False)),
                     ("dgrad", lambda: ops._lin_dgrad_raw(g, ww, sp, 1.0)),
                     ("wgrad", lambda: ops._lin_wgrad_raw(x, g, gww, sp, 1.0))):
        us = timeit(fn)
        print(f"stride scan K 384 dim 3 rows {3 * n} d_in {d_in:5d} {name:6s}: {us:7.1f} us  {fl / us / 1e6:6.1f} TF/s")

# the per-edge last layer of a radial MLP without the knot table (the protein score net: ~18 000 edges, 64 hidden, W = 1 920):
# output-stream bound -- 138 MB written per launch
for rows, W in ((18000, 1920), (72000, 1920)):
    sp = ops.LinearSpec(64, W, [ops.LinInstr(0, 0, 64, W, 1, 0, 1.0)], "e3nn", "e3nn", [], True, True, 64 * W)
    x = torch.randn(rows, 64, device=dev)
    ww = torch.randn(64 * W, device=dev)
    out = torch.empty(rows, W, device=dev)
    us = timeit(lambda: ops._lin_fwd_raw(x, ww, None, out, sp, 1.0, False))
    print(f"radial last layer rows {rows} K 64 N {W}: {us:7.1f} us  {rows * W * 4 / us / 1e6:6.2f} TB/s written  {2.0 * rows * 64 * W / us / 1e6:6.1f} TF/s")
