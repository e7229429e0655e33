#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "equivariant-nn-zoo_amd")):
    sys.path.insert(0, p)
import torch
from e3_layers_amd.backend import ops
dev = torch.device("cuda:0")
E = int(os.environ.get("E3K_ROWS", "69484"))
K, N = int(sys.argv[1]), int(sys.argv[2])
spec = ops.LinearSpec(K, N, [ops.LinInstr(0, 0, K, N, 1, 0, 1.0)], "e3nn", "e3nn", [], True, True, K * N)
h = torch.randn(E, K, device=dev); w = torch.randn(K * N, device=dev)
for _ in range(3): y = ops.strided_linear(h, w, None, spec)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): y = ops.strided_linear(h, w, None, spec)
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) * 100
print(f"K={K} N={N} env={[k for k in os.environ if k.startswith('E3K_')]} {us:.1f} us {2*E*K*N/us/1e6:.1f} TF/s")

if os.environ.get("E3K_STAMPS"):
    import ctypes, numpy as np
    from e3_layers_amd.backend import lib as L
    lib = L.load()
    nblk = ((E + 127) // 128) * ((((N + 63) // 64) + 7) // 8)
    n = nblk * 4 * 8
    buf = (ctypes.c_ulonglong * n)()
    lib.e3k_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    y = ops.strided_linear(h, w, None, spec)
    lib.e3k_debug_stamps(buf, n)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8).astype(np.float64)
    names = ["fetch_problem", "tables + A tile + first B load issue", "LDS write B (waits loads)", "barrier 1", "issue next B + MFMA", "store_acc", "barrier 2", "-"]
    print("   waves", a.shape[0], "total cycles/wave", a.sum(1).mean())
    for i, nm in enumerate(names):
        print(f"   {nm:40s} mean {a[:, i].mean():10.0f}  p50 {np.median(a[:, i]):10.0f} cycles per wave")
