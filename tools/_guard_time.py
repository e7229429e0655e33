import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "equivariant-nn-zoo_amd"))
import torch
from e3_layers_amd.backend import lib as L
dev = torch.device("cuda:0")
lib = L.load()
for n, rows, W in ((5, 641, 1920), (1, 641, 1920), (5, 641, 64)):
    tabs_t = [torch.randn(rows, W, device=dev) for _ in range(n)]
    st_t = [torch.zeros(4, device=dev) for _ in range(n)]
    sc_t = [torch.empty(16 * W, device=dev) for _ in range(n)]
    tabs = (C.c_void_p * n)(*[t.data_ptr() for t in tabs_t]); states = (C.c_void_p * n)(*[t.data_ptr() for t in st_t])
    scr = (C.c_void_p * n)(*[t.data_ptr() for t in sc_t]); widths = (C.c_int32 * n)(*[W] * n)
    def run():
        L.check(lib.e3k_rtable_guard(tabs, states, scr, widths, n, rows, 2.0 ** -7, 0.05, 1, L.stream_ptr()), "guard")
    for _ in range(5): run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): run()
    b.record(); torch.cuda.synchronize()
    print(f"guard of {n} tables [{rows}, {W}]: {a.elapsed_time(b) * 1e3 / 50:.1f} us; state {st_t[0].tolist()}")
