"""Would a BLOCK-MAJOR layout of the tensor product's output (`mid` / `g_mid`: one dense [N (2l+1), mul] matrix per irrep block instead of
26 KB rows holding all blocks) speed up the trailing Linear's GEMMs?  Same problems, same bytes; only the row stride of the mid-side
operand changes (forward and weight gradient: A; input gradient: C).  python3 tools/micro/blocked_mid.py [molecules]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "equivariant-nn-zoo_amd")):
    sys.path.insert(0, p)
import ctypes as C
import torch
from e3_layers_amd.backend import lib as L, ops
from e3_layers_amd.configs import config_energy
from e3_layers_amd.utils import build

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = 18 * B
model = build(config_energy.get_config(l_max=2).model_config).to(dev)
lin = model.layer3.conv.tp.linear
spec = lin.spec("cf", "cf")
w = lin.weight.detach()
mid = torch.randn(n, spec.d_in, device=dev)
gy = torch.randn(n, spec.d_out, device=dev)
y = torch.empty(n, spec.d_out, device=dev)
gx = torch.empty(n, spec.d_in, device=dev)
gw = torch.zeros_like(w)
blocks = {}
for ins in spec.instr:
    blocks[ins.in_off] = ins.mul_in * ins.dim       # block start (floats within the row) -> block row width


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


def patched(templates, side: str):
    """Copies of the template arrays with the mid-side operand block-major: offset blk_off -> blk_off * n, row stride d_in -> block width."""
    out = []
    for arr, cnt in templates.rounds:
        new = (L.GemmProblem * cnt)()
        for i in range(cnt):
            C.memmove(C.byref(new, i * C.sizeof(L.GemmProblem)), C.byref(arr, i * C.sizeof(L.GemmProblem)), C.sizeof(L.GemmProblem))
            p = new[i]
            if side == "A":
                off = (p.A or 0) // 4
                p.A, p.a_r1 = 4 * off * n, blocks[off]
            else:
                off = (p.C or 0) // 4
                p.C, p.c_r1 = 4 * off * n, blocks[off]
        out.append((new, cnt))
    return out


def run(rounds, a, b, c, wgrad=False):
    lib, st = L.load(), L.stream_ptr()
    for arr, cnt in rounds:
        L.check(lib.e3k_gemm_rebased(arr, cnt, a, None, b, c, None, n, int(wgrad), st), "gemm")


flops = sum(2.0 * n * ins.dim * ins.mul_in * ins.mul_out for ins in spec.instr)
t_f = ops._lin_fwd_templates(spec, 0.3, False, 0, 1.0, False)
t_d = ops._lin_dgrad_templates(spec, 0.3, False)
t_w = ops._lin_wgrad_templates(spec, 0.3)
for name, tm, side, args, wg in (("fwd", t_f, "A", (mid.data_ptr(), w.data_ptr(), y.data_ptr()), False),
                                 ("dgrad", t_d, "C", (gy.data_ptr(), w.data_ptr(), gx.data_ptr()), False),
                                 ("wgrad", t_w, "A", (mid.data_ptr(), gw.data_ptr(), gy.data_ptr()), True)):
    rows = timeit(lambda: run(tm.rounds, *args, wgrad=wg))
    blk = timeit(lambda: run(patched(tm, side), *args, wgrad=wg)) if False else None
    pr = patched(tm, side)
    blk = timeit(lambda: run(pr, *args, wgrad=wg))
    print(f"post-linear {name:6s}: rows of all blocks {rows:6.1f} us ({flops / rows / 1e6:5.1f} TF/s)   block-major {blk:6.1f} us ({flops / blk / 1e6:5.1f} TF/s)")
