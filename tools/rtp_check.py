"""Radial-fused TP kernels vs the GEMM + e3k_tp_* pair on one convolution signature: parity and time.

    python tools/rtp_check.py [--mol 256] [--lmax 2] [--layer 3]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "equivariant-nn-zoo_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch

from e3_layers_amd.backend import ops
from e3_layers_amd.backend.graph import build_topology
from e3_layers_amd.configs import config_energy
from e3_layers_amd.data.synthetic import synth_qm9
from e3_layers_amd.utils import build


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(n):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mol", type=int, default=256)
    ap.add_argument("--lmax", type=int, default=2)
    ap.add_argument("--layer", type=int, default=3)
    ap.add_argument("--stamps", action="store_true", help="diagnostic library (make -C csrc dbg; E3K_LIB=.../libe3k_dbg.so): "
                    "in-kernel cycle sums per phase")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = build(config_energy.get_config(l_max=args.lmax).model_config).to(dev)
    conv = getattr(model, f"layer{args.layer}").conv
    tp = conv.tp.tp
    plan = tp.plan
    batch = synth_qm9(1000, args.mol).to(dev)
    n, e = batch["pos"].shape[0], batch["edge_index"].shape[1]
    topo = build_topology(batch["edge_index"], n)
    x = torch.randn(n, plan.d_in, device=dev)
    vec = torch.randn(e, 3, device=dev)
    sh = ops.spherical_harmonics(vec, [0, 1, 2], True, "component")
    h = torch.randn(e, 64, device=dev)
    wl = torch.randn(64, plan.w_numel, device=dev)
    scale = 1.0 / 8.0
    print(f"N={n} E={e} d_in={plan.d_in} W={plan.w_numel} d_mid={plan.d_mid} rtp_supported={plan.rtp_supported(dev)}")
    w = (h @ wl) * scale
    ref = ops._tp_fwd_raw(x, sh, w, topo, plan)
    out = ops._rtp_fwd_raw(h, wl, scale, x, sh, topo, plan)
    err = float((out - ref).norm() / ref.norm())
    print(f"fwd rel err vs gemm+tp_fwd: {err:.3e}  max abs {float((out - ref).abs().max()):.3e}")
    if args.stamps:
        import ctypes as C

        from e3_layers_amd.backend import lib as L

        lib = L.load()
        n_tiles = (e + 63) // 64
        buf = torch.zeros(n_tiles * 4 * 5, dtype=torch.int64, device=dev)
        lib.e3k_rtp_set_debug_buffer.argtypes = [C.c_void_p]
        lib.e3k_rtp_set_debug_buffer.restype = None
        lib.e3k_rtp_set_debug_buffer(buf.data_ptr())
        for _ in range(3):
            ops._rtp_fwd_raw(h, wl, scale, x, sh, topo, plan)
        torch.cuda.synchronize()
        st = buf.view(n_tiles, 4, 5).double()
        names = ["prologue", "matrix", "wait1", "vector", "wait2"]
        tot = st.sum(-1)
        print("stamps (s_memtime ticks = 10 ns at 100 MHz), mean over tiles of the per-wave sums; tile total mean "
              f"{tot.mean():.0f}, max {tot.max():.0f}")
        for i, nm in enumerate(names):
            print(f"  {nm:9s} mean {st[..., i].mean():9.0f}  max-wave mean {st[..., i].max(1).values.mean():9.0f}")
        lib.e3k_rtp_set_debug_buffer(None)
    t_gemm = timed(lambda: torch.mm(h, wl))
    t_tp = timed(lambda: ops._tp_fwd_raw(x, sh, w, topo, plan))
    t_rtp = timed(lambda: ops._rtp_fwd_raw(h, wl, scale, x, sh, topo, plan))
    flops = 2.0 * e * 64 * plan.w_numel
    print(f"fwd: rocblas gemm {t_gemm:.1f} us + tp_fwd {t_tp:.1f} us  vs  rtp_fwd {t_rtp:.1f} us "
          f"({flops / t_rtp / 1e6:.1f} TFLOP/s of f32 MFMA)")


if __name__ == "__main__":
    main()
