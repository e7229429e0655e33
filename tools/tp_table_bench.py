#!/usr/bin/env python3
"""The table-form tensor-product kernels of config_energy layer 3 (l_max 2, 256 molecules) in isolation, on the batch's real edge
lengths, for several knot counts; `--pmc` runs each a few times only (for rocprofv3 --pmc passes: TCC_HIT_sum TCC_MISS_sum ...).
    python3 tools/tp_table_bench.py [--pmc] [knots ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "equivariant-nn-zoo_amd")):
    sys.path.insert(0, p)
import torch
from e3_layers_amd.backend import conv_force, ops, radial_table
from e3_layers_amd.backend.graph import build_topology
from e3_layers_amd.configs import config_energy
from e3_layers_amd.data.synthetic import synth_qm9
from e3_layers_amd.utils import build

args = [a for a in sys.argv[1:] if not a.startswith("--")]
pmc = "--pmc" in sys.argv
bonds = "clustered" if "--clustered" in sys.argv else "uniform"
targets = [int(a) for a in args] or [512]
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = build(config_energy.get_config(l_max=2).model_config).to(dev)
batch = synth_qm9(1000, 256, bonds=bonds).to(dev)
n, e = batch["pos"].shape[0], batch["edge_index"].shape[1]
topo = build_topology(batch["edge_index"], n)
ei = batch["edge_index"]
r = (batch["pos"][ei[1]] - batch["pos"][ei[0]]).norm(dim=1)
tp = model.layer3.conv.tp.tp.plan
x = torch.randn(n, tp.d_in, device=dev)
sh = torch.randn(e, tp.d_sh, device=dev)
g = torch.randn(n, tp.d_mid, device=dev)
w = torch.randn(e, tp.w_numel, device=dev)


def timeit(fn, reps=20):
    if pmc:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        return 0.0
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    t.record()
    torch.cuda.synchronize()
    return s.elapsed_time(t) * 1e3 / reps


print(f"N={n} E={e} W={tp.w_numel} d_in={tp.d_in} d_mid={tp.d_mid} bonds={bonds}")
if "--knot-order" in sys.argv:
    # VERDICT r4 item 3 (the g_w[E, W] round trip): a kernel that forms g_w[e] in registers while walking the edges in KNOT order
    # (so that the four stencil sums of a knot segment stay in registers) has to gather x[src] AND g_mid[dst] per edge with no
    # destination locality.  Its dominant cost is emulated with the existing streamed tp_bwd_x kernel walking a fake CSR whose
    # "nodes" are the <= 64-edge knot segments of the transposed interpolation: per edge it gathers the same g_mid[dst] rows
    # (26 KB) plus one w row (7.7 KB, standing in for the x[src] gather of 4.6 KB) -- a LOWER bound of the fused kernel's gather
    # traffic (results meaningless) -- to be compared with tp_bwd_w + the table transpose it would replace.
    from e3_layers_amd.backend.graph import GraphTopo
    bins = radial_table.build_bins(r, 4.0, 512)
    seg_ptr = []
    ptr = bins.ptr.cpu().tolist()
    for b in range(len(ptr) - 1):
        for s0 in range(ptr[b], ptr[b + 1], 64):
            seg_ptr.append(s0)
    seg_ptr.append(e)
    seg_ptr = torch.tensor(seg_ptr, dtype=torch.int32, device=dev)
    fake = GraphTopo(topo.src, topo.dst, topo.dst_ptr, topo.dst_perm, seg_ptr, bins.perm)
    n_seg = seg_ptr.numel() - 1
    gx = torch.zeros(n_seg, tp.d_in, device=dev)
    from e3_layers_amd.backend import lib as L_
    def knot_walk():
        L_.check(L_.load().e3k_tp_bwd_x(tp.handle(dev), L_.ptr(sh), L_.ptr(w), L_.ptr(g), L_.ptr(fake.dst), L_.ptr(fake.src_ptr), L_.ptr(fake.src_perm),
                                        n_seg, e, L_.ptr(gx), L_.stream_ptr()), "knot walk")
    t_k = timeit(knot_walk)
    t_w = timeit(lambda: ops._tp_bwd_w_raw(x, sh, None, g, topo, tp, False, True))
    gw, _ = ops._tp_bwd_w_raw(x, sh, None, g, topo, tp, False, True)
    t_t = timeit(lambda: radial_table.interp_bwd_raw(gw, bins))
    t_x = timeit(lambda: ops._tp_bwd_x_raw(sh, w, g, topo, tp))
    print(f"knot-order walk ({n_seg} segments of <= 64 edges; gathers g_mid[dst] + one 7.7 KB row per edge): {t_k:7.1f} us")
    print(f"what it would replace: tp_bwd_w {t_w:7.1f} us + table transpose {t_t:7.1f} us = {t_w + t_t:7.1f} us   (tp_bwd_x in source order, same gather volume: {t_x:7.1f} us)")
    sys.exit(0)
if "--ablate" in sys.argv:      # dbg library only (E3K_LIB=.../libe3k_dbg.so): timing-only masks of the packed forward
    import ctypes
    from e3_layers_amd.backend import lib as L
    lib = ctypes.CDLL(os.environ["E3K_LIB"])
    bins = radial_table.build_bins(r, 4.0, 512)
    T = torch.randn(bins.knots + 1, tp.w_numel, device=dev) * 1e-3
    P = radial_table.pack_raw(T, bins.knots)
    for mask, what in ((0, "full"), (1, "no table loads"), (2, "no x loads"), (3, "no table, no x loads"), (4, "no CG arithmetic"), (7, "loop skeleton")):
        assert lib.e3k_dbg_tp_ablate(mask) == 0
        print(f"packed tp_fwd, {what:22s}: {timeit(lambda: conv_force._tp_fwd_ptable(x, sh, P, bins, topo, tp)):7.1f} us")
    for mask, what in ((0, "full"), (8, "g[dst] rows from the walker's own node (L1 hits)"), (1, "no table loads"), (9, "neither")):
        assert lib.e3k_dbg_tp_ablate(mask) == 0
        print(f"packed tp_bwd_x, {what:50s}: {timeit(lambda: conv_force._tp_bwd_x_ptable(sh, P, bins, g, topo, tp)):7.1f} us   "
              f"with g_w: {timeit(lambda: conv_force._tp_bwd_xw_ptable(x, sh, P, bins, g, topo, tp)):7.1f} us")
    lib.e3k_dbg_tp_ablate(0)
    sys.exit(0)
print(f"streamed  tp_fwd {timeit(lambda: ops._tp_fwd_raw(x, sh, w, topo, tp)):7.1f} us   tp_bwd_x {timeit(lambda: ops._tp_bwd_x_raw(sh, w, g, topo, tp)):7.1f} us"
      f"   tp_bwd_w {timeit(lambda: ops._tp_bwd_w_raw(x, sh, None, g, topo, tp, False, True)):7.1f} us")
gw, _ = ops._tp_bwd_w_raw(x, sh, None, g, topo, tp, False, True)
for target in targets:
    t_bins = timeit(lambda: radial_table.build_bins(r, 4.0, target))
    bins = radial_table.build_bins(r, 4.0, target)
    T = torch.randn(bins.knots + 1, tp.w_numel, device=dev)
    cnt = (bins.ptr[1:] - bins.ptr[:-1]).float()
    t_f = timeit(lambda: conv_force._tp_fwd_table(x, sh, T, bins, topo, tp))
    t_x = timeit(lambda: conv_force._tp_bwd_x_table(sh, T, bins, g, topo, tp))
    P = radial_table.pack_raw(T, bins.knots)
    t_p = timeit(lambda: radial_table.pack_raw(T, bins.knots))
    t_fp = timeit(lambda: conv_force._tp_fwd_ptable(x, sh, P, bins, topo, tp))
    t_xp = timeit(lambda: conv_force._tp_bwd_x_ptable(sh, P, bins, g, topo, tp))
    t_xw = timeit(lambda: conv_force._tp_bwd_xw_ptable(x, sh, P, bins, g, topo, tp))
    gx_a = conv_force._tp_bwd_x_ptable(sh, P, bins, g, topo, tp)
    gx_b, gw_b = conv_force._tp_bwd_xw_ptable(x, sh, P, bins, g, topo, tp)
    gx_c = conv_force._tp_bwd_x_ptable(sh, P, bins, g, topo, tp)
    print(f"  (two runs of the unfused kernel bit-equal {bool(torch.equal(gx_a, gx_c))}; fused vs unfused max |d| {float((gx_a - gx_b).abs().max()):.2e} of {float(gx_a.abs().max()):.2e})")
    print(f"  fused input + weight gradient {t_xw:7.1f} us: g_x bit-equal {bool(torch.equal(gx_a, gx_b))}, "
          f"g_w vs tp_bwd_w max |d| {float((gw_b - gw).abs().max()):.2e} of {float(gw.abs().max()):.2e}")
    t_i = timeit(lambda: radial_table.interp_fwd_raw(T, bins))
    t_t = timeit(lambda: radial_table.interp_bwd_raw(gw, bins))
    print(f"knots {bins.knots:5d} (table {4e-6 * (bins.knots + 1) * tp.w_numel:5.1f} MB, edges/knot mean {float(cnt[cnt > 0].mean()):6.1f} max {int(cnt.max()):5d}): "
          f"tp_fwd_table {t_f:7.1f} us  tp_bwd_x_table {t_x:7.1f} us  PACKED fwd {t_fp:7.1f} us  bwd_x {t_xp:7.1f} us  pack {t_p:5.1f} us  interp_fwd {t_i:6.1f} us  interp_bwd {t_t:6.1f} us  bins {t_bins:5.1f} us")
