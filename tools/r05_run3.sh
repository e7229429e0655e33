python3 - <<'PY' 2>&1 | tail -40
import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "equivariant-nn-zoo_amd"), os.path.join(os.getcwd(), "tests")]
import torch
from e3_layers_amd.backend import radial_table, conv_force, ops
from e3_layers_amd.backend.graph import build_topology
from e3_layers_amd.nn import TensorProductExpansion
dev = torch.device("cuda:0")
torch.manual_seed(21)
for left, out in (("64x0e", "64x0e+64x1o+64x2e"), ("64x0e+64x0o+64x1e+64x1o+64x2e+64x2o", "64x0e+64x0o+64x1e+64x1o+64x2e+64x2o")):
    n, knots, r_max = 300, 256, 5.0
    gen = torch.Generator().manual_seed(3)
    src = torch.randint(0, n, (n * 12,), generator=gen); dst = torch.randint(0, n, (n * 12,), generator=gen)
    ei = torch.stack([src, dst])
    e = ei.shape[1]
    mod = TensorProductExpansion(left, ("1x0e+1x1o+1x2e", "edge_spherical"), (out, "edge_features"), "uvu", internal_weight=False).to(dev)
    plan = mod.tp.plan
    topo = build_topology(ei.to(dev), n)
    x = torch.randn(n, plan.d_in, device=dev); sh = torch.randn(e, 9, device=dev); g_out = torch.randn(n, plan.d_mid, device=dev)
    r = 0.3 + torch.rand(e) * 4.6
    bins = radial_table.build_bins(r.to(dev), r_max, knots)
    radii = torch.arange(bins.knots + 1, dtype=torch.float64) * bins.spacing
    cols = torch.arange(plan.w_numel, dtype=torch.float64)
    table = (torch.sin(radii[:, None] * (1.0 + 5.0 * cols[None, :] / plan.w_numel)) * torch.exp(-0.2 * radii[:, None])).float().to(dev)
    P = radial_table.pack_raw(table, bins.knots)
    wp = radial_table.interp_packed_raw(P, bins)
    w4 = radial_table.interp_fwd_raw(table, bins)
    print(left, "W", plan.w_numel, "wp finite", bool(torch.isfinite(wp).all()), "wp-w4", float((wp - w4).abs().max()))
    o_p = conv_force._tp_fwd_ptable(x, sh, P, bins, topo, plan)
    o_r = ops._tp_fwd_raw(x, sh, wp, topo, plan)
    torch.cuda.synchronize()
    bad = ~torch.isfinite(o_p)
    print(" fwd: nonfinite", int(bad.sum()), "of", o_p.numel(), "equal", bool(torch.equal(o_p, o_r)), "maxdiff", float((o_p - o_r)[~bad].abs().max()), "ref max", float(o_r.abs().max()))
    if bad.any():
        colsbad = bad.any(0).nonzero().flatten()
        print(" bad cols", colsbad[:20].tolist(), len(colsbad), "bad rows", int(bad.any(1).sum()))
    g_p = conv_force._tp_bwd_x_ptable(sh, P, bins, g_out, topo, plan)
    g_r = ops._tp_bwd_x_raw(sh, wp, g_out, topo, plan)
    torch.cuda.synchronize()
    print(" bwd_x: nonfinite", int((~torch.isfinite(g_p)).sum()), "maxdiff", float((g_p - g_r).abs().max()), "ref max", float(g_r.abs().max()))
PY
