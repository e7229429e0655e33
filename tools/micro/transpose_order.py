"""Is the knot-table transpose bound by the ORDER of its g_w rows?  Same sizes, radii in edge order (the step's: rows scattered over
the buffer) against radii sorted by knot (rows of a segment adjacent).  Prints us per launch and TB/s on E 4 W bytes."""
import sys
sys.path.insert(0, "/root/repo/equivariant-nn-zoo_amd")
import torch
from e3_layers_amd.backend import radial_table as rt

dev = torch.device("cuda:0")
E, W = 70656, int(sys.argv[1]) if len(sys.argv) > 1 else 1408
torch.manual_seed(0)
r = (0.9 + 3.0 * torch.rand(E, device=dev)).contiguous()
gw = torch.randn(E, W, device=dev)
for name, rr in (("edge order", r), ("knot order", torch.sort(r).values.contiguous())):
    bins = rt.build_bins(rr, 4.0, 512)
    for _ in range(3):
        rt.interp_bwd_raw(gw, bins)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    a.record()
    for _ in range(n):
        rt.interp_bwd_raw(gw, bins)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / n
    print(f"{name}: {us:.1f} us per transpose (partial + combine), {E * W * 4 / us / 1e6:.2f} TB/s on g_w")
