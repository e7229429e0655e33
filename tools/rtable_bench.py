"""Radial knot table: time of the interpolation kernels and of the per-edge GEMMs they replace (E edges, W weights)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "equivariant-nn-zoo_amd")):
    sys.path.insert(0, p)
import torch
from e3_layers_amd import nn as pnn
from e3_layers_amd.backend import radial_table
from e3_layers_amd.data.synthetic import synth_qm9
from e3_layers_amd.nn.core import FullyConnectedNet
from e3_layers_amd.utils.utils import activations
dev = torch.device("cuda:0")
W = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
batch = synth_qm9(1000, 256).to(dev)
ei = batch["edge_index"]
r = (batch["pos"][ei[1]] - batch["pos"][ei[0]]).norm(dim=1)
E = r.numel()
enc = pnn.RadialBasisEncoding(r_max=4.0, trainable=True, irreps_out=("8x0e", "edge_radial"), irreps_in=("1x0e", "edge_length")).to(dev)
fc = FullyConnectedNet([8, 64, 64, 64, W], activations["ssp"]).to(dev)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
emb = enc({"input": r}, {"input": ("edge", "1x0e")})[0]["radial_embedding"]
src = radial_table.source_of(emb)
bin32, t, ptr, perm = src.bins()
cnt = (ptr[1:] - ptr[:-1]).float()
print(f"E={E} W={W} knots={radial_table.KNOTS}: edges per knot mean {cnt.mean():.1f} max {int(cnt.max())} nonempty {int((cnt > 0).sum())}")
seed = torch.randn(E, W, device=dev)
params = list(fc.parameters())
def per_edge():
    e1 = enc({"input": r}, {"input": ("edge", "1x0e")})[0]["radial_embedding"]
    w = fc(e1)
    torch.autograd.grad(w, params + [enc.basis.bessel_weights], seed)
def table():
    e2 = enc({"input": r}, {"input": ("edge", "1x0e")})[0]["radial_embedding"]
    w = radial_table.table_weights(fc, e2)
    torch.autograd.grad(w, params + [enc.basis.bessel_weights], seed)
with torch.no_grad():
    T = fc(src.knot_basis()).detach()
def f_only():
    radial_table.RadialTableFn.apply(T, src)
Tg = T.clone().requires_grad_(True)
wt = radial_table.RadialTableFn.apply(Tg, src)
def b_only():
    torch.autograd.grad(wt, Tg, seed, retain_graph=True)
def bins_only():
    src._bins = None
    src.bins()
print(f"per-edge MLP fwd+bwd {timeit(per_edge):.1f} us | table fwd+bwd {timeit(table):.1f} us | interp fwd {timeit(f_only):.1f} us, "
      f"interp bwd {timeit(b_only):.1f} us, bins {timeit(bins_only):.1f} us")
