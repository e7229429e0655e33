python3 - <<'PY' 2>&1 | tail -30
import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "equivariant-nn-zoo_amd"), os.path.join(os.getcwd(), "tests")]
import torch, numpy as np
from e3_layers_amd.backend import radial_table
dev = torch.device("cuda:0")
W = 1920; knots = 256; r_max = 5.0
gen = torch.Generator().manual_seed(3)
r = (0.3 + torch.rand(20000, generator=gen) * 4.6)
bins = radial_table.build_bins(r.to(dev), r_max, knots)
K = bins.knots; h = bins.spacing
radii = torch.arange(K + 1, dtype=torch.float64) * h
cols = torch.arange(W, dtype=torch.float64)
table = (torch.sin(radii[:, None] * (1.0 + 5.0 * cols[None, :] / W)) * torch.exp(-0.2 * radii[:, None])).float().to(dev)
P = radial_table.pack_raw(table, K)
wp = radial_table.interp_packed_raw(P, bins).cpu().double().numpy()
w4 = radial_table.interp_fwd_raw(table, bins).cpu().double().numpy()
T = table.cpu().double().numpy()
x = r.double().numpy() / h; i = np.floor(x).astype(int); t = x - i
assert (i == bins.bin.cpu().numpy()).all()
a, b, c, d = T[i - 1], T[i], T[i + 1], T[i + 2]
t_ = t[:, None]
ref = a * (-t_ * (t_ - 1) * (t_ - 2) / 6) + b * ((t_ + 1) * (t_ - 1) * (t_ - 2) / 2) + c * (-(t_ + 1) * t_ * (t_ - 2) / 2) + d * ((t_ + 1) * t_ * (t_ - 1) / 6)
print("w4 err", np.abs(w4 - ref).max(), "wp err", np.abs(wp - ref).max())
err = np.abs(wp - ref)
print("per-column max (every 240th)", err.max(0)[::240])
e, cidx = np.unravel_index(err.argmax(), err.shape)
print("worst edge", e, "col", cidx, "t", t[e], "i", i[e], "r", float(r[e]))
Pn = P.cpu().numpy()
row = Pn[i[e]].reshape(-1)
d0 = row[:2 * W].view(np.float32)[2 * cidx]; d1 = row[:2 * W].view(np.float32)[2 * cidx + 1]
pk = row[2 * W + cidx]
hh = np.array([pk], dtype=np.int32).view(np.float16)
print("record d0", d0, "d1", d1, "D2", hh[0], "D3", hh[1])
aa, bb, cc, dd = a[e, cidx], b[e, cidx], c[e, cidx], d[e, cidx]
c1 = -aa / 3 - bb / 2 + cc - dd / 6; c2 = aa / 2 - bb + cc / 2; c3 = -aa / 6 + bb / 2 - cc / 2 + dd / 6
print("expect d0", bb + c1 / 2 + c2 / 4 + c3 / 8, "d1", c1 + c2 + .75 * c3, "D2", (c2 + 1.5 * c3) * 1024, "D3", c3 * 65536)
coef = bins.coef.cpu().numpy()[e]
s = (coef[2] - coef[0] + 2 * coef[3]) - 0.5
print("s kernel", s, "s true", t[e] - 0.5)
val = d0 + s * (d1 + (s / 1024) * (float(hh[0]) + (s / 64) * float(hh[1])))
print("host eval of record", val, "gpu", wp[e, cidx], "ref", ref[e, cidx])
PY
