"""The knot-table guard's per-column bound (e3k_rtable_guard) on the shipped models at random init, for several column floors:
floor 1.0 reproduces the global bound of rounds 2-4 (max|d4 T| / max|T|); smaller floors let small-amplitude columns speak for
themselves.  Run on the GPU box; writes one line per (model, layer, floor)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "equivariant-nn-zoo_amd")):
    sys.path.insert(0, p)
import torch

from e3_layers_amd.backend import lib as L
from e3_layers_amd.backend import ops, radial_table
from e3_layers_amd.configs import config_energy, config_energy_force
from e3_layers_amd.utils import build

dev = torch.device("cuda:0")
lib = L.load()


def est(table, floor):
    state = torch.zeros(4, device=dev)
    scratch = torch.empty(2 * table.shape[1], device=dev)
    L.check(lib.e3k_rtable_guard((C.c_void_p * 1)(table.data_ptr()), (C.c_void_p * 1)(state.data_ptr()), (C.c_void_p * 1)(scratch.data_ptr()),
                                 (C.c_int32 * 1)(table.shape[1]), 1, table.shape[0], float(floor), 0.0, 1, L.stream_ptr()), "guard")
    torch.cuda.synchronize()
    return float(state[3]) if floor < 1.0 else float(state[1])      # per-column ratio; floor 1.0: the table-wide ratio


for name, tree, r_max in (("config_energy l_max 2", config_energy.get_config(l_max=2).model_config, 4.0),
                          ("config_energy l_max 3", config_energy.get_config(l_max=3).model_config, 4.0),
                          ("config_energy_force", config_energy_force.get_config().model_config, 5.0)):
    for seed in (0, 1):
        torch.manual_seed(seed)
        model = build(tree).to(dev)
        net = getattr(model, "func", model)
        enc = net.radial_basis
        b, c = enc.basis, enc.cutoff
        radii = radial_table.knot_radii(r_max, radial_table.KNOTS, dev)
        with torch.no_grad():
            rows = ops.radial_basis(radii, b.bessel_weights, b.r_max, b.r_min, c.p, b.one_over_r, c.cutoff.kind)
            for i in range(tree.num_layers):
                fc = getattr(net, f"layer{i}").conv.fc
                table = fc(rows).contiguous()
                t = table.double()
                d4 = t[4:] - 4 * t[3:-1] + 6 * t[2:-2] - 4 * t[1:-3] + t[:-4]
                ref_global = float(d4.abs().amax() * 3 / 128 / t.abs().amax())
                col = (d4.abs().amax(0) * 3 / 128 / t.abs().amax(0).clamp_min(2.0 ** -7 * t.abs().amax()))
                print(f"{name} seed {seed} layer {i} W {table.shape[1]}: torch global {ref_global:.3e} torch col(2^-7) {float(col.max()):.3e} | kernel "
                      + " ".join(f"floor {fl:g}: {est(table, fl):.3e}" for fl in (1.0, 2.0 ** -4, 2.0 ** -7, 2.0 ** -10, 0.0)), flush=True)
