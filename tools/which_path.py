"""Which path the message-passing layers of a configuration take on this batch: the fused layer block (and whether the native
executor serves it) or the composed per-op path, with the reason.   python tools/which_path.py [energy_force|diffusion|diffusion_CA]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "equivariant-nn-zoo_amd")):
    sys.path.insert(0, p)
import torch

from e3_layers_amd.backend import conv_block, conv_native, radial_table
from e3_layers_amd.configs import config_diffusion, config_diffusion_CA, config_energy_force
from e3_layers_amd.data.synthetic import synth_protein, synth_qm9, synth_qm9_diffusion
from e3_layers_amd.nn import message_passing as mp
from e3_layers_amd.nn.core import get_row_key
from e3_layers_amd.run.sde_utils import VPSDE, sde_loss
from e3_layers_amd.utils import build

kind = sys.argv[1] if len(sys.argv) > 1 else "diffusion_CA"
dev = torch.device("cuda", 0)
if kind == "diffusion_CA":
    cfg, batch, sde = config_diffusion_CA.get_config(), synth_protein(1, 4, n_res=384), VPSDE({"CA": 3})
elif kind == "diffusion":
    cfg, batch, sde = config_diffusion.get_config(), synth_qm9_diffusion(1, 128), VPSDE({"pos": 3})
else:
    cfg, batch, sde = config_energy_force.get_config(), synth_qm9(1, 64, r_max=5.0), None
torch.manual_seed(0)
model = build(cfg.model_config).to(dev)
batch = batch.to(dev)
orig = mp.MessagePassing._forward_block
log = []


def traced(self, data, out_cf):
    out = orig(self, data, out_cf)
    x, sh, radial = data["input_features"], data["edge_spherical"], data["edge_radial"]
    why = "block"
    if out is None:
        plan = self._block_plan()
        if sh.requires_grad:
            why = "composed: edge_spherical requires grad (forces)"
        elif plan is None:
            why = "composed: no block plan for this layer structure"
        elif self.conv.sc is not None and not self.conv.sc.keyed_pays(get_row_key(data["node_attrs"]), x.shape[0]):
            key = get_row_key(data["node_attrs"])
            why = f"composed: node_attrs not keyed / too few rows per key (key = {None if key is None else key[1]}, rows = {x.shape[0]})"
        else:
            why = "composed: ?"
    else:
        plan = self._block_plan()
        fc = list(self.conv.fc.children())
        why = (f"block, native executor = {conv_native.ENABLED and conv_native.native_layer(plan) is not None}, "
               f"knot table = {radial_table.applicable(radial, fc[-1].weight)}, radial requires grad = {radial.requires_grad}")
    log.append(f"N={x.shape[0]} E={radial.shape[0]} {why}")
    return out


mp.MessagePassing._forward_block = traced
if sde is None:
    out = model(batch)
    loss = out["total_energy"].square().mean() + out["forces"].square().mean()
else:
    loss = sde_loss(sde, model, batch)[0]
loss.backward()
torch.cuda.synchronize()
print(kind)
for line in log:
    print("  ", line)
