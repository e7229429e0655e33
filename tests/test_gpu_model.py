"""GPU parity of whole networks: the product (HIP kernels behind the e3_layers module API) vs the
float64 oracle built from the same config tree with the same parameters."""
import copy

import pytest
import torch

from tests.util import batch_to_oracle, oracle_like, rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-5   # north-star forward tolerance (fp32 vs float64 oracle, normwise)
GTOL = 5e-5  # gradients: sums over all nodes/edges in fp32 (atomics in the weight-gradient GEMMs)


def _energy_tree(l_max, n_dim, num_layers, shifts=None, node_attrs="20x0e"):
    from e3_layers_amd.configs.layer_configs import addEnergyOutput, featureModel

    cfg = featureModel(n_dim=n_dim, l_max=l_max, edge_spherical="1x0e+1x1o+1x2e", node_attrs=node_attrs,
                       edge_radial="8x0e", num_types=10, num_layers=num_layers, r_max=4.0)
    return addEnergyOutput(cfg, shifts)


def _build_pair(tree, dev):
    from e3_layers_amd.utils import build

    torch.manual_seed(0)
    prod = build(tree).to(dev)
    return prod, oracle_like(prod, tree)


@pytest.mark.parametrize("l_max,n_dim,num_layers,n_mol", [(2, 16, 3, 4), (3, 32, 3, 3), (2, 64, 3, 6)])
def test_energy_forward_backward(dev, l_max, n_dim, num_layers, n_mol):
    from e3_layers_amd.data.synthetic import synth_qm9

    tree = _energy_tree(l_max, n_dim, num_layers)
    prod, orc = _build_pair(tree, dev)
    batch = synth_qm9(3, n_mol)
    data, attrs = batch_to_oracle(batch)
    out_ref, _ = orc(data, attrs)
    out = prod(batch.clone().to(dev))
    for key in ("edge_spherical", "edge_radial", "node_attrs", "node_features", "energy", "total_energy"):
        assert rel_err(out[key], out_ref[key]) < TOL, key
    # training loss of config_energy: 1e3 * MSE(total_energy)  (e3_layers/configs/config_energy.py:27)
    target = batch["total_energy"]
    loss = 1e3 * torch.nn.functional.mse_loss(out["total_energy"], target.to(dev))
    loss_ref = 1e3 * torch.nn.functional.mse_loss(out_ref["total_energy"], target.double())
    assert abs(float(loss) - float(loss_ref)) <= 1e-5 * abs(float(loss_ref))
    loss.backward()
    loss_ref.backward()
    ref_params = dict(orc.named_parameters())
    worst = 0.0
    for name, p in prod.named_parameters():
        r = ref_params["mods." + name]
        assert p.grad is not None, name
        if float(r.grad.norm()) == 0.0:
            assert float(p.grad.norm()) == 0.0, name
            continue
        err = rel_err(p.grad, r.grad)
        worst = max(worst, err)
        assert err < GTOL, (name, err)
    assert worst > 0.0


def test_shipped_config_energy_forward(dev):
    """config_energy exactly as shipped (n_dim 64, l_max 3, 5 layers, per-species shifts)."""
    from e3_layers_amd.configs import config_energy
    from e3_layers_amd.data.synthetic import synth_qm9

    tree = config_energy.get_config().model_config
    prod, orc = _build_pair(tree, dev)
    batch = synth_qm9(5, 3, config_energy.QM9_SHIFTS)
    data, attrs = batch_to_oracle(batch)
    out_ref, _ = orc(data, attrs)
    with torch.no_grad():
        out = prod(batch.clone().to(dev))
    assert rel_err(out["node_features"], out_ref["node_features"]) < TOL
    assert rel_err(out["total_energy"], out_ref["total_energy"]) < TOL
    assert set(out.keys()) >= {"total_energy", "energy", "node_features", "edge_index", "_n_nodes"}


def test_forces_by_autograd(dev):
    """config_energy_force: forces = -dE/dpos through the HIP backward kernels (eval mode)."""
    from e3_layers_amd.configs.layer_configs import addEnergyOutput, addForceOutput, featureModel
    from e3_layers_amd.data.synthetic import synth_qm9
    from e3_layers_amd.utils import build
    from oracle import e3ref

    cfg = featureModel(n_dim=16, l_max=2, edge_spherical="1x0e+1x1o+1x2e", node_attrs="16x0e", edge_radial="8x0e",
                       num_types=10, num_layers=3, r_max=4.0)
    cfg = addForceOutput(addEnergyOutput(cfg, None, output_key="energy_total"), y="energy_total")
    torch.manual_seed(0)
    prod = build(cfg).to(dev).eval()
    orc = e3ref.build(cfg)
    orc.load_state_dict({k.replace("func.", "func.mods.", 1): v.cpu() for k, v in prod.state_dict().items()})
    orc = orc.double().eval()
    batch = synth_qm9(7, 3)
    data, attrs = batch_to_oracle(batch)
    out_ref, _ = orc(data, attrs)
    out = prod(batch.clone().to(dev))
    assert rel_err(out["energy_total"], out_ref["energy_total"]) < TOL
    assert rel_err(out["forces"], out_ref["forces"]) < 5e-5
    # translation invariance: forces of a molecule sum to zero
    seg = out["_node_segment"]
    tot = torch.zeros(len(batch), 3, device=dev).index_add_(0, seg, out["forces"])
    assert float(tot.abs().max()) < 1e-4 * float(out["forces"].abs().max()) + 1e-7


def test_batch_additivity(dev):
    """A batch of k molecules == k single-molecule runs (also pins graph-parallel sharding)."""
    from e3_layers_amd.data.synthetic import synth_qm9
    from e3_layers_amd.utils import build

    tree = _energy_tree(2, 16, 3)
    torch.manual_seed(0)
    prod = build(tree).to(dev)
    batch = synth_qm9(11, 5)
    with torch.no_grad():
        full = prod(batch.clone().to(dev))["total_energy"]
        singles = torch.cat([prod(batch[[i]].to(dev))["total_energy"] for i in range(5)])
    assert rel_err(full, singles) < 1e-5
