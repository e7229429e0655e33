"""Full-size GPU checks at BASELINE.json's configurations, through size-independent properties (the float64
oracle cannot run these sizes in seconds): O(3) invariance / equivariance, translation and edge-order invariance,
additivity over the molecules of a batch, Newton's third law for the forces — the invariants SURVEY.md §8c lists."""
import math

import pytest
import torch

from tests.util import rel_err

pytestmark = pytest.mark.gpu


def _rotation(seed):
    g = torch.Generator().manual_seed(seed)
    q, r = torch.linalg.qr(torch.randn(3, 3, generator=g, dtype=torch.float64))
    q = q * torch.sign(torch.diagonal(r))
    if torch.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q.float()


def _energy_model(dev, l_max):
    from e3_layers_amd.configs import config_energy
    from e3_layers_amd.utils import build

    torch.manual_seed(0)
    return build(config_energy.get_config(l_max=l_max).model_config).to(dev).eval()


@pytest.mark.parametrize("l_max,n_mol", [(2, 256), (3, 64)])
def test_config_energy_full_size_invariances(dev, l_max, n_mol):
    """BASELINE configs[1] (l_max 2, 256 molecules; also the shipped l_max 3): energies are invariant under a proper
    rotation, an inversion, a translation and a permutation of the edge list, and additive over the batch."""
    from e3_layers_amd.configs import config_energy
    from e3_layers_amd.data.synthetic import synth_qm9

    model = _energy_model(dev, l_max)
    batch = synth_qm9(1000, n_mol, config_energy.QM9_SHIFTS).to(dev)
    with torch.no_grad():
        e0 = model(batch.clone())["total_energy"]
        scale = float(e0.abs().mean())
        rot = _rotation(1).to(dev)
        b = batch.clone()
        b["pos"] = batch["pos"] @ rot.t()
        assert float((model(b)["total_energy"] - e0).abs().max()) < 2e-5 * scale
        b = batch.clone()
        b["pos"] = -batch["pos"]                                   # parity: 'o' irreps flip, scalars do not
        assert float((model(b)["total_energy"] - e0).abs().max()) < 2e-5 * scale
        b = batch.clone()
        b["pos"] = batch["pos"] + torch.tensor([3.0, -2.0, 0.5], device=dev)
        assert float((model(b)["total_energy"] - e0).abs().max()) < 2e-5 * scale
        # edge order inside each graph shuffled (the CSR build sorts them back; sums change only in rounding)
        b = batch.clone()
        ei, seg = batch["edge_index"], batch["_edge_segment"]
        key = seg.double() + torch.rand(ei.shape[1], device=dev, dtype=torch.float64) * 0.5
        perm = torch.argsort(key)
        b["edge_index"] = ei[:, perm]
        assert float((model(b)["total_energy"] - e0).abs().max()) < 2e-5 * scale
        # additivity: two halves of the batch run separately
        half = n_mol // 2
        lo = model(batch[list(range(half))].to(dev))["total_energy"]
        hi = model(batch[list(range(half, n_mol))].to(dev))["total_energy"]
        assert float((torch.cat([lo, hi]) - e0).abs().max()) < 2e-5 * scale


def test_config_energy_force_full_size_equivariance(dev):
    """BASELINE configs[2] (config_energy_force model, 64 molecules): forces rotate with the frame, sum to zero per
    molecule, and are the negative finite-difference slope of the energy along a random direction."""
    from e3_layers_amd.configs import config_energy_force
    from e3_layers_amd.data.synthetic import synth_qm9
    from e3_layers_amd.utils import build

    torch.manual_seed(0)
    cfg = config_energy_force.get_config()
    model = build(cfg.model_config).to(dev).eval()
    batch = synth_qm9(2000, 64, r_max=5.0).to(dev)
    out = model(batch.clone())
    f0 = out["forces"].detach()
    fscale = float(f0.abs().mean())
    rot = _rotation(2).to(dev)
    b = batch.clone()
    b["pos"] = batch["pos"] @ rot.t()
    f1 = model(b)["forces"].detach()
    assert float((f1 - f0 @ rot.t()).abs().max()) < 5e-5 * fscale + 1e-6
    seg = out["_node_segment"]
    tot = torch.zeros(64, 3, device=dev).index_add_(0, seg, f0)
    assert float(tot.abs().max()) < 1e-4 * float(f0.abs().max())
    # F = -dE/dx: move one atom along x and difference its molecule's energy.  The per-species shifts (-620 per
    # atom) would bury a 1e-5 energy change under fp32 resolution: zero them for this part (forces do not see them).
    for m in model.modules():
        if hasattr(m, "shifts") and isinstance(m.shifts, torch.Tensor):
            m.shifts.zero_()
    atom, eps = 5, 1e-2
    mol = int(seg[atom])
    with torch.no_grad():
        bp, bm = batch.clone(), batch.clone()
        bp["pos"][atom, 0] += eps
        bm["pos"][atom, 0] -= eps
        ep = model.func(bp)["energy"][mol, 0]
        em = model.func(bm)["energy"][mol, 0]
    slope = float((ep - em) / (2 * eps))
    pred = -float(f0[atom, 0])
    assert abs(slope - pred) < 3e-2 * abs(pred) + 1e-2 * fscale


def test_config_diffusion_full_size_score_equivariance(dev):
    """BASELINE configs[3] (config_diffusion score net, 128 fully connected molecules): the score is a vector field —
    it rotates with the frame, flips under inversion, ignores translations."""
    from e3_layers_amd.configs import config_diffusion
    from e3_layers_amd.data.synthetic import synth_qm9_diffusion
    from e3_layers_amd.utils import build

    torch.manual_seed(0)
    model = build(config_diffusion.get_config().model_config).to(dev).eval()
    batch = synth_qm9_diffusion(7, 128).to(dev)
    with torch.no_grad():
        s0 = model(batch.clone())["score"]
        rot = _rotation(4).to(dev)
        b = batch.clone()
        b["pos"] = batch["pos"] @ rot.t()
        assert rel_err(model(b)["score"], s0 @ rot.t()) < 3e-4      # fp32 edge vectors re-rounded in the new frame
        b = batch.clone()
        b["pos"] = -batch["pos"]
        assert rel_err(model(b)["score"], -s0) < 1e-5                # exact sign symmetry of every kernel
        b = batch.clone()
        b["pos"] = batch["pos"] + 1.5
        assert rel_err(model(b)["score"], s0) < 3e-4
