"""Batch preparation ahead of the step (``SequentialGraphNetwork.prepare_data``, ``run/graph_step.PipelinedBucketedStep``):
what the forward derives from the batch alone may be computed before the step -- and must change nothing."""
import copy

import pytest
import torch

from tests.util import rel_err

pytestmark = pytest.mark.gpu


def _energy_model(dev, n_dim=64, layers=3):
    from e3_layers_amd.configs.layer_configs import addEnergyOutput, featureModel
    from e3_layers_amd.utils import build

    torch.manual_seed(4)
    tree = addEnergyOutput(featureModel(n_dim=n_dim, l_max=2, edge_spherical="1x0e+1x1o+1x2e", node_attrs="20x0e", edge_radial="8x0e",
                                        num_types=10, num_layers=layers, r_max=4.0), None)
    return build(tree).to(dev).train()


def test_prepared_forward_equals_the_plain_forward(dev):
    """``prepare_data`` runs edge vectors / spherical harmonics / one-hot / CSR / species groups / knot bins / edge records ahead;
    the forward then skips exactly those layers and finds the builds -- outputs and every parameter gradient bit-identical to
    the un-prepared forward; a batch whose ``pos`` is REPLACED after the preparation (a denoising loss perturbs it) gets its
    geometry layers run again."""
    from e3_layers_amd.backend import radial_table
    from e3_layers_amd.data.synthetic import synth_qm9

    model = _energy_model(dev)
    batch = synth_qm9(5, 96).to(dev)
    assert batch["edge_index"].shape[1] >= radial_table.MIN_EDGES_PER_KNOT * (radial_table.KNOTS + 1)

    def run(b):
        for p in model.parameters():
            p.grad = None
        out = model(b)
        out["total_energy"].square().sum().backward()
        torch.cuda.synchronize()
        return out, {n: p.grad.clone() for n, p in model.named_parameters()}

    ref, g_ref = run(batch.clone())
    prepared = batch.clone()
    done = model.prepare_data(prepared)
    assert {"edge_vector", "onehot", "spharm_edges"} <= set(done), done
    assert radial_table.prepared_bins(prepared["edge_length"]) is not None
    bins = radial_table.prepared_bins(prepared["edge_length"])[radial_table.KNOTS]
    assert len(bins._rec) == 2                               # both walks' edge records were built ahead
    out, g = run(prepared)
    assert len(bins._rec) == 2                               # ... and the step reused them
    for key in ("edge_vector", "edge_spherical", "node_attrs", "node_features", "total_energy"):
        assert torch.equal(out[key], ref[key]), key
    for n in g_ref:
        assert rel_err(g[n], g_ref[n]) < 2e-6, n             # (fp32 atomics in the weight-gradient GEMMs: not bit-reproducible)

    # a second forward on the same object does not skip anything (the record is consumed) and still agrees
    out2 = model(prepared.view())
    assert torch.equal(out2["total_energy"], ref["total_energy"])

    # ``pos`` replaced after the preparation: the record of the geometry layers no longer matches the tensors
    moved = batch.clone()
    model.prepare_data(moved)
    stale_vec = moved["edge_vector"]
    moved["pos"] = moved["pos"] * 1.01
    for k in ("edge_vector", "edge_length", "edge_spherical"):
        moved.data.pop(k)
    out3 = model(moved)
    fresh = batch.clone()
    fresh["pos"] = fresh["pos"] * 1.01
    ref3 = model(fresh)
    assert not torch.equal(out3["edge_vector"], stale_vec)
    assert torch.equal(out3["total_energy"], ref3["total_energy"])


def test_pipelined_preparation_follows_the_plain_replay(dev):
    """Six optimizer steps through ``PipelinedBucketedStep`` (the batch-only work of batch t + 1 as its own graph on a second stream
    beside the step of batch t, two static buffers) against the same six steps through ``BucketedStep`` (everything in one graph)
    from the same initial weights: losses, parameters, EMA; an un-announced batch is prepared in line."""
    from e3_layers_amd.backend import ops
    from e3_layers_amd.data.synthetic import synth_qm9
    from e3_layers_amd.run.graph_step import BucketedStep, PipelinedBucketedStep, bucket_capacity, pad_batch
    from e3_layers_amd.run.optim import FusedAdamEMA
    from e3_layers_amd.run.parallel import flat_param_order

    base = _energy_model(dev)
    host = [synth_qm9(31 + k, 48) for k in range(3)]
    order = [0, 1, 2, 1, 0, 2]
    n_cap, e_cap = bucket_capacity([(b["pos"].shape[0], b["edge_index"].shape[1]) for b in host])
    padded = [pad_batch(b, n_cap, e_cap).to(dev) for b in host]

    def trajectory(pipelined: bool):
        model = copy.deepcopy(base)
        opt = FusedAdamEMA(flat_param_order(model), lr=1e-3, ema_decay=0.99)
        opt.grads.enable_direct_accumulation()
        try:
            start = opt.flat.detach().clone()
            state0 = {k: getattr(opt, k).detach().clone() for k in ("exp_avg", "exp_avg_sq", "ema", "state")}

            def train_on(batch):
                target, weight = batch["total_energy"], batch["_graph_weight"]
                loss = ops.sq_error(model(batch)["total_energy"], target, weight, 1e3)
                opt.zero_grad()
                loss.backward()
                opt.step()
                return loss

            step = (PipelinedBucketedStep(model.prepare_data, train_on, padded[0], warmup=2) if pipelined
                    else BucketedStep(train_on, padded[0], warmup=2))
            with torch.no_grad():      # (the warm-ups and the captures took optimizer steps: rewind)
                opt.flat.copy_(start)
                for k, v in state0.items():
                    getattr(opt, k).copy_(v)
            losses = []
            for i, k in enumerate(order):
                if pipelined:
                    # announce the next batch -- except once (i == 2), where the step must prepare its batch in line
                    nxt = padded[order[i + 1]] if (i + 1 < len(order) and i != 2) else None
                    losses.append(float(step(padded[k], nxt=nxt).detach()))
                else:
                    losses.append(float(step(padded[k]).detach()))
            ops.join_side_streams()
            torch.cuda.synchronize()
            return losses, opt.flat.detach().clone(), opt.ema.detach().clone(), start
        finally:
            opt.grads.disable_direct_accumulation()

    l_p, flat_p, ema_p, start = trajectory(True)
    l_b, flat_b, ema_b, _ = trajectory(False)
    for a, b in zip(l_p, l_b):
        assert abs(a - b) <= 2e-5 * abs(b), (l_p, l_b)
    assert rel_err(flat_p - start, flat_b - start) < 2e-3      # (Adam amplifies the rounding of tiny gradients: the updates' bulk)
    assert rel_err(ema_p, ema_b) < 1e-6
