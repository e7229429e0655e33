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


@pytest.mark.parametrize("refine", [True, False])
def test_pipelined_step_records_itself_again_after_a_guard_trips(dev, monkeypatch, refine):
    """A guard that trips doubles the knot counts (``backend/radial_table._refine``) or, with no finer table left, switches that
    MLP's table off.  Either way both buffers' steps record themselves again -- each AFTER its preparation graph was recorded again
    on fresh tensors (``PipelinedBucketedStep._record_again``: the finer resolution's bins and edge records; and a step recorded
    on a buffer whose inputs were written to since its preparation was recorded would rebuild the per-tensor memos eagerly in its
    warm-up and replay the batch it was recorded on -- the first version of this did, found by this test).  The replayed losses
    equal the eager model's on the same batches afterwards, whichever batch a buffer held when it was recorded."""
    import warnings

    from e3_layers_amd.backend import ops, radial_table
    from e3_layers_amd.data.synthetic import synth_qm9
    from e3_layers_amd.run.graph_step import PipelinedBucketedStep, bucket_capacity, pad_batch

    every = 2
    monkeypatch.setattr(radial_table, "GUARD_EVERY", every)
    knots0 = radial_table.KNOTS
    if not refine:
        monkeypatch.setattr(radial_table, "KNOTS_MAX", knots0)
    model = _energy_model(dev)
    host = [synth_qm9(61 + k, 128) for k in range(3)]
    n_cap, e_cap = bucket_capacity([(b["pos"].shape[0], b["edge_index"].shape[1]) for b in host])
    assert e_cap >= radial_table.MIN_EDGES_PER_KNOT * (2 * knots0 + 1)
    padded = [pad_batch(b, n_cap, e_cap).to(dev) for b in host]

    def loss_on(batch):
        target, weight = batch["total_energy"], batch["_graph_weight"]
        loss = ops.sq_error(model(batch)["total_energy"], target, weight, 1e3)
        for p in model.parameters():
            p.grad = None
        loss.backward()
        return loss

    step = PipelinedBucketedStep(model.prepare_data, loss_on, padded[0], warmup=2)
    for i in range(4):
        step(padded[i % 3], nxt=padded[(i + 1) % 3])
    torch.cuda.synchronize()
    assert step.recaptures == 0 and radial_table.REFINEMENTS == 0
    first = list(model.layer1.conv.fc.children())[0].weight
    key = radial_table.last_weight(model.layer1.conv.fc)
    scale, i = 1.0, 4
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        while step.recaptures == 0 and scale < 40.0:
            with torch.no_grad():
                first.mul_(1.25)
            scale *= 1.25
            for _ in range(2 * every + 2):
                step(padded[i % 3], nxt=padded[(i + 1) % 3])
                i += 1
                torch.cuda.synchronize()
    if refine:
        assert radial_table.REFINEMENTS == 1 and radial_table.KNOTS == 2 * knots0 and radial_table.guard_ok(key), \
            (scale, [str(w.message)[:160] for w in caught])
        assert any("are rebuilt on" in str(w.message) for w in caught)
    else:
        assert radial_table.REFINEMENTS == 0 and radial_table.KNOTS == knots0 and not radial_table.guard_ok(key), scale
        assert any("per edge from now on" in str(w.message) for w in caught)
    losses = []
    for _ in range(7):                                          # both buffers have recorded themselves again by now, or do so here;
        losses.append((i % 3, float(step(padded[i % 3], nxt=padded[(i + 1) % 3]).detach())))      # every batch meets every buffer
        i += 1
        torch.cuda.synchronize()
    assert step.recaptures == 2                                 # one per buffer
    for b in range(2):                                          # the new preparation graphs build the bins of the resolution in force
        assert set(radial_table.prepared_bins(step.static[b]["edge_length"])) == {radial_table.KNOTS}
    ops.join_side_streams()
    want = [float(loss_on(padded[k].clone()).detach()) for k in range(3)]
    for k, got in losses:
        assert abs(got - want[k]) <= 2e-5 * abs(want[k]), (k, got, want, losses)


def test_inputs_overwritten_in_place_after_the_preparation_are_prepared_again(dev):
    """``prepare_data(batch)`` and then the batch's tensors are WRITTEN to (another batch copied in): the version counters say so, the
    forward drops what the layers run ahead left in the batch (edge vectors -- which ``computeEdgeVector`` would otherwise keep, as
    the reference's does --, spherical harmonics, one-hot, the CSR views) and runs them again."""
    from e3_layers_amd.data.synthetic import synth_qm9
    from e3_layers_amd.run.graph_step import bucket_capacity, pad_batch

    model = _energy_model(dev)
    host = [synth_qm9(61 + k, 64) for k in range(2)]
    n_cap, e_cap = bucket_capacity([(b["pos"].shape[0], b["edge_index"].shape[1]) for b in host])
    padded = [pad_batch(b, n_cap, e_cap).to(dev) for b in host]
    a = padded[0].clone()
    model.prepare_data(a)
    for k in padded[1].keys():
        if torch.is_tensor(padded[1][k]):
            a[k].copy_(padded[1][k])
    with torch.no_grad():
        got = model(a.view())["total_energy"]
        want = model(padded[1].clone())["total_energy"]
    assert torch.equal(got, want)
