"""Parity AFTER the weights have moved.

Every other model-level parity test compares the HIP path with the float64 oracle at random initialisation.  The radial knot
table -- a cubic interpolant whose two small Taylor coefficients are stored as an fp16 pair -- is an approximation whose
a-posteriori guard (``backend/radial_table.py``: 1e-6 table-wide, 2e-5 per column) was calibrated there; trained radial MLPs are
sharper than random ones.  Here the bench's own step -- the HIP-graph replay of padded fresh batches, ``FusedAdamEMA`` at the
SHIPPED learning rate (``e3_layers/configs/config_energy.py:15``: 1e-2) -- is taken >= 200 times, the parameters are copied into
the oracle, and the network's own output (per-species shifts zeroed: they are -1e4 eV per molecule and would turn a relative
bound into 0.1 eV of slack), its node features and every parameter gradient are compared with the table ON; the guard's two
ratios before / after go to ``profiles/r06_parity_measured.jsonl`` (``E3K_PARITY_LOG``).
"""
import pytest
import torch

from oracle import e3ref
from tests.util import batch_to_oracle, oracle_like, record_measured, rel_err, zero_shifts as _zero_shifts

pytestmark = pytest.mark.gpu

TOL = 1e-5    # forward quantities (north star)
GTOL = 5e-5   # parameter gradients
STEPS = 200


def _guard_ratios(model, r_max, knots, dev):
    """Per message-passing layer: (table-wide ratio, per-column ratio, kernel's last estimate, guard still on) of the knot table its
    radial MLP produces NOW -- the two quantities ``e3k_rtable_guard`` bounds (3/128 max|d4 T| against max|T|, resp. against the
    column's own maximum floored at 2^-7 of the table's), formed here in float64 from the fp32 table."""
    from e3_layers_amd.backend import ops, radial_table
    from e3_layers_amd.nn.message_passing import MessagePassing

    net = getattr(model, "func", model)
    enc = net.radial_basis
    b, c = enc.basis, enc.cutoff
    radii = radial_table.knot_radii(r_max, knots, dev)
    out = []
    with torch.no_grad():
        rows = ops.radial_basis(radii, b.bessel_weights, b.r_max, b.r_min, c.p, b.one_over_r, c.cutoff.kind)
        for name, m in net.named_children():
            if not isinstance(m, MessagePassing):
                continue
            t = m.conv.fc(rows).double()
            d4 = t[4:] - 4 * t[3:-1] + 6 * t[2:-2] - 4 * t[1:-3] + t[:-4]
            top = t.abs().amax()
            wide = float(d4.abs().amax() * 3 / 128 / top)
            col = float((d4.abs().amax(0) * 3 / 128 / t.abs().amax(0).clamp_min(radial_table.GUARD_COL_FLOOR * top)).max())
            key = radial_table.last_weight(m.conv.fc)
            out.append({"layer": name, "table_wide": wide, "per_column": col, "kernel_estimate": radial_table.guard_error(key),
                        "ok": bool(radial_table.guard_ok(key)), "max_abs": float(top)})
    return out


def _train_replayed(model, opt, padded, loss_of, steps, generators=()):
    """``steps`` optimizer steps of the bench's default launch mode: the whole step as ONE replayed graph, a different padded batch
    every step.  Returns the BucketedStep (its ``captured.recaptures`` counts knot-table vetoes)."""
    from e3_layers_amd.run.graph_step import BucketedStep

    def train_on(batch):
        loss = loss_of(batch)
        opt.step()
        return loss

    step = BucketedStep(train_on, padded[0], warmup=2, generators=generators)
    losses = []
    for k in range(steps):
        loss = step(padded[k % len(padded)])
        if k % 50 == 0 or k == steps - 1:
            losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    return step, losses


def test_energy_bench_step_after_training_meets_the_oracle(dev):
    """config_energy l_max 2 (BASELINE configs[1]'s network), 4 x 64 molecules, 200 replayed FusedAdamEMA steps at lr 1e-2."""
    from e3_layers_amd.backend import ops, radial_table
    from e3_layers_amd.configs import config_energy
    from e3_layers_amd.data.synthetic import synth_qm9
    from e3_layers_amd.run.graph_step import bucket_capacity, pad_batch
    from e3_layers_amd.run.optim import FusedAdamEMA
    from e3_layers_amd.run.parallel import flat_param_order, param_names
    from e3_layers_amd.utils import build

    cfg = config_energy.get_config(l_max=2)
    tree = cfg.model_config
    torch.manual_seed(0)
    model = build(tree).to(dev).train()
    order = flat_param_order(model)
    opt = FusedAdamEMA(order, lr=cfg.learning_rate, names=param_names(model, order), ema_decay=cfg.ema_decay,
                       ema_use_num_updates=cfg.ema_use_num_updates)
    assert opt.lr == 1e-2
    flat = opt.grads
    flat.enable_direct_accumulation()
    try:
        start = opt.flat.detach().clone()
        before = _guard_ratios(model, 4.0, radial_table.KNOTS, dev)
        host = [synth_qm9(500 + 17 * k, 64, config_energy.QM9_SHIFTS) for k in range(4)]
        n_cap, e_cap = bucket_capacity([(b["pos"].shape[0], b["edge_index"].shape[1]) for b in host])
        padded = [pad_batch(b, n_cap, e_cap).to(dev) for b in host]
        assert e_cap >= radial_table.MIN_EDGES_PER_KNOT * (radial_table.KNOTS + 1)

        def loss_of(batch):      # bench.py's: 1e3 * MSE over the real graphs, loss and its gradient in one launch
            target, weight = batch["total_energy"], batch["_graph_weight"]      # (the model writes its prediction under the same key)
            loss = ops.sq_error(model(batch)["total_energy"], target, weight, 1e3)
            flat.zero()
            loss.backward()
            return loss

        step, losses = _train_replayed(model, opt, padded, loss_of, STEPS)
        radial_table.drain_guards()
        after = _guard_ratios(model, 4.0, radial_table.KNOTS, dev)
        moved = rel_err(opt.flat, start)
        record_measured("trained_energy_guard", steps=STEPS + 3, lr=opt.lr, losses=losses, parameters_moved_rel=moved,
                        recaptures=step.captured.recaptures, before=before, after=after)
        assert moved > 0.05, moved                      # the weights did move
        assert losses[-1] < 0.2 * losses[0], losses     # ... towards the targets
        assert step.captured.recaptures == 0 and all(g["ok"] for g in after), after      # the table stayed on
        assert max(g["per_column"] for g in after) < radial_table.GUARD_TOL_COL
        assert max(g["table_wide"] for g in after) < radial_table.GUARD_TOL

        # ---- the trained network against the oracle: eager training-mode pass on an un-padded batch, shifts zeroed -----
        orc = oracle_like(model, tree)
        _zero_shifts(model, orc)
        batch = host[1]
        dbatch = batch.clone().to(dev)
        out = model(dbatch)
        assert radial_table.applicable(out["edge_radial"])
        probe = torch.randn(batch["total_energy"].shape, generator=torch.Generator().manual_seed(3))
        loss = (probe.to(dev) * out["total_energy"]).sum()
        flat.zero()
        loss.backward()
        ops.join_side_streams()
        torch.cuda.synchronize()
        grads = {name: p.grad.detach().clone() for name, p in model.named_parameters()}
    finally:
        flat.disable_direct_accumulation()
    data, attrs = batch_to_oracle(batch)
    threads = torch.get_num_threads()
    torch.set_num_threads(min(threads, 32))
    try:
        out_ref, _ = orc(data, attrs)
        (probe.double() * out_ref["total_energy"]).sum().backward()
    finally:
        torch.set_num_threads(threads)
    assert float(out_ref["total_energy"].abs().mean()) < 1e3      # (shifts zeroed: the bound below is on the learned part)
    e_err = rel_err(out["total_energy"], out_ref["total_energy"])
    a_err = rel_err(out["energy"], out_ref["energy"])
    f_err = rel_err(out["node_features"], out_ref["node_features"])
    ref_params = dict(orc.named_parameters())
    worst, worst_name, checked = 0.0, None, 0
    for name, g in grads.items():
        r = ref_params["mods." + name].grad
        if r is None or float(r.norm()) == 0.0:
            continue
        err = rel_err(g, r)
        if err > worst:
            worst, worst_name = err, name
        checked += 1
    record_measured("trained_energy_vs_f64_oracle", steps=STEPS + 3, total_energy=e_err, atom_energy=a_err, node_features=f_err,
                    worst_param_grad=worst, worst_param=worst_name, params_checked=checked)
    assert e_err < TOL and a_err < TOL and f_err < TOL, (e_err, a_err, f_err)
    assert worst < GTOL and checked >= 40, (worst_name, worst, checked)


def test_force_training_step_after_training_meets_the_oracle(dev):
    """config_energy_force as shipped (r_max 5: 641-row value AND slope tables), 4 x 24 molecules, 200 replayed steps of the
    energy + force loss (double backward) at the shipped learning rate, then energies / forces / every parameter gradient of a
    training-mode pass against the float64 oracle with both tables ON."""
    from e3_layers_amd.backend import conv_force, ops, radial_table
    from e3_layers_amd.configs import config_energy_force
    from e3_layers_amd.data.synthetic import synth_qm9
    from e3_layers_amd.run.graph_step import bucket_capacity, pad_batch
    from e3_layers_amd.run.optim import FusedAdamEMA
    from e3_layers_amd.run.parallel import backward_parameters, flat_param_order, param_names
    from e3_layers_amd.utils import build

    cfg = config_energy_force.get_config()
    tree = cfg.model_config
    torch.manual_seed(0)
    model = build(tree).to(dev).train()
    order = flat_param_order(model)
    opt = FusedAdamEMA(order, lr=cfg.learning_rate, names=param_names(model, order))
    flat = opt.grads
    flat.enable_direct_accumulation()
    gen = torch.Generator(device=dev)
    gen.manual_seed(9)
    try:
        start = opt.flat.detach().clone()
        before = _guard_ratios(model, 5.0, radial_table.KNOTS_SLOPE, dev)
        host = [synth_qm9(2100 + 17 * k, 24, config_energy_force.SHIFTS, r_max=5.0) for k in range(4)]
        n_cap, e_cap = bucket_capacity([(b["pos"].shape[0], b["edge_index"].shape[1]) for b in host])
        padded = [pad_batch(b, n_cap, e_cap).to(dev) for b in host]
        for p in padded:
            p["forces_target"] = torch.randn(p["pos"].shape, device=dev, generator=gen)
            p.attrs["forces_target"] = ("node", "1x1o")
        rows = radial_table.layout(5.0, radial_table.KNOTS_SLOPE)[0] + 1
        assert rows == 641 and e_cap >= radial_table.MIN_EDGES_PER_KNOT * rows

        def loss_of(batch):      # bench.py --config energy_force (config_energy_force.py:18 loss_coeffs)
            e_t, f_t = batch["total_energy"], batch["forces_target"]
            out = model(batch)
            loss = (ops.sq_error(out["energy"], e_t, batch["_graph_weight"], 1e3)
                    + ops.sq_error(out["forces"], f_t, batch["_node_weight"], 3e4 / 3.0))
            flat.zero()
            backward_parameters(loss, opt.params)
            return loss

        step, losses = _train_replayed(model, opt, padded, loss_of, STEPS)
        radial_table.drain_guards()
        after = _guard_ratios(model, 5.0, radial_table.KNOTS_SLOPE, dev)
        net = model.func
        slope_ok = [bool(radial_table.guard_ok(radial_table.last_weight(getattr(net, f"layer{i}").conv.fc), slope=True))
                    for i in range(tree.num_layers)]
        slope_est = [radial_table.guard_error(radial_table.last_weight(getattr(net, f"layer{i}").conv.fc), slope=True)
                     for i in range(tree.num_layers)]
        moved = rel_err(opt.flat, start)
        record_measured("trained_force_guard", steps=STEPS + 3, lr=opt.lr, losses=losses, parameters_moved_rel=moved,
                        recaptures=step.captured.recaptures, before=before, after=after, slope_ok=slope_ok, slope_estimate=slope_est)
        assert moved > 0.02, moved
        assert step.captured.recaptures == 0 and all(g["ok"] for g in after) and all(slope_ok), (after, slope_ok)

        orc = e3ref.build(tree)
        orc.load_state_dict({k.replace("func.", "func.mods.", 1): v.cpu() for k, v in model.state_dict().items()})
        orc = orc.double().train()
        _zero_shifts(model, orc)      # (-3.7 eV per atom: the energy bound below is on the learned part)
        batch = synth_qm9(2100 + 17, 24, r_max=5.0)
        data, attrs = batch_to_oracle(batch)
        threads = torch.get_num_threads()
        torch.set_num_threads(min(threads, 32))
        try:
            o, _ = orc(data, attrs)
            g2 = torch.Generator().manual_seed(5)
            f_target = o["forces"].detach() + torch.randn(batch["pos"].shape, dtype=torch.float64, generator=g2)
            e_target = o["energy"].detach() + torch.randn(o["energy"].shape, dtype=torch.float64, generator=g2)
            (((o["forces"] - f_target) ** 2).mean() + ((o["energy"] - e_target) ** 2).mean()).backward()
        finally:
            torch.set_num_threads(threads)
        stats = list(conv_force.STATS)
        out = model(batch.clone().to(dev))
        loss = ((out["forces"] - f_target.float().to(dev)) ** 2).mean() + ((out["energy"] - e_target.float().to(dev)) ** 2).mean()
        flat.zero()
        backward_parameters(loss, list(model.parameters()))
        ops.join_side_streams()
        torch.cuda.synchronize()
        assert [a - b for a, b in zip(conv_force.STATS, stats)] == [5, 5, 5]      # all five layers on the force block (tables on)
        err_f, err_e = rel_err(out["forces"], o["forces"]), rel_err(out["energy"], o["energy"])
        ref_params = dict(orc.named_parameters())
        checked, worst, worst_name = 0, 0.0, None
        for name, p in model.named_parameters():
            rp = ref_params[name.replace("func.", "func.mods.", 1)]
            if rp.grad is None or float(rp.grad.abs().max()) == 0.0:
                continue
            err = rel_err(p.grad, rp.grad)
            if err > worst:
                worst, worst_name = err, name
            checked += 1
    finally:
        flat.disable_direct_accumulation()
    record_measured("trained_force_vs_f64_oracle", steps=STEPS + 3, forces=err_f, energy=err_e, worst_param_grad=worst,
                    worst_param=worst_name, params_checked=checked)
    assert err_e < TOL and err_f < TOL, (err_e, err_f)
    assert worst < GTOL and checked >= 40, (worst_name, worst)
