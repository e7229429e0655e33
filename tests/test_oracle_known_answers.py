"""Pins of the oracle (CPU).  The reference has no tests and its e3nn dependency cannot be
imported, so the restatement is pinned by the closed forms of SURVEY.md
appendix A and by group-theoretic invariants: what e3nn 0.4.4 documents for these operators."""
import math

import pytest
import torch

from oracle import e3ref

torch.set_default_dtype(torch.float32)


def rot(a, b, c):
    def rz(t):
        return torch.tensor([[math.cos(t), -math.sin(t), 0], [math.sin(t), math.cos(t), 0], [0, 0, 1]], dtype=torch.float64)

    def ry(t):
        return torch.tensor([[math.cos(t), 0, math.sin(t)], [0, 1, 0], [-math.sin(t), 0, math.cos(t)]], dtype=torch.float64)

    return rz(a) @ ry(b) @ rz(c)


def wigner_D(l, R):
    """D^l(R) in the SH basis, recovered from Y^l(R v) = D^l(R) Y^l(v) on random directions."""
    g = torch.Generator().manual_seed(l)
    v = torch.randn(200, 3, generator=g, dtype=torch.float64)
    y0 = e3ref.spherical_harmonics([l], v)
    y1 = e3ref.spherical_harmonics([l], v @ R.T)
    return torch.linalg.lstsq(y0, y1).solution.T


# ---- A.2 Wigner 3j -------------------------------------------------------------------------
def test_wigner_known_answers():
    w = e3ref.wigner_3j
    assert torch.allclose(w(0, 0, 0), torch.ones(1, 1, 1, dtype=torch.float64))
    for l in range(4):
        assert torch.allclose(w(l, 0, l)[:, 0, :], torch.eye(2 * l + 1, dtype=torch.float64) / math.sqrt(2 * l + 1))
    assert torch.allclose(w(1, 1, 0)[:, :, 0], torch.eye(3, dtype=torch.float64) / math.sqrt(3))
    c = w(1, 1, 1)
    eps = torch.zeros(3, 3, 3, dtype=torch.float64)
    for i, j, k, s in [(0, 1, 2, 1), (1, 2, 0, 1), (2, 0, 1, 1), (0, 2, 1, -1), (2, 1, 0, -1), (1, 0, 2, -1)]:
        eps[i, j, k] = s
    assert torch.allclose(c, eps / math.sqrt(6))
    assert abs(float(c[0, 1, 2]) - 0.40825) < 1e-5
    c = w(1, 1, 2)
    expect = {(0, 0, 2): -0.18257, (2, 2, 2): -0.18257, (1, 1, 2): 0.36515, (0, 2, 0): 0.31623, (2, 0, 0): 0.31623,
              (0, 1, 1): 0.31623, (1, 0, 1): 0.31623, (1, 2, 3): 0.31623, (2, 1, 3): 0.31623, (0, 0, 4): -0.31623,
              (2, 2, 4): 0.31623}
    for idx, val in expect.items():
        assert abs(float(c[idx]) - val) < 1e-5, idx
    assert int((c.abs() > 1e-12).sum()) == len(expect)


@pytest.mark.parametrize("l1,l2,l3", [(1, 1, 2), (2, 1, 3), (2, 2, 2), (3, 2, 1), (3, 2, 3), (2, 2, 0)])
def test_wigner_norm_and_invariance(l1, l2, l3):
    c = e3ref.wigner_3j(l1, l2, l3)
    assert abs(float(c.norm()) - 1.0) < 1e-12
    R = rot(0.4, 1.2, -0.9)
    d1, d2, d3 = wigner_D(l1, R), wigner_D(l2, R), wigner_D(l3, R)
    rotated = torch.einsum("ia,jb,kc,abc->ijk", d1, d2, d3, c)
    assert torch.allclose(rotated, c, atol=1e-9)


def test_product_package_wigner_equals_oracle():
    from e3_layers_amd.o3 import wigner_3j

    for l1 in range(4):
        for l2 in range(3):
            for l3 in range(abs(l1 - l2), l1 + l2 + 1):
                a = torch.from_numpy(wigner_3j(l1, l2, l3).copy())
                assert torch.allclose(a, e3ref.wigner_3j(l1, l2, l3), atol=1e-13), (l1, l2, l3)


# ---- A.3 spherical harmonics ------------------------------------------------------------------
def test_sh_closed_forms_and_norm():
    g = torch.Generator().manual_seed(0)
    v = torch.randn(50, 3, generator=g, dtype=torch.float64)
    u = v / v.norm(dim=1, keepdim=True)
    x, y, z = u[:, 0], u[:, 1], u[:, 2]
    sh = e3ref.spherical_harmonics([0, 1, 2, 3], v)
    assert torch.allclose(sh[:, 0], torch.ones(50, dtype=torch.float64))
    assert torch.allclose(sh[:, 1:4], math.sqrt(3) * u)
    y2 = torch.stack([math.sqrt(15) * x * z, math.sqrt(15) * x * y, math.sqrt(5) * (y * y - 0.5 * (x * x + z * z)),
                      math.sqrt(15) * y * z, 0.5 * math.sqrt(15) * (z * z - x * x)], dim=1)
    assert torch.allclose(sh[:, 4:9], y2)
    for l, sl in [(0, slice(0, 1)), (1, slice(1, 4)), (2, slice(4, 9)), (3, slice(9, 16))]:
        assert torch.allclose((sh[:, sl] ** 2).sum(1), torch.full((50,), 2.0 * l + 1, dtype=torch.float64))
    # polar axis is y: Y^l(0,1,0) is concentrated on m = 0 (the middle component), positive
    pole = e3ref.spherical_harmonics([1, 2, 3], torch.tensor([[0.0, 1.0, 0.0]], dtype=torch.float64))[0]
    assert torch.allclose(pole[0:3], torch.tensor([0, math.sqrt(3), 0], dtype=torch.float64))
    assert torch.allclose(pole[3:8], torch.tensor([0, 0, math.sqrt(5), 0, 0], dtype=torch.float64))
    assert torch.allclose(pole[8:15], torch.tensor([0, 0, 0, math.sqrt(7), 0, 0, 0], dtype=torch.float64))


@pytest.mark.parametrize("l", [0, 1, 2])
def test_sh_recurrence_with_3j(l):
    """Y^{l+1} = c_l * C^{l,1,l+1} . Y^l . Y^1 with c_l > 0: ties the 3j sign convention to the SH one."""
    g = torch.Generator().manual_seed(1)
    v = torch.randn(40, 3, generator=g, dtype=torch.float64)
    yl = e3ref.spherical_harmonics([l], v)
    y1 = e3ref.spherical_harmonics([1], v)
    yn = e3ref.spherical_harmonics([l + 1], v)
    t = torch.einsum("ijk,zi,zj->zk", e3ref.wigner_3j(l, 1, l + 1), yl, y1)
    ratio = (yn * t).sum(1) / (t * t).sum(1)
    assert float(ratio.min()) > 0
    assert torch.allclose(t * ratio[:, None], yn, atol=1e-10)
    assert float(ratio.std()) < 1e-10


def test_sh_normalizations_and_odd_parity():
    v = torch.randn(10, 3, dtype=torch.float64)
    comp = e3ref.spherical_harmonics([2], v, True, "component")
    assert torch.allclose(e3ref.spherical_harmonics([2], v, True, "integral"), comp / math.sqrt(4 * math.pi))
    assert torch.allclose(e3ref.spherical_harmonics([2], v, True, "norm"), comp / math.sqrt(5))
    for l in (1, 2, 3):
        a = e3ref.spherical_harmonics([l], v)
        b = e3ref.spherical_harmonics([l], -v)
        assert torch.allclose(b, (-1) ** l * a)


# ---- A.4 / A.5 -----------------------------------------------------------------------------------
def test_radial_basis_known_answers():
    c = e3ref.poly_cutoff
    assert float(c(torch.tensor([0.0]), 0.25)) == 1.0
    assert float(c(torch.tensor([4.0]), 0.25)) == 0.0
    assert abs(float(c(torch.tensor([3.999], dtype=torch.float64), 0.25))) < 1e-9
    x = torch.tensor([0.5], dtype=torch.float64)
    assert abs(float(c(x * 4, 0.25)) - (1 - 28 * 0.5 ** 6 + 48 * 0.5 ** 7 - 21 * 0.5 ** 8)) < 1e-12
    s = e3ref.symmetric_cutoff
    assert float(s(torch.tensor([0.0]), 1.0)) == 1.0 and float(s(torch.tensor([-1.5]), 1.0)) == 0.0
    b = e3ref.BesselBasis(4.0, 0, 8, trainable=True).double()
    r = torch.tensor([1.3], dtype=torch.float64)
    n = torch.arange(1, 9, dtype=torch.float64)
    assert torch.allclose(b(r)[0], (2.0 / 4.0) * torch.sin(n * math.pi * 1.3 / 4.0) / 1.3)
    assert torch.allclose(b.bessel_weights.detach(), (n * math.pi).to(b.bessel_weights.dtype))


def test_activation_constants():
    # SURVEY.md A.5 (values obtained with this torch): ssp 1.87820, silu 1.67918, tanhlu 1.15019, tanh 1.59373
    expect = {"ssp": 1.87820, "silu": 1.67918, "tanhlu": 1.15019, "tanh": 1.59373, "abs": 1.00111}
    for name, val in expect.items():
        assert abs(e3ref.act_norm_const(name) - val) < 2e-5, (name, e3ref.act_norm_const(name))
    from e3_layers_amd.utils import act_second_moment_const

    for name in expect:
        assert act_second_moment_const(name) == e3ref.act_norm_const(name)
    assert e3ref.act_parity("tanhlu", -1) == -1 and e3ref.act_parity("abs", -1) == 1
    with pytest.raises(ValueError):
        e3ref.act_parity("silu", -1)


def test_linear_semantics():
    torch.manual_seed(0)
    lin = e3ref.Linear("4x0e+3x1o+2x0e", "5x0e+2x1o+1x2e", biases=True).double()
    assert lin.weight.numel() == 4 * 5 + 3 * 2 + 2 * 5 and lin.bias.numel() == 5
    x = torch.randn(7, 4 + 9 + 2, dtype=torch.float64)
    y = lin(x)
    w = lin.weight.detach()
    w00, w11, w20 = w[:20].view(4, 5), w[20:26].view(3, 2), w[26:36].view(2, 5)
    pw0 = 1 / math.sqrt(6)  # fan-in of the 0e output = 4 + 2
    expect0 = pw0 * (x[:, :4] @ w00 + x[:, 13:15] @ w20) + lin.bias
    assert torch.allclose(y[:, :5], expect0)
    x1 = x[:, 4:13].view(7, 3, 3)
    expect1 = torch.einsum("uw,zum->zwm", w11, x1) / math.sqrt(3)
    assert torch.allclose(y[:, 5:11], expect1.reshape(7, 6))
    assert torch.all(y[:, 11:] == 0)  # 2e has no input path


def test_fctp_scalar_attrs_semantics():
    torch.manual_seed(0)
    tp = e3ref.FullyConnectedTensorProduct("3x1o+2x0e", "4x0e", "5x1o+2x0e").double()
    x, a = torch.randn(6, 11, dtype=torch.float64), torch.randn(6, 4, dtype=torch.float64)
    y = tp(x, a)
    w = tp.weight.detach()
    w1 = w[:60].view(3, 4, 5)
    expect = torch.einsum("uvw,zuk,zv->zwk", w1, x[:, :9].view(6, 3, 3), a) / math.sqrt(12)
    assert torch.allclose(y[:, :15], expect.reshape(6, 15))


def test_uvu_path_normalisation_and_variance():
    """component normalisation: unit-variance inputs and weights give ~unit-variance outputs."""
    torch.manual_seed(0)
    mod = e3ref.TensorProductExpansion("32x0e+32x1o", ("1x0e+1x1o+1x2e", "s"), ("32x0e+32x1o+32x2e", "o"), "uvu", False)
    z = 4000
    x = torch.randn(z, 32 * 4)
    sh = e3ref.spherical_harmonics([0, 1, 2], torch.randn(z, 3))
    w = torch.randn(z, mod.tp.weight_numel)
    y = mod.tp(x, sh, w)
    assert 0.8 < float(y.pow(2).mean()) < 1.25
    assert mod.tp.coeff == pytest.approx([math.sqrt(2 * lo + 1) for lo in [mod.tp.out[i][1] for *_, i, _ in [(0, 0, ins[2], 0) for ins in mod.tp.instr]]])


def test_gate_layout():
    g = e3ref.Gate("2x0e+1x0o", ["silu", "tanhlu"], "2x0e", ["silu", "silu"], "1x1o+1x2e")
    assert e3ref.irreps_str(g.irreps_in) == "2x0e+1x0o+2x0e+1x1o+1x2e"
    assert e3ref.irreps_str(g.irreps_out) == "2x0e+1x0o+1x1o+1x2e"
    x = torch.randn(3, 2 + 1 + 2 + 3 + 5, dtype=torch.float64)
    y = g(x)
    cs, ct = e3ref.act_norm_const("silu"), e3ref.act_norm_const("tanhlu")
    assert torch.allclose(y[:, :2], cs * torch.nn.functional.silu(x[:, :2]))
    assert torch.allclose(y[:, 2], ct * torch.tanh(x[:, 2]) * x[:, 2].abs())
    assert torch.allclose(y[:, 3:6], x[:, 5:8] * (cs * torch.nn.functional.silu(x[:, 3:4])))
    assert torch.allclose(y[:, 6:11], x[:, 8:13] * (cs * torch.nn.functional.silu(x[:, 4:5])))


def test_scatter_matches_index_add_order():
    src = torch.randn(10, 3)
    idx = torch.tensor([0, 2, 2, 1, 0, 2, 1, 1, 0, 2])
    out = e3ref.scatter(src, idx, dim_size=4)
    manual = torch.zeros(4, 3)
    for e in range(10):
        manual[idx[e]] += src[e]
    assert torch.equal(out, manual)
    assert torch.all(out[3] == 0)
    mean = e3ref.scatter(src, idx, dim_size=4, reduce="mean")
    assert torch.allclose(mean[2], src[idx == 2].mean(0))


# ---- whole-network invariants --------------------------------------------------------------------
def _small_oracle(l_max=2):
    from e3_layers_amd.configs.layer_configs import addEnergyOutput, featureModel

    tree = addEnergyOutput(featureModel(n_dim=8, l_max=l_max, edge_spherical="1x0e+1x1o+1x2e", node_attrs="6x0e",
                                        edge_radial="8x0e", num_types=10, num_layers=3, r_max=4.0), None)
    torch.manual_seed(0)
    return e3ref.build(tree).double(), tree


def _inputs(n_mol=3, seed=2):
    from e3_layers_amd.data.synthetic import synth_qm9

    b = synth_qm9(seed, n_mol)
    return {k: (v.double() if v.is_floating_point() else v) for k, v in b.data.items()}, dict(b.attrs)


@pytest.mark.parametrize("l_max", [2, 3])
def test_network_equivariance_parity_translation(l_max):
    net, _ = _small_oracle(l_max)
    data, attrs = _inputs()
    out, oattrs = net(data, attrs)
    R = rot(0.3, 1.1, -0.7)
    for transform, sign in ((lambda p: p @ R.T + 0.7, 1), (lambda p: -(p @ R.T), -1)):
        d2 = dict(data)
        d2["pos"] = transform(data["pos"])
        out2, _ = net(d2, attrs)
        assert torch.allclose(out2["total_energy"], out["total_energy"], atol=1e-12)
        pos = 0
        for mul, l, p in e3ref.parse_irreps(oattrs["node_features"][1]):
            d = 2 * l + 1
            a = out["node_features"][:, pos:pos + mul * d].reshape(-1, mul, d)
            b = out2["node_features"][:, pos:pos + mul * d].reshape(-1, mul, d)
            D = wigner_D(l, R) * (p if sign == -1 else 1)
            assert torch.allclose(b, a @ D.T, atol=1e-10), (l, p)
            pos += mul * d


def test_network_permutation_and_additivity():
    net, _ = _small_oracle()
    data, attrs = _inputs(3)
    out, _ = net(data, attrs)
    # permute the edges: node outputs unchanged
    perm = torch.randperm(data["edge_index"].shape[1], generator=torch.Generator().manual_seed(0))
    d2 = dict(data)
    d2["edge_index"] = data["edge_index"][:, perm]
    out2, _ = net(d2, attrs)
    assert torch.allclose(out2["node_features"], out["node_features"], atol=1e-12)
    # a batch of k molecules == k single-molecule runs
    from e3_layers_amd.data.synthetic import synth_qm9

    b = synth_qm9(2, 3)
    singles = []
    for i in range(3):
        s = b[[i]]
        sd = {k: (v.double() if v.is_floating_point() else v) for k, v in s.data.items()}
        singles.append(net(sd, dict(s.attrs))[0]["total_energy"])
    assert torch.allclose(torch.cat(singles), out["total_energy"], atol=1e-12)


def test_linear_commutes_with_scatter():
    """The re-ordering the fused design relies on: Linear(scatter(x)) == scatter(Linear(x))."""
    torch.manual_seed(0)
    lin = e3ref.Linear("6x0e+4x1o", "3x0e+5x1o").double()
    x = torch.randn(40, 18, dtype=torch.float64)
    idx = torch.randint(7, (40,))
    assert torch.allclose(lin(e3ref.scatter(x, idx, 7)), e3ref.scatter(lin(x), idx, 7), atol=1e-12)


def test_forces_match_finite_differences():
    from e3_layers_amd.configs.layer_configs import addEnergyOutput, addForceOutput, featureModel

    cfg = featureModel(n_dim=8, l_max=2, edge_spherical="1x0e+1x1o+1x2e", node_attrs="6x0e", edge_radial="8x0e",
                       num_types=10, num_layers=3, r_max=4.0)
    cfg = addForceOutput(addEnergyOutput(cfg, None, output_key="e_tot"), y="e_tot")
    torch.manual_seed(0)
    net = e3ref.build(cfg).double().eval()
    data, attrs = _inputs(1, seed=4)
    out, _ = net(data, attrs)
    f = out["forces"]
    eps = 1e-5
    for (i, c) in [(0, 0), (2, 1), (5, 2)]:
        dp, dm = dict(data), dict(data)
        dp["pos"] = data["pos"].clone()
        dm["pos"] = data["pos"].clone()
        dp["pos"][i, c] += eps
        dm["pos"][i, c] -= eps
        ep = net(dp, attrs)[0]["e_tot"].sum()
        em = net(dm, attrs)[0]["e_tot"].sum()
        fd = -(ep - em) / (2 * eps)
        assert abs(float(fd) - float(f[i, c])) < 1e-6 * max(1.0, abs(float(fd)))
