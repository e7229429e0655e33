"""CPU tests of the host side: irreps algebra, config trees, path tables, Data/Batch, edge
construction (bit-exact integer contract), factory helpers.  No kernel is launched."""
import math
from functools import partial

import pytest
import torch

from oracle import e3ref


# ---- irreps ------------------------------------------------------------------------------------
def test_irreps_algebra():
    from e3_layers_amd.o3 import Irrep, Irreps

    ir = Irreps("64x0e+64x1o + 2x2e")
    assert ir.dim == 64 + 192 + 10 and ir.num_irreps == 130 and ir.lmax == 2
    assert str(ir) == "64x0e+64x1o+2x2e" and ir == "64x0e+64x1o+2x2e"
    assert Irrep("1o") in ir and Irrep("1e") not in ir and "2e" in ir
    assert [s.start for s in ir.slices()] == [0, 64, 256]
    assert str(Irreps("4x0e+8x0e+1x1o+2x0e").simplify()) == "12x0e+1x1o+2x0e"   # adjacent only
    assert list(Irrep("1o") * Irrep("2e")) == [Irrep("1o"), Irrep("2o"), Irrep("3o")]
    s, p, inv = Irreps("1x1e+1x0e+1x1o+1x0o").sort()
    assert str(s) == "1x0o+1x0e+1x1o+1x1e" and p == (3, 1, 2, 0)   # (l, p) tuple order: odd first
    assert str(Irreps("2x1o") + Irreps("3x0e")) == "2x1o+3x0e"
    assert Irreps("0x0e").dim == 0
    with pytest.raises(ValueError):
        Irreps("3x1q")
    for a in ("64x0e+64x1o", "8x0e+8x0o+8x1e"):
        assert e3ref.irreps_str(e3ref.parse_irreps(a)) == str(Irreps(a))
    srt, perm = e3ref.irreps_sort(e3ref.parse_irreps("1x1e+1x0e+1x1o+1x0o"))
    assert e3ref.irreps_str(srt) == str(s) and tuple(perm) == p


def test_tp_path_exists_agrees_with_oracle():
    from e3_layers_amd.utils import tp_path_exists

    cases = [("64x0e", "1x0e+1x1o+1x2e", "1o"), ("64x0e", "1x0e+1x1o+1x2e", "1e"), ("8x1o+8x2e", "1x1o", "0o"),
             ("8x1o", "1x2e", "3o"), ("8x1o", "1x2e", "3e"), ("8x0o", "1x0e", "0o")]
    for a, b, c in cases:
        assert tp_path_exists(a, b, c) == e3ref.tp_path_exists(a, b, c), (a, b, c)


# ---- config trees / path tables ---------------------------------------------------------------------
def test_config_energy_tables_match_survey_appendix_b():
    """weight_numel / D_mid / path counts / conv_out dims / parameter totals of SURVEY.md appendix B."""
    from e3_layers_amd.configs import config_energy
    from e3_layers_amd.utils import build, countParameters

    model = build(config_energy.get_config().model_config)
    expect = {0: (3, 192, 576, 704), 1: (18, 1152, 4608, 2368), 2: (45, 2880, 12992, 2432),
              3: (48, 3072, 13568, 2432), 4: (48, 3072, 13568, 2432)}
    for i, (paths, w, dmid, dout) in expect.items():
        conv = getattr(model, f"layer{i}").conv
        assert len(conv.tp.tp.paths) == paths and conv.tp.tp.weight_numel == w and conv.tp.tp.d_mid == dmid
        assert conv.tp.linear.irreps_out.dim == dout
    sc = [getattr(model, f"layer{i}").conv.sc.weight.numel() for i in range(5)]
    assert sc == [245760, 737280, 1064960, 1146880, 1146880]
    assert abs(countParameters(model) - 6.13e6) < 0.01e6
    m2 = build(config_energy.get_config(l_max=2).model_config)
    assert [getattr(m2, f"layer{i}").conv.tp.tp.weight_numel for i in range(5)] == [192, 960, 1728, 1920, 1920]
    assert [getattr(m2, f"layer{i}").conv.tp.tp.d_mid for i in range(5)] == [576, 3264, 5952, 6528, 6528]


@pytest.mark.parametrize("name", ["config_energy", "config_energy_force", "config_diffusion", "config_diffusion_CA",
                                  "config_diffusion_backbone"])
def test_config_trees_build_and_share_state_dict_names_with_oracle(name):
    import importlib

    from e3_layers_amd.utils import build

    tree = importlib.import_module(f"e3_layers_amd.configs.{name}").get_config().model_config
    torch.manual_seed(0)
    prod = build(tree)
    orc = e3ref.build(tree)
    strip = lambda k: k.replace("mods.", "")
    a = {k: tuple(v.shape) for k, v in prod.state_dict().items()}
    b = {strip(k): tuple(v.shape) for k, v in orc.state_dict().items()}
    assert {k: math.prod(v) for k, v in a.items()} == {k: math.prod(v) for k, v in b.items()}


def test_tp_group_tables_cover_every_path_once():
    from e3_layers_amd.nn.core import UVUTensorProduct, tp_slots

    tp = UVUTensorProduct("64x0e+64x0o+64x1e+64x1o+64x2e+64x2o+64x3e+64x3o", "1x0e+1x1o+1x2e",
                          "64x0e+64x0o+64x1e+64x1o+64x2e+64x2o+64x3e+64x3o")
    seen_w, seen_out = set(), set()
    n_paths = 0
    for g in tp.plan.groups:
        slots = tp_slots(g.l1)
        for q, (l2, l3) in enumerate(slots):
            if g.mask >> q & 1:
                n_paths += 1
                assert g.w_off[q] not in seen_w
                seen_w.add(g.w_off[q])
                for k in range(2 * l3 + 1):
                    for u in (0, g.mul - 1):
                        idx = g.out_off[q] + k * g.out_stride[q] + u
                        assert idx not in seen_out and 0 <= idx < tp.d_mid
                        seen_out.add(idx)
                assert abs(g.coeff[q] - math.sqrt(2 * l3 + 1)) < 1e-6
                assert g.y_off[l2] == [0, 1, 4][l2]
    assert n_paths == len(tp.paths) == 48 and len(seen_w) == 48
    assert sorted(seen_w) == [64 * i for i in range(48)]


def test_unsupported_degree_fails_loudly():
    from e3_layers_amd.nn.core import UVUTensorProduct

    with pytest.raises(NotImplementedError, match="CG tables"):
        UVUTensorProduct("8x4e", "1x0e", "8x4e")
    from e3_layers_amd.nn import FullyConnectedTensorProduct, MessagePassing

    with pytest.raises(NotImplementedError):
        FullyConnectedTensorProduct("8x0e", "4x1o", "8x1o")


def test_config_dict_stand_in():
    from e3_layers_amd.configs import ConfigDict

    c = ConfigDict()
    c.a = 1
    c.sub = {"x": 2}
    assert isinstance(c.sub, ConfigDict) and c.sub.x == 2 and c["a"] == 1 and "a" in c
    c.update({"sub": {"y": 3}, "b": [1, 2]})
    assert c.sub.x == 2 and c.sub.y == 3 and c.to_dict() == {"a": 1, "sub": {"x": 2, "y": 3}, "b": [1, 2]}
    with pytest.raises(AttributeError):
        c.missing


def test_build_prunes_kwargs_and_keymap():
    from e3_layers_amd.utils import build, keyMap, pruneArgs, insertAfter, replace

    def f(a, b=2):
        return a + b

    assert build({"module": f, "a": 1, "zzz": 9}) == 3
    assert build((f, 5)) == 7 and build(f, a=1, b=1, c=1) == 2
    assert pruneArgs(prefix="conv", conv_x=1, other=2) == {"x": 1}
    assert keyMap({"p": 1, "q": 2}, {"p": "r"}) == {"r": 1, "q": 2}
    assert keyMap({"p": 1}, {"p": ["a", "b"]}) == {"a": 1, "b": 1}
    lst = [("a", 1), ("b", 2)]
    assert insertAfter(lst, "a", ("c", 3)) == [("a", 1), ("c", 3), ("b", 2)]
    assert replace(lst, "b", ("d", 4)) == [("a", 1), ("d", 4)]
    with pytest.raises(ValueError):
        insertAfter(lst, "zz", ("c", 3))


# ---- Data / Batch --------------------------------------------------------------------------------------
def _samples():
    attrs = {"pos": ("node", "1x1o"), "species": ("node", "1x0e"), "total_energy": ("graph", "1x0e"), "bond": ("edge", "1x0e")}
    s0 = {"pos": torch.arange(9.0).view(3, 3), "species": torch.tensor([1, 6, 8]), "total_energy": torch.tensor([1.5]),
          "edge_index": torch.tensor([[0, 1, 2], [1, 2, 0]]), "bond": torch.tensor([1, 2, 3])}
    s1 = {"pos": torch.arange(6.0).view(2, 3) + 100, "species": torch.tensor([[1], [1]]), "total_energy": torch.tensor([[2.5]]),
          "edge_index": torch.tensor([[0, 1], [1, 0]]), "bond": torch.tensor([[4], [5]])}
    return [s0, s1], attrs


def test_batch_from_data_list_and_back():
    from e3_layers_amd.data import Batch

    lst, attrs = _samples()
    b = Batch.from_data_list(lst, attrs)
    assert len(b) == 2 and b["pos"].shape == (5, 3) and b["species"].shape == (5, 1) and b["species"].dtype == torch.int64
    assert b["pos"].dtype == torch.float32 and b["total_energy"].shape == (2, 1)
    assert b["_n_nodes"].view(-1).tolist() == [3, 2] and b["_n_edges"].view(-1).tolist() == [3, 2]   # not 0 (appendix C)
    assert b["edge_index"].tolist() == [[0, 1, 2, 3, 4], [1, 2, 0, 4, 3]]
    assert b["_node_segment"].tolist() == [0, 0, 0, 1, 1] and b["_edge_segment"].tolist() == [0, 0, 0, 1, 1]
    one = b[1]
    assert one["edge_index"].tolist() == [[0, 1], [1, 0]] and one["pos"][0, 0] == 100 and one["bond"].view(-1).tolist() == [4, 5]
    again = b[[1, 0]]
    assert again["_n_nodes"].view(-1).tolist() == [2, 3] and again["edge_index"].tolist() == [[0, 1, 2, 3, 4], [1, 0, 3, 4, 2]]
    assert b[0:1]["pos"].shape == (3, 3) and len(b[torch.tensor([True, False])]) == 1
    c = b.clone()
    c["pos"][0, 0] = -1
    assert b["pos"][0, 0] == 0
    b["extra"] = [1.0, 2.0]
    assert isinstance(b["extra"], torch.Tensor)
    v = b.view()                                    # same tensors, independent key / attrs dicts
    v["new_key"] = torch.zeros(5, 1)
    v.attrs["new_key"] = ("node", "1x0e")
    assert "new_key" not in b and "new_key" not in b.attrs and v["pos"].data_ptr() == b["pos"].data_ptr()
    assert len(v) == len(b) and v["_node_segment"] is b["_node_segment"]


def test_data_autoreshape_by_irreps():
    from e3_layers_amd.data import Data

    d = Data({"f": ("node", "2x1o")}, f=torch.arange(12.0))
    assert d["f"].shape == (2, 6) and "f" in d and list(d.keys()) == ["f"]
    d.update({"g": torch.zeros(3)})
    assert d["g"].shape == (3,)


# ---- computeEdgeIndex: bit-exact integer contract ---------------------------------------------------
def _ref_edge_index_loops(pos, n_nodes, r_max):
    """Independent brute force: graphs concatenated, (i, j) lexicographic, |pos_i - pos_j| < r_max, i != j."""
    src, dst, start = [], [], 0
    for n in n_nodes:
        for i in range(start, start + n):
            for j in range(start, start + n):
                if i != j and float(torch.linalg.norm(pos[i] - pos[j])) < r_max:
                    src.append(i)
                    dst.append(j)
        start += n
    return torch.tensor([src, dst], dtype=torch.long).view(2, -1)


def test_compute_edge_index_order_and_strict_cutoff():
    from e3_layers_amd.data import computeEdgeIndex

    g = torch.Generator().manual_seed(0)
    n_nodes = [4, 1, 6, 3]
    pos = torch.randn(sum(n_nodes), 3, generator=g) * 1.5
    pos[1] = pos[0] + torch.tensor([2.0, 0.0, 0.0])     # exactly r_max apart: excluded by the strict '<'
    data = {"pos": pos, "_n_nodes": torch.tensor(n_nodes).view(-1, 1)}
    out, attrs = computeEdgeIndex(data, {}, r_max=2.0)
    ref = _ref_edge_index_loops(pos, n_nodes, 2.0)
    assert torch.equal(out["edge_index"], ref)
    assert data["_n_edges"].view(-1).tolist() == [int(((ref[0] >= s) & (ref[0] < s + n)).sum()) for s, n in
                                                 zip([0, 4, 5, 11], n_nodes)]
    assert data["_n_edges"][1].item() == 0      # single-atom graph: no edges
    # the oracle restatement agrees bit for bit
    odata = {"pos": pos, "_n_nodes": torch.tensor(n_nodes).view(-1, 1)}
    o, _ = e3ref.compute_edge_index(odata, {}, r_max=2.0)
    assert torch.equal(o["edge_index"], out["edge_index"]) and torch.equal(odata["_n_edges"], data["_n_edges"])


def test_compute_edge_index_criteria_and_existing_edges():
    from e3_layers_amd.data import computeEdgeIndex

    pos = torch.tensor([[0.0, 0, 0], [1.0, 0, 0], [5.0, 0, 0], [0.0, 0, 0], [0.5, 0, 0]])
    n_nodes = torch.tensor([[3], [2]])
    old = torch.tensor([[0, 2], [2, 0]])                      # a long "bond" beyond the cutoff
    bond = torch.tensor([[7.0], [9.0]])
    for fn in (computeEdgeIndex, e3ref.compute_edge_index):
        data = {"pos": pos, "_n_nodes": n_nodes, "edge_index": old.clone(), "bond": bond.clone()}
        attrs = {"bond": ("edge", "1x0e")}
        out, attrs = fn(data, attrs, r_max=1.5)
        assert out["edge_index"].tolist() == [[0, 0, 1, 2, 3, 4], [1, 2, 0, 0, 4, 3]]
        assert data["bond"].view(-1).tolist() == [0.0, 7.0, 0.0, 9.0, 0.0, 0.0]
        assert data["_n_edges"].view(-1).tolist() == [4, 2]
    crit = lambda data, ei: (ei[0] - ei[1]).abs() == 2        # extra criterion: |i - j| == 2
    for fn in (computeEdgeIndex, e3ref.compute_edge_index):
        data = {"pos": pos, "_n_nodes": n_nodes}
        out, _ = fn(data, {}, r_max=1.5, criteria=crit)
        assert out["edge_index"].tolist() == [[0, 0, 1, 2, 3, 4], [1, 2, 0, 0, 4, 3]]


def test_synthetic_qm9_is_seeded_and_plausible():
    from e3_layers_amd.data.synthetic import synth_qm9

    a, b = synth_qm9(0, 8), synth_qm9(0, 8)
    assert torch.equal(a["pos"], b["pos"]) and torch.equal(a["edge_index"], b["edge_index"])
    n = a["_n_nodes"].view(-1)
    assert int(n.min()) >= 3 and int(n.max()) <= 29
    assert set(a["species"].view(-1).tolist()) <= {1, 6, 7, 8, 9}
    ei = a["edge_index"]
    d = (a["pos"][ei[0]] - a["pos"][ei[1]]).norm(dim=1)
    assert float(d.max()) < 4.0 and float(d.min()) >= 0.95 - 1e-6
    assert torch.equal(a["_node_segment"][ei[0]], a["_node_segment"][ei[1]])   # no cross-graph edges
    # symmetric graph, grouped by source in ascending order (row-major all-pairs construction)
    assert torch.all(ei[0][1:] >= ei[0][:-1])
    fwd = set(map(tuple, ei.t().tolist()))
    assert all((j, i) in fwd for i, j in fwd)


def test_partition_by_edges_balances_and_covers():
    from e3_layers_amd.run.parallel import partition_by_edges

    counts = [100, 10, 10, 10, 90, 40, 40, 100]
    for w in (1, 2, 3, 4, 8):
        parts = partition_by_edges(counts, w)
        assert len(parts) == w and sum(parts, []) == list(range(8)) and all(parts)
    parts = partition_by_edges(counts, 2)
    loads = [sum(counts[i] for i in p) for p in parts]
    assert abs(loads[0] - loads[1]) <= 100


def test_protein_preprocess_masked2indexed_and_crop():
    """config_diffusion_CA's dataset-side preprocess functions (e3_layers/configs/config_diffusion_CA.py:11-56)."""
    from e3_layers_amd.configs import config_diffusion_CA as cfg
    from e3_layers_amd.data import Batch

    g = torch.Generator().manual_seed(0)
    n = 600
    steps = torch.randn(n, 3, generator=g)
    ca = torch.cumsum(3.8 * steps / steps.norm(dim=1, keepdim=True), 0)
    attrs = {k: ("node", "1x1o") for k in ("N", "CA", "C", "O")}
    attrs.update({"species": ("node", "1x0e"), "chain_id": ("node", "1x0e"), "mask": ("node", "1x0e")})
    mask = torch.ones(n, 1, dtype=torch.long)
    mask[::7] = 0
    raw = Batch(dict(attrs), N=ca + 0.1, CA=ca.clone(), C=ca - 0.1, O=ca + 0.2, species=torch.randint(0, 21, (n, 1), generator=g),
                chain_id=(torch.arange(n) // 300).view(-1, 1), mask=mask, _n_nodes=torch.tensor([[n]]))
    b = cfg.masked2indexed(raw)
    kept = int(mask.sum())
    assert int(b["_n_nodes"]) == kept and b["CA"].shape == (kept, 3)
    assert torch.equal(b["id"].view(-1), torch.arange(n)[mask.view(-1).bool()])
    data, a2 = cfg.crop(dict(b.data), dict(b.attrs), max_nodes=384, generator=torch.Generator().manual_seed(1))
    m = int(data["_n_nodes"])
    assert 0 < m <= 384 and "N" not in data and "O" not in a2
    assert data["CA"].shape == (m, 3) and data["id"].shape[0] == m and data["species"].shape[0] == m
    # the kept residues form a ball: every kept CA is closer to the (unknown) centre than every dropped one for SOME kept centre
    kept_ids = set(data["id"].view(-1).tolist())
    ca_all, ids_all = b["CA"], b["id"].view(-1).tolist()
    ok = False
    for c in range(m):
        d = (ca_all - data["CA"][c]).norm(dim=1)
        inside = d[[i in kept_ids for i in ids_all]]
        outside = d[[i not in kept_ids for i in ids_all]]
        if float(inside.max()) < float(outside.min()):
            ok = True
            break
    assert ok
    small, _ = cfg.crop(dict(b.data), dict(b.attrs), max_nodes=10_000)
    assert int(small["_n_nodes"]) == kept          # nothing to crop
    tree = cfg.get_config()
    assert [getattr(f, "__name__", getattr(getattr(f, "func", None), "__name__", "")) for f in tree.data_config.preprocess] == ["masked2indexed", "crop"]


def test_backbone_config_keeps_four_atoms_and_scaler_round_trips():
    """config_diffusion_backbone (e3_layers/configs/config_diffusion_backbone.py): crop keeps N/CA/C/O (:53-54), the
    scaler expresses C, N relative to CA and O relative to C, centres CA per protein and divides by std (:100-102);
    the inverse scaler undoes everything but the centring; the tree has the concat3 layer after layer3 (:169-176) and
    one score head per diffused atom (:179-189)."""
    from e3_layers_amd.configs import config_diffusion_backbone as cfg
    from e3_layers_amd.data import Batch

    g = torch.Generator().manual_seed(3)
    n = 500
    steps = torch.randn(n, 3, generator=g)
    ca = torch.cumsum(3.8 * steps / steps.norm(dim=1, keepdim=True), 0)
    attrs = {k: ("node", "1x1o") for k in ("N", "CA", "C", "O")}
    attrs.update({"species": ("node", "1x0e"), "chain_id": ("node", "1x0e"), "mask": ("node", "1x0e")})
    off = {k: torch.randn(n, 3, generator=g) for k in ("N", "C", "O")}
    raw = Batch(dict(attrs), N=ca + off["N"], CA=ca.clone(), C=ca + off["C"], O=ca + off["C"] + off["O"],
                species=torch.randint(0, 21, (n, 1), generator=g), chain_id=(torch.arange(n) // 250).view(-1, 1),
                mask=torch.ones(n, 1, dtype=torch.long), _n_nodes=torch.tensor([[n]]))
    conf = cfg.get_config()
    b = conf.data_config.preprocess[0](raw)
    data, a2 = conf.data_config.preprocess[1](dict(b.data), dict(b.attrs), generator=torch.Generator().manual_seed(0))
    m = int(data["_n_nodes"])
    assert 0 < m <= 384
    for atom in ("N", "CA", "C", "O"):
        assert data[atom].shape == (m, 3) and atom in a2
    cropped = Batch(a2, **{k: v for k, v in data.items() if k not in ("_node_segment", "_edge_segment")})
    scaled = conf.data_config.scaler(cropped)
    std = conf.data_config.std
    keep = data["id"].view(-1)
    assert torch.allclose(scaled["C"], off["C"][keep] / std, atol=1e-5)
    assert torch.allclose(scaled["N"], off["N"][keep] / std, atol=1e-5)
    assert torch.allclose(scaled["O"], off["O"][keep] / std, atol=1e-5)
    assert float(scaled["CA"].mean(0).abs().max()) < 1e-5
    back = conf.data_config.inverse_scaler(scaled)
    shift = cropped["CA"].mean(0, keepdim=True)
    for atom in ("N", "CA", "C", "O"):
        assert torch.allclose(back[atom], cropped[atom] - shift, atol=2e-4), atom
    names = [k for k, _ in conf.model_config.layers]
    assert names.index("concat3") == names.index("layer3") + 1
    assert names[-4:] == ["score_CA", "score_C", "score_O", "score_N"]
    assert conf.diffusion_keys == {"CA": 3, "C": 3, "O": 3, "N": 3}


def test_batch_index_select_is_one_gather_per_tensor():
    """Batch.index_select / slicing / boolean masks (e3_layers/data/batch.py:133-162, 180-186) through segment arithmetic:
    identical, tensor for tensor, to rebuilding the sub-batch from per-graph Data objects; order and repeats honoured."""
    import torch
    from e3_layers_amd.data import Batch
    from e3_layers_amd.data.synthetic import synth_qm9

    b = synth_qm9(3, 40)
    attrs = {k: v for k, v in b.attrs.items() if k not in ("_node_segment", "_edge_segment")}
    mask = torch.zeros(40, dtype=torch.bool)
    mask[[2, 11, 39]] = True
    for sel, ids in ([[5, 3, 3, 38, -1], [5, 3, 3, 38, 39]], [slice(4, 20, 3), list(range(4, 20, 3))], [mask, [2, 11, 39]],
                     [torch.tensor([7, 0]), [7, 0]]):
        new = b[sel]
        old = Batch.from_data_list([b.get(i) for i in ids], dict(attrs))
        assert set(new.data) == set(old.data)
        for k in old.data:
            assert new.data[k].dtype == old.data[k].dtype and torch.equal(new.data[k], old.data[k]), k
    assert len(b[[]]) == 0 and b[[]]["pos"].shape == (0, 3)
    with pytest.raises(IndexError):
        b[[40]]
    # per-graph access after the counts tensor was replaced sees the new offsets (no stale cache)
    first = b[1]["pos"].clone()
    sub = b[[1, 2]]
    assert torch.equal(sub[0]["pos"], first)


def test_optimizer_checkpoint_layout_is_verified_or_remapped_by_name():
    """``FusedAdamEMA.state_dict()`` used to hold bare flat vectors; a checkpoint written under
    ``flat_param_order(model)`` loaded into an optimizer built from ``model.parameters()`` (same numel) silently permuted
    weights and moments.  The layout now travels with the checkpoint: identical -> plain copy, named on both sides ->
    re-mapped slice by slice, anything else raises."""
    from e3_layers_amd.run.optim import layout_moves

    a = [(0, 6, (2, 3), "w1"), (64, 4, (4,), "b"), (128, 6, (2, 3), "w2")]
    assert layout_moves(a, list(a)) is None
    # the same parameters in another order (what flat_param_order does to the radial MLPs)
    b = [(0, 4, (4,), "b"), (64, 6, (2, 3), "w1"), (128, 6, (2, 3), "w2")]
    assert layout_moves(a, b) == [(0, 64, 6), (64, 0, 4), (128, 128, 6)]
    # same shapes at every position but the names say the order differs: re-mapped, not trusted
    c = [(0, 6, (2, 3), "w2"), (64, 4, (4,), "b"), (128, 6, (2, 3), "w1")]
    assert layout_moves(a, c) == [(0, 128, 6), (64, 64, 4), (128, 0, 6)]
    unnamed = [(o, n, sh, None) for o, n, sh, _ in b]
    with pytest.raises(ValueError):
        layout_moves(a, unnamed)            # different layout, no names on one side
    with pytest.raises(ValueError):
        layout_moves(a, None)               # a pre-round-4 checkpoint
    with pytest.raises(ValueError):
        layout_moves(a, [(0, 6, (2, 3), "w1"), (64, 4, (4,), "b"), (128, 6, (2, 3), "other")])
    with pytest.raises(ValueError):
        layout_moves(a, [(0, 6, (3, 2), "w1"), (64, 4, (4,), "b"), (128, 6, (2, 3), "w2")][::-1])
    # same shape sequence, names on one side only, two parameters of one shape: the order cannot be verified ...
    with pytest.raises(ValueError):
        layout_moves([(o, n, sh, None) for o, n, sh, _ in a], a)
    assert layout_moves([(o, n, sh, None) for o, n, sh, _ in a], a, assume_same_order=True) is None      # ... unless the caller vouches
    assert layout_moves(a, None, assume_same_order=True) is None                   # the way in for pre-round-4 checkpoints
    d = [(0, 6, (2, 3), "w1"), (64, 4, (4,), "b")]
    assert layout_moves([(o, n, sh, None) for o, n, sh, _ in d], d) is None       # all shapes distinct: the sequence determines the order


def test_linear_terms_that_add_onto_one_block_become_one_k_chain(monkeypatch):
    """``ops._chain_rounds`` (round 6): the input gradient of the trailing Linear of a convolution (``e3_layers/nn/message_passing.py:58``,
    ``o3.Linear``) has two terms for ``0e`` -- it feeds the scalars AND the gates -- which used to be an accumulating second launch.  The
    template builder now hangs the second term behind the first as a K-chain follower (``e3k_gemm_problem.chain``): one round, the head's
    ``chain`` counts its followers, a follower repeats the head's output block; the forward has one writer per block and stays as it was;
    with the knob off the rounds are back."""
    import torch

    from e3_layers_amd.backend import ops
    from e3_layers_amd.configs import config_energy
    from e3_layers_amd.utils import build

    model = build(config_energy.get_config(l_max=2).model_config)
    spec = model.layer2.conv.tp.linear.spec("cf", "cf")
    by_in = {}
    for ins in spec.instr:
        by_in.setdefault(ins.in_off, []).append(ins)
    shared = [v for v in by_in.values() if len(v) > 1]
    assert len(shared) == 1 and len(shared[0]) == 2          # `0e` -> 64x0e and 256x0e
    t = ops._lin_dgrad_templates(spec, 0.3, False)
    assert len(t.rounds) == 1
    arr, n = t.rounds[0]
    assert n == len(spec.instr)
    heads = [i for i in range(n) if arr[i].chain > 0]
    assert len(heads) == 1 and arr[heads[0]].chain == 1
    h, f = arr[heads[0]], arr[heads[0] + 1]
    assert f.chain == 0 and (f.C, f.N, f.M2, f.c_r1, f.c_r2, f.c_n) == (h.C, h.N, h.M2, h.c_r1, h.c_r2, h.c_n)
    assert {h.K, f.K} == {64, 256} and h.A != f.A and h.B != f.B
    fwd = ops._lin_fwd_templates(spec, 0.3, False, 0, 1.0, False)
    assert len(fwd.rounds) == 1 and all(fwd.rounds[0][0][i].chain == 0 for i in range(fwd.rounds[0][1]))
    monkeypatch.setenv("E3K_GEMM_CHAIN", "0")
    plain = ops._lin_dgrad_templates(spec, 0.3, False)
    assert [cnt for _, cnt in plain.rounds] == [n - 1, 1]
    assert all(plain.rounds[r][0][i].chain == 0 for r in range(2) for i in range(plain.rounds[r][1]))
    assert plain.rounds[1][0][0].accumulate == 1 and plain.rounds[1][0][0].C == h.C
