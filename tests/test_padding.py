"""Host logic of the bucketed graph replay (run/graph_step.py): padding a batch to a size bucket with a ghost graph.
The replay itself is a GPU test (tests/test_gpu_model.py::test_bucketed_graph_replay_with_fresh_padded_batches_equals_eager)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "equivariant-nn-zoo_amd"))

from e3_layers_amd.data.data import Batch  # noqa: E402
from e3_layers_amd.data.loader import samples_of  # noqa: E402
from e3_layers_amd.data.synthetic import synth_qm9  # noqa: E402
from e3_layers_amd.run.graph_step import GHOST_DEGREE, bucket_capacity, ghost_sample, pad_batch  # noqa: E402


def test_pad_batch_appends_one_ghost_graph_of_zero_weight():
    b = synth_qm9(3, 8)
    n, e, g = b["pos"].shape[0], b["edge_index"].shape[1], len(b)
    n_cap, e_cap = bucket_capacity([(n, e)])
    p = pad_batch(b, n_cap, e_cap)
    assert p["pos"].shape[0] == n_cap and p["edge_index"].shape[1] == e_cap and len(p) == g + 1
    # the real graphs are untouched, in place
    assert torch.equal(p["pos"][:n], b["pos"]) and torch.equal(p["edge_index"][:, :e], b["edge_index"])
    assert torch.equal(p["species"][:n], b["species"]) and torch.equal(p["total_energy"][:g], b["total_energy"])
    assert p["_n_nodes"].view(-1).tolist() == b["_n_nodes"].view(-1).tolist() + [n_cap - n]
    assert p["_n_edges"].view(-1).tolist() == b["_n_edges"].view(-1).tolist() + [e_cap - e]
    # loss weights: a mean over the real graphs / the real nodes, nothing for the ghost
    w, wn = p["_graph_weight"].view(-1), p["_node_weight"].view(-1)
    assert w[:g].tolist() == pytest.approx([1.0 / g] * g) and float(w[-1]) == 0.0
    assert float(wn[:n].sum()) == pytest.approx(1.0) and float(wn[n:].abs().sum()) == 0.0
    # ghost geometry: edges between ghost nodes only, all lengths inside (0, r_max), spread over many distinct values
    ei = p["edge_index"][:, e:]
    assert int(ei.min()) >= n and int(ei.max()) < n_cap
    length = (p["pos"][ei[0]] - p["pos"][ei[1]]).norm(dim=1)
    assert float(length.min()) >= 1.0 - 1e-5 and float(length.max()) < 3.5 + 1e-5
    assert torch.unique((length * 1e3).round()).numel() >= min(n_cap - n - 1, 8)
    # ghost degree stays near the target
    deg = torch.bincount(ei[1] - n, minlength=n_cap - n)
    assert int(deg.max()) <= 4 * GHOST_DEGREE
    # the padded batch is a regular Batch: it splits back into samples
    assert len(samples_of(p)) == g + 1 and isinstance(p, Batch)


def test_bucket_capacity_fits_every_batch_with_bounded_ghost_degree():
    sizes = [(517, 7070), (618, 9812), (560, 8300)]
    n_cap, e_cap = bucket_capacity(sizes)
    assert e_cap % 1024 == 0 and n_cap % 32 == 0 and e_cap >= 9812
    for n, e in sizes:
        ghosts, ghost_edges = n_cap - n, e_cap - e
        assert ghosts >= 2 and ghost_edges <= GHOST_DEGREE * ghosts + GHOST_DEGREE


def test_padding_refuses_what_does_not_fit():
    b = synth_qm9(4, 4)
    n, e = b["pos"].shape[0], b["edge_index"].shape[1]
    with pytest.raises(ValueError):
        pad_batch(b, n + 1, e + 100)          # fewer than two ghost nodes
    with pytest.raises(ValueError):
        pad_batch(b, n + 64, e - 1)           # not enough edge capacity
    with pytest.raises(ValueError):
        ghost_sample(samples_of(b)[0], 1, 5)  # edges need two nodes


def test_flat_param_order_puts_the_radial_mlps_behind_the_layer_slices():
    """run/parallel.flat_param_order: every MessagePassing layer's node-side weights stay one contiguous run (its early
    all-reduce slice), the radial MLPs of all layers follow at the tail (their gradients arrive together when the MLPs run as
    one stack)."""
    from e3_layers_amd.configs import config_energy
    from e3_layers_amd.nn.message_passing import MessagePassing
    from e3_layers_amd.run.parallel import flat_param_order
    from e3_layers_amd.utils import build

    model = build(config_energy.get_config(l_max=2, num_layers=3).model_config)
    order = flat_param_order(model)
    assert len(order) == len(list(model.parameters())) and {id(p) for p in order} == {id(p) for p in model.parameters()}
    pos = {id(p): i for i, p in enumerate(order)}
    radial = [p for m in model.modules() if isinstance(m, MessagePassing) for p in m.conv.fc.parameters()]
    assert radial and sorted(pos[id(p)] for p in radial) == list(range(len(order) - len(radial), len(order)))
    for m in model.modules():
        if isinstance(m, MessagePassing):
            fc = {id(p) for p in m.conv.fc.parameters()}
            idx = sorted(pos[id(p)] for p in m.parameters() if id(p) not in fc)
            assert idx == list(range(idx[0], idx[0] + len(idx)))
