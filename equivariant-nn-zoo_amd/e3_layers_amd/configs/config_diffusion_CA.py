"""Residue-level (C-alpha) protein score network — the model tree of
``e3_layers/configs/config_diffusion_CA.py:66-194`` (BASELINE.json configs[4] names it
"config_diffusion_protein", which the reference does not register).
n_dim 64, l_max 2, 8 layers, 32 radial / 32 node-attr channels, avg_num_neighbors 100,
LayerNormalization on, relative-position encoding of same-chain residue pairs, time encoding.
``masked2indexed`` (:11-24), ``crop`` (:26-56) and ``criteria`` (:58-64) are the dataset-side preprocess functions of
that file (host plumbing on CPU tensors, as in the reference); the HDF5 reader itself is out of scope.
"""
from functools import partial

import torch

from ..data import computeEdgeIndex, computeEdgeVector
from ..nn import Broadcast, Concat, PointwiseLinear, RadialBasisEncoding, RelativePositionEncoding, symmetricCutoff
from ..utils import getScaler, insertAfter, replace
from .config_dict import ConfigDict
from .layer_configs import featureModel


def masked2indexed(batch):
    """Keep the residues whose ``mask`` is set; ``id`` records their position in the original chain (:11-24)."""
    from ..data import Batch

    n = int(batch["_n_nodes"].reshape(-1)[0])
    mask = batch["mask"].view(-1).bool()
    data = {"id": torch.arange(n)[mask].view(-1, 1), "_n_nodes": mask.sum().view(-1, 1),
            "species": batch["species"][mask], "chain_id": batch["chain_id"][mask]}
    attrs = {"id": ("node", "1x0e")}
    for atom in ("N", "CA", "C", "O"):
        data[atom] = batch[atom][mask]
    attrs.update(batch.attrs)
    return Batch(attrs, **data)


def crop(data, attrs, max_nodes, generator=None, atoms=("CA",)):
    """Drop the backbone atoms other than ``atoms`` (CA only here; ``config_diffusion_backbone`` keeps all four); if
    the protein has more than ``max_nodes`` residues keep those inside a ball around a random residue whose radius a
    bisection over [20, 70] A (0.5 A resolution) picks so that at most ``max_nodes`` remain (:26-56).  ``generator``:
    a torch CPU generator for the centre (the reference uses the global numpy stream)."""
    for key in ("N", "C", "O"):
        if key not in atoms:
            data.pop(key)
            attrs.pop(key)
    n = int(data["_n_nodes"].reshape(-1)[0])
    if n <= max_nodes:
        return data, attrs
    x = int(torch.randint(n, (1,), generator=generator))
    distance = torch.linalg.norm(data["CA"] - data["CA"][x], dim=-1)
    lo, hi = 20.0, 70.0
    while hi - lo >= 0.5:                       # the reference's recursion, iteratively
        mid = 0.5 * (lo + hi)
        count = int((distance < mid).sum())
        if count > max_nodes:
            hi = mid
        elif count < max_nodes:
            lo = mid
        else:
            lo = mid
            break
    mask = (distance < lo).view(-1)
    data["_n_nodes"] = mask.sum().view(-1, 1)
    for key in ("id", "species", "chain_id") + tuple(atoms):
        data[key] = data[key][mask]
    return data, attrs


def criteria(data, edge_index):
    """same chain and |i - j| < 5, or a 2 % random subset (:58-64; drawn from the generator of the device that holds the
    batch — the reference draws on the CPU and copies; a 590 k-candidate H2D copy per model call)."""
    mask = (data["chain_id"][edge_index[0]] == data["chain_id"][edge_index[1]]).view(-1)
    mask = torch.logical_and(mask, (edge_index[0] - edge_index[1]).abs() < 5)
    extra = torch.rand((edge_index.shape[1],), device=mask.device)
    return torch.logical_or(mask, extra < 0.02)


def get_config(spec="", l_max=2, num_layers=8, n_dim=64):
    data = ConfigDict()
    data.std = 25.83
    data.scaler = getScaler([("CA", ("shift", "mean")), ("CA", ("scale", 1 / data.std))])
    data.inverse_scaler = getScaler([("CA", ("scale", data.std))])
    data.preprocess = [masked2indexed, partial(crop, max_nodes=384)]
    return score_config({"CA": 3}, data, l_max, num_layers, n_dim)


def score_config(diffusion_keys, data, l_max, num_layers, n_dim, side_atoms=()):
    """The tree both protein configs share (``config_diffusion_CA.py:66-194``, ``config_diffusion_backbone.py:64-194``):
    one ``score_{key}`` head per diffusion key; ``side_atoms`` (backbone: C, N, O relative positions) are mixed into the
    node features after ``layer3`` by a ``Concat`` (``config_diffusion_backbone.py:169-176``)."""
    config = ConfigDict()
    model = ConfigDict()
    config.data_config, config.model_config = data, model
    config.diffusion_keys = dict(diffusion_keys)
    config.update(dict(learning_rate=2e-3, batch_size=4, grad_acc=4, grad_clid_norm=1.0))

    model.n_dim, model.l_max, model.r_max, model.num_layers = n_dim, l_max, 5.0, num_layers
    model.edge_radial, model.node_attrs, model.jit = "32x0e", "32x0e", True
    num_types = 21
    data.key_map = {}

    features = "+".join(f"{model.n_dim}x{l}e+{model.n_dim}x{l}o" for l in range(model.l_max + 1))
    lc = featureModel(n_dim=model.n_dim, l_max=model.l_max, edge_spherical="1x0e+1x1o+1x2e",
                      node_attrs=model.node_attrs, edge_radial=model.edge_radial, num_types=num_types,
                      num_layers=model.num_layers, r_max=model.r_max, avg_num_neighbors=100, normalize=True)
    lc.layers = replace(lc.layers, "edge_vector", ("edge_vector", partial(computeEdgeVector, key="CA")))
    rel_radial = {"module": RadialBasisEncoding, "r_max": 150, "cutoff": symmetricCutoff, "trainable": True,
                  "one_over_r": False}
    relative_position = ("relative_position", {"module": RelativePositionEncoding, "segment": ("1x0e", "chain_id"),
                                               "id": ("1x0e", "id"), "irreps_out": (model.edge_radial, "rel_pos_embed"),
                                               "radial_encoding": rel_radial})
    concat1 = ("concat1", {"module": Concat, "rel_pos": (model.edge_radial, "rel_pos_embed"),
                           "edge_radial": (model.edge_radial, "edge_radial"),
                           "irreps_out": (model.edge_radial, "edge_radial")})
    lc.layers = [relative_position] + list(lc.layers)
    lc.layers = insertAfter(lc.layers, "radial_basis", concat1)
    time_encoding = ("time_encoding", {"module": RadialBasisEncoding, "r_max": 1.0, "trainable": True,
                                       "irreps_in": ("1x0e", "t"), "one_over_r": False,
                                       "irreps_out": (f"{model.n_dim}x0e", "time_encoding")})
    lc.layers = insertAfter(lc.layers, "embedding", time_encoding)
    graph2node = ("graph2node", {"module": Broadcast, "irreps_in": (f"{model.n_dim}x0e", "time_encoding"),
                                 "irreps_out": (f"{model.n_dim}x0e", "time_encoding"), "to": "node"})
    lc.layers = insertAfter(lc.layers, "time_encoding", graph2node)
    concat2 = ("concat2", {"module": Concat, "node_attrs": (model.node_attrs, "node_attrs"),
                           "time_encoding": (f"{model.n_dim}x0e", "time_encoding"),
                           "irreps_out": (model.node_attrs, "node_attrs")})
    lc.layers = insertAfter(lc.layers, "graph2node", concat2)
    if side_atoms:
        concat3 = {"module": Concat, "node_features": (lc.node_features, "node_features")}
        concat3.update({atom: ("1x1o", atom) for atom in side_atoms})
        concat3["irreps_out"] = (lc.node_features, "node_features")
        lc.layers = insertAfter(lc.layers, "layer3", ("concat3", concat3))
    for key in config.diffusion_keys:
        lc.layers = list(lc.layers) + [(f"score_{key}", {"module": PointwiseLinear,
                                                         "irreps_in": (features, "node_features"),
                                                         "irreps_out": ("1x1o", f"score_{key}")})]
    lc.layers = [("edge_index", partial(computeEdgeIndex, r_max=8.0 / data.std, key="CA", criteria=criteria))] + list(lc.layers)
    model.update(lc)
    return config
