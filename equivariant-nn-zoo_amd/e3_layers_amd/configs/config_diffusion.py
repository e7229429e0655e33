"""Small-molecule VP-SDE score network — the tree of ``e3_layers/configs/config_diffusion.py:32-119``."""
from functools import partial

from ..data import computeEdgeIndex
from ..nn import Broadcast, Concat, OneHotEncoding, PointwiseLinear, RadialBasisEncoding
from ..utils import insertAfter
from .config_dict import ConfigDict
from .elements import chemical_symbols
from .layer_configs import addEnergyOutput, addForceOutput, featureModel


def get_config(spec=""):
    spec = spec or ""
    config = ConfigDict()
    data, model = ConfigDict(), ConfigDict()
    config.data_config, config.model_config = data, model
    config.update(dict(learning_rate=1e-2, batch_size=128, grad_clid_norm=1.0, grad_acc=1, optimizer_name="Adam"))

    model.n_dim, model.l_max, model.num_layers = 32, 2, 4
    model.edge_radial, model.node_attrs, model.r_max, model.jit = "8x0e", "16x0e", 8.0, True
    num_types = 18
    data.std = 1.4
    data.r_max = model.r_max / data.std
    data.type_names = chemical_symbols[:num_types]
    data.key_map = {"Z": "species", "R": "pos", "U": "total_energy", "edge_attr": "bond_type"}
    data.preprocess = [partial(computeEdgeIndex, r_max=9999)]
    # harness additions (the shipped config lacks them)
    config.diffusion_keys = {"pos": 3}

    features = "+".join(f"{model.n_dim}x{l}e+{model.n_dim}x{l}o" for l in range(model.l_max + 1))
    lc = featureModel(n_dim=model.n_dim, l_max=model.l_max, edge_spherical="1x0e+1x1o+1x2e",
                      node_attrs=model.node_attrs, edge_radial=model.edge_radial, num_types=num_types,
                      num_layers=model.num_layers, r_max=model.r_max / data.std)
    bond_onehot = ("bond_onehot", {"module": OneHotEncoding, "num_types": 4, "irreps_in": ("1x0e", "bond_type"),
                                   "irreps_out": ("4x0e", "bond_type_onehot")})
    concat1 = ("concat1", {"module": Concat, "bondtype": ("4x0e", "bond_type_onehot"),
                           "edge_radial": (model.edge_radial, "edge_radial"),
                           "irreps_out": (model.edge_radial, "edge_radial")})
    lc.layers = insertAfter(lc.layers, "radial_basis", bond_onehot)
    lc.layers = insertAfter(lc.layers, "bond_onehot", concat1)
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
    if "nll" in spec:
        lc = addEnergyOutput(lc, shifts=None, output_key="nll")
        lc = addForceOutput(lc, y="nll", gradients="score")
    else:
        lc.layers = list(lc.layers) + [("score_output", {"module": PointwiseLinear,
                                                         "irreps_in": (features, "node_features"),
                                                         "irreps_out": ("1x1o", "score")})]
    model.update(lc)
    return config
