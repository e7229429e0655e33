"""Layer-graph builders: the ``model_config`` trees of the reference's configs.

Same functions, arguments and resulting ``(name, {module: cls, **kwargs})`` lists as
``e3_layers/configs/layer_configs.py``: ``featureModel`` (:10-101), ``embedCategorial``
(:104-118), ``addEnergyOutput`` (:121-147), ``addForceOutput`` (:150-166).  ``addMatrixOutput``
(legacy Hamiltonian head) is out of scope.  A tree produced by the reference's own functions
(with its classes swapped for the ones of ``e3_layers_amd.nn``) builds identically.
"""
from __future__ import annotations

from copy import deepcopy

from ..data import computeEdgeVector
from ..nn import (FactorizedConvolution, GradientOutput, MessagePassing, OneHotEncoding, PerTypeScaleShift,
                  PointwiseLinear, Pooling, RadialBasisEncoding, SequentialGraphNetwork, SphericalEncoding)
from ..o3 import Irreps
from ..utils import tp_path_exists
from .config_dict import ConfigDict


def embedCategorial(num_types, irreps_in, irreps_out):
    return {
        "onehot": {"module": OneHotEncoding, "num_types": num_types,
                   "irreps_out": (f"{num_types}x0e", "onehot"), "irreps_in": irreps_in},
        "embedding": {"module": PointwiseLinear, "irreps_in": (f"{num_types}x0e", "onehot"), "irreps_out": irreps_out},
    }


def featureModel(n_dim, l_max, edge_radial, num_types, num_layers, r_max, node_attrs, edge_spherical=None,
                 avg_num_neighbors=10, normalize=False):
    config = ConfigDict()
    full = "+".join(f"{n_dim}x{l}e+{n_dim}x{l}o" for l in range(l_max + 1))
    if edge_spherical is None:
        edge_spherical = "+".join(f"1x{l}{'e' if l % 2 == 0 else 'o'}" for l in range(l_max + 1))
    config.update(dict(n_dim=n_dim, l_max=l_max, edge_spherical=edge_spherical, edge_radial=edge_radial,
                       num_types=num_types, num_layers=num_layers, r_max=r_max, module=SequentialGraphNetwork,
                       node_features=full, node_attrs=node_attrs))

    layers = {"edge_vector": computeEdgeVector}
    layers.update(embedCategorial(num_types, ("1x0e", "species"), (node_attrs, "node_attrs")))
    layers["node_features"] = {"module": PointwiseLinear, "irreps_in": (f"{num_types}x0e", "onehot"),
                               "irreps_out": (f"{n_dim}x0e", "node_features")}
    layers["spharm_edges"] = {"module": SphericalEncoding, "irreps_out": (edge_spherical, "edge_spherical"),
                              "irreps_in": ("1x1o", "edge_vector")}
    layers["radial_basis"] = {"module": RadialBasisEncoding, "r_max": r_max, "trainable": True,
                              "polynomial_degree": 6, "irreps_in": ("1x0e", "edge_length"),
                              "irreps_out": (edge_radial, "edge_radial")}
    conv = {"module": FactorizedConvolution, "avg_num_neighbors": avg_num_neighbors, "use_sc": True,
            "invariant_layers": 3, "invariant_neurons": n_dim}
    template = {
        "module": MessagePassing, "resnet": False, "convolution": conv, "nonlinearity_type": "gate",
        "nonlinearity_scalars": {"e": "silu", "o": "tanhlu"}, "nonlinearity_gates": {"e": "silu", "o": "tanhlu"},
        "normalize": normalize,
        "node_attrs": node_attrs, "input_features": [full, "node_features"], "edge_radial": edge_radial,
        "edge_spherical": edge_spherical, "output_features": [full, "node_features"],
    }
    current = Irreps(f"{n_dim}x0e")
    target = Irreps(full)
    for i in range(num_layers):
        layer = deepcopy(template)
        layer["input_features"][0] = str(current)
        current = Irreps([mi for mi in target if tp_path_exists(current, edge_spherical, mi.ir)])
        layer["output_features"][0] = str(current)
        layers[f"layer{i}"] = layer
    config.layers = list(layers.items())
    return config


def addEnergyOutput(config, shifts=None, output_key="total_energy"):
    layers = {"output_linear": {"module": PointwiseLinear, "irreps_in": (config.node_features, "node_features"),
                                "irreps_out": ("1x0e", "energy")}}
    if shifts is not None:
        layers["rescale"] = {"module": PerTypeScaleShift, "num_types": config.num_types, "shifts": shifts,
                             "scales": None, "irreps_in": ("1x0e", "energy"), "irreps_out": ("1x0e", "energy"),
                             "species": ("1x0e", "atom_types")}
    layers["reduce"] = {"module": Pooling, "reduce": "sum", "irreps_in": ("1x0e", "energy"),
                        "irreps_out": ("1x0e", output_key)}
    config.layers = list(config.layers) + list(layers.items())
    return config


def addForceOutput(config, gradients="forces", y="energy", sign=-1.0):
    plain = config.to_dict()
    layers = plain.pop("layers")
    module = plain.pop("module")
    out = ConfigDict(plain)
    out.func = {"module": module, "layers": layers}
    out.update({"module": GradientOutput, "x": ("1x1o", "pos"), "y": ("1x0e", y),
                "gradients": ("1x1o", gradients), "sign": sign})
    return out
