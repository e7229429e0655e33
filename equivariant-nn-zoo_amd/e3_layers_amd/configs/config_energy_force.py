"""Energy + forces model — the tree of ``e3_layers/configs/config_energy_force.py:37-77``."""
from functools import partial

from ..data import computeEdgeIndex
from .config_dict import ConfigDict
from .elements import chemical_symbols
from .layer_configs import addEnergyOutput, addForceOutput, featureModel

SHIFTS = [-3.7204, -2.2483, -3.7204, -3.7204, -3.7204, -3.7204, -7.6108, -4.0182, -5.2651, -3.7204, -3.7204,
          -3.7204, -3.7204, -3.7204, -3.7204, -3.7204, -3.2213, -3.7204, -3.7204, -3.7204]


def get_config(spec=None):
    config = ConfigDict()
    data, model = ConfigDict(), ConfigDict()
    config.data_config, config.model_config = data, model
    config.update(dict(learning_rate=1e-2, batch_size=64, optimizer_name="Adam"))
    config.loss_coeffs = {"energy": [1e3, "MSELoss"], "forces": [3e4, "MSELoss"]}
    config.metrics_components = {"energy": ["mae"], "forces": ["mae"]}

    model.n_dim, model.l_max, model.r_max, model.num_layers = 64, 2, 5.0, 5
    model.jit, model.node_attrs = True, "16x0e"
    num_types = 20
    data.type_names = chemical_symbols[:num_types]
    data.preprocess = [partial(computeEdgeIndex, r_max=model.r_max)]

    layer_configs = featureModel(n_dim=model.n_dim, l_max=model.l_max, edge_spherical="1x0e+1x1o+1x2e",
                                 node_attrs=model.node_attrs, edge_radial="8x0e", num_types=num_types,
                                 num_layers=model.num_layers, r_max=model.r_max)
    layer_configs = addEnergyOutput(layer_configs, SHIFTS, output_key="energy")
    layer_configs = addForceOutput(layer_configs)
    model.update(layer_configs)
    return config
