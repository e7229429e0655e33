"""QM9 total-energy model — the tree of ``e3_layers/configs/config_energy.py`` (model part:
:34-40,52-80; training hyper-parameters kept as read-only metadata)."""
from functools import partial

from ..data import computeEdgeIndex
from .config_dict import ConfigDict
from .elements import chemical_symbols
from .layer_configs import addEnergyOutput, featureModel

QM9_SHIFTS = [-620.4502, -16.4435, -620.4502, -620.4502, -620.4502, -620.4502, -1036.0271, -1489.8005,
              -2046.9702, -2717.4263]


def get_config(spec=None, l_max=3, n_dim=64, num_layers=5):
    config = ConfigDict()
    data, model = ConfigDict(), ConfigDict()
    config.data_config, config.model_config = data, model
    config.update(dict(epoch_subdivision=1, learning_rate=1e-2, batch_size=128, use_ema=True, ema_decay=0.99,
                       ema_use_num_updates=True, metric_key="validation_loss", max_epochs=int(1e6),
                       optimizer_name="Adam", lr_scheduler_name="ReduceLROnPlateau", lr_scheduler_patience=1,
                       lr_scheduler_factor=0.8))
    config.loss_coeffs = {"total_energy": [1e3, "MSELoss"]}
    config.metrics_components = {"total_energy": ["mae"]}

    model.n_dim, model.l_max, model.r_max, model.num_layers = n_dim, l_max, 4.0, num_layers
    model.node_attrs, model.jit = "20x0e", True
    num_types = 10

    data.n_train, data.n_val = 120000, 10831
    data.train_val_split, data.shuffle = "random", True
    data.type_names = chemical_symbols[:num_types]
    data.key_map = {"Z": "species", "R": "pos", "U0": "total_energy"}
    data.preprocess = [partial(computeEdgeIndex, r_max=model.r_max)]

    layer_configs = featureModel(n_dim=model.n_dim, l_max=model.l_max, edge_spherical="1x0e+1x1o+1x2e",
                                 node_attrs=model.node_attrs, edge_radial="8x0e", num_types=num_types,
                                 num_layers=model.num_layers, r_max=model.r_max, normalize=False)
    layer_configs = addEnergyOutput(layer_configs, QM9_SHIFTS)
    model.update(layer_configs)
    return config
