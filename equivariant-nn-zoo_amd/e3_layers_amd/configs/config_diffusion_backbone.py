"""Backbone (N, CA, C, O) protein score network — the model tree of
``e3_layers/configs/config_diffusion_backbone.py:64-194`` (registered at ``configs/__init__.py:7``).
Same residue-level graph and message-passing stack as ``config_diffusion_CA``; the differences are on the data side
(all four backbone atoms are kept and diffused; C and N are expressed relative to CA, O relative to C, :100-102) and
two additions to the tree: a ``Concat`` after ``layer3`` that mixes the three relative atom positions (``1x1o`` each)
into the node features (:169-176) and one ``score_{CA,C,O,N}`` head per diffused key (:179-189).
"""
from functools import partial

from ..utils import getScaler
from .config_dict import ConfigDict
from .config_diffusion_CA import crop, masked2indexed, score_config

ATOMS = ("N", "CA", "C", "O")


def get_config(spec="", l_max=2, num_layers=8, n_dim=64):
    data = ConfigDict()
    data.std = 25.83
    data.scaler = getScaler([("O", ("shift", "C", -1)), ("C", ("shift", "CA", -1)), ("N", ("shift", "CA", -1)),
                             ("CA", ("shift", "mean")), (["CA", "C", "N", "O"], ("scale", 1 / data.std))])
    data.inverse_scaler = getScaler([(["C", "CA", "N", "O"], ("scale", data.std)), ("C", ("shift", "CA")),
                                     ("N", ("shift", "CA")), ("O", ("shift", "C"))])
    data.preprocess = [masked2indexed, partial(crop, max_nodes=384, atoms=ATOMS)]
    return score_config({"CA": 3, "C": 3, "O": 3, "N": 3}, data, l_max, num_layers, n_dim, side_atoms=("C", "N", "O"))
