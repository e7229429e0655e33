from .utils import (
    setSeed, getScaler, insertAfter, replace, activations, tp_path_exists, build, pruneArgs, keyMap, countParameters,
    act_second_moment_const, activation_name,
)
