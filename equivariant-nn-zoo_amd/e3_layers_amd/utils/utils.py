"""Factory and small helpers of the layer-graph protocol.

Mirrors the public names of ``e3_layers/utils/utils.py`` that the hot path and the shipped
configs use: ``build`` (:99-116), ``pruneArgs`` (:119-136), ``keyMap`` (:139-156),
``tp_path_exists`` (:87-96), the ``activations`` table (:78-84), ``insertAfter``/``replace``
(:49-61), ``getScaler`` (:15-47), ``setSeed`` (:9-13).  Activations are *names with kernel
ids* here — the arithmetic lives in ``csrc/e3k_node.hip`` — plus the second-moment constants
e3nn attaches to them (``normalize2mom``).
"""
from __future__ import annotations

import inspect
import math
from functools import lru_cache
from typing import Callable, Dict

import numpy as np
import torch

from ..o3 import Irrep, Irreps


def setSeed(seed: int) -> None:
    torch.manual_seed(seed)
    np.random.seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


# ---- activations -------------------------------------------------------------------------
class Activation:
    """A named activation; calling it runs the HIP elementwise kernel (unnormalised)."""

    def __init__(self, name: str):
        self.name = name
        self.__name__ = name

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        from ..backend import ops

        return ops.activation(x, self.name, 1.0)

    def __repr__(self):
        return f"Activation({self.name})"


activations: Dict[str, Activation] = {n: Activation(n) for n in ("abs", "tanh", "ssp", "silu", "tanhlu")}


def activation_name(act) -> str:
    if isinstance(act, Activation):
        return act.name
    if isinstance(act, str):
        if act not in activations:
            raise KeyError(f"unknown activation {act!r}")
        return act
    name = getattr(act, "__name__", None)
    if name in activations:
        return name
    raise KeyError(f"unknown activation {act!r}")


_HOST_ACTS: Dict[str, Callable] = {
    "abs": torch.abs,
    "tanh": torch.tanh,
    "ssp": lambda x: torch.nn.functional.softplus(x) - math.log(2.0),
    "silu": torch.nn.functional.silu,
    "tanhlu": lambda x: torch.tanh(x) * torch.abs(x),
}


@lru_cache(maxsize=None)
def act_second_moment_const(name: str) -> float:
    """``normalize2mom`` constant: (E_{z~N(0,1)} act(z)^2)^(-1/2), estimated the way e3nn 0.4.4
    does (1e6 float64 samples, CPU generator seeded with 0); 1.0 when within 1e-4 of one."""
    gen = torch.Generator(device="cpu").manual_seed(0)
    z = torch.randn(1_000_000, generator=gen, dtype=torch.float64)
    c = float(_HOST_ACTS[name](z).pow(2).mean().pow(-0.5))
    return 1.0 if abs(c - 1.0) < 1e-4 else c


def act_output_parity(name: str, p_in: int) -> int:
    """Parity of act(x) for a scalar of parity p_in (the check ``e3nn.nn.Activation`` makes)."""
    if p_in == 1:
        return 1
    x = torch.linspace(0.0, 10.0, 256, dtype=torch.float64)
    f = _HOST_ACTS[name]
    a, b = f(x), f(-x)
    if float((a - b).abs().max()) < 1e-10:
        return 1
    if float((a + b).abs().max()) < 1e-10:
        return -1
    raise ValueError(f"activation {name!r} is neither even nor odd, it cannot act on an odd scalar")


# ---- irreps helpers ----------------------------------------------------------------------
def tp_path_exists(irreps_in1, irreps_in2, ir_out) -> bool:
    a = Irreps(irreps_in1).simplify()
    b = Irreps(irreps_in2).simplify()
    target = Irrep(ir_out)
    return any(target in set(x.ir * y.ir) for x in a for y in b)


# ---- config-tree factory -----------------------------------------------------------------
def _is_mapping(node) -> bool:
    return hasattr(node, "keys") and hasattr(node, "__getitem__")


def pruneArgs(_func=None, prefix: str = "", **kwargs):
    if prefix:
        kwargs = {k[len(prefix) + 1:]: v for k, v in kwargs.items() if k.startswith(prefix)}
    if _func is None:
        return kwargs
    params = inspect.signature(_func).parameters
    if any(p.kind is inspect.Parameter.VAR_KEYWORD for p in params.values()):
        return kwargs
    return {k: v for k, v in kwargs.items() if k in params}


def build(node, **kwargs):
    """Instantiate the layer described by a config node: a mapping with a ``module`` entry
    (class or function) whose remaining entries are keyword arguments, a ``(func, *args)``
    sequence, or a bare callable.  Keyword arguments the target does not accept are dropped."""
    args = []
    if _is_mapping(node):
        func = node["module"]
        kwargs.update({k: node[k] for k in node.keys()})
    elif isinstance(node, (list, tuple)):
        func, args = node[0], list(node[1:])
    else:
        func = node
    kwargs.pop("module", None)
    return func(*args, **pruneArgs(func, **kwargs))


def keyMap(dic, key_mapping):
    if isinstance(dic, dict):
        out = {}
        for key, value in dic.items():
            target = key_mapping.get(key, key)
            if isinstance(target, str):
                out[target] = value
            else:
                for t in target:
                    out[t] = value
        return out
    return type(dic)(keyMap(dic.attrs, key_mapping), **keyMap(dic.data, key_mapping))


def insertAfter(lst, key, item):
    for i, layer in enumerate(lst):
        if layer[0] == key:
            return list(lst[: i + 1]) + [item] + list(lst[i + 1:])
    raise ValueError(f"Key {key} not found.")


def replace(lst, key, item):
    for i, layer in enumerate(lst):
        if layer[0] == key:
            return list(lst[:i]) + [item] + list(lst[i + 1:])
    raise ValueError(f"Key {key} not found.")


def getScaler(operations):
    """Batch -> Batch normaliser built from ``(key, ('scale', c))`` / ``(key, ('shift', 'mean'|other_key[, sign]))``."""

    def scaler(batch):
        batch = batch.clone()
        seg = batch["_node_segment"] if "_node_segment" in batch else batch.nodeSegment()
        for key, op in operations:
            if op[0] == "scale":
                for k in (key if isinstance(key, (tuple, list)) else (key,)):
                    batch[k] = batch[k] * op[1]
            elif op[0] == "shift":
                if op[1] == "mean":
                    n = batch["_n_nodes"].view(-1, 1).to(batch[key].dtype)
                    center = torch.zeros(n.shape[0], batch[key].shape[1], dtype=batch[key].dtype, device=batch[key].device)
                    center.index_add_(0, seg, batch[key])
                    batch[key] = batch[key] - (center / n)[seg]
                elif op[1] in batch:
                    sign = op[2] if len(op) == 3 else 1
                    batch[key] = batch[key] + sign * batch[op[1]]
                else:
                    raise ValueError(op)
            else:
                raise ValueError(op)
        return batch

    return scaler


def countParameters(model) -> int:
    return sum(p.numel() for p in model.parameters() if p.requires_grad)
