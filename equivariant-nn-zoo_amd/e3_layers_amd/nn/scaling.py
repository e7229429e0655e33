"""``PerTypeScaleShift`` — interface of ``e3_layers/nn/scaling.py:9-67`` (index + fma plumbing)."""
from __future__ import annotations

from typing import List, Optional

import torch

from .sequential import Module


class PerTypeScaleShift(Module):
    def __init__(self, num_types: int, shifts: Optional[List[float]], scales: Optional[List[float]],
                 scales_trainable: bool = False, shifts_trainable: bool = False,
                 irreps_in="1x0e", irreps_out="1x0e", species="1x0e"):
        super().__init__()
        self.num_types = num_types
        self.init_irreps(input=irreps_in, output=irreps_out, species=species, output_keys=["output"])
        self.has_shifts, self.has_scales = shifts is not None, scales is not None
        for name, value, trainable in (("shifts", shifts, shifts_trainable), ("scales", scales, scales_trainable)):
            if value is None:
                continue
            t = torch.as_tensor(value, dtype=torch.get_default_dtype()).reshape(-1)
            if t.numel() == 1:
                t = t.repeat(num_types)
            if t.shape != (num_types,):
                raise ValueError(f"invalid shape of {name}: {tuple(t.shape)}")
            if trainable:
                setattr(self, name, torch.nn.Parameter(t.clone()))
            else:
                self.register_buffer(name, t.clone())

    def forward(self, data, attrs):
        species, x = data["species"].view(-1), data["input"]
        if self.has_scales:
            x = self.scales.to(x.device)[species].view(-1, 1) * x
        if self.has_shifts:
            x = self.shifts.to(x.device)[species].view(-1, 1) + x
        return {"output": x}, {"output": (attrs["input"][0], self.irreps_out["output"])}
