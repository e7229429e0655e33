#!/usr/bin/env python3
"""Runs one GEMM shape a few times (for rocprofv3 --pmc passes).  argv[1]: fwd | dgrad | wgrad | sc"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "equivariant-nn-zoo_amd")):
    sys.path.insert(0, p)
import torch
from e3_layers_amd.backend import ops
dev = torch.device("cuda:0")
E = 69484
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
spec = ops.LinearSpec(64, 1920, [ops.LinInstr(0, 0, 64, 1920, 1, 0, 1.0)], "e3nn", "e3nn", [], True, True, 64 * 1920)
h = torch.randn(E, 64, device=dev, requires_grad=(which == "dgrad"))
w = torch.randn(64 * 1920, device=dev, requires_grad=(which == "wgrad"))
for _ in range(3):
    y = ops.strided_linear(h, w, None, spec)
    if which != "fwd":
        y.backward(torch.ones_like(y))
torch.cuda.synchronize()
