#!/usr/bin/env python3
"""Vendor reference point: torch.matmul (rocBLAS / hipBLASLt, fp32) on the GEMM shapes of the radial MLP's last layer and
of the node Linears — compare with tools/gemm_time.py / tools/gemm_profile.py for the hand-written kernels."""
import torch
dev=torch.device("cuda:0")
E=69484
h=torch.randn(E,64,device=dev); W=torch.randn(64,1920,device=dev); g=torch.randn(E,1920,device=dev)
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)*1e3/reps
fl=2.0*E*64*1920
for name,fn in (("fwd  h@W", lambda: h@W), ("dgrad g@W^T", lambda: g@W.t()), ("wgrad h^T@g", lambda: h.t()@g)):
    us=timeit(fn); print(f"rocBLAS {name:14s} {us:8.1f} us {fl/us/1e6:6.1f} TF/s")
x=torch.randn(4623,1152,device=dev); W2=torch.randn(1152,1152,device=dev)
us=timeit(lambda: x@W2); print(f"rocBLAS node 4623x1152x1152 {us:8.1f} us {2.0*4623*1152*1152/us/1e6:6.1f} TF/s")
a=torch.randn(69484,1024,device=dev); b=torch.randn(1024,1024,device=dev)
us=timeit(lambda: a@b); print(f"rocBLAS 69484x1024x1024 {us:8.1f} us {2.0*69484*1024*1024/us/1e6:6.1f} TF/s")
