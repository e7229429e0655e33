import torch, time
dev = torch.device("cuda:0")
n, d_mid, d_conv = 4608, 6528, 1408
mid = torch.randn(n, d_mid, device=dev); g = torch.randn(n, d_conv, device=dev)
# the seven problems of the trailing Linear (layer 3): (in_off, mul_in, dim, out_off, mul_out)
probs = [(0, 128, 1, 64, 64), (128, 192, 1, 0, 64), (128, 192, 1, 128, 256), (320, 384, 3, 576, 64), (1472, 320, 3, 384, 64), (2432, 320, 5, 1088, 64), (4032, 384, 5, 768, 64)]
outs = [torch.zeros(k, no, device=dev) for (_, k, _, _, no) in probs]
def run():
    for (io, k, dim, oo, no), w in zip(probs, outs):
        for m in range(dim):
            a = mid[:, io + m * k: io + (m + 1) * k]          # [n, k], row stride d_mid
            b = g[:, oo + m * no: oo + (m + 1) * no]         # e3nn/cf layout detail ignored: same shapes and strides
            torch.addmm(w, a.t(), b, out=w) if False else w.addmm_(a.t(), b)
for _ in range(5): run()
torch.cuda.synchronize()
a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a0.record()
for _ in range(30): run()
a1.record(); torch.cuda.synchronize()
flops = sum(2.0 * n * dim * k * no for (_, k, dim, _, no) in probs)
us = a0.elapsed_time(a1) * 1e3 / 30
print(f"rocBLAS/hipBLASLt via torch.addmm_: {sum(d for _,_,d,_,_ in probs)} calls, {us:.1f} us per set, {flops/us/1e6:.1f} TF/s")
