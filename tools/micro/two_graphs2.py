"""The event pattern of PipelinedBucketedStep with synthetic graphs: main graph M (long), prep graph P (a chain of small kernels),
two buffers.  Per-iteration time: M alone vs the pipelined pattern vs P then M in line."""
import sys, time
import torch
sys.path.insert(0, "equivariant-nn-zoo_amd")
from e3_layers_amd.run.graph_step import stream_beside
dev = torch.device("cuda:0")
x = torch.randn(64, 1 << 20, device=dev)
small = [torch.randn(256, 256, device=dev) for _ in range(2)]
main = torch.cuda.current_stream(dev)
prep = stream_beside(main)
M = [torch.cuda.CUDAGraph() for _ in range(2)]
P = [torch.cuda.CUDAGraph() for _ in range(2)]
x.mul_(1.0); small[0].add_(0.0); torch.cuda.synchronize()
for b in range(2):
    with torch.cuda.graph(P[b]):
        for _ in range(100):
            small[b].add_(1e-9)
    with torch.cuda.graph(M[b]):
        y = small[b].sum()
        for _ in range(40):
            x.mul_(1.0000001)
torch.cuda.synchronize()
ev_p = [torch.cuda.Event() for _ in range(2)]
ev_m = [torch.cuda.Event() for _ in range(2)]

def loop(n, mode):
    ran = [False, False]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    P[0].replay()
    for t in range(n):
        b = t % 2
        if mode == "pipe":
            if t > 0:
                main.wait_event(ev_p[b])
            M[b].replay(); ev_m[b].record(main); ran[b] = True
            o = 1 - b
            if ran[o]:
                prep.wait_event(ev_m[o])
            else:
                prep.wait_stream(main)
            with torch.cuda.stream(prep):
                P[o].replay(); ev_p[o].record(prep)
        elif mode == "inline":
            if t > 0:
                P[b].replay()
            M[b].replay()
        else:
            M[b].replay()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n

for mode in ("only-M", "inline", "pipe", "only-M", "inline", "pipe"):
    print(mode, f"{loop(40, mode):.3f} ms / iteration")

# where does P run relative to M?  timing events around both, pipelined pattern
ran = [False, False]
marks = []
torch.cuda.synchronize()
P[0].replay()
for t in range(12):
    b = t % 2
    if t > 0:
        main.wait_event(ev_p[b])
    m0, m1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    m0.record(main); M[b].replay(); m1.record(main); ev_m[b].record(main); ran[b] = True
    o = 1 - b
    if ran[o]:
        prep.wait_event(ev_m[o])
    else:
        prep.wait_stream(main)
    with torch.cuda.stream(prep):
        p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        p0.record(prep); P[o].replay(); p1.record(prep); ev_p[o].record(prep)
    marks.append((m0, m1, p0, p1))
torch.cuda.synchronize()
base = marks[4][0]
for t in range(4, 10):
    m0, m1, p0, p1 = marks[t]
    print(f"t={t}: M [{base.elapsed_time(m0):7.3f}, {base.elapsed_time(m1):7.3f}]  P(next) [{base.elapsed_time(p0):7.3f}, {base.elapsed_time(p1):7.3f}] ms iteration")
