"""Do two single-stream HIP graphs replayed on two streams overlap on the device?  (round 6: PipelinedBucketedStep)"""
import sys, time
import torch
dev = torch.device("cuda:0")
prio = int(sys.argv[1]) if len(sys.argv) > 1 else 0
x = torch.randn(64, 1 << 20, device=dev)       # a kernel that takes ~100 us on 64 of 256 CUs' worth of work
y = torch.randn(64, 1 << 20, device=dev)
small_a = torch.randn(256, 256, device=dev)
small_b = torch.randn(256, 256, device=dev)

def long_chain(t, n):
    for _ in range(n):
        t.mul_(1.0000001)

def small_chain(t, n):
    for _ in range(n):
        t.add_(1e-9)

s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev, priority=prio)
g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
long_chain(x, 2); small_chain(small_a, 2); torch.cuda.synchronize()
with torch.cuda.graph(g1):
    long_chain(x, 40)
with torch.cuda.graph(g2):
    small_chain(small_a, 40)
torch.cuda.synchronize()

def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n

def only1():
    with torch.cuda.stream(s1): g1.replay()
def only2():
    with torch.cuda.stream(s2): g2.replay()
def both():
    with torch.cuda.stream(s1): g1.replay()
    with torch.cuda.stream(s2): g2.replay()
def both_eager():
    with torch.cuda.stream(s1): long_chain(x, 40)
    with torch.cuda.stream(s2): small_chain(small_a, 40)
def graph_and_eager():
    with torch.cuda.stream(s1): g1.replay()
    with torch.cuda.stream(s2): small_chain(small_a, 40)
print(f"priority {prio}: g1 alone {timeit(only1):.3f} ms, g2 alone {timeit(only2):.3f} ms, both graphs {timeit(both):.3f} ms, "
      f"both eager {timeit(both_eager):.3f} ms, graph + eager {timeit(graph_and_eager):.3f} ms")
