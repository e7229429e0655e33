import ctypes as C, torch
hip = C.CDLL("libamdhip64.so")
hip.hipGetErrorString.restype = C.c_char_p
def ck(what, rc):
    print(f"{what:50s} -> {hip.hipGetErrorString(rc).decode()}")
a, b = C.c_void_p(), C.c_void_p()
ck("hipEventCreate a", hip.hipEventCreate(C.byref(a)))
ck("hipEventCreate b", hip.hipEventCreate(C.byref(b)))
x = torch.randn(1 << 24, device="cuda")
x.mul_(1.0); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
for mode in ("global", "thread_local", "relaxed"):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode=mode):
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        ck(f"[{mode}] record a external", hip.hipEventRecordWithFlags(a, st, 1))
        for _ in range(20):
            x.mul_(1.0000001)
        ck(f"[{mode}] record b external", hip.hipEventRecordWithFlags(b, st, 1))
    g.replay(); torch.cuda.synchronize()
    ms = C.c_float(-1)
    ck("elapsed", hip.hipEventElapsedTime(C.byref(ms), a, b)); print("   ms", ms.value)
