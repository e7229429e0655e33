// Does hipEventRecordWithFlags(hipEventRecordExternal) work inside a stream capture on this runtime, and do the recorded events
// time the kernels of a replay?  hipcc --offload-arch=gfx950 tools/micro/ext_event.hip -o /tmp/ext_event && /tmp/ext_event
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(float* p, int n) {
  float v = p[threadIdx.x];
  for (int i = 0; i < n; ++i) v = v * 1.000001f + 0.5f;
  p[threadIdx.x] = v;
}
#define CK(x) do { hipError_t e_ = (x); printf("%-60s -> %s\n", #x, hipGetErrorString(e_)); } while (0)
int main() {
  float* d; hipMalloc(&d, 1024);
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int mode = 0; mode < 3; ++mode) {
    hipStreamCaptureMode m = mode == 0 ? hipStreamCaptureModeGlobal : mode == 1 ? hipStreamCaptureModeThreadLocal : hipStreamCaptureModeRelaxed;
    printf("capture mode %d\n", mode);
    CK(hipStreamBeginCapture(st, m));
    CK(hipEventRecordWithFlags(a, st, hipEventRecordExternal));
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, d, 200000);
    CK(hipEventRecordWithFlags(b, st, hipEventRecordExternal));
    hipGraph_t g; CK(hipStreamEndCapture(st, &g));
    hipGraphExec_t ge; CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 2; ++r) {
      CK(hipGraphLaunch(ge, st));
      CK(hipStreamSynchronize(st));
      float ms = -1; CK(hipEventElapsedTime(&ms, a, b));
      printf("   replay %d: %.3f ms between the external events\n", r, ms);
    }
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
  }
  return 0;
}
