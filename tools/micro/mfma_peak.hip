// What the f32 matrix pipe sustains on this box: every SIMD of every CU runs `waves` waves, each a chain of
// v_mfma_f32_32x32x2_f32 on `NACC` independent accumulators, no memory traffic.  Prints TFLOP/s per configuration.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void chain(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int n = 0; n < NACC; ++n)
    for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
  float a = a0 + threadIdx.x * 1e-9f, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[n], 0, 0, 0);
  }
  float s = 0.f;
  for (int n = 0; n < NACC; ++n)
    for (int i = 0; i < 16; ++i) s += acc[n][i];
  if (s == 12345.678f) out[0] = s;
}

template <int NACC>
void run(int blocks_per_cu, int iters, float* d) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int blocks = 256 * blocks_per_cu;
  hipLaunchKernelGGL(chain<NACC>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f, 1e-30f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(chain<NACC>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f, 1e-30f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)blocks * 4 * iters * 8 * NACC * 4096.0;
  printf("accumulators %d, waves/SIMD %d, %d MFMAs per wave: %.3f ms  %.1f TFLOP/s\n", NACC, blocks_per_cu, iters * 8 * NACC, ms,
         flop / ms / 1e9);
}

int main() {
  float* d;
  hipMalloc(&d, 4);
  for (int iters : {200, 2000, 20000}) {
    run<1>(1, iters, d);
    run<2>(1, iters, d);
    run<4>(1, iters, d);
    run<2>(2, iters, d);
    run<2>(4, iters, d);
  }
  return 0;
}
