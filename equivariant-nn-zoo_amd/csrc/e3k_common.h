// Shared helpers for libe3k (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/e3k.h"

#define E3K_WAVE 64

// Tuning constants of the library.  The PRODUCT build (make, __graft_entry__.build()) compiles them in: no environment variable
// changes what libe3k.so computes or how it launches.  `make dbg` (-DE3K_DEBUG_KNOBS, libe3k_dbg.so, loaded only through
// E3K_LIB=...) reads the same names from the environment for experiments (tools/): sweeps of tile counts, and the timing-only
// ablation mask E3K_ABLATE, with which the library skips kernel families and returns WRONG results by design.
#ifdef E3K_DEBUG_KNOBS
#include <cstdlib>
#define E3K_KNOB_INT(var, name, dflt) static const long long var = getenv(name) ? atoll(getenv(name)) : (dflt)
#else
#define E3K_KNOB_INT(var, name, dflt) static constexpr long long var = (dflt)
#endif

#define E3K_CHECK_LAUNCH()                          \
  do {                                              \
    hipError_t _e = hipGetLastError();              \
    if (_e != hipSuccess) return E3K_ERR_LAUNCH;    \
  } while (0)

namespace e3k {

__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// XCD-aware remap of a linear block id: blocks b and b+8 share an XCD (round-robin dispatch,
// MI355X_MICROARCH "Workgroup dispatch"), so give XCD x the contiguous slice
// [x*per, (x+1)*per) of the work list: neighbouring work items (same graph, same rows of x)
// then hit the same 4 MiB L2.  Bijective for any grid size; speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int b, int nblocks) {
  const int nx = 8;
  const int per = nblocks / nx, rem = nblocks % nx;
  const int x = b % nx, i = b / nx;
  // XCD x owns per (+1 if x < rem) items
  const int start = x * per + (x < rem ? x : rem);
  return start + i;
}

// Zero-fill as a KERNEL, not hipMemsetAsync: memset nodes recorded while a stream is being captured did not replay
// correctly here (a HIP graph holding e3k_csr_build faulted on its second replay with "write access to a read-only
// page"; with this kernel in place of the three memsets it replays) -- and one launch path is one thing less to reason about.
static __global__ __launch_bounds__(256) void zero_words_kernel(uint32_t* __restrict__ p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = 0u;
}
static inline int zero_fill(void* p, int64_t bytes, hipStream_t st) {
  if (bytes <= 0) return 0;
  if ((reinterpret_cast<uintptr_t>(p) & 3) || (bytes & 3)) return -1;
  const int64_t n = bytes / 4;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)blocks), dim3(256), 0, st, static_cast<uint32_t*>(p), n);
  return 0;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

}  // namespace e3k
