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
// 16-byte stores, eight per thread and trip: a 7 MB buffer is 216 workgroups instead of 4 096 -- these fills sit on the main
// stream in front of kernels that accumulate (tp_bwd_x of plans that share input blocks, the keyed self-connection's weight
// gradient), and 16 k one-store waves queued behind a side stream's big grid took 56-78 us to get through on the protein
// net (round 4's launch census) where the bytes need 3.
static __global__ __launch_bounds__(256) void zero_words_kernel(uint32_t* __restrict__ p, int64_t n) {
  // [0, head): words in front of the first 16-byte boundary; [head, head + 4 * n16): uint4 body; the rest: tail words
  const int64_t head = ((16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15) / 4 < n ? ((16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15) / 4 : n;
  const int64_t n16 = (n - head) / 4;
  uint4* __restrict__ q = reinterpret_cast<uint4*>(p + head);
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) q[i] = z;
  if (blockIdx.x == 0) {
    if ((int64_t)threadIdx.x < head) p[threadIdx.x] = 0u;
    const int64_t t0 = head + 4 * n16;
    if (t0 + threadIdx.x < n) p[t0 + threadIdx.x] = 0u;      // (fewer than four tail words)
  }
}
static inline int zero_fill(void* p, int64_t bytes, hipStream_t st) {
  if (bytes <= 0) return 0;
  if ((reinterpret_cast<uintptr_t>(p) & 3) || (bytes & 3)) return -1;
  const int64_t n = bytes / 4;
  int64_t blocks = (bytes + 32767) / 32768;      // 256 threads x 8 x 16 bytes per workgroup and trip
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)blocks), dim3(256), 0, st, static_cast<uint32_t*>(p), n);
  return 0;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

}  // namespace e3k
