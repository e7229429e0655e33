// Shared helpers for libe3k (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/e3k.h"

#define E3K_WAVE 64

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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

}  // namespace e3k
