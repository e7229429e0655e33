// Per-batch graph topology on the device for gfx950: int32 endpoints and the two CSR views of the edge list
// (by destination for the forward reduce and the weight gradients, by source for the feature gradients), with the
// edge ids ASCENDING inside every segment -- the order of a stable sort by endpoint, i.e. the summation order of the
// reference's CPU scatter (torch_runstats.scatter -> index_add_: e3_layers/nn/message_passing.py:109) -- plus the
// tile ownership lists of the radial-fused kernels.  SURVEY.md 8f-1; replaces, per batch, two stable argsorts, two
// bincounts, two cumsums, two searchsorteds and their dtype conversions (about 25 launches) by five:
//   count   one thread per edge: int32 endpoints out, atomic histogram of both endpoints
//   scan    one workgroup per CSR: exclusive scan of the histogram into the row pointers (+ a cursor copy)
//   fill    one thread per edge: claims a slot in its source row and its destination row (order arbitrary)
//   sort    one wave per (node, CSR): ranks the row's edge ids -> ascending order (rows are a node's degree: register tiles)
// The result does not depend on the order in which the atomics of `fill` land.
#include "e3k_common.h"

namespace e3k {

__global__ __launch_bounds__(256) void csr_count_kernel(const int64_t* __restrict__ edge_index, int64_t E, int32_t N,
                                                        int32_t* __restrict__ src, int32_t* __restrict__ dst,
                                                        int32_t* __restrict__ dst_ptr, int32_t* __restrict__ src_ptr,
                                                        int32_t* __restrict__ bad) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  const int64_t s = edge_index[e], d = edge_index[E + e];
  if (s < 0 || s >= N || d < 0 || d >= N) {   // reported by the host wrapper after the build (no sync here)
    *bad = 1;
    src[e] = 0;
    dst[e] = 0;
    return;
  }
  src[e] = (int32_t)s;
  dst[e] = (int32_t)d;
  atomicAdd(src_ptr + s + 1, 1);
  atomicAdd(dst_ptr + d + 1, 1);
}

// ptr[0] = 0, ptr[i + 1] = counts of nodes 0..i (counts arrive in ptr[1..N]); cursor[i] = ptr[i].  One workgroup.
__global__ __launch_bounds__(1024) void csr_scan_kernel(int32_t* __restrict__ dst_ptr, int32_t* __restrict__ src_ptr,
                                                        int32_t* __restrict__ dst_cur, int32_t* __restrict__ src_cur, int32_t N) {
  int32_t* ptr = blockIdx.x == 0 ? dst_ptr : src_ptr;
  int32_t* cur = blockIdx.x == 0 ? dst_cur : src_cur;
  __shared__ int32_t wave_tot[16];
  __shared__ int32_t carry_s;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  if (t == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < N; base += 1024) {
    const int i = base + t;
    const int32_t v = i < N ? ptr[i + 1] : 0;
    int32_t x = v;      // inclusive scan inside the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int32_t y = __shfl_up(x, off, 64);
      if (lane >= off) x += y;
    }
    if (lane == 63) wave_tot[w] = x;
    __syncthreads();
    int32_t before = carry_s;
    for (int k = 0; k < w; ++k) before += wave_tot[k];
    const int32_t incl = before + x;
    if (i < N) {
      ptr[i + 1] = incl;
      cur[i] = incl - v;
    }
    __syncthreads();
    if (t == 1023) carry_s = incl;
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void csr_fill_kernel(const int32_t* __restrict__ src, const int32_t* __restrict__ dst, int64_t E,
                                                       int32_t* __restrict__ dst_cur, int32_t* __restrict__ src_cur,
                                                       int32_t* __restrict__ dst_tmp, int32_t* __restrict__ src_tmp) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  dst_tmp[atomicAdd(dst_cur + dst[e], 1)] = (int32_t)e;
  src_tmp[atomicAdd(src_cur + src[e], 1)] = (int32_t)e;
}

// one wave per (node, CSR): out[beg + rank(v)] = v with rank = number of smaller ids in the row (ids are distinct)
__global__ __launch_bounds__(256) void csr_sort_kernel(const int32_t* __restrict__ dst_ptr, const int32_t* __restrict__ src_ptr,
                                                       const int32_t* __restrict__ dst_tmp, const int32_t* __restrict__ src_tmp,
                                                       int32_t* __restrict__ dst_perm, int32_t* __restrict__ src_perm, int32_t N) {
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= 2 * (int64_t)N) return;
  const bool by_src = item >= N;
  const int node = (int)(by_src ? item - N : item);
  const int32_t* ptr = by_src ? src_ptr : dst_ptr;
  const int32_t* in = by_src ? src_tmp : dst_tmp;
  int32_t* out = by_src ? src_perm : dst_perm;
  const int beg = uniform(ptr[node]), len = uniform(ptr[node + 1]) - beg;
  const int lane = threadIdx.x & 63;
  if (len <= 64) {
    const int32_t v = lane < len ? in[beg + lane] : 0x7fffffff;
    int rank = 0;
    for (int j = 0; j < len; ++j) rank += __shfl(v, j, 64) < v ? 1 : 0;
    if (lane < len) out[beg + rank] = v;
  } else {
    // long rows (node degree > 64: dense graphs): tiles of 64 candidates (one per lane) are ranked against tiles of 64 row
    // elements held in registers and broadcast lane by lane -- len^2 / 64 register-only steps (round 3: every lane re-read the
    // whole row from memory per element, len^2 / 64 dependent loads; the knot bins of the radial table, where rows reach
    // thousands, no longer come through here at all: csrc/e3k_rtable.hip groups them with a stable counting sort)
    for (int i0 = 0; i0 < len; i0 += 64) {
      const bool mine = i0 + lane < len;
      const int32_t v = mine ? in[beg + i0 + lane] : 0x7fffffff;
      int rank = 0;
      for (int j0 = 0; j0 < len; j0 += 64) {
        const int32_t u = j0 + lane < len ? in[beg + j0 + lane] : 0x7fffffff;
        const int m = len - j0 < 64 ? len - j0 : 64;
        for (int j = 0; j < m; ++j) rank += __shfl(u, j, 64) < v ? 1 : 0;
      }
      if (mine) out[beg + rank] = v;
    }
  }
}

}  // namespace e3k

extern "C" int64_t e3k_csr_workspace_ints(int64_t N, int64_t E) { return 2 * N + 2 * E + 2; }

extern "C" int e3k_csr_build(const int64_t* edge_index, int64_t N, int64_t E, int32_t* src, int32_t* dst, int32_t* dst_ptr,
                             int32_t* dst_perm, int32_t* src_ptr, int32_t* src_perm, int32_t* workspace, int32_t* bad_flag,
                             void* stream) {
  if (N < 0 || E < 0) return E3K_ERR_INVALID;
  if (N >= 0x7fffffffLL || E >= 0x7fffffffLL) return E3K_ERR_UNSUPPORTED;
  if (!dst_ptr || !src_ptr || !bad_flag) return E3K_ERR_INVALID;
  if (E > 0 && (!edge_index || !src || !dst || !dst_perm || !src_perm || !workspace)) return E3K_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  if (dst_ptr == bad_flag + 1 && src_ptr == dst_ptr + (N + 1)) {   // one block [flag | dst_ptr | src_ptr]: one fill
    if (e3k::zero_fill(bad_flag, sizeof(int32_t) * (2 * (N + 1) + 1), st)) return E3K_ERR_INVALID;
  } else {
    if (e3k::zero_fill(dst_ptr, sizeof(int32_t) * (N + 1), st) || e3k::zero_fill(src_ptr, sizeof(int32_t) * (N + 1), st) ||
        e3k::zero_fill(bad_flag, sizeof(int32_t), st))
      return E3K_ERR_INVALID;
  }
  if (E > 0 && N > 0) {
    int32_t* dst_cur = workspace;
    int32_t* src_cur = workspace + N;
    int32_t* dst_tmp = workspace + 2 * N;
    int32_t* src_tmp = workspace + 2 * N + E;
    const unsigned eb = (unsigned)((E + 255) / 256);
    hipLaunchKernelGGL(e3k::csr_count_kernel, dim3(eb), dim3(256), 0, st, edge_index, E, (int32_t)N, src, dst, dst_ptr, src_ptr, bad_flag);
    hipLaunchKernelGGL(e3k::csr_scan_kernel, dim3(2), dim3(1024), 0, st, dst_ptr, src_ptr, dst_cur, src_cur, (int32_t)N);
    hipLaunchKernelGGL(e3k::csr_fill_kernel, dim3(eb), dim3(256), 0, st, src, dst, E, dst_cur, src_cur, dst_tmp, src_tmp);
    hipLaunchKernelGGL(e3k::csr_sort_kernel, dim3((unsigned)((2 * N + 3) / 4)), dim3(256), 0, st, dst_ptr, src_ptr, dst_tmp, src_tmp,
                       dst_perm, src_perm, (int32_t)N);
  }
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

// ------------------------------------------------------------------------------------------
// rows grouped by a small categorical key (the keyed self-connection: nodes by species)
// ------------------------------------------------------------------------------------------
// perm = row ids sorted by key, STABLE (ascending row id inside a key); bounds[k] = {start, count}; reps[k] = first row of
// key k (row 0 for an absent key).  One workgroup, one launch, no host sync: replaces a stable argsort (radix sort:
// ~8 launches), a scatter_add, a cumsum, a gather and their dtype conversions per batch.  Rows are placed chunk by
// chunk (1024 rows): every wave ranks its lanes per key with ballots (no barrier), the 16 x K per-wave counts are
// prefixed by one thread each, and a running offset per key carries over to the next chunk.
namespace e3k {
constexpr int GR_MAXK = 256;
constexpr int GR_EPT = 16 * GR_MAXK / 1024;      // (wave, key) table entries per thread

__global__ __launch_bounds__(1024) void group_rows_kernel(const int64_t* __restrict__ key, int32_t R, int32_t K,
                                                          int32_t* __restrict__ perm, int32_t* __restrict__ bounds,
                                                          int64_t* __restrict__ reps, int32_t* __restrict__ bad) {
  __shared__ int32_t count[GR_MAXK], start[GR_MAXK], running[GR_MAXK];
  __shared__ int32_t hist[16][GR_MAXK];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  if (t < GR_MAXK) count[t] = 0, running[t] = 0;
  if (t == 0) *bad = 0;      // (one workgroup: cleared here, in front of the barrier, instead of by a fill launch of its own)
  __syncthreads();
  for (int i = t; i < R; i += 1024) {
    const int64_t k = key[i];
    if (k < 0 || k >= K) *bad = 1;
    else atomicAdd(&count[(int)k], 1);
  }
  __syncthreads();
  if (t == 0) {
    int acc = 0;
    for (int k = 0; k < K; ++k) {
      start[k] = acc;
      bounds[2 * k] = acc;
      bounds[2 * k + 1] = count[k];
      acc += count[k];
    }
  }
  __syncthreads();
  for (int base = 0; base < R; base += 1024) {
    const int i = base + t;
    int k = -1;
    if (i < R) {
      const int64_t kk = key[i];
      k = (kk >= 0 && kk < K) ? (int)kk : -1;
    }
    int rank = 0;
    for (int q = 0; q < K; ++q) {
      const unsigned long long b = __ballot(k == q);
      if (k == q) rank = __popcll(b & ((1ull << lane) - 1ull));
      if (lane == 0) hist[w][q] = __popcll(b);
    }
    __syncthreads();
    // table entry (tw, tq): rows of key tq in the waves before tw; 16 x K entries over 1024 threads
    int before[GR_EPT];
#pragma unroll
    for (int j = 0; j < GR_EPT; ++j) {
      const int e = t + 1024 * j, tw = e / K, tq = e - tw * K;
      before[j] = 0;
      if (e < 16 * K)
        for (int v = 0; v < tw; ++v) before[j] += hist[v][tq];
    }
    __syncthreads();
    int total_q[GR_EPT];
#pragma unroll
    for (int j = 0; j < GR_EPT; ++j) {
      const int e = t + 1024 * j, tw = e / K, tq = e - tw * K;
      total_q[j] = 0;
      if (e < 16 * K) {
        if (tw == 15) total_q[j] = before[j] + hist[15][tq];
        hist[tw][tq] = before[j];            // now: offset of wave tw inside the chunk's rows of key tq
      }
    }
    __syncthreads();
    if (k >= 0) perm[start[k] + running[k] + hist[w][k] + rank] = i;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < GR_EPT; ++j) {
      const int e = t + 1024 * j, tw = e / K, tq = e - tw * K;
      if (e < 16 * K && tw == 15) running[tq] += total_q[j];
    }
    __syncthreads();
  }
  if (t < K) reps[t] = count[t] > 0 ? perm[start[t]] : 0;
}
}  // namespace e3k

// ---- edge records (round 5): everything the tensor-product kernels need to know about the t-th edge of a CSR walk in ONE aligned
// 64-byte block -- {neighbour node, knot, four interpolation weights, nine spherical harmonics, edge id} -- in the order of the walk.
// The kernels used to chase perm[t] -> e -> {nbr[e], bin[e], coef[e], sh[e]} (three dependent scalar round trips per edge and wave,
// in front of the row loads whose addresses they feed); a record is one s_load_dwordx16 whose address follows from the loop
// counter, so the next edge's record is fetched while this edge is computed.  Built once per batch and direction (4.4 MB at 70 k
// edges), shared by every layer.
namespace e3k {
__global__ __launch_bounds__(256) void edge_records_kernel(const int32_t* __restrict__ perm, const int32_t* __restrict__ nbr,
                                                           const int32_t* __restrict__ bin, const float* __restrict__ coef,
                                                           const float* __restrict__ sh, int d_sh, int64_t E, int32_t* __restrict__ rec) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;      // one thread per dword of the records
  if (q >= E * 16) return;
  const int64_t t = q >> 4;
  const int f = (int)(q & 15);
  const int e = perm[t];
  int32_t v = 0;
  if (f == 0) v = nbr[e];
  else if (f == 1) v = bin ? bin[e] : 0;
  else if (f < 6) v = coef ? __float_as_int(coef[4 * (int64_t)e + (f - 2)]) : 0;
  else if (f < 15) v = (f - 6) < d_sh ? __float_as_int(sh[(int64_t)e * d_sh + (f - 6)]) : 0;
  else v = e;
  rec[q] = v;
}
}  // namespace e3k

extern "C" int e3k_edge_records(const int32_t* perm, const int32_t* nbr, const int32_t* bin, const float* coef, const float* sh,
                                int32_t d_sh, int64_t E, int32_t* rec, void* stream) {
  if (E < 0 || d_sh <= 0 || d_sh > 9 || (bin != nullptr) != (coef != nullptr)) return E3K_ERR_INVALID;
  if (E == 0) return E3K_OK;
  if (!perm || !nbr || !sh || !rec || (reinterpret_cast<uintptr_t>(rec) & 63)) return E3K_ERR_INVALID;
  hipLaunchKernelGGL(e3k::edge_records_kernel, dim3((unsigned)((E * 16 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, perm, nbr, bin,
                     coef, sh, d_sh, E, rec);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int e3k_group_rows(const int64_t* key, int64_t R, int32_t K, int32_t* perm, int32_t* bounds, int64_t* reps,
                              int32_t* bad_flag, void* stream) {
  if (R < 0 || K <= 0) return E3K_ERR_INVALID;
  if (K > e3k::GR_MAXK || R >= 0x7fffffffLL) return E3K_ERR_UNSUPPORTED;
  if (!bounds || !reps || !bad_flag || (R > 0 && (!key || !perm))) return E3K_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(e3k::group_rows_kernel, dim3(1), dim3(1024), 0, st, key, (int32_t)R, K, perm, bounds, reps, bad_flag);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}
