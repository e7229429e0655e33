// Grouped strided GEMM on the gfx950 f32 matrix pipe (v_mfma_f32_32x32x2_f32: exact fp32 FMA
// chain, 64 FLOP/clk/SIMD — cdna_hip_programming.md §3 "FP32-input MFMA").
//
// Replaces on the node / edge side of a convolution layer (paths relative to /root/reference):
//   o3.Linear                          e3_layers/nn/message_passing.py:58,102; nn/pointwise.py:18,87,142
//   FullyConnectedNet layers           e3_layers/nn/message_passing.py:74,93      (radial MLP)
//   FullyConnectedTensorProduct (sc)   e3_layers/nn/message_passing.py:83,100     (outer mode)
//
// forward / dgrad kernel  (e3k_gemm):   C[(r1,r2), n] = alpha * sum_k Aeff[(r1,r2), k] B[k, n]
//   tile 128 rows x 64 cols x 32 k, 256 threads = 4 waves, wave w owns rows [32w, 32w+32) x 64 cols
//   (two 32x32 accumulators).  A tile in LDS row-major with an odd row stride (33) so that the
//   MFMA A fragment (lane -> row l&31, k l>>5) is bank-conflict free; B tile row-major [k][n].
//   dgrad (dA = dC . B^T) is the same kernel with b_k / b_n swapped by the caller.
//   Outer mode: Aeff[(r1,r2), u*V+v] = X[(r1,r2), u] * attrs[r1, v]; X and attrs tiles sit in
//   LDS for the whole K loop and the product is formed when the fragment is read.
// wgrad kernel (e3k_gemm_wgrad):         B[k, n] += alpha * sum_rows Aeff[row, k] G[row, n]
//   tile 64 k x 64 n, rows split over blocks, 64-row chunks staged in natural row-major order
//   (both MFMA operands are then conflict-free), fp32 atomics on the small output.
#include "e3k_common.h"

namespace e3k {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GEMM_MAXP = 8;
struct GemmBatch {
  int n;
  int tile_start[GEMM_MAXP + 1];
  int flags[GEMM_MAXP];  // bit0: A float4-loadable, bits1-2: B mode (0 scalar, 1 n-contiguous vec, 2 k-contiguous vec)
  int splits[GEMM_MAXP]; // wgrad only
  e3k_gemm_problem p[GEMM_MAXP];
};

constexpr int BM = 128, BN = 64, BK = 32;
constexpr int LDA = BK + 1;
constexpr int LDB = BN;
constexpr int XU = 64;          // outer mode: channels of X kept per super-step
constexpr int LDX = XU + 1;
constexpr int VMAX = 32;        // outer mode: max attrs width
constexpr int LDV = VMAX + 1;

__device__ __forceinline__ void find_problem(const GemmBatch& gb, int bid, int& pi, int& local) {
  pi = 0;
#pragma unroll
  for (int i = 1; i < GEMM_MAXP; ++i)
    if (i < gb.n && bid >= gb.tile_start[i]) pi = i;
  local = bid - gb.tile_start[pi];
}

__device__ __forceinline__ void stage_b(const e3k_gemm_problem& P, int bmode, int k0, int n0, float* Bs) {
  const int t = threadIdx.x;
  if (bmode == 1) {
    // rows of B contiguous along n: float4 loads, 16-byte LDS stores
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int k = (t >> 4) + 16 * pass, nq = (t & 15) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k0 + k < P.K && n0 + nq < P.N) v = *reinterpret_cast<const float4*>(P.B + (int64_t)(k0 + k) * P.b_k + (n0 + nq));
      *reinterpret_cast<float4*>(Bs + k * LDB + nq) = v;
    }
  } else if (bmode == 2) {
    // B^T view: contiguous along k
    const int n = t & 63, kq = (t >> 6) * 8;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      const int k = kq + 4 * h;
      if (n0 + n < P.N && k0 + k < P.K) v = *reinterpret_cast<const float4*>(P.B + (int64_t)(n0 + n) * P.b_n + (k0 + k));
      Bs[(k + 0) * LDB + n] = v.x;
      Bs[(k + 1) * LDB + n] = v.y;
      Bs[(k + 2) * LDB + n] = v.z;
      Bs[(k + 3) * LDB + n] = v.w;
    }
  } else {
    const int n = t & 63, kb = t >> 6;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = kb + 4 * j;
      float v = 0.f;
      if (n0 + n < P.N && k0 + k < P.K) v = P.B[(int64_t)(k0 + k) * P.b_k + (int64_t)(n0 + n) * P.b_n];
      Bs[k * LDB + n] = v;
    }
  }
}

__device__ __forceinline__ void epilogue(const e3k_gemm_problem& P, const f32x16 (&acc)[2], int row0, int n0) {
  const int lane = threadIdx.x & 63, wr = threadIdx.x >> 6;
  const int M = P.M1 * P.M2;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + j * 32 + (lane & 31);
    if (n >= P.N) continue;
    const float bias = P.bias ? P.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int R = row0 + wr * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
      if (R >= M) continue;
      const int r1 = R / P.M2, r2 = R - r1 * P.M2;
      float* c = P.C + (int64_t)r1 * P.c_r1 + (int64_t)r2 * P.c_r2 + (int64_t)n * P.c_n;
      float v = P.alpha * acc[j][i] + bias;
      if (P.accumulate) v += *c;
      *c = v;
    }
  }
}

// ---------------------------------------------------------------------------------------
// plain forward / dgrad
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemm_kernel(const GemmBatch gb) {
  __shared__ float As[BM * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[BK * LDB];
  int pi, local;
  find_problem(gb, blockIdx.x, pi, local);
  const e3k_gemm_problem& P = gb.p[pi];
  const int flags = gb.flags[pi];
  const int M = P.M1 * P.M2;
  const int tiles_n = (P.N + BN - 1) / BN;
  const int row0 = (local / tiles_n) * BM, n0 = (local % tiles_n) * BN;
  const int t = threadIdx.x, lane = t & 63, wr = t >> 6;

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;

  const bool avec = flags & 1;
  const int bmode = (flags >> 1) & 3;
  // per-thread source rows (fixed over the K loop)
  const float* arow[4];
  if (avec) {
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int R = row0 + (t >> 3) + 32 * pass;
      if (R < M) {
        const int r1 = R / P.M2, r2 = R - r1 * P.M2;
        arow[pass] = P.A + (int64_t)r1 * P.a_r1 + (int64_t)r2 * P.a_r2;
      } else {
        arow[pass] = nullptr;
      }
    }
  } else {
    const int R = row0 + (t >> 1);
    if (R < M) {
      const int r1 = R / P.M2, r2 = R - r1 * P.M2;
      arow[0] = P.A + (int64_t)r1 * P.a_r1 + (int64_t)r2 * P.a_r2;
    } else {
      arow[0] = nullptr;
    }
  }

  for (int k0 = 0; k0 < P.K; k0 += BK) {
    if (avec) {
      const int kq = (t & 7) * 4;
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (arow[pass] && k0 + kq < P.K) v = *reinterpret_cast<const float4*>(arow[pass] + k0 + kq);
        float* d = As + ((t >> 3) + 32 * pass) * LDA + kq;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      }
    } else {
      const int kb = (t & 1) * 16;
      float* d = As + (t >> 1) * LDA + kb;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        float v = 0.f;
        if (arow[0] && k0 + kb + j < P.K) v = arow[0][(int64_t)(k0 + kb + j) * P.a_k];
        d[j] = v;
      }
    }
    stage_b(P, bmode, k0, n0, Bs);
    __syncthreads();
    const float* ap = As + (wr * 32 + (lane & 31)) * LDA + (lane >> 5);
    const float* bp = Bs + (lane >> 5) * LDB + (lane & 31);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const float a = ap[kk];
      const float b0 = bp[kk * LDB], b1 = bp[kk * LDB + 32];
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
    }
    __syncthreads();
  }
  epilogue(P, acc, row0, n0);
}

// ---------------------------------------------------------------------------------------
// outer-mode forward:  Aeff[(r1,r2), u*V+v] = X[(r1,r2),u] * attrs[r1,v]
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemm_outer_kernel(const GemmBatch gb) {
  __shared__ float Xs[BM * LDX];
  __shared__ float Vs[BM * LDV];
  __shared__ __attribute__((aligned(16))) float Bs[BK * LDB];
  int pi, local;
  find_problem(gb, blockIdx.x, pi, local);
  const e3k_gemm_problem& P = gb.p[pi];
  const int flags = gb.flags[pi];
  const int bmode = (flags >> 1) & 3;
  const int M = P.M1 * P.M2, V = P.V, U = P.K / P.V;
  const int tiles_n = (P.N + BN - 1) / BN;
  const int row0 = (local / tiles_n) * BM, n0 = (local % tiles_n) * BN;
  const int t = threadIdx.x, lane = t & 63, wr = t >> 6;

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;

  // attrs tile: row r -> attrs[r1(r), 0..V)
  {
    const int r = t >> 1, R = row0 + r;
    const int vb = (t & 1) * 16;
    if (R < M) {
      const int r1 = R / P.M2;
      const float* src = P.A2 + (int64_t)r1 * P.a2_r1;
#pragma unroll
      for (int j = 0; j < 16; ++j) Vs[r * LDV + vb + j] = (vb + j < V) ? src[vb + j] : 0.f;
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) Vs[r * LDV + vb + j] = 0.f;
    }
  }
  const float* xrow = nullptr;
  {
    const int R = row0 + (t >> 1);
    if (R < M) {
      const int r1 = R / P.M2, r2 = R - r1 * P.M2;
      xrow = P.A + (int64_t)r1 * P.a_r1 + (int64_t)r2 * P.a_r2;
    }
  }
  const float* xp = Xs + (wr * 32 + (lane & 31)) * LDX;
  const float* vp = Vs + (wr * 32 + (lane & 31)) * LDV;
  const float* bp = Bs + (lane >> 5) * LDB + (lane & 31);

  for (int u0 = 0; u0 < U; u0 += XU) {
    __syncthreads();  // previous super-step finished reading Xs
    {
      const int ub = (t & 1) * 32;
      float* d = Xs + (t >> 1) * LDX + ub;
#pragma unroll 8
      for (int j = 0; j < 32; ++j) {
        float v = 0.f;
        if (xrow && u0 + ub + j < U) v = xrow[(int64_t)(u0 + ub + j) * P.a_k];
        d[j] = v;
      }
    }
    const int uend = (u0 + XU < U) ? u0 + XU : U;
    const int kbeg = u0 * V, kend = uend * V;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
      __syncthreads();
      stage_b(P, bmode, k0, n0, Bs);  // rows >= K are zero-filled; rows in [kend, K) belong to the next super-step
      __syncthreads();
      int kg = k0 + (lane >> 5);
      int u = kg / V, v = kg - u * V;
      u -= u0;
#pragma unroll 4
      for (int kk = 0; kk < BK; kk += 2) {
        float a = 0.f;
        if (kg < kend) a = xp[u] * vp[v];
        const float b0 = bp[kk * LDB], b1 = bp[kk * LDB + 32];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
        kg += 2;
        v += 2;
        while (v >= V) {
          v -= V;
          ++u;
        }
      }
    }
  }
  epilogue(P, acc, row0, n0);
}

// ---------------------------------------------------------------------------------------
// wgrad:  B[k, n] += alpha * sum_rows Aeff[row, k] * G[row, n]      (G passed in P.C)
// ---------------------------------------------------------------------------------------
constexpr int WK = 64, WN = 64, WR = 64;  // output tile 64x64, 64-row chunks
constexpr int LDWA = WK + 4, LDWG = WN + 4;

template <bool OUTER>
__global__ __launch_bounds__(256) void gemm_wgrad_kernel(const GemmBatch gb) {
  __shared__ __attribute__((aligned(16))) float As[WR * LDWA];
  __shared__ __attribute__((aligned(16))) float Gs[WR * LDWG];
  __shared__ float Vs[OUTER ? WR * LDV : 1];
  int pi, local;
  find_problem(gb, blockIdx.x, pi, local);
  const e3k_gemm_problem& P = gb.p[pi];
  const int flags = gb.flags[pi];
  const int M = P.M1 * P.M2;
  const int tiles_k = (P.K + WK - 1) / WK, tiles_n = (P.N + WN - 1) / WN;
  const int splits = gb.splits[pi];
  const int tile = local % (tiles_k * tiles_n), split = local / (tiles_k * tiles_n);
  const int k0 = (tile / tiles_n) * WK, n0 = (tile % tiles_n) * WN;
  const int chunk_rows = ((M + splits - 1) / splits + WR - 1) / WR * WR;
  const int rbeg = split * chunk_rows;
  const int rend = (rbeg + chunk_rows < M) ? rbeg + chunk_rows : M;
  if (rbeg >= M) return;  // block-uniform: this split has no rows
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int wk = wv >> 1, wn = wv & 1;

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  const bool avec = flags & 1;
  const bool gvec = flags & 8;
  // outer mode: this lane's k index -> (u, v), fixed for the whole loop
  int V = 1, ulo = 0, uw = 0, lu = 0, lv = 0;
  bool kvalid = true;
  if constexpr (OUTER) {
    V = P.V;
    ulo = k0 / V;
    const int klast = (k0 + WK - 1 < P.K - 1) ? k0 + WK - 1 : P.K - 1;
    uw = klast / V - ulo + 1;  // <= WK/2 + 1 < LDWA
    const int ki = k0 + wk * 32 + (lane & 31);
    kvalid = ki < P.K;
    lu = kvalid ? ki / V - ulo : 0;
    lv = kvalid ? ki % V : 0;
  }

  for (int r0 = rbeg; r0 < rend; r0 += WR) {
    // ---- stage the operand chunk (rows r0 .. r0+63)
    if constexpr (!OUTER) {
      if (avec) {
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
          const int r = (t >> 4) + 16 * pass, kq = (t & 15) * 4, R = r0 + r;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (R < rend && k0 + kq < P.K) {
            const int r1 = R / P.M2, r2 = R - r1 * P.M2;
            v = *reinterpret_cast<const float4*>(P.A + (int64_t)r1 * P.a_r1 + (int64_t)r2 * P.a_r2 + k0 + kq);
          }
          *reinterpret_cast<float4*>(As + r * LDWA + kq) = v;
        }
      } else {
        const int r = t >> 2, kb = (t & 3) * 16, R = r0 + r;
        const float* src = nullptr;
        if (R < rend) {
          const int r1 = R / P.M2, r2 = R - r1 * P.M2;
          src = P.A + (int64_t)r1 * P.a_r1 + (int64_t)r2 * P.a_r2;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          float v = 0.f;
          if (src && k0 + kb + j < P.K) v = src[(int64_t)(k0 + kb + j) * P.a_k];
          As[r * LDWA + kb + j] = v;
        }
      }
    } else {
      // X columns ulo .. ulo+uw-1 and the attrs row of each staged row
      const int r = t >> 2, R = r0 + r;
      const float* src = nullptr;
      const float* asrc = nullptr;
      if (R < rend) {
        const int r1 = R / P.M2, r2 = R - r1 * P.M2;
        src = P.A + (int64_t)r1 * P.a_r1 + (int64_t)r2 * P.a_r2;
        asrc = P.A2 + (int64_t)r1 * P.a2_r1;
      }
      for (int j = (t & 3); j < uw; j += 4) As[r * LDWA + j] = src ? src[(int64_t)(ulo + j) * P.a_k] : 0.f;
      for (int j = (t & 3); j < V; j += 4) Vs[r * LDV + j] = asrc ? asrc[j] : 0.f;
    }
    if (gvec) {
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int r = (t >> 4) + 16 * pass, nq = (t & 15) * 4, R = r0 + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (R < rend && n0 + nq < P.N) {
          const int r1 = R / P.M2, r2 = R - r1 * P.M2;
          v = *reinterpret_cast<const float4*>(P.C + (int64_t)r1 * P.c_r1 + (int64_t)r2 * P.c_r2 + n0 + nq);
        }
        *reinterpret_cast<float4*>(Gs + r * LDWG + nq) = v;
      }
    } else {
      const int r = t >> 2, nb = (t & 3) * 16, R = r0 + r;
      const float* src = nullptr;
      if (R < rend) {
        const int r1 = R / P.M2, r2 = R - r1 * P.M2;
        src = P.C + (int64_t)r1 * P.c_r1 + (int64_t)r2 * P.c_r2;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        float v = 0.f;
        if (src && n0 + nb + j < P.N) v = src[(int64_t)(n0 + nb + j) * P.c_n];
        Gs[r * LDWG + nb + j] = v;
      }
    }
    __syncthreads();
    const float* gp = Gs + (lane >> 5) * LDWG + wn * 32 + (lane & 31);
    if constexpr (!OUTER) {
      const float* ap = As + (lane >> 5) * LDWA + wk * 32 + (lane & 31);
#pragma unroll 8
      for (int rr = 0; rr < WR; rr += 2) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[rr * LDWA], gp[rr * LDWG], acc, 0, 0, 0);
      }
    } else {
      const float* ap = As + (lane >> 5) * LDWA + lu;
      const float* vp = Vs + (lane >> 5) * LDV + lv;
#pragma unroll 8
      for (int rr = 0; rr < WR; rr += 2) {
        const float a = kvalid ? ap[rr * LDWA] * vp[rr * LDV] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, gp[rr * LDWG], acc, 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // ---- atomic accumulate into B
  const int n = n0 + wn * 32 + (lane & 31);
  if (n < P.N) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int k = k0 + wk * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
      if (k < P.K) atomicAdd(const_cast<float*>(P.B) + (int64_t)k * P.b_k + (int64_t)n * P.b_n, P.alpha * acc[i]);
    }
  }
}

// ---------------------------------------------------------------------------------------
// small helpers: column sums and the self-connection backward reduction
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ G, int64_t rows, int cols, int64_t ld,
                                                      float* __restrict__ out) {
  // block = 256 threads: 64 columns x 4 row phases; grid.x tiles columns, grid.y splits rows
  __shared__ float part[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
  float s = 0.f;
  if (c < cols)
    for (int64_t r = (int64_t)blockIdx.y * 4 + ph; r < rows; r += (int64_t)gridDim.y * 4) s += G[r * ld + c];
  part[ph][threadIdx.x & 63] = s;
  __syncthreads();
  if (ph == 0 && c < cols) atomicAdd(out + c, part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// one wave per (r1, r2) row: lanes over u; H row = [U*V] (u-major, v-minor)
__global__ __launch_bounds__(256) void fctp_reduce_kernel(const float* __restrict__ H, const float* __restrict__ X,
                                                           const float* __restrict__ A2, int M1, int M2, int U, int V,
                                                           int64_t x_r1, int64_t x_r2, int64_t a2_r1,
                                                           float* __restrict__ dX, int dx_acc, float* __restrict__ dA2) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)M1 * M2) return;
  const int lane = threadIdx.x & 63;
  const int r1 = (int)(row / M2), r2 = (int)(row - (int64_t)r1 * M2);
  const float* h = H + row * (int64_t)U * V;
  const float* a = A2 + (int64_t)r1 * a2_r1;
  const float* x = X + (int64_t)r1 * x_r1 + (int64_t)r2 * x_r2;
  float* dx = dX + (int64_t)r1 * x_r1 + (int64_t)r2 * x_r2;
  float da[VMAX];
#pragma unroll
  for (int v = 0; v < VMAX; ++v) da[v] = 0.f;
  for (int u = lane; u < U; u += 64) {
    const float xv = x[u];
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < VMAX; ++v) {
      if (v < V) {
        const float hv = h[(int64_t)u * V + v];
        s = fmaf(a[v], hv, s);
        da[v] = fmaf(xv, hv, da[v]);
      }
    }
    dx[u] = dx_acc ? dx[u] + s : s;
  }
#pragma unroll
  for (int v = 0; v < VMAX; ++v) {
    if (v < V) {
      const float tot = wave_sum(da[v]);
      if (lane == 0) atomicAdd(dA2 + (int64_t)r1 * a2_r1 + v, tot);
    }
  }
}

}  // namespace e3k

// ---------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------
namespace {
bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int validate(const e3k_gemm_problem& P, bool wgrad) {
  if (P.M1 < 0 || P.M2 <= 0 || P.N <= 0 || P.K <= 0) return E3K_ERR_INVALID;
  if (!P.A || !P.B || !P.C) return E3K_ERR_INVALID;
  if (P.V < 0 || P.V > e3k::VMAX) return E3K_ERR_UNSUPPORTED;
  if (P.V > 0 && (!P.A2 || P.K % P.V != 0)) return E3K_ERR_INVALID;
  if (wgrad && P.bias) return E3K_ERR_INVALID;
  return E3K_OK;
}

int a_flags(const e3k_gemm_problem& P) {
  int f = 0;
  if (P.V == 0 && P.a_k == 1 && P.K % 4 == 0 && P.a_r1 % 4 == 0 && P.a_r2 % 4 == 0 && aligned16(P.A)) f |= 1;
  return f;
}
}  // namespace

extern "C" int e3k_gemm(const e3k_gemm_problem* problems, int n_problems, void* stream) {
  if (n_problems < 0 || (n_problems && !problems)) return E3K_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  for (int mode = 0; mode < 2; ++mode) {  // 0: plain, 1: outer
    int i = 0;
    while (i < n_problems) {
      e3k::GemmBatch gb{};
      int tiles = 0;
      while (i < n_problems && gb.n < e3k::GEMM_MAXP) {
        const e3k_gemm_problem& P = problems[i];
        ++i;
        const int rc = validate(P, false);
        if (rc != E3K_OK) return rc;
        if ((P.V > 0) != (mode == 1)) continue;
        const int64_t M = (int64_t)P.M1 * P.M2;
        if (M == 0) continue;
        int f = a_flags(P);
        if (P.b_n == 1 && P.N % 4 == 0 && P.b_k % 4 == 0 && aligned16(P.B)) f |= 1 << 1;
        else if (P.b_k == 1 && P.K % 4 == 0 && P.b_n % 4 == 0 && aligned16(P.B)) f |= 2 << 1;
        gb.p[gb.n] = P;
        gb.flags[gb.n] = f;
        gb.tile_start[gb.n] = tiles;
        tiles += (int)((M + e3k::BM - 1) / e3k::BM) * ((P.N + e3k::BN - 1) / e3k::BN);
        ++gb.n;
      }
      gb.tile_start[gb.n] = tiles;
      if (!tiles) continue;
      if (mode == 0) hipLaunchKernelGGL(e3k::gemm_kernel, dim3(tiles), dim3(256), 0, st, gb);
      else hipLaunchKernelGGL(e3k::gemm_outer_kernel, dim3(tiles), dim3(256), 0, st, gb);
    }
  }
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int e3k_gemm_wgrad(const e3k_gemm_problem* problems, int n_problems, void* stream) {
  if (n_problems < 0 || (n_problems && !problems)) return E3K_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  for (int mode = 0; mode < 2; ++mode) {
    int i = 0;
    while (i < n_problems) {
      e3k::GemmBatch gb{};
      int blocks = 0;
      while (i < n_problems && gb.n < e3k::GEMM_MAXP) {
        const e3k_gemm_problem& P = problems[i];
        ++i;
        const int rc = validate(P, true);
        if (rc != E3K_OK) return rc;
        if ((P.V > 0) != (mode == 1)) continue;
        const int64_t M = (int64_t)P.M1 * P.M2;
        if (M == 0) continue;
        int f = a_flags(P);
        if (P.c_n == 1 && P.N % 4 == 0 && P.c_r1 % 4 == 0 && P.c_r2 % 4 == 0 && aligned16(P.C)) f |= 8;
        const int tiles = ((P.K + e3k::WK - 1) / e3k::WK) * ((P.N + e3k::WN - 1) / e3k::WN);
        int64_t splits = (1024 + tiles - 1) / tiles;
        const int64_t max_splits = (M + 4 * e3k::WR - 1) / (4 * e3k::WR);
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
        gb.p[gb.n] = P;
        gb.flags[gb.n] = f;
        gb.splits[gb.n] = (int)splits;
        gb.tile_start[gb.n] = blocks;
        blocks += tiles * (int)splits;
        ++gb.n;
      }
      gb.tile_start[gb.n] = blocks;
      if (!blocks) continue;
      if (mode == 0) hipLaunchKernelGGL(e3k::gemm_wgrad_kernel<false>, dim3(blocks), dim3(256), 0, st, gb);
      else hipLaunchKernelGGL(e3k::gemm_wgrad_kernel<true>, dim3(blocks), dim3(256), 0, st, gb);
    }
  }
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int e3k_colsum(const float* G, int64_t rows, int32_t cols, int64_t ld, float* out, void* stream) {
  if (rows < 0 || cols <= 0 || !out) return E3K_ERR_INVALID;
  if (rows == 0) return E3K_OK;
  if (!G) return E3K_ERR_INVALID;
  int gy = (int)((rows + 255) / 256);
  if (gy > 256) gy = 256;
  hipLaunchKernelGGL(e3k::colsum_kernel, dim3((cols + 63) / 64, gy), dim3(256), 0, (hipStream_t)stream, G, rows, cols,
                     ld, out);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int e3k_fctp_reduce_bwd(const float* H, const float* X, const float* A2, int32_t M1, int32_t M2, int32_t U,
                                   int32_t V, int64_t x_r1, int64_t x_r2, int64_t a2_r1, float* dX,
                                   int32_t dx_accumulate, float* dA2, void* stream) {
  if (M1 < 0 || M2 <= 0 || U <= 0 || V <= 0 || V > e3k::VMAX) return E3K_ERR_INVALID;
  if (M1 == 0) return E3K_OK;
  if (!H || !X || !A2 || !dX || !dA2) return E3K_ERR_INVALID;
  const int64_t rows = (int64_t)M1 * M2;
  hipLaunchKernelGGL(e3k::fctp_reduce_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, H, X,
                     A2, M1, M2, U, V, x_r1, x_r2, a2_r1, dX, dx_accumulate, dA2);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}
