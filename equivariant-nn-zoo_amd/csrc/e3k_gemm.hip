// Grouped strided GEMM on the gfx950 f32 matrix pipe (v_mfma_f32_32x32x2_f32: exact fp32 FMA
// chain, 64 FLOP/clk/SIMD — cdna_hip_programming.md §3 "FP32-input MFMA").
//
// Replaces on the node / edge side of a convolution layer (paths relative to /root/reference):
//   o3.Linear                          e3_layers/nn/message_passing.py:58,102; nn/pointwise.py:18,87,142
//   FullyConnectedNet layers           e3_layers/nn/message_passing.py:74,93      (radial MLP)
//   FullyConnectedTensorProduct (sc)   e3_layers/nn/message_passing.py:83,100     (outer mode)
//
// One problem:  C[(r1,r2), n] = alpha * sum_k Aeff[(r1,r2), k] B[k, n]  (+ C) (+ bias[n]),  rows are
// (r1 < M1 nodes/edges, r2 < M2 = 2l+1 components) with independent strides, so the same kernels
// serve the e3nn layout ([mul][2l+1]) and the channel-fastest layout ([2l+1][mul]).
//
// Kernels (256 threads = 4 waves, 32x32x2 f32 MFMA accumulators):
//   gemm_kernel<WM>     forward / dgrad.  Tile (32*WM) x 64 x 32: WM=4 -> 128x64 (wave = 32 rows x 64 cols,
//                       two accumulators), WM=2 -> 64x64 (wave = 32 x 32) for launches that would not fill
//                       the chip with 128-row tiles.  A tile in LDS at an odd row stride (33): the MFMA A
//                       fragment (lane -> row l&31, k l>>5) is bank-conflict free; B tile [k][n].  Global
//                       loads are 16 B/lane into registers one K-step ahead of the MFMAs (issue early,
//                       write LDS late).  dgrad (dA = dC . B^T) = same kernel, b_k / b_n swapped.
//   gemm_smallk_kernel  K <= 64 (radial MLP, 64-channel node Linears): the A tile (128 x K) stays in LDS
//                       while the block walks up to 8 column tiles, so A is read once per 512 columns and
//                       the stores of one column tile drain under the MFMAs of the next: this GEMM is
//                       bound by its output stream (E x W floats), not by the matrix pipe.
//   gemm_outer_kernel   self-connection: Aeff[(r1,r2), u*V+v] = X[(r1,r2),u] * attrs[r1,v]; X and attrs
//                       tiles sit in LDS for the whole K loop, the product is formed when the fragment is
//                       read, so x (x) node_attrs (N x 1280 x (2l+1)) is never materialised.
//   gemm_wgrad_kernel   B[k,n] += alpha * sum_rows Aeff[row,k] G[row,n]: tile 64 k x (64 or 128) n, rows
//                       split over blocks, 64-row chunks staged in natural row-major order (both MFMA
//                       operands conflict-free), fp32 atomics on the small output.
//
// Descriptors: up to 20 problems per launch, passed by value (3.85 KB of kernel arguments); keyed and plain problems mix; every
// workgroup finds its problem and pulls it into scalar registers with s_load (the batch is wave-uniform).
#include <algorithm>
#include <cstdlib>

#include "e3k_common.h"

namespace e3k {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// 20 x 168 B of descriptors + tables = 3.85 KB of the 4 KB kernel-argument segment (round 6: 16 -> 20 -- the three weight gradients of a
// layer are 7 + 7 + 6 problems: with 16 per launch linear_1's last four went out as a second, 23 us launch of their own)
constexpr int GEMM_MAXP = 20;
static_assert(true, "");
struct GemmBatch {
  int n;
  int reps[GEMM_MAXP];              // > 1: the problem stands for `reps` key groups (its tile range is reps equal sub-ranges);
  long long key_stride[GEMM_MAXP];  //      key t uses B + t*key_stride and the device pair group_dev + 2*t
  int tile_start[GEMM_MAXP + 1];
  int flags[GEMM_MAXP];  // bit0: A float4-loadable, bits1-2: B mode (0 scalar, 1 n-contiguous vec, 2 k-contiguous vec), bit3: G float4-loadable (wgrad)
  int aux[GEMM_MAXP];    // wgrad: row splits; smallk: column tiles per block
  e3k_gemm_problem p[GEMM_MAXP];
};
static_assert(sizeof(GemmBatch) % 4 == 0 && sizeof(e3k_gemm_problem) % 4 == 0, "word-copyable descriptors");
static_assert(sizeof(GemmBatch) + 16 <= 4096, "the batch travels by value in the kernel-argument segment");

constexpr int BN = 64, BK = 32;
constexpr int LDA = BK + 1;
constexpr int LDB = BN;
constexpr int XU = 64;          // outer mode: channels of X kept per super-step
constexpr int LDX = XU + 1;
constexpr int VMAX = 32;        // outer mode: max attrs width
constexpr int LDV = VMAX + 1;
constexpr int SK_KMAX = 64;     // small-K kernel: max K
constexpr int SK_LDA = SK_KMAX + 1;
constexpr int SK_CT = 8;        // small-K kernel: column tiles per block

#ifdef E3K_STAMPS
__device__ unsigned long long e3k_dbg_buf[1 << 20];
#define STAMP_DECL unsigned long long st_last = __builtin_readcyclecounter(), st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define STAMP(i)                                                  \
  do {                                                            \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    st_acc[i] += now_ - st_last;                                  \
    st_last = now_;                                               \
  } while (0)
#define STAMP_FLUSH()                                                                             \
  do {                                                                                            \
    if ((threadIdx.x & 63) == 0) {                                                                \
      const size_t slot = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8;                      \
      if (slot + 8 <= (1u << 20))                                                                 \
        for (int i_ = 0; i_ < 8; ++i_) e3k_dbg_buf[slot + i_] = st_acc[i_];                       \
    }                                                                                             \
  } while (0)
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_FLUSH()
#endif

struct BlockProblem {
  e3k_gemm_problem P;
  int flags, aux, local;
  int pi, key;      // index of the problem in the batch; key group of a keyed problem (0 otherwise): what a K-chain's followers reuse
};

// COMPACT keyed grids (round 6; flags bit 6, gemm_kernel and gemm_smallk_kernel): a keyed problem used to get `reps` x the tiles of its
// row bound M1 -- every key a full-size grid, the workgroups past a key's last row exit after reading their descriptor: 8 377
// workgroups for ~1 300 tiles of work in a layer's linear_1 + self-connection launch, nine rounds of empty workgroups through the
// CUs.  The key groups PARTITION the rows, so sum_k ceil(count_k M2 / bm) <= ceil(M1 M2 / bm) + reps row tiles suffice: a workgroup
// finds its key by walking the (<= 32) device-side counts.  `bm`: rows per tile; `cols(P, aux)`: workgroups per row tile.
template <class Cols>
__device__ __forceinline__ BlockProblem fetch_problem(const GemmBatch& gb, int bm, Cols cols);
struct NoCols {
  __device__ int operator()(const e3k_gemm_problem&, int) const { return 1; }
};
__device__ __forceinline__ BlockProblem fetch_problem(const GemmBatch& gb) { return fetch_problem(gb, 0, NoCols{}); }

template <class Cols>
__device__ __forceinline__ BlockProblem fetch_problem(const GemmBatch& gb, int bm, Cols cols) {
  // The batch lives in the kernel-argument segment: everything here is wave-uniform, so the compiler reads it with
  // scalar loads (s_load_dwordxN at a uniform dynamic offset) straight into SGPRs — no LDS copy, no barrier.
  const int n = gb.n;
  int pi = 0;
#pragma unroll
  for (int i = 1; i < GEMM_MAXP; ++i)
    if (i < n && (int)blockIdx.x >= gb.tile_start[i]) pi = i;
  pi = uniform(pi);
  BlockProblem out;
  out.local = blockIdx.x - gb.tile_start[pi];
  out.flags = gb.flags[pi];
  out.aux = gb.aux[pi];
  out.P = gb.p[pi];
  out.pi = pi;
  out.key = 0;
  const int reps = gb.reps[pi];
  if (reps > 1 && bm != 0 && (out.flags & 64)) {      // compact keyed grid
    if (bm < 0) bm = out.aux;                           // (the weight gradient's row tile = the rows of one split: in aux)
    const int c = cols(out.P, out.aux);
    const int rt = out.local / c, col = out.local - rt * c;
    int key = -1, rt_local = 0, base = 0;
    for (int k = 0; k < reps; ++k) {
      int cnt = uniform(out.P.group_dev[2 * k + 1]);
      cnt = cnt < out.P.M1 ? cnt : out.P.M1;
      const int t = (cnt * out.P.M2 + bm - 1) / bm;
      if (key < 0 && rt < base + t) {
        key = k;
        rt_local = rt - base;
      }
      base += t;
    }
    if (key < 0) {      // surplus workgroup (the grid is sized by the bound)
      out.local = -1;
      return out;
    }
    out.local = rt_local * c + col;
    out.P.B += (int64_t)key * gb.key_stride[pi];
    out.P.group_dev += 2 * key;
    out.key = key;
  } else if (reps > 1) {  // keyed problem: which key group this workgroup belongs to
    const int per_key = (gb.tile_start[pi + 1] - gb.tile_start[pi]) / reps;
    const int key = out.local / per_key;
    out.local -= key * per_key;
    out.P.B += (int64_t)key * gb.key_stride[pi];
    out.P.group_dev += 2 * key;
    out.key = key;
  }
  if (out.P.row_index && out.P.group_dev) {  // device-side {start, count} of this key group
    const int start = uniform(out.P.group_dev[0]), count = uniform(out.P.group_dev[1]);
    out.P.row_index += start;
    out.P.M1 = count < out.P.M1 ? count : out.P.M1;
  }
  return out;
}

// K-chain (round 6): problem pi + j, j = 1 .. P.chain of the head, continues the head's K loop into the SAME accumulators -- another A
// block times another B block summed into the same C tile (the input gradient of an irrep that feeds two outputs of a Linear:
// `64x0e` and the gates' `256x0e` both read `0e`; until now the second one was an accumulating second launch, 21 us of a
// layer's backward for 0.4 GFLOP).  A follower has its head's rows, columns, C and key groups; it brings A, B, K, their strides
// and alpha.  Zero tiles of its own: no workgroup ever finds it through tile_start.
__device__ __forceinline__ void fetch_follower(const GemmBatch& gb, int idx, const BlockProblem& head, BlockProblem& out) {
  out.P = gb.p[idx];
  out.flags = gb.flags[idx];
  if (gb.reps[idx] > 1) {
    out.P.B += (int64_t)head.key * gb.key_stride[idx];
    out.P.group_dev += 2 * head.key;
  }
  if (out.P.row_index && out.P.group_dev) {
    const int start = uniform(out.P.group_dev[0]), count = uniform(out.P.group_dev[1]);
    out.P.row_index += start;
    out.P.M1 = count < out.P.M1 ? count : out.P.M1;
  }
}

__device__ __forceinline__ float epilogue_act(const e3k_gemm_problem& P, float v) {
  if (P.act == 1) v = P.act_cst * (fmaxf(v, 0.f) + log1pf(expf(-fabsf(v))) - 0.6931471805599453f);  // shifted softplus
  return v;
}

struct BRegs {
  float4 v[2];
};

// B tile (BK x BN) global -> registers (vector modes only; issued early, written to LDS late).  ONE load statement per
// register whatever the mode: with a load in each arm of `if (bmode == 1) ... else ...` the two arms write the same
// registers, the structurised control flow keeps an edge from one arm into the other, and the compiler protects the second
// arm's zero-initialisation with s_waitcnt vmcnt(0) -- which, on the forward path, waited for the A loads issued just
// before, i.e. the K-step prefetch never overlapped the MFMAs (found in round 4 with timing-only ablations: the phases of
// these kernels added up).
__device__ __forceinline__ void load_b_regs(const e3k_gemm_problem& P, int bmode, int k0, int n0, BRegs& r) {
  const int t = threadIdx.x;
  const bool m1 = bmode == 1;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    // mode 1: row k = (t >> 4) + 16 pass, columns nq..nq+3;  mode 2 (B^T view, contiguous along k): column n = t & 63, k = kq + 4 pass
    const int k = m1 ? (t >> 4) + 16 * pass : (t >> 6) * 8 + 4 * pass;
    const int n = m1 ? (t & 15) * 4 : (t & 63);
    const int64_t off = m1 ? (int64_t)(k0 + k) * P.b_k + (n0 + n) : (int64_t)(n0 + n) * P.b_n + (k0 + k);
    r.v[pass] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k0 + k < P.K && n0 + n < P.N) r.v[pass] = *reinterpret_cast<const float4*>(P.B + off);
  }
}

__device__ __forceinline__ void store_b_regs(int bmode, const BRegs& r, float* Bs) {
  const int t = threadIdx.x;
  if (bmode == 1) {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int k = (t >> 4) + 16 * pass, nq = (t & 15) * 4;
      *reinterpret_cast<float4*>(Bs + k * LDB + nq) = r.v[pass];
    }
  } else {
    const int n = t & 63, kq = (t >> 6) * 8;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = kq + 4 * h;
      Bs[(k + 0) * LDB + n] = r.v[h].x;
      Bs[(k + 1) * LDB + n] = r.v[h].y;
      Bs[(k + 2) * LDB + n] = r.v[h].z;
      Bs[(k + 3) * LDB + n] = r.v[h].w;
    }
  }
}

// generic (scalar) B staging straight to LDS
__device__ __forceinline__ void stage_b_scalar(const e3k_gemm_problem& P, int k0, int n0, float* Bs) {
  const int t = threadIdx.x;
  const int n = t & 63, kb = t >> 6;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = kb + 4 * j;
    float v = 0.f;
    if (n0 + n < P.N && k0 + k < P.K) v = P.B[(int64_t)(k0 + k) * P.b_k + (int64_t)(n0 + n) * P.b_n];
    Bs[k * LDB + n] = v;
  }
}

// row tables of a tile: element offsets of its A and C rows (-1 = out of range); one integer
// division per row per block instead of one per staged / stored element
template <int ROWS>
__device__ __forceinline__ void fill_row_tables(const e3k_gemm_problem& P, int row0, int M, long long* rowA,
                                                long long* rowC) {
  const int t = threadIdx.x;
  if (t < ROWS) {
    const int R = row0 + t;
    long long oa = -1, oc = -1;
    if (R < M) {
      int r1 = R / P.M2;
      const int r2 = R - r1 * P.M2;
      if (P.row_index) r1 = P.row_index[r1];
      oa = (long long)r1 * P.a_r1 + (long long)r2 * P.a_r2;
      oc = (long long)r1 * P.c_r1 + (long long)r2 * P.c_r2;
    }
    rowA[t] = oa;
    rowC[t] = oc;
  }
}

// stores NT 32x32 accumulators of one wave: rows wrow0.., columns ncol0 + 32*j.  Accumulating problems (C += ...) read ALL
// their old values first and store afterwards: written as load-add-store per element the compiler must assume that a store
// aliases the next element's load and waits for every load on its own -- 32 dependent memory round trips per wave (the
// input gradient of linear_1 on top of the self-connection's: 137 us for 0.7 GFLOP, found in round 4's launch census).
template <int NT>
__device__ __forceinline__ void store_acc(const e3k_gemm_problem& P, const f32x16 (&acc)[NT], const long long* rowC,
                                          int wrow0, int ncol0) {
  const int lane = threadIdx.x & 63;
  int n[NT];
  bool ok[NT];
  float bias[NT];
  long long cn[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    n[j] = ncol0 + 32 * j + (lane & 31);
    ok[j] = n[j] < P.N;
    bias[j] = (P.bias && ok[j]) ? P.bias[n[j]] : 0.f;
    cn[j] = (long long)n[j] * P.c_n;
  }
  long long off[16];
  bool inside = true;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    off[i] = rowC[wrow0 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)];
    inside = inside && off[i] >= 0;
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) inside = inside && ok[j];
  if (__all(inside)) {
    // interior tile (almost all of them): straight-line code -- every old value requested, then every result stored; with a
    // predicate around each element the compiler waits (vmcnt(0), which also covers the stores issued so far) per element
    float v[NT][16];
    if (P.accumulate) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) v[j][i] = P.C[off[i] + cn[j]];
#pragma unroll
      for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) v[j][i] += fmaf(P.alpha, acc[j][i], bias[j]);
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) v[j][i] = fmaf(P.alpha, acc[j][i], bias[j]);
    }
    if (P.act == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) v[j][i] = epilogue_act(P, v[j][i]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) P.C[off[i] + cn[j]] = v[j][i];
    return;
  }
  // edge tile: per-element predicates
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if (off[i] < 0) continue;
    float* c = P.C + off[i];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if (ok[j]) {
        float v = fmaf(P.alpha, acc[j][i], bias[j]);
        if (P.accumulate) v += c[cn[j]];
        c[cn[j]] = epilogue_act(P, v);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// plain forward / dgrad, tile (32*WM) x 64 x 32
// ---------------------------------------------------------------------------------------
// CHAIN: the launch holds K-chains (e3k.h: e3k_gemm_problem.chain).  A second instantiation: the loop over a chain's links around the
// body costs the compiler 56 more registers (100 -> 156: three waves per SIMD instead of five); launches without chains -- all but a
// layer's input gradients -- keep the lean form.
template <int WM, bool CHAIN>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmBatch gb) {
  constexpr int BM_ = 32 * WM;          // rows per tile
  constexpr int WN = 4 / WM;            // waves along n
  constexpr int NT = (BN / WN) / 32;    // accumulators per wave
  constexpr int PASSES = BM_ / 32;      // float4 passes of the A staging
  constexpr int TPR = 256 / BM_;        // threads per row in the scalar path
  constexpr int KSEG = BK / TPR;
  __shared__ float As[BM_ * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[BK * LDB];
  __shared__ long long rowA[BM_];
  __shared__ long long rowC[BM_];
  const BlockProblem head_ = fetch_problem(gb, BM_, [](const e3k_gemm_problem& Q, int) { return (Q.N + BN - 1) / BN; });
  if (head_.local < 0) return;      // (block-uniform: surplus workgroup of a compact keyed grid)
  BlockProblem bp_ = head_;
  const e3k_gemm_problem& P = bp_.P;
  const int local = bp_.local;
  const int M = P.M1 * P.M2;
  const int tiles_n = (P.N + BN - 1) / BN;
  const int row0 = (local / tiles_n) * BM_, n0 = (local % tiles_n) * BN;
  if (row0 >= M) return;  // block-uniform: surplus workgroup of a device-sized group
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wm = w % WM, wn = w / WM;
  const int chain = CHAIN ? head_.P.chain : 0;

  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;

  for (int link = 0;; ++link) {      // (one trip unless the problem heads a K-chain)
  fill_row_tables<BM_>(P, row0, M, rowA, rowC);
  __syncthreads();

  const int flags = bp_.flags;
  const bool avec = flags & 1;
  const int bmode = (flags >> 1) & 3;
  const float* ap = As + (wm * 32 + (lane & 31)) * LDA + (lane >> 5);
  const float* bp = Bs + (lane >> 5) * LDB + wn * (BN / WN) + (lane & 31);
  auto mfma_tile = [&]() {
#ifdef E3K_DEBUG_KNOBS
    if (flags & 16) return;    // timing only: no MFMAs
#endif
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const float a = ap[kk];
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[kk * LDB + 32 * j], acc[j], 0, 0, 0);
    }
  };

  if (avec && bmode != 0) {
    const float* arow[PASSES];
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
      const long long off = rowA[(t >> 3) + 32 * pass];
      arow[pass] = off >= 0 ? P.A + off : nullptr;
    }
    const int kq = (t & 7) * 4;
    float4 ra[PASSES];
    BRegs rb;
    auto gload = [&](int k0) {
#pragma unroll
      for (int pass = 0; pass < PASSES; ++pass) {
        ra[pass] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (arow[pass] && k0 + kq < P.K) ra[pass] = *reinterpret_cast<const float4*>(arow[pass] + k0 + kq);
      }
      load_b_regs(P, bmode, k0, n0, rb);
    };
    gload(0);
    for (int k0 = 0; k0 < P.K; k0 += BK) {
#pragma unroll
      for (int pass = 0; pass < PASSES; ++pass) {
        float* d = As + ((t >> 3) + 32 * pass) * LDA + kq;
        d[0] = ra[pass].x; d[1] = ra[pass].y; d[2] = ra[pass].z; d[3] = ra[pass].w;
      }
      store_b_regs(bmode, rb, Bs);
      __syncthreads();
      if (k0 + BK < P.K) gload(k0 + BK);
      mfma_tile();
      __syncthreads();
    }
  } else {
    const long long off = rowA[t / TPR];
    const float* arow = off >= 0 ? P.A + off : nullptr;
    const int kb = (t % TPR) * KSEG;
    for (int k0 = 0; k0 < P.K; k0 += BK) {
      float* d = As + (t / TPR) * LDA + kb;
#pragma unroll
      for (int j = 0; j < KSEG; ++j) {
        float v = 0.f;
        if (arow && k0 + kb + j < P.K) v = arow[(int64_t)(k0 + kb + j) * P.a_k];
        d[j] = v;
      }
      if (bmode != 0) {
        BRegs rb;
        load_b_regs(P, bmode, k0, n0, rb);
        store_b_regs(bmode, rb, Bs);
      } else {
        stage_b_scalar(P, k0, n0, Bs);
      }
      __syncthreads();
      mfma_tile();
      __syncthreads();
    }
  }
  if (!CHAIN || link == chain) break;
  {      // the next link: acc holds sum / alpha of what is behind; alpha is applied once, in the epilogue, with the LAST link's value
    const float behind = P.alpha;
    fetch_follower(gb, head_.pi + link + 1, head_, bp_);
    const float ratio = behind / P.alpha;
    if (ratio != 1.0f) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] *= ratio;
    }
  }
  }
#ifdef E3K_DEBUG_KNOBS
  if (bp_.flags & 32) return;      // timing only: no stores
#endif
  store_acc<NT>(P, acc, rowC, wm * 32, n0 + wn * (BN / WN));
}

#ifdef E3K_DEBUG_KNOBS
#include "../../tools/experiments/gemm_persist_kernel.inc"      // (debug build only: the persistent work-list experiment)
#endif  // E3K_DEBUG_KNOBS

// ---------------------------------------------------------------------------------------
// few rows, long K (the radial MLP's dgrad on the knot table: 4097 x 64 outputs, K ~ 2000): a 64-row tile grid would
// occupy a quarter of the chip with one serial K loop each.  Here a workgroup owns 32 x 32 outputs and its four waves
// split K; operands go global -> registers (both k-contiguous, no LDS), the four partial tiles are summed through
// LDS in wave order (deterministic).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemm_splitk_kernel(const GemmBatch gb) {
  constexpr int SKC = 64;                       // k per super-chunk: 8 per lane half and MFMA group, 4 groups in flight
  __shared__ float red[4][32 * 33];
  __shared__ long long rowA[32];
  __shared__ long long rowC[32];
  const BlockProblem bp_ = fetch_problem(gb);
  const e3k_gemm_problem& P = bp_.P;
  const int M = P.M1 * P.M2, K = P.K;
  const int tiles_n = (P.N + 31) / 32;
  const int row0 = (bp_.local / tiles_n) * 32, n0 = (bp_.local % tiles_n) * 32;
  if (row0 >= M) return;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  fill_row_tables<32>(P, row0, M, rowA, rowC);
  __syncthreads();
  const int r = lane & 31, hh = lane >> 5;
  // rows / columns beyond the problem read a valid row instead (their outputs are never stored): no guards in the loop
  const long long offa = rowA[r] >= 0 ? rowA[r] : rowA[0];
  const float* arow = P.A + offa;
  const float* brow = P.B + (int64_t)(n0 + r < P.N ? n0 + r : n0) * P.b_n;
  const int chunks = K / SKC, per = (chunks + 3) / 4;          // K % 64 == 0 (dispatch condition)
  const int c_beg = uniform(w * per), c_end = uniform((c_beg + per < chunks) ? c_beg + per : chunks);   // (scalar loop control)
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float4 a0[8], b0[8], a1[8], b1[8];
#define SPLITK_LOAD(c, xa, xb)                                                        \
  _Pragma("unroll") for (int g = 0; g < 4; ++g) _Pragma("unroll") for (int q = 0; q < 2; ++q) { \
    const int k = (c) * SKC + 16 * g + 8 * hh + 4 * q;                                \
    xa[2 * g + q] = *reinterpret_cast<const float4*>(arow + k);                       \
    xb[2 * g + q] = *reinterpret_cast<const float4*>(brow + k);                       \
  }
#define SPLITK_MFMA(xa, xb)                                                           \
  _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                     \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[j].x, xb[j].x, acc, 0, 0, 0);       \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[j].y, xb[j].y, acc, 0, 0, 0);       \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[j].z, xb[j].z, acc, 0, 0, 0);       \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[j].w, xb[j].w, acc, 0, 0, 0);       \
  }
  // two chunks per trip, every load and MFMA of the trip unconditional (a conditional load makes the compiler wait for
  // ALL outstanding loads before the MFMAs of the other buffer); loads past the wave's range re-read its last chunk
  const int n_chunks = c_end - c_beg;
  if (n_chunks > 0) {
    const int last = c_end - 1;
    SPLITK_LOAD(c_beg, a0, b0)
    for (int i = 0; i + 1 < n_chunks; i += 2) {
      const int c1 = c_beg + i + 1, c2 = (c_beg + i + 2 < last) ? c_beg + i + 2 : last;
      // (the fences keep the scheduler from sinking the loads to their uses: the point is a full chunk in flight)
      SPLITK_LOAD(c1, a1, b1)
      __builtin_amdgcn_sched_barrier(0);
      SPLITK_MFMA(a0, b0)
      __builtin_amdgcn_sched_barrier(0);
      SPLITK_LOAD(c2, a0, b0)
      __builtin_amdgcn_sched_barrier(0);
      SPLITK_MFMA(a1, b1)
      __builtin_amdgcn_sched_barrier(0);
    }
    if (n_chunks & 1) { SPLITK_MFMA(a0, b0) }
  }
#undef SPLITK_LOAD
#undef SPLITK_MFMA
#pragma unroll
  for (int i = 0; i < 16; ++i) red[w][((i & 3) + 8 * (i >> 2) + 4 * hh) * 33 + r] = acc[i];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int o = t + 256 * j, row = o >> 5, col = o & 31;
    const long long offc = rowC[row];
    const int n = n0 + col;
    if (offc < 0 || n >= P.N) continue;
    float v = ((red[0][row * 33 + col] + red[1][row * 33 + col]) + red[2][row * 33 + col]) + red[3][row * 33 + col];
    v = fmaf(P.alpha, v, P.bias ? P.bias[n] : 0.f);
    float* c = P.C + offc + (long long)n * P.c_n;
    if (P.accumulate) v += *c;
    *c = epilogue_act(P, v);
  }
}

// (An fp32 GEMM on the bf16 matrix pipe -- every operand split into three bf16 planes, six 32x32x16 bf16 MFMAs per k-block
// instead of eight 32x32x2 f32 ones -- was built and measured in round 1: 2.7x cheaper on the matrix pipe and as accurate,
// but +12 % at best (K = N = 1024) and -12 % on the radial shapes: the split costs VALU per staged element and these GEMMs
// are not matrix-pipe-bound.  Removed in round 2)

// ---------------------------------------------------------------------------------------
// small-K forward (K <= 64, A k-contiguous, B n-contiguous): A tile resident, walk column tiles
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemm_smallk_kernel(const GemmBatch gb) {
  constexpr int BM_ = 128;
  __shared__ float As[BM_ * SK_LDA];
  __shared__ __attribute__((aligned(16))) float Bs[SK_KMAX * LDB];
  __shared__ long long rowA[BM_];
  __shared__ long long rowC[BM_];
  STAMP_DECL
  const BlockProblem bp_ = fetch_problem(gb, BM_, [](const e3k_gemm_problem& Q, int aux) {
    const int tn = (Q.N + BN - 1) / BN;
    return (tn + aux - 1) / aux;
  });
  if (bp_.local < 0) return;      // (block-uniform: surplus workgroup of a compact keyed grid)
  STAMP(0);
  const e3k_gemm_problem& P = bp_.P;
  const int local = bp_.local, ct = bp_.aux;
  const int M = P.M1 * P.M2, K = P.K;
  const int tiles_n = (P.N + BN - 1) / BN;
  const int col_groups = (tiles_n + ct - 1) / ct;
  const int row0 = (local / col_groups) * BM_;
  if (row0 >= M) return;  // block-uniform: surplus workgroup of a device-sized group
  const int tile_beg = (local % col_groups) * ct;
  const int tile_end = (tile_beg + ct < tiles_n) ? tile_beg + ct : tiles_n;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int kpad = (K + 3) & ~3;  // K % 4 == 0 is a precondition of this kernel (float4 rows)

  fill_row_tables<BM_>(P, row0, M, rowA, rowC);
  __syncthreads();
  // ---- A tile: 2 threads per row, each up to 8 float4
  {
    const long long off = rowA[t >> 1];
    const float* arow = off >= 0 ? P.A + off : nullptr;
    float* d = As + (t >> 1) * SK_LDA;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = ((t & 1) * 8 + j) * 4;
      if (k < kpad) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (arow && k < K) v = *reinterpret_cast<const float4*>(arow + k);
        d[k] = v.x; d[k + 1] = v.y; d[k + 2] = v.z; d[k + 3] = v.w;
      }
    }
  }
  // ---- B tile registers: K x 64 floats = up to 4 float4 per thread
  float4 rb[4];
  auto gload_b = [&](int n0) {
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int k = (t >> 4) + 16 * pass, nq = (t & 15) * 4;
      rb[pass] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k < K && n0 + nq < P.N) rb[pass] = *reinterpret_cast<const float4*>(P.B + (int64_t)k * P.b_k + (n0 + nq));
    }
  };
  const float* ap = As + (w * 32 + (lane & 31)) * SK_LDA + (lane >> 5);
  const float* bp = Bs + (lane >> 5) * LDB + (lane & 31);
  gload_b(tile_beg * BN);
  STAMP(1);
  for (int tile = tile_beg; tile < tile_end; ++tile) {
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int k = (t >> 4) + 16 * pass, nq = (t & 15) * 4;
      if (k < kpad) *reinterpret_cast<float4*>(Bs + k * LDB + nq) = rb[pass];
    }
    STAMP(2);
    __syncthreads();
    STAMP(3);
    if (tile + 1 < tile_end) gload_b((tile + 1) * BN);
    // The MFMA is issued with the operands EXCHANGED (weights as the row operand, activations as the column operand):
    // the accumulator then holds C^T — lane = output row (edge), four consecutive registers = four consecutive
    // output columns — so every lane stores 16 contiguous bytes straight from its registers.  (The natural
    // orientation gives a lane one column and 16 scattered rows: 4-byte stores, or a trip through LDS, which
    // measured as long as the MFMAs of the tile.)
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    int kk = 0;
    for (; kk + 16 <= kpad; kk += 16) {  // 16 MFMAs per trip, all fragment reads issued up front
      float a[8], b0[8], b1[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        a[s] = ap[kk + 2 * s];
        b0[s] = bp[(kk + 2 * s) * LDB];
        b1[s] = bp[(kk + 2 * s) * LDB + 32];
      }
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[s], a[s], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[s], a[s], acc[1], 0, 0, 0);
      }
    }
    for (; kk < kpad; kk += 2) {
      const float a = ap[kk];
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(bp[kk * LDB], a, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(bp[kk * LDB + 32], a, acc[1], 0, 0, 0);
    }
    STAMP(4);
    {
      const long long off = rowC[w * 32 + (lane & 31)];
      if (off >= 0) {
        float* crow = P.C + off;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int n = tile * BN + 32 * j + 8 * q + 4 * (lane >> 5);
            if (n < P.N) {
              float4 o = make_float4(P.alpha * acc[j][4 * q], P.alpha * acc[j][4 * q + 1], P.alpha * acc[j][4 * q + 2],
                                     P.alpha * acc[j][4 * q + 3]);
              if (P.bias) {
                const float4 b4 = *reinterpret_cast<const float4*>(P.bias + n);
                o.x += b4.x; o.y += b4.y; o.z += b4.z; o.w += b4.w;
              }
              float4* dst = reinterpret_cast<float4*>(crow + n);
              if (P.accumulate) {
                const float4 old = *dst;
                o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
              }
              o.x = epilogue_act(P, o.x); o.y = epilogue_act(P, o.y); o.z = epilogue_act(P, o.z); o.w = epilogue_act(P, o.w);
              *dst = o;
            }
          }
        }
      }
    }
    STAMP(5);
    __syncthreads();
    STAMP(6);
  }
  STAMP_FLUSH();
}

// ---------------------------------------------------------------------------------------
// outer-mode forward:  Aeff[(r1,r2), u*V+v] = X[(r1,r2),u] * attrs[r1,v]
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemm_outer_kernel(const GemmBatch gb) {
  constexpr int BM_ = 128;
  __shared__ float Xs[BM_ * LDX];
  __shared__ float Vs[BM_ * LDV];
  __shared__ __attribute__((aligned(16))) float Bs[BK * LDB];
  __shared__ long long rowA[BM_];
  __shared__ long long rowC[BM_];
  const BlockProblem bp_ = fetch_problem(gb);
  const e3k_gemm_problem& P = bp_.P;
  const int flags = bp_.flags, local = bp_.local;
  const int bmode = (flags >> 1) & 3;
  const int M = P.M1 * P.M2, V = P.V, U = P.K / P.V;
  const int tiles_n = (P.N + BN - 1) / BN;
  const int row0 = (local / tiles_n) * BM_, n0 = (local % tiles_n) * BN;
  const int t = threadIdx.x, lane = t & 63, wr = t >> 6;

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;

  fill_row_tables<BM_>(P, row0, M, rowA, rowC);
  // attrs tile: row r -> attrs[r1(r), 0..V)
  {
    const int r = t >> 1, R = row0 + r;
    const int vb = (t & 1) * 16;
    if (R < M) {
      int r1 = R / P.M2;
      if (P.row_index) r1 = P.row_index[r1];
      const float* src = P.A2 + (int64_t)r1 * P.a2_r1;
#pragma unroll
      for (int j = 0; j < 16; ++j) Vs[r * LDV + vb + j] = (vb + j < V) ? src[vb + j] : 0.f;
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) Vs[r * LDV + vb + j] = 0.f;
    }
  }
  __syncthreads();
  const long long xoff = rowA[t >> 1];
  const float* xrow = xoff >= 0 ? P.A + xoff : nullptr;
  const float* xp = Xs + (wr * 32 + (lane & 31)) * LDX;
  const float* vp = Vs + (wr * 32 + (lane & 31)) * LDV;
  const float* bp = Bs + (lane >> 5) * LDB + (lane & 31);

  for (int u0 = 0; u0 < U; u0 += XU) {
    {
      // (the trailing barrier of the previous super-step guarantees nobody still reads Xs)
      const int ub = (t & 1) * 32;
      float* d = Xs + (t >> 1) * LDX + ub;
      if (xrow && P.a_k == 1 && u0 + ub + 32 <= U && ((reinterpret_cast<uintptr_t>(xrow + u0 + ub) & 15) == 0)) {
#pragma unroll
        for (int j = 0; j < 32; j += 4) {
          const float4 v = *reinterpret_cast<const float4*>(xrow + u0 + ub + j);
          d[j] = v.x; d[j + 1] = v.y; d[j + 2] = v.z; d[j + 3] = v.w;
        }
      } else {
#pragma unroll 8
        for (int j = 0; j < 32; ++j) {
          float v = 0.f;
          if (xrow && u0 + ub + j < U) v = xrow[(int64_t)(u0 + ub + j) * P.a_k];
          d[j] = v;
        }
      }
    }
    const int uend = (u0 + XU < U) ? u0 + XU : U;
    const int kbeg = u0 * V, kend = uend * V;
    BRegs rb;
    if (bmode != 0) load_b_regs(P, bmode, kbeg, n0, rb);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
      // rows of B beyond kend belong to the next super-step: their A operand is forced to zero below
      if (bmode != 0) store_b_regs(bmode, rb, Bs);
      else stage_b_scalar(P, k0, n0, Bs);
      __syncthreads();
      if (bmode != 0 && k0 + BK < kend) load_b_regs(P, bmode, k0 + BK, n0, rb);
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
      __syncthreads();
    }
  }
  store_acc<2>(P, acc, rowC, wr * 32, n0);
}

// ---------------------------------------------------------------------------------------
// wgrad:  B[k, n] += alpha * sum_rows Aeff[row, k] * G[row, n]      (G passed in P.C)
// ---------------------------------------------------------------------------------------
constexpr int WK = 64, WR = 64;  // output tile 64 k x (64*TN) n, 64-row chunks
constexpr int LDWA = WK + 4;

// a row cursor that advances by WR rows per iteration without integer division
struct RowCursor {
  int R, r1, r2;
  __device__ __forceinline__ void init(int row, int M2) {
    R = row;
    r1 = row / M2;
    r2 = row - r1 * M2;
  }
  __device__ __forceinline__ void advance(int q, int rem, int M2) {
    R += WR;
    r1 += q;
    r2 += rem;
    if (r2 >= M2) {
      r2 -= M2;
      ++r1;
    }
  }
};

template <bool OUTER, int TN>
__device__ __forceinline__ void gemm_wgrad_body(const BlockProblem& bp_, float* As, float* Gs, float* Vs) {
  constexpr int WN = 64 * TN;
  constexpr int LDWG = WN + 4;
  const e3k_gemm_problem& P = bp_.P;
  const int flags = bp_.flags, local = bp_.local;
  const int M = P.M1 * P.M2, M2 = P.M2;
  const int tiles_k = (P.K + WK - 1) / WK, tiles_n = (P.N + WN - 1) / WN;
  const int splits = bp_.aux;
  const int tile = local % (tiles_k * tiles_n), split = local / (tiles_k * tiles_n);
  const int k0 = (tile / tiles_n) * WK, n0 = (tile % tiles_n) * WN;
  const int chunk_rows = ((M + splits - 1) / splits + WR - 1) / WR * WR;
  const int rbeg = split * chunk_rows;
  const int rend = (rbeg + chunk_rows < M) ? rbeg + chunk_rows : M;
  if (rbeg >= M) return;  // block-uniform: this split has no rows
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int wk = wv >> 1, wn = wv & 1;  // wave = 32 k x (32*TN) n
  const int q64 = WR / M2, rem64 = WR - q64 * M2;

  f32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;

  const bool avec = flags & 1;
  const bool gvec = flags & 8;
  const float* gp = Gs + (lane >> 5) * LDWG + wn * (32 * TN) + (lane & 31);

  if constexpr (!OUTER) {
    if (avec && gvec) {
      // fast path: both operands by 16-byte loads, one chunk ahead
      RowCursor cur[4];
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) cur[pass].init(rbeg + (t >> 4) + 16 * pass, M2);
      const int cq = (t & 15) * 4;
      float4 ra[4], rg[4 * TN];
      auto gload = [&]() {
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
          ra[pass] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int j = 0; j < TN; ++j) rg[pass * TN + j] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (cur[pass].R < rend) {
            const int n1 = P.row_index ? P.row_index[cur[pass].r1] : cur[pass].r1;
            if (k0 + cq < P.K)
              ra[pass] = *reinterpret_cast<const float4*>(P.A + (int64_t)n1 * P.a_r1 + (int64_t)cur[pass].r2 * P.a_r2 + k0 + cq);
            const float* grow = P.C + (int64_t)n1 * P.c_r1 + (int64_t)cur[pass].r2 * P.c_r2 + n0 + cq;
#pragma unroll
            for (int j = 0; j < TN; ++j)
              if (n0 + cq + 64 * j < P.N) rg[pass * TN + j] = *reinterpret_cast<const float4*>(grow + 64 * j);
          }
          cur[pass].advance(q64, rem64, M2);
        }
      };
      const float* ap = As + (lane >> 5) * LDWA + wk * 32 + (lane & 31);
      gload();
      for (int r0 = rbeg; r0 < rend; r0 += WR) {
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
          const int r = (t >> 4) + 16 * pass;
          *reinterpret_cast<float4*>(As + r * LDWA + cq) = ra[pass];
#pragma unroll
          for (int j = 0; j < TN; ++j) *reinterpret_cast<float4*>(Gs + r * LDWG + cq + 64 * j) = rg[pass * TN + j];
        }
        __syncthreads();
        if (r0 + WR < rend) gload();
#pragma unroll 8
        for (int rr = 0; rr < WR; rr += 2) {
          const float a = ap[rr * LDWA];
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, gp[rr * LDWG + 32 * j], acc[j], 0, 0, 0);
        }
        __syncthreads();
      }
    } else {
      RowCursor cur;
      cur.init(rbeg + (t >> 2), M2);
      const int cb = (t & 3) * 16;
      const float* ap = As + (lane >> 5) * LDWA + wk * 32 + (lane & 31);
      for (int r0 = rbeg; r0 < rend; r0 += WR) {
        const bool ok = cur.R < rend;
        const int n1 = (ok && P.row_index) ? P.row_index[cur.r1] : cur.r1;
        const float* sa = P.A + (int64_t)n1 * P.a_r1 + (int64_t)cur.r2 * P.a_r2;
        const float* sg = P.C + (int64_t)n1 * P.c_r1 + (int64_t)cur.r2 * P.c_r2;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          float va = 0.f;
          if (ok && k0 + cb + j < P.K) va = sa[(int64_t)(k0 + cb + j) * P.a_k];
          As[(t >> 2) * LDWA + cb + j] = va;
        }
#pragma unroll
        for (int j = 0; j < 16 * TN; ++j) {
          float vg = 0.f;
          const int c = (t & 3) * 16 * TN + j;
          if (ok && n0 + c < P.N) vg = sg[(int64_t)(n0 + c) * P.c_n];
          Gs[(t >> 2) * LDWG + c] = vg;
        }
        cur.advance(q64, rem64, M2);
        __syncthreads();
#pragma unroll 8
        for (int rr = 0; rr < WR; rr += 2) {
          const float a = ap[rr * LDWA];
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, gp[rr * LDWG + 32 * j], acc[j], 0, 0, 0);
        }
        __syncthreads();
      }
    }
  } else {
    // outer mode: this lane's k index -> (u, v), fixed for the whole loop
    const int V = P.V;
    const int ulo = k0 / V;
    const int klast = (k0 + WK - 1 < P.K - 1) ? k0 + WK - 1 : P.K - 1;
    const int uw = klast / V - ulo + 1;  // <= WK/2 + 1 < LDWA
    const int ki = k0 + wk * 32 + (lane & 31);
    const bool kvalid = ki < P.K;
    const int lu = kvalid ? ki / V - ulo : 0;
    const int lv = kvalid ? ki % V : 0;
    RowCursor cur;
    cur.init(rbeg + (t >> 2), M2);
    const float* ap = As + (lane >> 5) * LDWA + lu;
    const float* vp = Vs + (lane >> 5) * LDV + lv;
    for (int r0 = rbeg; r0 < rend; r0 += WR) {
      const bool ok = cur.R < rend;
      const int n1 = (ok && P.row_index) ? P.row_index[cur.r1] : cur.r1;
      const float* sx = P.A + (int64_t)n1 * P.a_r1 + (int64_t)cur.r2 * P.a_r2;
      const float* sv = P.A2 + (int64_t)n1 * P.a2_r1;
      const float* sg = P.C + (int64_t)n1 * P.c_r1 + (int64_t)cur.r2 * P.c_r2;
      for (int j = (t & 3); j < uw; j += 4) As[(t >> 2) * LDWA + j] = ok ? sx[(int64_t)(ulo + j) * P.a_k] : 0.f;
      for (int j = (t & 3); j < V; j += 4) Vs[(t >> 2) * LDV + j] = ok ? sv[j] : 0.f;
      const int cb = (t & 3) * 16 * TN;
      if (gvec) {
#pragma unroll
        for (int j = 0; j < 16 * TN; j += 4) {
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (ok && n0 + cb + j < P.N) v = *reinterpret_cast<const float4*>(sg + n0 + cb + j);
          *reinterpret_cast<float4*>(Gs + (t >> 2) * LDWG + cb + j) = v;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 16 * TN; ++j) {
          float vg = 0.f;
          if (ok && n0 + cb + j < P.N) vg = sg[(int64_t)(n0 + cb + j) * P.c_n];
          Gs[(t >> 2) * LDWG + cb + j] = vg;
        }
      }
      cur.advance(q64, rem64, M2);
      __syncthreads();
#pragma unroll 8
      for (int rr = 0; rr < WR; rr += 2) {
        const float a = kvalid ? ap[rr * LDWA] * vp[rr * LDV] : 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, gp[rr * LDWG + 32 * j], acc[j], 0, 0, 0);
      }
      __syncthreads();
    }
  }
  // ---- atomic accumulate into B
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * (32 * TN) + 32 * j + (lane & 31);
    if (n < P.N) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int k = k0 + wk * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
        if (k < P.K) atomicAdd(const_cast<float*>(P.B) + (int64_t)k * P.b_k + (int64_t)n * P.b_n, P.alpha * acc[j][i]);
      }
    }
  }
}

// (one kernel serving both tile widths by a per-problem flag was built and measured: 164 VGPRs and 51 KB of LDS for every
//  problem -- three workgroups per CU -- made the 64-wide problems 55 % slower; a call issues one launch per width)
template <bool OUTER, int TN>
__global__ __launch_bounds__(256) void gemm_wgrad_kernel(const GemmBatch gb) {
  __shared__ __attribute__((aligned(16))) float As[WR * LDWA];
  __shared__ __attribute__((aligned(16))) float Gs[WR * (64 * TN + 4)];
  __shared__ float Vs[OUTER ? WR * LDV : 1];
  const BlockProblem bp_ = fetch_problem(gb);
  gemm_wgrad_body<OUTER, TN>(bp_, As, Gs, Vs);
}

// ---------------------------------------------------------------------------------------
// wgrad, software-pipelined form (round 4).  What the round's measurements say about the form above on the node-side
// problems (4 608 nodes, K = 384, N = 64, rows 3 or 5 per node): its time is a fixed 8-19 us per 64-row chunk and
// workgroup whatever the number of co-resident workgroups -- the loop is one load latency long per chunk (loads issued
// one chunk ahead, behind a barrier, waited for in front of the next) -- and it pays 16 KB of float atomics per 256 rows
// (the chip adds 1.3 TB/s of atomic bytes, MI355X_MICROARCH.md "Global float atomics").  Here: (i) two LDS stages, the
// loads of chunk c + 2 in flight while chunk c is in the matrix pipe: one barrier per chunk and two chunk times for a
// load to land; (ii) tile 128 k x 64 n (WKW = 4 waves along k, two accumulators each: the G fragment pair is reused by
// every wave, A is still read once) or 64 k x 64 n (WKW = 2) for K <= 64; (iii) the launch is sized to ONE round of
// workgroups -- three per CU, what LDS admits -- with EQUAL row ranges per workgroup (splits proportional to a problem's
// rows), so a workgroup adds its tile once per ~1/768 of the launch's work (measured: 256 / 512 / 768 / 1 024 workgroups
// 70 / 73 / 59 / 72 us on the trailing Linear: a second, partial round costs what it saves); (iv) row pointers advanced by
// constant 64-bit deltas instead of two 64-bit multiplies per load and chunk; gathered rows (IDX) through node indices
// fetched one chunk ahead.
// ---------------------------------------------------------------------------------------
constexpr int W2R = 32;   // rows per chunk
template <int WKW, bool IDX>
__device__ __forceinline__ void gemm_wgrad2_body(const BlockProblem& bp_, float* As_, float* Gs_) {
  constexpr int TK = 32 * WKW;            // k per tile
  constexpr int WNW = 4 / WKW;            // waves along n
  constexpr int NT = 2 / WNW;             // accumulators per wave (tile is 64 n wide)
  constexpr int LDA2 = TK + 4, LDG2 = 64 + 4;
  constexpr int APASS = (W2R * TK / 4) / 256;     // float4 loads of A per thread and chunk (4 or 2)
  constexpr int AROWS = 256 / (TK / 4);           // rows covered by one pass (8 or 16)
  float (*As)[W2R * LDA2] = reinterpret_cast<float (*)[W2R * LDA2]>(As_);
  float (*Gs)[W2R * LDG2] = reinterpret_cast<float (*)[W2R * LDG2]>(Gs_);
  const e3k_gemm_problem& P = bp_.P;
  const int local = bp_.local;
  const int M = P.M1 * P.M2, M2 = P.M2;
  const int tiles_k = (P.K + TK - 1) / TK, tiles_n = (P.N + 63) / 64;
  const int splits = bp_.aux;
  const int tile = local % (tiles_k * tiles_n), split = local / (tiles_k * tiles_n);
  const int k0 = (tile / tiles_n) * TK, n0 = (tile % tiles_n) * 64;
  // (compact keyed grid, flags bit 6: aux IS the rows per split -- the same for every key: fetch_problem)
  const int chunk_rows = (bp_.flags & 64) ? splits : ((M + splits - 1) / splits + W2R - 1) / W2R * W2R;
  const int rbeg = split * chunk_rows;
  const int rend = (rbeg + chunk_rows < M) ? rbeg + chunk_rows : M;
  if (rbeg >= M) return;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int wk = wv % WKW, wn = wv / WKW;
  const int qR = W2R / M2, remR = W2R - qR * M2;

  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;

  // this thread's rows of a chunk: A pass p -> row (t / (TK/4)) + AROWS*p, G pass p -> row (t >> 4) + 16*p.  Plain problems:
  // row pointers advance by constant 64-bit deltas (one chunk down, and the wrap of the component index r2).  Gathered
  // rows (IDX: the keyed self-connection, rows = the nodes of one key through row_index): the node index of a row is
  // fetched one chunk ahead, behind the chunk's float4 loads, so that it has landed with them
  int aR[APASS], ar1[APASS], ar2[APASS], gR[2], gr1[2], gr2[2];
  int ia[APASS], ig[2];
  const float* pa[APASS];
  const float* pg[2];
  const int64_t dA = (int64_t)qR * P.a_r1 + (int64_t)remR * P.a_r2, wA = P.a_r1 - (int64_t)M2 * P.a_r2;
  const int64_t dG = (int64_t)qR * P.c_r1 + (int64_t)remR * P.c_r2, wG = P.c_r1 - (int64_t)M2 * P.c_r2;
  const int acol = (t % (TK / 4)) * 4, gcol = (t & 15) * 4;
  const bool a_in = k0 + acol < P.K, g_in = n0 + gcol < P.N;
#pragma unroll
  for (int p = 0; p < APASS; ++p) {
    aR[p] = rbeg + t / (TK / 4) + AROWS * p;
    ar1[p] = aR[p] / M2;
    ar2[p] = aR[p] - ar1[p] * M2;
    pa[p] = P.A + (int64_t)ar1[p] * P.a_r1 + (int64_t)ar2[p] * P.a_r2 + k0 + acol;
    ia[p] = (IDX && aR[p] < rend) ? P.row_index[ar1[p]] : 0;
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    gR[p] = rbeg + (t >> 4) + 16 * p;
    gr1[p] = gR[p] / M2;
    gr2[p] = gR[p] - gr1[p] * M2;
    pg[p] = P.C + (int64_t)gr1[p] * P.c_r1 + (int64_t)gr2[p] * P.c_r2 + n0 + gcol;
    ig[p] = (IDX && gR[p] < rend) ? P.row_index[gr1[p]] : 0;
  }
  float4 ra[APASS], rg[2];
  auto gload = [&]() {
#pragma unroll
    for (int p = 0; p < APASS; ++p) {
      ra[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (IDX) {
        if (aR[p] < rend && a_in)
          ra[p] = *reinterpret_cast<const float4*>(P.A + (int64_t)ia[p] * P.a_r1 + (int64_t)ar2[p] * P.a_r2 + k0 + acol);
        aR[p] += W2R; ar1[p] += qR; ar2[p] += remR;
        if (ar2[p] >= M2) { ar2[p] -= M2; ++ar1[p]; }
      } else {
        if (aR[p] < rend && a_in) ra[p] = *reinterpret_cast<const float4*>(pa[p]);
        aR[p] += W2R; ar2[p] += remR; pa[p] += dA;
        if (ar2[p] >= M2) { ar2[p] -= M2; pa[p] += wA; }
      }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      rg[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (IDX) {
        if (gR[p] < rend && g_in)
          rg[p] = *reinterpret_cast<const float4*>(P.C + (int64_t)ig[p] * P.c_r1 + (int64_t)gr2[p] * P.c_r2 + n0 + gcol);
        gR[p] += W2R; gr1[p] += qR; gr2[p] += remR;
        if (gr2[p] >= M2) { gr2[p] -= M2; ++gr1[p]; }
      } else {
        if (gR[p] < rend && g_in) rg[p] = *reinterpret_cast<const float4*>(pg[p]);
        gR[p] += W2R; gr2[p] += remR; pg[p] += dG;
        if (gr2[p] >= M2) { gr2[p] -= M2; pg[p] += wG; }
      }
    }
    if constexpr (IDX) {      // node indices of the NEXT chunk's rows
#pragma unroll
      for (int p = 0; p < APASS; ++p) ia[p] = aR[p] < rend ? P.row_index[ar1[p]] : 0;
#pragma unroll
      for (int p = 0; p < 2; ++p) ig[p] = gR[p] < rend ? P.row_index[gr1[p]] : 0;
    }
  };
  auto lstore = [&](int st) {
#pragma unroll
    for (int p = 0; p < APASS; ++p)
      *reinterpret_cast<float4*>(&As[st][(t / (TK / 4) + AROWS * p) * LDA2 + acol]) = ra[p];
#pragma unroll
    for (int p = 0; p < 2; ++p) *reinterpret_cast<float4*>(&Gs[st][((t >> 4) + 16 * p) * LDG2 + gcol]) = rg[p];
  };
  const int n_chunks = (rend - rbeg + W2R - 1) / W2R;
  gload();
  lstore(0);
  if (n_chunks > 1) gload();
  __syncthreads();
  const int aoff = (lane >> 5) * LDA2 + wk * 32 + (lane & 31);
  const int goff = (lane >> 5) * LDG2 + wn * (32 * NT) + (lane & 31);
  for (int c = 0; c < n_chunks; ++c) {
    const int st = c & 1;
    if (c + 1 < n_chunks) lstore(st ^ 1);
#ifdef E3K_DEBUG_KNOBS
    if (c + 2 < n_chunks && !(bp_.flags & 32)) gload();
    if (bp_.flags & 16) { __syncthreads(); continue; }
#else
    if (c + 2 < n_chunks) gload();
#endif
    const float* ap = &As[st][aoff];
    const float* gp = &Gs[st][goff];
#pragma unroll
    for (int rr = 0; rr < W2R; rr += 2) {
      const float a = ap[rr * LDA2];
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, gp[rr * LDG2 + 32 * j], acc[j], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + wn * (32 * NT) + 32 * j + (lane & 31);
    if (n < P.N) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int k = k0 + wk * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
        if (k < P.K) atomicAdd(const_cast<float*>(P.B) + (int64_t)k * P.b_k + (int64_t)n * P.b_n, P.alpha * acc[j][i]);
      }
    }
  }
}

// one kernel for both tile shapes (a call's K > 64 and K <= 64 problems share a launch); registers and LDS are the wide form's
__global__ __launch_bounds__(256, 2) void gemm_wgrad2_kernel(const GemmBatch gb) {
  __shared__ __attribute__((aligned(16))) float As[2 * W2R * (128 + 4)];
  __shared__ __attribute__((aligned(16))) float Gs[2 * W2R * (64 + 4)];
  const BlockProblem bp_ = fetch_problem(gb, -1, [](const e3k_gemm_problem& Q, int) {
    return ((Q.K + (Q.K > 64 ? 127 : 63)) / (Q.K > 64 ? 128 : 64)) * ((Q.N + 63) / 64);
  });
  if (bp_.local < 0) return;      // (block-uniform: surplus workgroup of a compact keyed grid)
  if (bp_.P.row_index) {
    if (bp_.P.K > 64) gemm_wgrad2_body<4, true>(bp_, As, Gs);
    else gemm_wgrad2_body<2, true>(bp_, As, Gs);
  } else {
    if (bp_.P.K > 64) gemm_wgrad2_body<4, false>(bp_, As, Gs);
    else gemm_wgrad2_body<2, false>(bp_, As, Gs);
  }
}

#ifdef E3K_DEBUG_KNOBS
#include "../../tools/experiments/gemm_wgrad3_kernel.inc"       // (debug build only: the LDS-ring weight-gradient experiment)
#endif  // E3K_DEBUG_KNOBS

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

// Forward of a Linear with ONE output column (the energy head): C[row] = alpha <A[row, :], B> (+ bias) (+ C): a wave per row, the
// lanes walk K, one butterfly.  (On the 64 x 64 tile kernel: 74 workgroups, 10.7 us for 1.2 MB.)
__global__ __launch_bounds__(256) void gemm_n1_kernel(const float* __restrict__ A, const float* __restrict__ B, int64_t M1, int M2, int K,
                                                       int64_t a_r1, int64_t a_r2, int64_t a_k, int64_t b_k, int64_t c_r1, int64_t c_r2,
                                                       float alpha, const float* __restrict__ bias, int accumulate, int act, float act_cst,
                                                       float* __restrict__ C) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= M1 * M2) return;
  const int lane = threadIdx.x & 63;
  const int64_t r1 = r / M2, r2 = r - r1 * M2;
  const float* __restrict__ a = A + r1 * a_r1 + r2 * a_r2;
  float s = 0.f;
  for (int k = lane; k < K; k += 64) s = fmaf(a[k * a_k], B[k * b_k], s);
  s = wave_sum(s);
  if (lane == 0) {
    float* c = C + r1 * c_r1 + r2 * c_r2;
    float v = fmaf(alpha, s, bias ? bias[0] : 0.f);
    if (accumulate) v += *c;
    if (act == 1) v = act_cst * (fmaxf(v, 0.f) + log1pf(expf(-fabsf(v))) - 0.6931471805599453f);
    *c = v;
  }
}

// Weight gradient of a Linear with ONE output column (the energy head's `64x0e -> 1x0e`, e3_layers/nn/pointwise.py:18): B[k] += alpha
// sum_rows A[row, k] g[row] is a weighted column sum, not a GEMM -- on the MFMA tile kernels its 4 704 x 64 rows made 74 workgroups
// wait 22 us for 1.2 MB.  Same shape as colsum_kernel: 64 columns x 4 row phases per block, grid.y splits the rows.
__global__ __launch_bounds__(256) void wgrad_n1_kernel(const float* __restrict__ A, const float* __restrict__ g, int64_t M1, int M2,
                                                        int K, int64_t a_r1, int64_t a_r2, int64_t a_k, int64_t c_r1, int64_t c_r2,
                                                        int64_t b_k, float alpha, float* __restrict__ B) {
  __shared__ float part[4][64];
  const int k = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
  const int64_t rows = M1 * M2;
  float s = 0.f;
  if (k < K)
    for (int64_t r = (int64_t)blockIdx.y * 4 + ph; r < rows; r += (int64_t)gridDim.y * 4) {
      const int64_t r1 = r / M2, r2 = r - r1 * M2;
      s = fmaf(A[r1 * a_r1 + r2 * a_r2 + k * a_k], g[r1 * c_r1 + r2 * c_r2], s);
    }
  part[ph][threadIdx.x & 63] = s;
  __syncthreads();
  if (ph == 0 && k < K)
    atomicAdd(B + k * b_k, alpha * (part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]));
}

// Coalesced variant for V in {4, 8, 16, 32} (V/4 lanes share one u, a lane owns four consecutive v for the whole row):
// one wave per (r1, r2) row streams the row's U*V floats as 16-byte loads, 1 KB per wave-instruction.  (The generic
// kernel below lets every lane walk its own 4*V-byte segment: 64 cache lines in flight per instruction, 0.75 TB/s.)
template <int V>
__global__ __launch_bounds__(256) void fctp_reduce_vec_kernel(const float* __restrict__ H, const float* __restrict__ X,
                                                               const float* __restrict__ A2, int M1, int M2, int U,
                                                               int64_t x_r1, int64_t x_r2, int64_t a2_r1,
                                                               float* __restrict__ dX, int dx_acc, float* __restrict__ dA2) {
  constexpr int LPU = V / 4;          // lanes per u
  constexpr int UPI = 64 / LPU;       // u values per wave-iteration
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)M1 * M2) return;
  const int lane = threadIdx.x & 63;
  const int r1 = (int)(row / M2), r2 = (int)(row - (int64_t)r1 * M2);
  const float4* h4 = reinterpret_cast<const float4*>(H + row * (int64_t)U * V);
  const float* x = X + (int64_t)r1 * x_r1 + (int64_t)r2 * x_r2;
  float* dx = dX + (int64_t)r1 * x_r1 + (int64_t)r2 * x_r2;
  const int v0 = (lane % LPU) * 4, usub = lane / LPU;
  const float4 av = *reinterpret_cast<const float4*>(A2 + (int64_t)r1 * a2_r1 + v0);
  float4 da = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int ub = 0; ub < U; ub += UPI) {
    const int u = ub + usub;
    const bool ok = u < U;
    const float4 hv = ok ? h4[(int64_t)(ub / UPI) * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float xv = ok ? x[u] : 0.f;
    float s = fmaf(av.x, hv.x, fmaf(av.y, hv.y, fmaf(av.z, hv.z, av.w * hv.w)));
#pragma unroll
    for (int off = 1; off < LPU; off <<= 1) s += __shfl_xor(s, off, 64);
    da.x = fmaf(xv, hv.x, da.x);
    da.y = fmaf(xv, hv.y, da.y);
    da.z = fmaf(xv, hv.z, da.z);
    da.w = fmaf(xv, hv.w, da.w);
    if (ok && (lane % LPU) == 0) dx[u] = dx_acc ? dx[u] + s : s;
  }
#pragma unroll
  for (int off = LPU; off < 64; off <<= 1) {
    da.x += __shfl_xor(da.x, off, 64);
    da.y += __shfl_xor(da.y, off, 64);
    da.z += __shfl_xor(da.z, off, 64);
    da.w += __shfl_xor(da.w, off, 64);
  }
  if (lane < LPU) {
    float* o = dA2 + (int64_t)r1 * a2_r1 + v0;
    atomicAdd(o + 0, da.x);
    atomicAdd(o + 1, da.y);
    atomicAdd(o + 2, da.z);
    atomicAdd(o + 3, da.w);
  }
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

int cu_count() {      // compute units of the current device (launches sized to the chip)
  static int n = 0;
  if (!n) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  }
  return n;
}

template <class K>
int launch_batch(K kernel, const e3k::GemmBatch& gb, int blocks, hipStream_t st) {
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, st, gb);
  return E3K_OK;
}

int validate(const e3k_gemm_problem& P, bool wgrad) {
  if (P.M1 < 0 || P.M2 <= 0 || P.N <= 0 || P.K <= 0) return E3K_ERR_INVALID;
  if (P.M1 == 0) return E3K_OK;   // empty problem (a batch without edges): skipped, its pointers may be NULL
  if (!P.A || !P.B || !P.C) return E3K_ERR_INVALID;
  if (P.V < 0 || P.V > e3k::VMAX) return E3K_ERR_UNSUPPORTED;
  if (P.V > 0 && (!P.A2 || P.K % P.V != 0)) return E3K_ERR_INVALID;
  if (wgrad && (P.bias || P.act || P.chain)) return E3K_ERR_INVALID;
  if (P.act < 0 || P.act > 1) return E3K_ERR_INVALID;
  if (P.chain < 0 || P.chain >= e3k::GEMM_MAXP) return E3K_ERR_INVALID;
  return E3K_OK;
}

// followers of a K-chain (e3k.h): the head's rows, columns, output and key groups
bool follows(const e3k_gemm_problem& H, const e3k_gemm_problem& F) {
  return F.chain == 0 && F.V == 0 && H.V == 0 && !F.bias && F.M1 == H.M1 && F.M2 == H.M2 && F.N == H.N && F.C == H.C && F.c_r1 == H.c_r1 &&
         F.c_r2 == H.c_r2 && F.c_n == H.c_n && F.row_index == H.row_index && F.group_dev == H.group_dev && F.alpha != 0.f;
}

bool a_vec(const e3k_gemm_problem& P) {
  return P.V == 0 && P.a_k == 1 && P.K % 4 == 0 && P.a_r1 % 4 == 0 && (P.M2 == 1 || P.a_r2 % 4 == 0) && aligned16(P.A);
}
bool c_vec(const e3k_gemm_problem& P) {
  return P.c_n == 1 && P.N % 4 == 0 && P.c_r1 % 4 == 0 && (P.M2 == 1 || P.c_r2 % 4 == 0) && aligned16(P.C) &&
         (!P.bias || aligned16(P.bias));
}
int b_mode(const e3k_gemm_problem& P) {
  if (P.b_n == 1 && P.N % 4 == 0 && P.b_k % 4 == 0 && aligned16(P.B)) return 1;
  if (P.b_k == 1 && P.K % 4 == 0 && P.b_n % 4 == 0 && aligned16(P.B)) return 2;
  return 0;
}

enum FwdKind { FWD_PLAIN = 0, FWD_SMALLK, FWD_OUTER, FWD_SPLITK, FWD_PERSIST, FWD_KINDS, FWD_FOLLOWER };      // (a follower rides with its head)

struct Batcher {
  e3k::GemmBatch gb{};
  int blocks = 0;
  bool chained = false;      // the batch holds a K-chain
  void reset() {
    gb = e3k::GemmBatch{};
    blocks = 0;
    chained = false;
  }
};
constexpr int MAX_CALL = 64;   // problems per C-ABI call
}  // namespace

// reps[i] > 1: problem i is a keyed template (row_index = the key-sorted permutation, group_dev = the pair of key 0,
// B of key t at B + t * key_stride[i]); reps == nullptr: all plain
static int gemm_fwd_impl(const e3k_gemm_problem* problems, int n_problems, const int* reps, const long long* key_stride,
                         void* stream) {
  if (n_problems < 0 || (n_problems && !problems)) return E3K_ERR_INVALID;
  if (n_problems > MAX_CALL) return E3K_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  int kind[MAX_CALL];
  int64_t plain_tiles128 = 0;
  // The resident-A small-K kernel (gemm_smallk_kernel) is OFF since round 6: with the plain kernel back at five waves per SIMD and the
  // keyed grids compact, every workload measured runs faster with its K <= 64 problems on gemm_kernel<2> -- 256 molecules 3.88 -> 3.82 ms
  // (debug library), l_max 3 6.39 -> 6.28, 32 molecules 1.52 -> 1.47, force training 5.40 -> 5.33, config_diffusion (per-edge radial MLPs, the
  // kernel's original customer) 3.34 -> 3.29, config_diffusion_CA 8.03 -> 7.76.  Its 51 KB of LDS and 152 registers leave three workgroups per
  // CU where a launch has two rounds of them.  E3K_SK_MIN_ROWS=1024 (debug library) routes rows x K <= 64 problems to it again.
  E3K_KNOB_INT(sk_min_rows, "E3K_SK_MIN_ROWS", 1LL << 40);
  E3K_KNOB_INT(splitk_on, "E3K_SPLITK", 1);
  E3K_KNOB_INT(keyed_compact, "E3K_KEYED_COMPACT", 1);
  int follower_of[MAX_CALL];
  for (int i = 0; i < n_problems; ++i) follower_of[i] = -1;
  for (int i = 0; i < n_problems; ++i) {
    const e3k_gemm_problem& P = problems[i];
    const int rc = validate(P, false);
    if (rc != E3K_OK) return rc;
    if (P.chain > 0) {      // a K-chain: head and followers on the plain kernel, in one batch
      if (follower_of[i] >= 0 || i + P.chain >= n_problems) return E3K_ERR_INVALID;
      for (int j = 1; j <= P.chain; ++j) {
        if (!follows(P, problems[i + j]) || (reps && (reps[i + j] > 1) != (reps[i] > 1)) || (reps && reps[i + j] != reps[i])) return E3K_ERR_INVALID;
        const int rcf = validate(problems[i + j], false);
        if (rcf != E3K_OK) return rcf;
        follower_of[i + j] = i;
      }
    }
  }
  for (int i = 0; i < n_problems; ++i) {
    const e3k_gemm_problem& P = problems[i];
    const int64_t M = (int64_t)P.M1 * P.M2;
    if (follower_of[i] >= 0) {
      kind[i] = FWD_FOLLOWER;
      continue;
    }
    if (P.N == 1 && P.V == 0 && P.chain == 0 && !P.row_index && !(reps && reps[i] > 1)) {      // one output column: gemm_n1_kernel, at once
      if (M > 0)
        hipLaunchKernelGGL(e3k::gemm_n1_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, P.A, P.B, (int64_t)P.M1, P.M2, P.K, P.a_r1, P.a_r2,
                           P.a_k, P.b_k, P.c_r1, P.c_r2, P.alpha, P.bias, P.accumulate, P.act, P.act_cst, P.C);
      kind[i] = FWD_FOLLOWER;      // (nothing left for the batched kernels)
      continue;
    }
    if (P.chain > 0) {
      kind[i] = FWD_PLAIN;
      plain_tiles128 += ((M + 127) / 128) * ((P.N + e3k::BN - 1) / e3k::BN);
      continue;
    }
    if (P.V > 0) kind[i] = FWD_OUTER;
    else if (P.K <= e3k::SK_KMAX && a_vec(P) && b_mode(P) == 1 && c_vec(P) && M >= sk_min_rows) kind[i] = FWD_SMALLK;
    else if (splitk_on && a_vec(P) && b_mode(P) == 2 && P.K >= 256 && P.K % 64 == 0 && ((M + 63) / 64) * ((P.N + e3k::BN - 1) / e3k::BN) < 128)
      kind[i] = FWD_SPLITK;
    else {
      kind[i] = FWD_PLAIN;
      plain_tiles128 += ((M + 127) / 128) * ((P.N + e3k::BN - 1) / e3k::BN);   // keyed: the groups partition these rows
    }
  }
  // A split-K problem next to plain ones costs the call a launch of its own (the accumulating second round of a layer's input
  // gradients: the gate scalars' K = 256 block of the self-connection goes split-K, its sibling of the trailing Linear plain:
  // 13 + 14 us as two dependent launches in the replayed step).  When the call has plain problems anyway and the split-K one is
  // small (K <= 512), it rides in their launch as a plain problem: its 64-row tiles finish under the siblings' tiles.
  {
    bool any_plain = false;
    for (int i = 0; i < n_problems; ++i) any_plain = any_plain || kind[i] == FWD_PLAIN;
    if (any_plain)
      for (int i = 0; i < n_problems; ++i)
        if (kind[i] == FWD_SPLITK && problems[i].K <= 512) {
          kind[i] = FWD_PLAIN;
          plain_tiles128 += (((int64_t)problems[i].M1 * problems[i].M2 + 127) / 128) * ((problems[i].N + e3k::BN - 1) / e3k::BN);
        }
  }
  // ... and the other way round: a PLAIN problem of a few tiles with a long-ish K loop (the radial stack's last-layer input gradient of
  // layer 0: 9 tiles, K = 192, beside its siblings' K = 960 .. 1 920 which go split-K) waits 10 us in a launch of its own for nine
  // serial K loops.  When the call has split-K problems anyway, it joins them.
  {
    bool any_splitk = false;
    for (int i = 0; i < n_problems; ++i) any_splitk = any_splitk || kind[i] == FWD_SPLITK;
    if (any_splitk && splitk_on)
      for (int i = 0; i < n_problems; ++i) {
        const e3k_gemm_problem& P = problems[i];
        const int64_t M = (int64_t)P.M1 * P.M2;
        if (kind[i] == FWD_PLAIN && P.chain == 0 && !(reps && reps[i] > 1) && a_vec(P) && b_mode(P) == 2 && P.K >= 128 && P.K % 64 == 0 &&
            ((M + 63) / 64) * ((P.N + e3k::BN - 1) / e3k::BN) < 32) {
          kind[i] = FWD_SPLITK;
          plain_tiles128 -= ((M + 127) / 128) * ((P.N + e3k::BN - 1) / e3k::BN);
        }
      }
  }
  // plain problems with vector-loadable operands: the persistent kernel when the launch holds enough tiles for every
  // workgroup slot to walk a sequence of them (below that the one-tile-per-workgroup kernels start sooner)
#ifdef E3K_DEBUG_KNOBS
  E3K_KNOB_INT(kPersist, "E3K_GEMM_PERSIST", 0);
  E3K_KNOB_INT(kPersistMin, "E3K_GEMM_PERSIST_MIN_TILES", 1024);
  const int n_cu = cu_count();
  if (kPersist) {
    int64_t ptiles = 0;
    auto persistable = [&](int i) {
      const e3k_gemm_problem& P = problems[i];
      return kind[i] == FWD_PLAIN && a_vec(P) && b_mode(P) != 0 && !(reps && reps[i] > 1) && !P.row_index && (int64_t)P.M1 * P.M2 > 0;
    };
    for (int i = 0; i < n_problems; ++i)
      if (persistable(i)) ptiles += (((int64_t)problems[i].M1 * problems[i].M2 + 127) / 128) * ((problems[i].N + e3k::BN - 1) / e3k::BN);
    if (ptiles >= kPersistMin)
      for (int i = 0; i < n_problems; ++i)
        if (persistable(i)) {
          kind[i] = FWD_PERSIST;
          plain_tiles128 -= (((int64_t)problems[i].M1 * problems[i].M2 + 127) / 128) * ((problems[i].N + e3k::BN - 1) / e3k::BN);
        }
  }
#endif
  // 64-row tiles (gemm_kernel<2>) unless the launch holds at least kSmallGrid 128-row tiles.  Round 3 switched to 128-row
  // tiles from three per CU; re-measured in round 4 (tools/ab_bench.py, 2-3 interleaved rounds): the 64-row form wins at
  // every size met -- trailing Linear forward 56 vs 62 us (twice the tiles: the unequal-K problems of a launch spread more
  // evenly over the CUs), 256 molecules 4.65 vs 4.69 ms, l_max 3 7.03 vs 7.16, the protein net 8.99 vs 9.09 -- so the
  // threshold now sits above anything a layer issues and gemm_kernel<4> is the form kept for launches beyond it.
  E3K_KNOB_INT(kSmallGrid, "E3K_GEMM_SMALL_GRID_TILES", 1 << 20);
  const bool small_grid = plain_tiles128 < kSmallGrid;
  for (int k = 0; k < FWD_KINDS; ++k) {
    Batcher b;
    auto flush = [&]() -> int {
      if (!b.blocks) {
        b.reset();
        return E3K_OK;
      }
      b.gb.tile_start[b.gb.n] = b.blocks;
      int rc = E3K_OK;
      switch (k) {
        case FWD_PLAIN:
          if (b.chained)
            rc = small_grid ? launch_batch(e3k::gemm_kernel<2, true>, b.gb, b.blocks, st) : launch_batch(e3k::gemm_kernel<4, true>, b.gb, b.blocks, st);
          else
            rc = small_grid ? launch_batch(e3k::gemm_kernel<2, false>, b.gb, b.blocks, st) : launch_batch(e3k::gemm_kernel<4, false>, b.gb, b.blocks, st);
          break;
        case FWD_SMALLK: rc = launch_batch(e3k::gemm_smallk_kernel, b.gb, b.blocks, st); break;
        case FWD_SPLITK: rc = launch_batch(e3k::gemm_splitk_kernel, b.gb, b.blocks, st); break;
#ifdef E3K_DEBUG_KNOBS
        case FWD_PERSIST: {
          E3K_KNOB_INT(kPersistOcc, "E3K_GEMM_PERSIST_WG_PER_CU", 2);
          const int grid = b.blocks < (int)kPersistOcc * n_cu ? b.blocks : (int)kPersistOcc * n_cu;
          hipLaunchKernelGGL(e3k::gemm_persist_kernel, dim3(grid), dim3(256), 0, st, b.gb, b.blocks);
          break;
        }
#endif
        default: rc = launch_batch(e3k::gemm_outer_kernel, b.gb, b.blocks, st); break;
      }
      b.reset();
      return rc;
    };
    // longest-processing-time-first: workgroups are dispatched in blockIdx order, so the problems with the longest
    // K loops go first and the short ones fill the tail of the launch
    int order[MAX_CALL];
    for (int i = 0; i < n_problems; ++i) order[i] = i;
    auto k_total = [&](int a) {
      int kt = problems[a].K;
      for (int j = 1; j <= problems[a].chain; ++j) kt += problems[a + j].K;
      return kt;
    };
    std::stable_sort(order, order + n_problems, [&](int a, int b) { return k_total(a) > k_total(b); });
    for (int oi = 0; oi < n_problems; ++oi) {
      const int i = order[oi];
      if (kind[i] != k) continue;
      const e3k_gemm_problem& P = problems[i];
      const int64_t M = (int64_t)P.M1 * P.M2;
      if (M == 0) continue;
      if (P.chain > 0 && b.gb.n + 1 + P.chain > e3k::GEMM_MAXP) {      // a chain does not straddle two launches
        const int rc = flush();
        if (rc != E3K_OK) return rc;
      }
      const int rp = reps && reps[i] > 1 ? reps[i] : 1;
      const int tiles_n = (P.N + e3k::BN - 1) / e3k::BN;
      int64_t blocks;
      int aux = 0;
      if (k == FWD_SMALLK) {
        E3K_KNOB_INT(sk_ct, "E3K_SK_CT", e3k::SK_CT);
        // a block keeps its A tile and walks `aux` column tiles -- unless the problem has too few tiles to fill the chip
        // that way (the radial MLP's last layer on the knot table: 33 row tiles): then fewer columns per block
        const int64_t fill = ((M + 127) / 128) * tiles_n / 1024;
        aux = tiles_n < sk_ct ? tiles_n : sk_ct;
        if (fill < aux) aux = fill < 1 ? 1 : (int)fill;
        blocks = ((M + 127) / 128) * ((tiles_n + aux - 1) / aux);
      } else if (k == FWD_SPLITK) {
        blocks = ((M + 31) / 32) * ((P.N + 31) / 32);
      } else if (k == FWD_PLAIN && small_grid) {
        blocks = ((M + 63) / 64) * tiles_n;
      } else {
        blocks = ((M + 127) / 128) * tiles_n;
      }
      bool compact = false;
      if (rp > 1 && keyed_compact && P.group_dev && P.row_index && (k == FWD_SMALLK || (k == FWD_PLAIN))) {
        // (blocks = row tiles x workgroups per row tile for both kernels: the key groups partition the rows, see fetch_problem)
        const int64_t bm = k == FWD_SMALLK ? 128 : (small_grid ? 64 : 128);
        const int64_t row_tiles = (M + bm - 1) / bm, per_row = blocks / row_tiles;
        blocks = (row_tiles + rp) * per_row;
        compact = true;
      } else {
        blocks *= rp;
      }
      if (b.blocks + blocks > 0x7fffffffLL) return E3K_ERR_INVALID;
      e3k::GemmBatch& gb = b.gb;
      gb.p[gb.n] = P;
      gb.reps[gb.n] = rp;
      gb.key_stride[gb.n] = rp > 1 ? key_stride[i] : 0;
      E3K_KNOB_INT(kAblF, "E3K_GEMM_ABLATE", 0);
      gb.flags[gb.n] = (a_vec(P) ? 1 : 0) | (b_mode(P) << 1) | (int)kAblF | (compact ? 64 : 0);
      gb.aux[gb.n] = aux;
      gb.tile_start[gb.n] = b.blocks;
      b.blocks += (int)blocks;
      ++gb.n;
      if (P.chain > 0) b.chained = true;
      for (int j = 1; j <= P.chain; ++j) {      // the followers: right behind their head, no tiles of their own
        const e3k_gemm_problem& F = problems[i + j];
        gb.p[gb.n] = F;
        gb.p[gb.n].accumulate = P.accumulate;      // (the last link's descriptor runs the epilogue)
        gb.p[gb.n].bias = P.bias;
        gb.p[gb.n].act = P.act;
        gb.p[gb.n].act_cst = P.act_cst;
        gb.reps[gb.n] = rp;
        gb.key_stride[gb.n] = rp > 1 ? key_stride[i + j] : 0;
        gb.flags[gb.n] = (a_vec(F) ? 1 : 0) | (b_mode(F) << 1) | (int)kAblF;
        gb.aux[gb.n] = 0;
        gb.tile_start[gb.n] = b.blocks;
        ++gb.n;
      }
      if (gb.n == e3k::GEMM_MAXP) {
        const int rc = flush();
        if (rc != E3K_OK) return rc;
      }
    }
    const int rc = flush();
    if (rc != E3K_OK) return rc;
  }
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int e3k_gemm(const e3k_gemm_problem* problems, int n_problems, void* stream) {
  return gemm_fwd_impl(problems, n_problems, nullptr, nullptr, stream);
}

static int gemm_wgrad_impl(const e3k_gemm_problem* problems, int n_problems, const int* reps, const long long* key_stride,
                           void* stream) {
  if (n_problems < 0 || (n_problems && !problems)) return E3K_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  for (int i = 0; i < n_problems; ++i) {
    const int rc = validate(problems[i], true);
    if (rc != E3K_OK) return rc;
  }
  // Problems with 16-byte-loadable operands (plain and keyed; not the outer-product form): the pipelined kernel, ONE launch
  // sized to one round of workgroups (three per CU), every workgroup the same number of rows -- splits proportional to a
  // problem's rows.  A keyed problem's key groups partition its M1 rows: it counts once, every key gets the splits of the
  // whole, and the workgroups past a key's last row exit.
  E3K_KNOB_INT(kV2, "E3K_WGRAD2", 1);
  E3K_KNOB_INT(kBlocks, "E3K_WGRAD2_BLOCKS", 0);
  E3K_KNOB_INT(kMinChunks, "E3K_WGRAD2_MIN_CHUNKS", 2);
  E3K_KNOB_INT(kAbl, "E3K_WGRAD2_ABLATE", 0);
  E3K_KNOB_INT(kCompact, "E3K_KEYED_COMPACT", 1);
  bool taken[MAX_CALL] = {};
  if (n_problems > MAX_CALL) return E3K_ERR_INVALID;
  // one output column, plain rows: a weighted column sum (wgrad_n1_kernel)
  for (int i = 0; i < n_problems; ++i) {
    const e3k_gemm_problem& P = problems[i];
    if (P.N != 1 || P.V != 0 || P.row_index || (reps && reps[i] > 1) || (int64_t)P.M1 * P.M2 <= 0) continue;
    const int64_t rows = (int64_t)P.M1 * P.M2;
    int gy = (int)((rows + 31) / 32);
    if (gy > 1024) gy = 1024;
    hipLaunchKernelGGL(e3k::wgrad_n1_kernel, dim3((P.K + 63) / 64, gy), dim3(256), 0, st, P.A, P.C, (int64_t)P.M1, P.M2, P.K, P.a_r1,
                       P.a_r2, P.a_k, P.c_r1, P.c_r2, P.b_k, P.alpha, const_cast<float*>(P.B));
    taken[i] = true;
  }
  auto g_vec = [](const e3k_gemm_problem& P) {
    return P.c_n == 1 && P.N % 4 == 0 && P.c_r1 % 4 == 0 && (P.M2 == 1 || P.c_r2 % 4 == 0) && aligned16(P.C);
  };
  auto tiles_of = [](const e3k_gemm_problem& P) { return ((P.K + (P.K > 64 ? 127 : 63)) / (P.K > 64 ? 128 : 64)) * ((P.N + 63) / 64); };
  // launch(gb, blocks) for every batch of up to GEMM_MAXP eligible problems, `target` workgroups in total, `chunk` rows per
  // pipeline stage; gathered: keyed / row-indexed problems are eligible too
  auto equal_rows = [&](auto launch, double target, int chunk, bool gathered) -> int {
    auto eligible = [&](int i) {
      const e3k_gemm_problem& P = problems[i];
      if (!gathered && ((reps && reps[i] > 1) || P.row_index)) return false;
      return P.V == 0 && (int64_t)P.M1 * P.M2 > 0 && a_vec(P) && g_vec(P);
    };
    double total = 0;      // rows x tiles over the problems of this launch
    for (int i = 0; i < n_problems; ++i)
      if (eligible(i)) total += (double)problems[i].M1 * problems[i].M2 * tiles_of(problems[i]);
    Batcher b;
    auto flush = [&]() -> int {
      if (!b.blocks) { b.reset(); return E3K_OK; }
      b.gb.tile_start[b.gb.n] = b.blocks;
      const int rc = launch(b.gb, b.blocks);
      b.reset();
      return rc;
    };
    for (int i = 0; i < n_problems && total > 0; ++i) {
      if (!eligible(i)) continue;
      const e3k_gemm_problem& P = problems[i];
      const int64_t M = (int64_t)P.M1 * P.M2;
      const int rp = reps && reps[i] > 1 ? reps[i] : 1;
      taken[i] = true;
      int64_t splits = (int64_t)(target * (double)M / total);      // rounded down: the launch stays within one round
      const int64_t max_splits = (M + kMinChunks * chunk - 1) / (kMinChunks * chunk);
      if (splits > max_splits) splits = max_splits;
      if (splits < 1) splits = 1;
      e3k::GemmBatch& gb = b.gb;
      gb.p[gb.n] = P;
      gb.reps[gb.n] = rp;
      gb.key_stride[gb.n] = rp > 1 ? key_stride[i] : 0;
      gb.flags[gb.n] = 9 | (int)kAbl;
      gb.aux[gb.n] = (int)splits;
      gb.tile_start[gb.n] = b.blocks;
      if (rp > 1 && kCompact && P.group_dev && P.row_index) {
        // the key groups partition the M rows: `splits` splits of the whole (+ one per key for the remainders) instead of `splits` per key
        const int64_t rows = ((M + splits - 1) / splits + chunk - 1) / chunk * chunk;
        gb.flags[gb.n] |= 64;
        gb.aux[gb.n] = (int)rows;
        b.blocks += tiles_of(P) * (int)((M + rows - 1) / rows + rp);
      } else {
        b.blocks += tiles_of(P) * (int)splits * rp;
      }
      if (++gb.n == e3k::GEMM_MAXP) {
        const int rc = flush();
        if (rc != E3K_OK) return rc;
      }
    }
    return flush();
  };
  const int n_cu = cu_count();
#ifdef E3K_DEBUG_KNOBS
  if (kV2 == 2) {      // experiment: the LDS-direct ring kernel (plain problems only)
    E3K_KNOB_INT(kBlocks3, "E3K_WGRAD3_BLOCKS", 0);
    E3K_KNOB_INT(kCfg3, "E3K_WGRAD3_CFG", 0);
    const int rc = equal_rows(
        [&](const e3k::GemmBatch& gb, int blocks) {
          return kCfg3 == 1   ? launch_batch(e3k::gemm_wgrad3_kernel<64, 3, 1>, gb, blocks, st)
                 : kCfg3 == 2 ? launch_batch(e3k::gemm_wgrad3_kernel<32, 3, 2>, gb, blocks, st)
                              : launch_batch(e3k::gemm_wgrad3_kernel<32, 4, 1>, gb, blocks, st);
        },
        kBlocks3 > 0 ? (double)kBlocks3 : (double)n_cu * (kCfg3 == 2 ? 2 : 1), kCfg3 == 1 ? 64 : 32, false);
    if (rc != E3K_OK) return rc;
  }
#endif
  if (kV2 == 1) {
    const int rc = equal_rows([&](const e3k::GemmBatch& gb, int blocks) { return launch_batch(e3k::gemm_wgrad2_kernel, gb, blocks, st); },
                              kBlocks > 0 ? (double)kBlocks : 3.0 * n_cu, e3k::W2R, true);
    if (rc != E3K_OK) return rc;
  }
  for (int mode = 0; mode < 4; ++mode) {  // (outer: x (x) attrs formed on the fly?, 128-wide output tile?)
    const bool outer = mode & 1;
    const int tn = (mode & 2) ? 2 : 1;
    Batcher b;
    auto flush = [&]() -> int {
      if (!b.blocks) {
        b.reset();
        return E3K_OK;
      }
      b.gb.tile_start[b.gb.n] = b.blocks;
      int rc;
      if (!outer) rc = tn == 2 ? launch_batch(e3k::gemm_wgrad_kernel<false, 2>, b.gb, b.blocks, st)
                               : launch_batch(e3k::gemm_wgrad_kernel<false, 1>, b.gb, b.blocks, st);
      else rc = tn == 2 ? launch_batch(e3k::gemm_wgrad_kernel<true, 2>, b.gb, b.blocks, st)
                        : launch_batch(e3k::gemm_wgrad_kernel<true, 1>, b.gb, b.blocks, st);
      b.reset();
      return rc;
    };
    for (int i = 0; i < n_problems; ++i) {
      const e3k_gemm_problem& P = problems[i];
      if (taken[i]) continue;
      if ((P.V > 0) != outer) continue;
      if ((P.N >= 128 ? 2 : 1) != tn) continue;
      const int64_t M = (int64_t)P.M1 * P.M2;
      if (M == 0) continue;
      const int rp = reps && reps[i] > 1 ? reps[i] : 1;
      int f = a_vec(P) ? 1 : 0;
      if (P.c_n == 1 && P.N % 4 == 0 && P.c_r1 % 4 == 0 && (P.M2 == 1 || P.c_r2 % 4 == 0) && aligned16(P.C)) f |= 8;
      const int wn = 64 * tn;
      const int tiles = ((P.K + e3k::WK - 1) / e3k::WK) * ((P.N + wn - 1) / wn);
      E3K_KNOB_INT(kTarget, "E3K_WGRAD_TARGET", 1024);
      E3K_KNOB_INT(kChunks, "E3K_WGRAD_CHUNKS", 4);
      int64_t splits = (kTarget + tiles - 1) / tiles;
      // rows per workgroup: kChunks chunks of WR rows -- unless that leaves most of the chip idle (round 6: the one-hot embeddings'
      // and the energy head's weight gradients are ONE tile over 4 600 rows: 18 workgroups walking four chunks each took 17-22 us
      // apiece in the replayed step; a chunk per workgroup, 73 workgroups, is one load latency long)
      int chunks = kChunks;
      while (chunks > 1 && tiles * ((M + (int64_t)chunks * e3k::WR - 1) / ((int64_t)chunks * e3k::WR)) < n_cu) chunks >>= 1;
      const int64_t max_splits = (M + chunks * e3k::WR - 1) / (chunks * e3k::WR);
      if (splits > max_splits) splits = max_splits;
      if (splits < 1) splits = 1;
      e3k::GemmBatch& gb = b.gb;
      gb.p[gb.n] = P;
      gb.reps[gb.n] = rp;
      gb.key_stride[gb.n] = rp > 1 ? key_stride[i] : 0;
      gb.flags[gb.n] = f;
      gb.aux[gb.n] = (int)splits;
      gb.tile_start[gb.n] = b.blocks;
      b.blocks += tiles * (int)splits * rp;
      if (++gb.n == e3k::GEMM_MAXP) {
        const int rc = flush();
        if (rc != E3K_OK) return rc;
      }
    }
    const int rc = flush();
    if (rc != E3K_OK) return rc;
  }
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

#ifdef E3K_STAMPS
extern "C" int e3k_debug_stamps(unsigned long long* out, int n) {
  if (hipDeviceSynchronize() != hipSuccess) return E3K_ERR_LAUNCH;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(e3k::e3k_dbg_buf), sizeof(unsigned long long) * n) != hipSuccess) return E3K_ERR_LAUNCH;
  return E3K_OK;
}
#endif

extern "C" int e3k_gemm_wgrad(const e3k_gemm_problem* problems, int n_problems, void* stream) {
  return gemm_wgrad_impl(problems, n_problems, nullptr, nullptr, stream);
}

namespace {
// resolves one segment's templates into `out` (pointer fields of a template hold byte offsets; A2 / bias offset + 1)
int resolve_segment(const e3k_gemm_segment& sg, e3k_gemm_problem* out, int* reps, long long* key_stride, int cap) {
  if (sg.n_templates < 0 || sg.n_templates > cap || (sg.n_templates && !sg.templates)) return -1;
  if (sg.M1 > 0x7fffffffLL) return -1;
  const bool keyed = sg.n_keys > 0;
  if (keyed && (!sg.perm || !sg.groups_dev)) return -1;
  const auto off = [](const void* q) { return reinterpret_cast<uintptr_t>(q); };
  for (int i = 0; i < sg.n_templates; ++i) {
    e3k_gemm_problem p = sg.templates[i];
    p.A = reinterpret_cast<const float*>(off(sg.a_base) + off(p.A));
    p.B = reinterpret_cast<const float*>(off(sg.b_base) + off(p.B));
    p.C = reinterpret_cast<float*>(off(sg.c_base) + off(p.C));
    if (sg.M1 >= 0) {   // rebased templates (M1 < 0: the templates carry addresses and their own row counts)
      if (p.A2) {
        if (!sg.a2_base) return -1;
        p.A2 = reinterpret_cast<const float*>(off(sg.a2_base) + off(p.A2) - 1);
      }
      if (p.bias) {
        if (!sg.bias_base) return -1;
        p.bias = reinterpret_cast<const float*>(off(sg.bias_base) + off(p.bias) - 1);
      }
      p.M1 = (int32_t)sg.M1;
    }
    if (keyed) {
      p.row_index = sg.perm;
      p.group_dev = sg.groups_dev;
    }
    out[i] = p;
    reps[i] = keyed ? sg.n_keys : 1;
    key_stride[i] = keyed ? sg.b_key_stride : 0;
  }
  return sg.n_templates;
}
}  // namespace

extern "C" int e3k_gemm_multi(const e3k_gemm_segment* segments, int32_t n_segments, int32_t wgrad, void* stream) {
  if (n_segments < 0 || (n_segments && !segments)) return E3K_ERR_INVALID;
  e3k_gemm_problem buf[MAX_CALL];
  int reps[MAX_CALL];
  long long ks[MAX_CALL];
  int n = 0;
  for (int s = 0; s < n_segments; ++s) {
    const int got = resolve_segment(segments[s], buf + n, reps + n, ks + n, MAX_CALL - n);
    if (got < 0) return E3K_ERR_INVALID;
    n += got;
  }
  return wgrad ? gemm_wgrad_impl(buf, n, reps, ks, stream) : gemm_fwd_impl(buf, n, reps, ks, stream);
}

extern "C" int e3k_gemm_rebased(const e3k_gemm_problem* templates, int n_templates, const void* a_base,
                                const void* a2_base, const void* b_base, void* c_base, const void* bias_base,
                                int64_t M1, int32_t wgrad, void* stream) {
  if (M1 < 0) return E3K_ERR_INVALID;
  e3k_gemm_segment sg{};
  sg.templates = templates; sg.n_templates = n_templates;
  sg.a_base = a_base; sg.a2_base = a2_base; sg.b_base = b_base; sg.c_base = c_base; sg.bias_base = bias_base;
  sg.M1 = M1;
  return e3k_gemm_multi(&sg, 1, wgrad, stream);
}

extern "C" int e3k_gemm_grouped(const e3k_gemm_problem* templates, int n_templates, const int32_t* perm,
                                const int32_t* groups_dev, int32_t n_keys, int64_t b_key_stride, int32_t wgrad,
                                void* stream) {
  if (n_keys <= 0) return E3K_ERR_INVALID;
  e3k_gemm_segment sg{};
  sg.templates = templates; sg.n_templates = n_templates;
  sg.M1 = -1;
  sg.perm = perm; sg.groups_dev = groups_dev; sg.n_keys = n_keys; sg.b_key_stride = b_key_stride;
  return e3k_gemm_multi(&sg, 1, wgrad, stream);
}

extern "C" int e3k_gemm_grouped_rebased(const e3k_gemm_problem* templates, int n_templates, const void* a_base,
                                        const void* b_base, void* c_base, int64_t M1, const int32_t* perm,
                                        const int32_t* groups_dev, int32_t n_keys, int64_t b_key_stride, int32_t wgrad,
                                        void* stream) {
  if (M1 < 0 || n_keys <= 0) return E3K_ERR_INVALID;
  e3k_gemm_segment sg{};
  sg.templates = templates; sg.n_templates = n_templates;
  sg.a_base = a_base; sg.b_base = b_base; sg.c_base = c_base;
  sg.M1 = M1;
  sg.perm = perm; sg.groups_dev = groups_dev; sg.n_keys = n_keys; sg.b_key_stride = b_key_stride;
  return e3k_gemm_multi(&sg, 1, wgrad, stream);
}

extern "C" int e3k_colsum(const float* G, int64_t rows, int32_t cols, int64_t ld, float* out, void* stream) {
  if (rows < 0 || cols <= 0 || !out) return E3K_ERR_INVALID;
  if (rows == 0) return E3K_OK;
  if (!G) return E3K_ERR_INVALID;
  // (a thread's walk down its rows is a chain of load latencies: 64 rows per thread took 15 us for a 1 MB matrix -- 8 rows, 3 us)
  int gy = (int)((rows + 31) / 32);
  if (gy > 1024) gy = 1024;
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
  const dim3 grid((unsigned)((rows + 3) / 4));
  hipStream_t st = (hipStream_t)stream;
  // 16-byte streams need: H rows (U*V floats) and the A2 rows 16-byte aligned
  const bool vec = aligned16(H) && aligned16(A2) && a2_r1 % 4 == 0 && ((int64_t)U * V) % 4 == 0 && 64 % (V / 4 > 0 ? V / 4 : 1) == 0;
#define E3K_FCTP_VEC(VV)                                                                                              \
  hipLaunchKernelGGL(e3k::fctp_reduce_vec_kernel<VV>, grid, dim3(256), 0, st, H, X, A2, M1, M2, U, x_r1, x_r2, a2_r1, dX, \
                     dx_accumulate, dA2)
  if (vec && V == 32) E3K_FCTP_VEC(32);
  else if (vec && V == 16) E3K_FCTP_VEC(16);
  else if (vec && V == 8) E3K_FCTP_VEC(8);
  else if (vec && V == 4) E3K_FCTP_VEC(4);
  else
    hipLaunchKernelGGL(e3k::fctp_reduce_kernel, grid, dim3(256), 0, st, H, X, A2, M1, M2, U, V, x_r1, x_r2, a2_r1, dX,
                       dx_accumulate, dA2);
#undef E3K_FCTP_VEC
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}
