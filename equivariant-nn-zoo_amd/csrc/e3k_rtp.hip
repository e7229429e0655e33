// Radial-fused tensor product for gfx950 (MI355X): the per-edge path weights never touch HBM.
//
// Replaces, per convolution layer of the reference (paths relative to /root/reference):
//   weight = self.fc(edge_radial)      LAST layer of the radial MLP    e3_layers/nn/message_passing.py:74-79,93
//   x[edge_src] gather                                                 e3_layers/nn/message_passing.py:96,105
//   o3.TensorProduct 'uvu'                                             e3_layers/nn/pointwise.py:78-85,94-98
//   scatter over edge_dst                                              e3_layers/nn/message_passing.py:109
// In e3k_tp.hip the weights w[E, weight_numel] (7.7 KB per edge at n_dim 64, l_max 2) are written by a GEMM and read
// back by every tensor-product pass: six trips through HBM per layer and training step.  Here a workgroup owns a tile
// of 64 consecutive edges of the CSR order and alternates, per chunk of <= 2 paths of one input block:
//   matrix phase  w_tile[64 edges, 2 x 64 channels] = h_tile[64, 64] . Wl[64, chunk columns]      v_mfma_f32_32x32x2_f32
//                 (exact f32; A fragments = the tile's hidden activations, resident in registers for the whole tile;
//                 B fragments straight from the L2-resident weight matrix), accumulators -> LDS;
//   vector phase  the Clebsch-Gordan contraction of e3k_tp.hip, one wave per (node of the tile, chunk), lane = channel,
//                 weights read from LDS (a conflict-free row per edge and path), accumulation in registers in
//                 ascending edge order, one 256-B store per output row.
// The matrix pipe and the vector ALUs are separate: with several workgroups per CU one tile's matrix phase runs
// under another's vector phase.  A node whose in-edges straddle a tile border is completed with float atomics into
// a row that a small pre-kernel zeroes (at most one such node per tile); every other row is stored exactly once.
#include <cstdlib>

#include "e3k_tp_body.h"

namespace e3k {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int RT_TE = 64;   // edges per tile
constexpr int RT_LD = 65;   // LDS row stride in floats: lanes 0..31 reading one column of 32 rows hit 32 banks
constexpr int RT_K = 64;    // hidden width of the radial MLP (contraction length)

struct RtpArgs {
  const float* h;       // [E, 64] activations of the last hidden layer
  const float* wl;      // [64, W] last-layer weights (row = hidden unit)
  float w_scale;        // 1 / sqrt(64): e3nn's FullyConnectedNet normalisation
  const float* x;       // [N, d_in] cf
  const float* sh;      // [E, d_sh]
  const float* g_out;   // [N, d_mid] cf (backward)
  float* out;           // forward: [N, d_mid]; backward wrt x: g_x [N, d_in]
  float* g_h;           // [E, 64]
  float* g_wl;          // [64, W], accumulated
  const int32_t* nbr;   // src[e] (dst-CSR passes) or dst[e] (src-CSR pass)
  const int32_t* ptr;   // CSR row pointers [N+1]
  const int32_t* perm;  // CSR edge ids [E]
  const int32_t* own0;  // [n_tiles+1]: first node whose segment STARTS at or after the tile's first position
  int32_t d_in, d_sh, W, d_mid, E, n_tiles, n_chunks;
  int32_t y_off[3];     // column of each sh degree in a row of sh (-1: absent); the same for every group of the plan
#ifdef E3K_RTP_STAMPS
  unsigned long long* stamps;   // diagnostic build only: [n_tiles][4 waves][5] cycle sums (prologue, matrix, wait, vector, wait)
#endif
};

#ifdef E3K_RTP_STAMPS
#define RTP_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define RTP_STAMP(var)
#endif

// workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the wave's global stores and atomics
// (s_waitcnt vmcnt(0)): 1-3 k cycles per phase here, for rows that no other wave of the workgroup ever reads.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ int acc_row16(int i, int lane) { return (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5); }

// A fragments of the wave's 32 tile rows: af[s] = w_scale * h[edge(row), hh*32 + s]  (k permuted so that a lane reads a
// contiguous run; the B fragments use the same permutation)
__device__ __forceinline__ void rtp_load_a(const RtpArgs& a, int tile, int rb, int lane, float (&af)[32]) {
  const int row = tile * RT_TE + rb * 32 + (lane & 31);
#pragma unroll
  for (int s = 0; s < 32; ++s) af[s] = 0.0f;
  if (row < a.E) {
    const int e = a.perm[row];
    const float4* hp = reinterpret_cast<const float4*>(a.h + (int64_t)e * RT_K + (lane >> 5) * 32);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float4 v = hp[q];
      af[4 * q] = v.x * a.w_scale; af[4 * q + 1] = v.y * a.w_scale; af[4 * q + 2] = v.z * a.w_scale; af[4 * q + 3] = v.w * a.w_scale;
    }
  }
}

// matrix phase of one chunk: wave (rb, j) computes rows rb*32.. of column blocks j, j+2 (block cb = path cb>>1, half cb&1)
__device__ __forceinline__ void rtp_w_chunk(const RtpArgs& a, const e3k_rtp_chunk& c, const float (&af)[32], float* w_s,
                                            int wave, int lane) {
  const int rb = wave & 1, j = wave >> 1, hh = lane >> 5, n = lane & 31;
  const float* __restrict__ bp0 = a.wl + (int64_t)(hh * 32) * a.W + c.col[0] + j * 32 + n;
  float b0[32], b1[32];
#pragma unroll
  for (int s = 0; s < 32; ++s) b0[s] = bp0[(int64_t)s * a.W];
  if (c.np == 2) {     // the second block's fragments are in flight under the first block's MFMAs
    const float* __restrict__ bp1 = a.wl + (int64_t)(hh * 32) * a.W + c.col[1] + j * 32 + n;
#pragma unroll
    for (int s = 0; s < 32; ++s) b1[s] = bp1[(int64_t)s * a.W];
  }
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
#pragma unroll
  for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], b0[s], acc, 0, 0, 0);
  float* dst = w_s + (rb * 32) * RT_LD + j * 32 + n;
#pragma unroll
  for (int i = 0; i < 16; ++i) dst[acc_row16(i, lane) * RT_LD] = acc[i];
  if (c.np == 2) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
#pragma unroll
    for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], b1[s], acc, 0, 0, 0);
    dst += RT_TE * RT_LD;
#pragma unroll
    for (int i = 0; i < 16; ++i) dst[acc_row16(i, lane) * RT_LD] = acc[i];
  }
}

constexpr int RT_SH = 16;   // floats per staged sh row (d_sh <= 16)

// per-tile edge metadata staged once: the CG loops then depend on global memory only through the x rows, whose
// addresses are known up front (a perm -> src -> x chain of three dependent misses per edge otherwise)
struct TileLds {
  float* w;        // [2][RT_TE][RT_LD]
  float* sh;       // [RT_TE][RT_SH]
  int32_t* nbr;    // [RT_TE]
  int32_t* eid;    // [RT_TE]
  int32_t* ptr;    // [RT_TE + 2]: CSR pointers of the tile's nodes (first .. end inclusive)
  int32_t* next;   // [1]: next unclaimed node of the current vector phase
};
__device__ __forceinline__ void stage_tile(const RtpArgs& a, int tile, const TileLds& L) {
  const int t = threadIdx.x;
  if (t < RT_TE) {
    const int pos = tile * RT_TE + t;
    int e = 0, s = 0;
    if (pos < a.E) {
      e = a.perm[pos];
      s = a.nbr[e];
    }
    L.eid[t] = e;
    L.nbr[t] = s;
  }
  __syncthreads();
  // canonical row: [0] = l 0, [1..3] = l 1, [4..8] = l 2, whatever the column order of sh (absent degrees read zero)
  for (int idx = t; idx < RT_TE * 9; idx += 256) {
    const int te = idx / 9, j = idx - te * 9;
    const int l = j == 0 ? 0 : (j < 4 ? 1 : 2);
    const int col = a.y_off[l] + (j - (l == 0 ? 0 : (l == 1 ? 1 : 4)));
    L.sh[te * RT_SH + j] = (a.y_off[l] >= 0 && tile * RT_TE + te < a.E) ? a.sh[(int64_t)L.eid[te] * a.d_sh + col] : 0.0f;
  }
}
// (after tile_nodes) the row pointers of the tile's nodes: at most RT_TE + 1 nodes touch a tile
__device__ __forceinline__ void stage_ptr(const RtpArgs& a, int first, int end, const TileLds& L) {
  const int t = threadIdx.x;
  if (t <= end - first && t < RT_TE + 2) L.ptr[t] = a.ptr[first + t];
}
struct TileNodes {
  int first, end;   // nodes [first, end) have edges in (or start inside) the tile
  int t0, t1;       // CSR positions of the tile
};
__device__ __forceinline__ TileNodes tile_nodes(const RtpArgs& a, int tile) {
  TileNodes tn;
  tn.t0 = tile * RT_TE;
  tn.t1 = tn.t0 + RT_TE < a.E ? tn.t0 + RT_TE : a.E;
  const int n_lo = uniform(a.own0[tile]);
  tn.end = uniform(a.own0[tile + 1]);
  // the node that contains position t0 started earlier: it is completed here with atomics
  tn.first = uniform(a.ptr[n_lo]) > tn.t0 ? n_lo - 1 : n_lo;
  return tn;
}

// ------------------------------------------------------------------------------------------
// forward, vector phase: tp_fwd_body of e3k_tp.hip over the tile's slice of a node's in-edges, weights from LDS
// ------------------------------------------------------------------------------------------
template <int L1, int L3MAX>
__device__ __forceinline__ void rtp_fwd_cg(const RtpArgs& a, const e3k_rtp_chunk& c, const TileLds& L, const TileNodes& tn,
                                           int lane) {
  using S = Slots<L1>;
  constexpr int D1 = 2 * L1 + 1;
#ifndef RTP_PF2
#define RTP_PF2 2     // measured (layer 3, 256 molecules): batches of 2 edges without a second register set = 3 waves per SIMD,
#endif                // 457 us; batches of 3 with one = 2 waves per SIMD, 520 us
  constexpr int PF = L1 <= 1 ? RTP_PF2 + 1 : (L1 == 2 ? RTP_PF2 : RTP_PF2 - 1);   // edges per batch; the next batch's x rows load under this one's math
  const int mul = c.mul;
  const unsigned mask = c.mask;
  const unsigned xcol = (unsigned)(c.x_off + c.cchunk * 64 + lane);   // lane offset inside a row of x: 32-bit, the row base is scalar
  const int n_items = tn.end - tn.first;
  // which of the chunk's two paths a slot is (compile-time Q, run-time mask)
  auto pos_of = [&](int q) { return (mask & ((1u << q) - 1u)) ? 1 : 0; };
  for (;;) {
    int it = 0;
    if (lane == 0) it = atomicAdd(L.next, 1);      // nodes are claimed one at a time: degrees differ
    it = uniform(it);
    if (it >= n_items) break;
    const int node = tn.first + it;
    const int pb = uniform(L.ptr[it]), pe = uniform(L.ptr[it + 1]);
    const int beg = pb > tn.t0 ? pb : tn.t0, end = pe < tn.t1 ? pe : tn.t1;
    const bool whole = pb >= tn.t0 && pe <= tn.t1;
    if (!whole && beg >= end) continue;
    float acc[S::TOTAL];
#pragma unroll
    for (int i = 0; i < S::TOTAL; ++i) acc[i] = 0.0f;
    if (beg < end) {
      float xc[PF][D1], xn[PF][D1];
      auto issue = [&](int tb, float (&xb)[PF][D1]) {
        int sv[PF];
#pragma unroll
        for (int p = 0; p < PF; ++p) sv[p] = L.nbr[(tb + p < end ? tb + p : end - 1) - tn.t0];
#pragma unroll
        for (int p = 0; p < PF; ++p) {
          const float* __restrict__ xr = a.x + (int64_t)uniform(sv[p]) * a.d_in;
#pragma unroll
          for (int i = 0; i < D1; ++i) xb[p][i] = xr[xcol + (unsigned)(i * mul)];
        }
      };
#ifndef RTP_DB
#define RTP_DB 0      // 1: the next batch's x rows are requested before this batch's math (a second register set)
#endif
      if (RTP_DB) issue(beg, xc);
      for (int tb = beg; tb < end; tb += PF) {
        const bool more = RTP_DB && tb + PF < end;
        if (!RTP_DB) issue(tb, xc);
        if (more) issue(tb + PF, xn);
        // every LDS operand of the batch is read up front, at fixed offsets: one wait instead of one per use
        YRegs yc[PF];
        float wv0[PF], wv1[PF];
#pragma unroll
        for (int p = 0; p < PF; ++p) {
          const bool valid = tb + p < end;       // the tail of the last batch repeats its last edge with weight zero
          const int te = (valid ? tb + p : end - 1) - tn.t0;
          const float4* yr = reinterpret_cast<const float4*>(L.sh + te * RT_SH);
          const float4 ya = yr[0], yb = yr[1];
          yc[p].y0[0] = ya.x; yc[p].y1[0] = ya.y; yc[p].y1[1] = ya.z; yc[p].y1[2] = ya.w;
          yc[p].y2[0] = yb.x; yc[p].y2[1] = yb.y; yc[p].y2[2] = yb.z; yc[p].y2[3] = yb.w;
          yc[p].y2[4] = L.sh[te * RT_SH + 8];
          const float* wrow = L.w + te * RT_LD + lane;
          const float w0 = wrow[0], w1 = wrow[RT_TE * RT_LD];
          wv0[p] = valid ? w0 * c.coeff[0] : 0.0f;
          wv1[p] = (valid && c.np == 2) ? w1 * c.coeff[1] : 0.0f;
        }
        slot_for_part<S, L1, L3MAX, 2>([&](auto qc) {
          constexpr int Q = decltype(qc)::value;
          constexpr int L2 = S::L2[Q], L3 = S::L3[Q], OFF = S::OFF[Q];
          if (mask & (1u << Q)) {
            const int pos = pos_of(Q);
#pragma unroll
            for (int p = 0; p < PF; ++p) {
              float tt[2 * L3 + 1];
              CG<L1, L2, L3>::xy(xc[p], yref<L2>(yc[p]), tt);
              const float wq = pos ? wv1[p] : wv0[p];
#pragma unroll
              for (int k = 0; k < 2 * L3 + 1; ++k) acc[OFF + k] = fmaf(wq, tt[k], acc[OFF + k]);
            }
          }
        });
        if (more) {
#pragma unroll
          for (int p = 0; p < PF; ++p)
#pragma unroll
            for (int i = 0; i < D1; ++i) xc[p][i] = xn[p][i];
        }
      }
    }
    float* __restrict__ orow = a.out + (int64_t)node * a.d_mid + c.cchunk * 64 + lane;
    slot_for_part<S, L1, L3MAX, 2>([&](auto qc) {
      constexpr int Q = decltype(qc)::value;
      constexpr int L3 = S::L3[Q], OFF = S::OFF[Q];
      if (mask & (1u << Q)) {
        const int pos = pos_of(Q);
        float* dst = orow + (pos ? c.out_off[1] : c.out_off[0]);
        const int stride = pos ? c.out_stride[1] : c.out_stride[0];
#pragma unroll
        for (int k = 0; k < 2 * L3 + 1; ++k) {
          if (whole) dst[k * stride] = acc[OFF + k];
          else atomicAdd(dst + k * stride, acc[OFF + k]);
        }
      }
    });
  }
}

#define E3K_RTP_CASE(L, BODY, ...)                                        \
  if constexpr (MAXL >= L) {                                              \
    if (l1 == L) BODY<L, __VA_ARGS__>(a, c, L_, tn, lane);                \
  }

template <int MAXL, int L3MAX>
__global__ __launch_bounds__(256) void rtp_fwd_kernel(const RtpArgs a, const e3k_tp_group* __restrict__ groups,
                                                      const e3k_rtp_chunk* __restrict__ chunks) {
  __shared__ float w_s[2 * RT_TE * RT_LD];
  __shared__ __attribute__((aligned(16))) float sh_s[RT_TE * RT_SH];
  __shared__ int32_t nbr_s[RT_TE], eid_s[RT_TE], ptr_s[RT_TE + 2], next_s[1];
  const TileLds L_{w_s, sh_s, nbr_s, eid_s, ptr_s, next_s};
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  RTP_STAMP(s_begin);
  float af[32];
  rtp_load_a(a, tile, wave & 1, lane, af);
  stage_tile(a, tile, L_);
  const TileNodes tn = tile_nodes(a, tile);
  stage_ptr(a, tn.first, tn.end, L_);
#ifdef E3K_RTP_STAMPS
  unsigned long long acc_m = 0, acc_w1 = 0, acc_v = 0, acc_w2 = 0;
  unsigned long long s_prev = __builtin_amdgcn_s_memtime();
  const unsigned long long s_pro = s_prev - s_begin;
#endif
  for (int ci = 0; ci < a.n_chunks; ++ci) {
    const e3k_rtp_chunk c = chunks[ci];       // by value: scalar registers
    if (threadIdx.x == 0) next_s[0] = 0;      // (every wave left the previous vector phase at the barrier below)
    rtp_w_chunk(a, c, af, w_s, wave, lane);
    RTP_STAMP(s_a);
    lds_barrier();
    RTP_STAMP(s_b);
    const int l1 = c.l1;
    E3K_RTP_CASE(0, rtp_fwd_cg, L3MAX)
    E3K_RTP_CASE(1, rtp_fwd_cg, L3MAX)
    E3K_RTP_CASE(2, rtp_fwd_cg, L3MAX)
    E3K_RTP_CASE(3, rtp_fwd_cg, L3MAX)
    RTP_STAMP(s_c);
    lds_barrier();
#ifdef E3K_RTP_STAMPS
    const unsigned long long s_d = __builtin_amdgcn_s_memtime();
    acc_m += s_a - s_prev; acc_w1 += s_b - s_a; acc_v += s_c - s_b; acc_w2 += s_d - s_c;
    s_prev = s_d;
#endif
  }
#ifdef E3K_RTP_STAMPS
  if (a.stamps && lane == 0) {
    unsigned long long* o = a.stamps + ((int64_t)tile * 4 + wave) * 5;
    o[0] = s_pro; o[1] = acc_m; o[2] = acc_w1; o[3] = acc_v; o[4] = acc_w2;
  }
#endif
}

// rows that receive atomics (the node that contains a tile's first position, when it started in an earlier tile)
__global__ __launch_bounds__(256) void rtp_zero_rows_kernel(float* __restrict__ out, const int32_t* __restrict__ ptr,
                                                            const int32_t* __restrict__ own0, int n_tiles, int width) {
  const int tile = blockIdx.x;
  if (tile >= n_tiles) return;
  const int n_lo = own0[tile];
  if (ptr[n_lo] <= tile * RT_TE) return;
  float* row = out + (int64_t)(n_lo - 1) * width;
  for (int i = threadIdx.x; i < width; i += 256) row[i] = 0.0f;
}

}  // namespace e3k

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" int e3k_rtp_supported(const e3k_tp_plan* plan) {
  return (plan && plan->d_chunks && plan->n_chunks > 0) ? 1 : 0;
}

extern "C" int e3k_rtp_tile_edges(void) { return e3k::RT_TE; }

#ifdef E3K_RTP_STAMPS
static void* e3k_rtp_debug_buffer = nullptr;   // diagnostic build (make dbg): device buffer the next launches stamp into
extern "C" void e3k_rtp_set_debug_buffer(void* p) { e3k_rtp_debug_buffer = p; }
#endif

namespace {
int rtp_fill(e3k::RtpArgs& a, const e3k_tp_plan* plan, const float* h, const float* wl, int32_t k, float w_scale,
             const int32_t* nbr, const int32_t* ptr, const int32_t* perm, const int32_t* own0, int64_t N, int64_t E) {
  if (!plan || N < 0 || E < 0) return E3K_ERR_INVALID;
  if (!e3k_rtp_supported(plan) || k != e3k::RT_K) return E3K_ERR_UNSUPPORTED;
  if (E > 0x7fffffffLL - e3k::RT_TE || N > 0x7fffffffLL) return E3K_ERR_INVALID;
  if (E > 0 && (!h || !wl || !nbr || !ptr || !perm || !own0)) return E3K_ERR_INVALID;
  a.h = h; a.wl = wl; a.w_scale = w_scale; a.nbr = nbr; a.ptr = ptr; a.perm = perm; a.own0 = own0;
  a.d_in = plan->d_in; a.d_sh = plan->d_sh; a.W = plan->w_numel; a.d_mid = plan->d_mid;
  a.E = (int32_t)E;
  a.n_tiles = (int32_t)((E + e3k::RT_TE - 1) / e3k::RT_TE);
  a.n_chunks = plan->n_chunks;
  for (int l = 0; l < 3; ++l) a.y_off[l] = plan->y_off[l];
  return E3K_OK;
}
}  // namespace

extern "C" int e3k_rtp_fwd(const e3k_tp_plan* plan, const float* h, const float* wl, int32_t k, float w_scale,
                           const float* x, const float* sh, const int32_t* src, const int32_t* dst_ptr,
                           const int32_t* dst_perm, const int32_t* dst_own0, int64_t N, int64_t E, float* out,
                           void* stream) {
  e3k::RtpArgs a{};
  const int rc = rtp_fill(a, plan, h, wl, k, w_scale, src, dst_ptr, dst_perm, dst_own0, N, E);
  if (rc != E3K_OK) return rc;
  if (N == 0) return E3K_OK;
  if (!out || (E > 0 && (!x || !sh))) return E3K_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  if (E == 0) {
    if (hipMemsetAsync(out, 0, sizeof(float) * (size_t)N * plan->d_mid, st) != hipSuccess) return E3K_ERR_LAUNCH;
    return E3K_OK;
  }
  a.x = x; a.sh = sh; a.out = out;
#ifdef E3K_RTP_STAMPS
  a.stamps = (unsigned long long*)e3k_rtp_debug_buffer;
#endif
  hipLaunchKernelGGL(e3k::rtp_zero_rows_kernel, dim3(a.n_tiles), dim3(256), 0, st, out, dst_ptr, dst_own0, a.n_tiles, plan->d_mid);
  const dim3 grid(a.n_tiles), block(256);
  const bool low = plan->max_l3 <= plan->max_l1;
#define E3K_RTP_LAUNCH(ML, L3) hipLaunchKernelGGL((e3k::rtp_fwd_kernel<ML, L3>), grid, block, 0, st, a, plan->d_groups, plan->d_chunks)
  switch (plan->max_l1) {
    case 0: if (low) E3K_RTP_LAUNCH(0, 0); else E3K_RTP_LAUNCH(0, 3); break;
    case 1: if (low) E3K_RTP_LAUNCH(1, 1); else E3K_RTP_LAUNCH(1, 3); break;
    case 2: if (low) E3K_RTP_LAUNCH(2, 2); else E3K_RTP_LAUNCH(2, 3); break;
    default: E3K_RTP_LAUNCH(3, 3); break;
  }
#undef E3K_RTP_LAUNCH
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}
