// Fused 'uvu' Clebsch-Gordan tensor product + destination reduce for gfx950 (MI355X).
//
// Replaces, per convolution layer of the reference:
//   x[edge_src] gather            e3_layers/nn/message_passing.py:96,105   (never materialised here)
//   o3.TensorProduct 'uvu'        e3_layers/nn/pointwise.py:78-85,94-98   (per-edge external weights)
//   scatter over edge_dst         e3_layers/nn/message_passing.py:109     (no atomics here)
//
// Execution shape (one wave = one work item, no LDS, no atomics):
//   work item = (node n, group g, 64-channel chunk c); lane = channel u.
//   A group is one input irrep block (degree L1, template parameter) with up to NQ(L1) paths,
//   at most one per (l2, l3) slot.  The wave walks the in-edges of n (CSR by destination,
//   edge ids ascending => same summation order as the reference's CPU index_add_), keeps the
//   whole group output  sum_q (2 l3_q + 1)  in registers at compile-time slot offsets and
//   writes each output row exactly once, as 256-byte rows in the channel-fastest layout.
//   Per edge it reads: 2 L1 + 1 coalesced 256-B rows of x[src] (L2-resident: a graph's rows
//   stay on one XCD thanks to xcd_remap), one 256-B row of w per path (the HBM stream: every
//   weight element is read exactly once per pass) and <= 9 wave-uniform floats of sh[e]
//   (scalar loads).  Wigner-3j coefficients are instruction literals (e3k_cg_gen.h).
//
// Backward: bwd_w has the forward's shape (per destination node; writes grad_w[e] rows,
// optionally reduces grad_sh[e] over lanes), bwd_x walks the out-edges of a source node
// (CSR by source) and accumulates grad_x in registers.
#include <cstdlib>

#include "e3k_tp_body.h"

// the per-edge weights are read once per pass and the weight gradients written once: streamed past the caches
// (the cache lines stay for the gathered node rows)
#ifdef E3K_NO_NT
#define E3K_STREAM_LOAD(p) (*(p))
#define E3K_STREAM_STORE(v, p) (*(p) = (v))
#define E3K_STREAM_AUX 0
#else
#define E3K_STREAM_AUX 2   // buffer-load cache policy bit 1 = nt
#define E3K_STREAM_LOAD(p) __builtin_nontemporal_load(p)
#define E3K_STREAM_STORE(v, p) __builtin_nontemporal_store((v), (p))
#endif

namespace e3k {

// FULL: every group of the plan has a multiple of 64 channels and every (l2, l3) slot its degrees allow (all output
// parities present: the inner layers of the shipped models), so no lane is ever idle and no slot is ever skipped: the
// `active` selects, the exec-masked branches around the loads and the per-slot mask branches compile away, and an
// edge's loads and arithmetic are one straight-line block.  Row addresses are formed as (wave-uniform pointer)[lane channel]:
// the uniform part stays on the scalar unit and the loads take the SGPR-base + 32-bit-lane-offset form (no 64-bit vector
// add per load); sh[e] is read first so that waiting for it does not mean waiting for the rows issued after it.
// ---- FULL instantiations: buffer addressing ---------------------------------------------------------------------------
// A row of an edge is addressed as (descriptor of the edge's row block: wave-uniform base, rebuilt per edge with two scalar
// adds) + (ONE loop-invariant SGPR: the row's byte offset inside the block) + (one loop-invariant VGPR: 4 * lane channel).
// The flat form `(uniform pointer)[u]` costs an SGPR pair per row, all live at once because every row of an edge is
// requested before the first is consumed: tp_bwd_x (22-31 gradient rows + 6-8 weight rows per edge) ran out of scalar
// registers -- 37-74 spilled to VGPR lanes and read back with v_readlane inside the edge loop -- and the k > 0 rows were
// formed with 64-bit VECTOR adds.  sh[e] is nine unconditional scalar loads (FULL plans keep the degrees 0, 1, 2 at columns
// 0, 1, 4 of a 9-wide row: checked when the plan is built), no branch on y_off inside the loop.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(const float* base, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float buf_ld(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ float buf_ld_stream(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, E3K_STREAM_AUX));
}
__device__ __forceinline__ void buf_st_stream(float v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, E3K_STREAM_AUX);
}
__device__ __forceinline__ void load_y_full(YRegs& y, const float* __restrict__ yr) {
  y.y0[0] = yr[0];
#pragma unroll
  for (int j = 0; j < 3; ++j) y.y1[j] = yr[1 + j];
#pragma unroll
  for (int j = 0; j < 5; ++j) y.y2[j] = yr[4 + j];
}

// MODE 1 (table): the edge's path weights are not read from w[e] but interpolated here from the four rows of the radial knot
// table around the edge's radius with the edge's weights coef[e, 0..3] (same products, same order as
// rtable_interp_fwd_kernel): w[E, W] is never written or read; the table (2-4 MB a layer) is served by L2.
// MODE 2 (table, second-order forms of force training): a second operand set (x2, sh2, s2) rides along and the kernel forms
// the sum of the three terms in which exactly one of (x, sh, w) is replaced by its partner -- x2, sh2, and s2[e] * dw/dr[e]
// with dw/dr interpolated from the slope table D (a.w2) by the same four weights: the derivative of the trilinear product
// along (x2, sh2, s2 in r) -- in one walk over the edges instead of three.
struct KnotRows {
  __amdgpu_buffer_rsrc_t ra, rb, rc, rd;
  float c0, c1, c2, c3;
};
struct KnotRows2 {      // the same four rows of the slope table
  __amdgpu_buffer_rsrc_t ra, rb, rc, rd;
};
__device__ __forceinline__ KnotRows2 knot_rows2(const TpArgs& a, int e, int row_w) {
  const int i = uniform(a.bin[e]);
  const float* __restrict__ base = a.w2 + (int64_t)(i - 1) * a.W;
  KnotRows2 k;
  k.ra = row_rsrc(base, row_w);
  k.rb = row_rsrc(base + a.W, row_w);
  k.rc = row_rsrc(base + 2 * a.W, row_w);
  k.rd = row_rsrc(base + 3 * a.W, row_w);
  return k;
}
__device__ __forceinline__ float sload(const float* __restrict__ p) { return __uint_as_float(uniform((int)__float_as_uint(*p))); }
__device__ __forceinline__ KnotRows knot_rows(const TpArgs& a, const float* __restrict__ coef, int e, int row_w) {
  const int i = uniform(a.bin[e]);
  const float* __restrict__ base = a.w + (int64_t)(i - 1) * a.W;
  const float* __restrict__ cp = coef + 4 * (int64_t)e;
  KnotRows k;
  k.ra = row_rsrc(base, row_w);
  k.rb = row_rsrc(base + a.W, row_w);
  k.rc = row_rsrc(base + 2 * a.W, row_w);
  k.rd = row_rsrc(base + 3 * a.W, row_w);
  k.c0 = sload(cp); k.c1 = sload(cp + 1); k.c2 = sload(cp + 2); k.c3 = sload(cp + 3);
  return k;
}
__device__ __forceinline__ float knot_mix(float c0, float c1, float c2, float c3, float va, float vb, float vc, float vd) {
  return fmaf(c3, vd, fmaf(c2, vc, fmaf(c1, vb, c0 * va)));
}

#ifdef E3K_DEBUG_KNOBS
static int g_tp_ablate_host = 0;      // timing-only ablation mask of the packed-table forward (tools/tp_table_bench.py --ablate): 1 no table
                                      // loads, 2 no x row loads, 4 no CG arithmetic -- wrong results by design; travels in TpArgs.ablate
#endif
// MODE 4 (packed table, round 5): the same cubic from 12 bytes per (knot, weight) in ONE table row -- {d0, d1: f32; d2 * 2^10,
// d3 * 2^16: f16}, the Taylor coefficients about the middle of the knot interval (e3k_rtable_pack; a row = its W (d0, d1) pairs, then
// its W f16 pairs) -- one dwordx2 + one dword load per slot instead of four dword loads out of four rows: 23 instead of 31 KB of
// table per edge through L1, half the table's VMEM instructions.  a.w is then the packed table [K + 1, 3 W] (dwords).
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
struct PackedRec {
  float d0, d1;
  unsigned int pk;
};
struct KnotPacked {
  __amdgpu_buffer_rsrc_t r;
  float s, s2, s3;
};
// the t-th edge of the walk as one aligned 64-byte record (e3k_edge_records): a single s_load_dwordx16 whose address follows from
// the loop counter -- the packed-table kernels fetch edge t + 1's record while edge t is computed
struct EdgeRec {
  int nbr, bin;
  float c0, c1, c2, c3;
  float y[9];
  int e;
};
__device__ __forceinline__ EdgeRec load_rec(const int32_t* __restrict__ erec, int t) {
  const int32_t* __restrict__ p = static_cast<const int32_t*>(__builtin_assume_aligned(erec + 16 * (int64_t)t, 64));
  EdgeRec r;
  r.nbr = uniform(p[0]);
  r.bin = uniform(p[1]);
  r.c0 = __int_as_float(uniform(p[2])); r.c1 = __int_as_float(uniform(p[3]));
  r.c2 = __int_as_float(uniform(p[4])); r.c3 = __int_as_float(uniform(p[5]));
#pragma unroll
  for (int j = 0; j < 9; ++j) r.y[j] = __int_as_float(uniform(p[6 + j]));
  r.e = uniform(p[15]);
  return r;
}
__device__ __forceinline__ void rec_y(YRegs& y, const EdgeRec& r) {
  y.y0[0] = r.y[0];
#pragma unroll
  for (int j = 0; j < 3; ++j) y.y1[j] = r.y[1 + j];
#pragma unroll
  for (int j = 0; j < 5; ++j) y.y2[j] = r.y[4 + j];
}
__device__ __forceinline__ KnotPacked knot_packed_rec(const TpArgs& a, const EdgeRec& r, int row_p) {
  KnotPacked k;
  k.r = row_rsrc(a.w + 3 * (int64_t)r.bin * a.W, row_p);
  k.s = fmaf(2.f, r.c3, r.c2 - r.c0) - 0.5f;      // t = sum_k x_k L_k(t), x = (-1, 0, 1, 2); s = t - 1/2
  k.s2 = k.s * (1.f / 1024.f);
  k.s3 = k.s * (1.f / 64.f);
  return k;
}
// slot at weight offset woff4 (bytes of a 4-byte column) of the row, lane channel u4 = 4 u; pk_base = 8 W (bytes)
__device__ __forceinline__ PackedRec buf_ld_rec(__amdgpu_buffer_rsrc_t r, int u4, int woff4, int pk_base, int ablate = 0) {
#ifdef E3K_DEBUG_KNOBS
  if (ablate & 1) {      // timing only: no table loads
    PackedRec p;
    p.d0 = __int_as_float(u4 + woff4); p.d1 = 1.f; p.pk = (unsigned)pk_base;
    return p;
  }
#endif
  const auto v = __builtin_amdgcn_raw_buffer_load_b64(r, u4 * 2, woff4 * 2, 0);
  PackedRec p;
  p.d0 = __uint_as_float(v[0]);
  p.d1 = __uint_as_float(v[1]);
  p.pk = __builtin_amdgcn_raw_buffer_load_b32(r, u4, pk_base + woff4, 0);
  return p;
}
__device__ __forceinline__ float packed_mix(const KnotPacked& k, const PackedRec& v) {      // the order of e3k::packed_eval (e3k_rtable.hip)
  const f16x2 h = __builtin_bit_cast(f16x2, v.pk);
  return fmaf(k.s, fmaf(k.s2, fmaf(k.s3, (float)h.y, (float)h.x), v.d1), v.d0);
}

template <int L1, int L3MAX, int MODE, int PART>
__device__ __forceinline__ void tp_fwd_body_full(const TpArgs& a, const e3k_tp_group& g, const int node, const int u) {
  using S = Slots<L1>;
  constexpr int D1 = 2 * L1 + 1;
  constexpr bool TABLE = MODE == 1 || MODE == 2, JVP = MODE == 2 || MODE == 3;      // MODE 3: the JVP form on STREAMED w[e], dw/dr[e] rows
  constexpr bool PACKED = MODE == 4;
  const int u4 = u * 4;
  const int xoff4 = uniform(g.x_off * 4), mul4 = uniform(g.mul * 4);
  int woff4[S::NQ];
  float cf[S::NQ];
  slot_for_part<S, L1, L3MAX, PART>([&](auto qc) {
    constexpr int Q = decltype(qc)::value;
    woff4[Q] = uniform(g.w_off[Q] * 4);
    cf[Q] = g.coeff[Q];
  });
  const int row_x = a.d_in * 4, row_w = a.W * 4;
  float acc[S::TOTAL];
#pragma unroll
  for (int i = 0; i < S::TOTAL; ++i) acc[i] = 0.0f;
  const int beg = uniform(a.ptr[node]), end = uniform(a.ptr[node + 1]);
  EdgeRec cur{};
  if constexpr (PACKED) {
    if (beg < end) cur = load_rec(a.erec, beg);
  }
  for (int t = beg; t < end; ++t) {
    int e, s;
    YRegs yc, y2;
    EdgeRec nxt{};
    if constexpr (PACKED) {      // this edge's record arrived during the previous edge; the next one's is requested now
      nxt = load_rec(a.erec, t + 1 < end ? t + 1 : t);
      e = cur.e;
      s = cur.nbr;
      rec_y(yc, cur);
    } else {
      e = uniform(a.perm[t]);
      s = uniform(a.nbr[e]);
      load_y_full(yc, a.sh + (int64_t)e * a.d_sh);
    }
    if constexpr (JVP) load_y_full(y2, a.sh2 + (int64_t)e * a.d_sh);
    const __amdgpu_buffer_rsrc_t rx = row_rsrc(a.x + (int64_t)s * a.d_in, row_x);
    float xc[D1], x2c[JVP ? D1 : 1], wc[S::NQ], w2[JVP ? S::NQ : 1];
#ifdef E3K_DEBUG_KNOBS
    if (PACKED && (a.ablate & 2)) {
#pragma unroll
      for (int i = 0; i < D1; ++i) xc[i] = __int_as_float(s + i);
    } else
#endif
#pragma unroll
    for (int i = 0; i < D1; ++i) xc[i] = buf_ld(rx, u4, xoff4 + i * mul4);
    if constexpr (JVP) {
      const __amdgpu_buffer_rsrc_t rx2 = row_rsrc(a.x2 + (int64_t)s * a.d_in, row_x);
#pragma unroll
      for (int i = 0; i < D1; ++i) x2c[i] = buf_ld(rx2, u4, xoff4 + i * mul4);
    }
    if constexpr (PACKED) {
      const KnotPacked kp = knot_packed_rec(a, cur, row_w * 3);
      PackedRec rec[S::NQ];
      slot_for_part<S, L1, L3MAX, PART>([&](auto qc) {
        constexpr int Q = decltype(qc)::value;
        rec[Q] = buf_ld_rec(kp.r, u4, woff4[Q], row_w * 2, a.ablate);
      });
      slot_for_part<S, L1, L3MAX, PART>([&](auto qc) {
        constexpr int Q = decltype(qc)::value;
        wc[Q] = packed_mix(kp, rec[Q]);
      });
    } else if constexpr (TABLE) {
      const KnotRows kr = knot_rows(a, a.coef, e, row_w);
      float wa[S::NQ], wb[S::NQ], wcc[S::NQ], wd[S::NQ];
      slot_for_part<S, L1, L3MAX, PART>([&](auto qc) {
        constexpr int Q = decltype(qc)::value;
        wa[Q] = buf_ld(kr.ra, u4, woff4[Q]);
        wb[Q] = buf_ld(kr.rb, u4, woff4[Q]);
        wcc[Q] = buf_ld(kr.rc, u4, woff4[Q]);
        wd[Q] = buf_ld(kr.rd, u4, woff4[Q]);
      });
      slot_for_part<S, L1, L3MAX, PART>([&](auto qc) {
        constexpr int Q = decltype(qc)::value;
        wc[Q] = knot_mix(kr.c0, kr.c1, kr.c2, kr.c3, wa[Q], wb[Q], wcc[Q], wd[Q]);
      });
      if constexpr (JVP) {
        const KnotRows2 k2 = knot_rows2(a, e, row_w);
        const float sc = sload(a.s2 + e);
        slot_for_part<S, L1, L3MAX, PART>([&](auto qc) {
          constexpr int Q = decltype(qc)::value;
          w2[Q] = sc * knot_mix(kr.c0, kr.c1, kr.c2, kr.c3, buf_ld(k2.ra, u4, woff4[Q]), buf_ld(k2.rb, u4, woff4[Q]),
                                buf_ld(k2.rc, u4, woff4[Q]), buf_ld(k2.rd, u4, woff4[Q]));
        });
      }
    } else {
      const __amdgpu_buffer_rsrc_t rw = row_rsrc(a.w + (int64_t)e * a.W, row_w);
      slot_for_part<S, L1, L3MAX, PART>([&](auto qc) {
        constexpr int Q = decltype(qc)::value;
        wc[Q] = JVP ? buf_ld(rw, u4, woff4[Q]) : buf_ld_stream(rw, u4, woff4[Q]);      // (second-order: the rows are read by several kernels)
      });
      if constexpr (JVP) {
        const __amdgpu_buffer_rsrc_t rd = row_rsrc(a.w2 + (int64_t)e * a.W, row_w);
        const float sc = sload(a.s2 + e);
        slot_for_part<S, L1, L3MAX, PART>([&](auto qc) {
          constexpr int Q = decltype(qc)::value;
          w2[Q] = sc * buf_ld(rd, u4, woff4[Q]);
        });
      }
    }
    slot_for_part<S, L1, L3MAX, PART>([&](auto qc) {
      constexpr int Q = decltype(qc)::value;
      constexpr int L2 = S::L2[Q], L3 = S::L3[Q], OFF = S::OFF[Q];
      const float wv = wc[Q] * cf[Q];
      float tt[2 * L3 + 1];
#ifdef E3K_DEBUG_KNOBS
      if (PACKED && (a.ablate & 4)) {
#pragma unroll
        for (int k = 0; k < 2 * L3 + 1; ++k) tt[k] = xc[k % D1] + yc.y0[0];
      } else
#endif
      CG<L1, L2, L3>::xy(xc, yref<L2>(yc), tt);
      if constexpr (JVP) {
        // d/d(eps) [ w(r + eps s2) * xy(x + eps x2, y + eps y2) ] at eps = 0
        const float wv2 = w2[Q] * cf[Q];
        float ta[2 * L3 + 1], tb[2 * L3 + 1];
        CG<L1, L2, L3>::xy(x2c, yref<L2>(yc), ta);
        CG<L1, L2, L3>::xy(xc, yref<L2>(y2), tb);
#pragma unroll
        for (int k = 0; k < 2 * L3 + 1; ++k) acc[OFF + k] = fmaf(wv, ta[k] + tb[k], fmaf(wv2, tt[k], acc[OFF + k]));
      } else {
#pragma unroll
        for (int k = 0; k < 2 * L3 + 1; ++k) acc[OFF + k] = fmaf(wv, tt[k], acc[OFF + k]);
      }
    });
    if constexpr (PACKED) cur = nxt;
  }
  float* __restrict__ orow = a.out + (int64_t)node * a.d_mid;
  slot_for_part<S, L1, L3MAX, PART>([&](auto qc) {
    constexpr int Q = decltype(qc)::value;
    constexpr int L3 = S::L3[Q], OFF = S::OFF[Q];
#pragma unroll
    for (int k = 0; k < 2 * L3 + 1; ++k) (orow + g.out_off[Q] + k * g.out_stride[Q])[u] = acc[OFF + k];
  });
}

// HALF (plans that are not channel-complete, groups of at most 32 channels -- the 32-channel score nets): lanes 32-63 of the
// wave would idle behind `u < mul`; instead the two lane halves walk the node's edge list two edges at a time (half h takes
// edges t + h), everything that was wave-uniform per edge -- edge id, source row, sh, the weight row -- becomes a per-lane
// value (two distinct addresses per wave instruction, 128 contiguous bytes each), and the halves' partial sums are added
// across lanes l and l ^ 32 once per node.
template <int L1, int L3MAX, int MODE, int PART, bool FULL, bool HALF = false>
__device__ __forceinline__ void tp_fwd_body(const TpArgs& a, const e3k_tp_group& g, const int node, const int u) {
  static_assert(FULL || MODE == 0, "the table forms exist for FULL plans only");
  if constexpr (FULL) {
    tp_fwd_body_full<L1, L3MAX, MODE, PART>(a, g, node, u);
    return;
  }
  using S = Slots<L1>;
  constexpr int D1 = 2 * L1 + 1;
  const int mul = g.mul;
  if constexpr (!FULL && !HALF) {
    if (mul <= 32) {
      tp_fwd_body<L1, L3MAX, MODE, PART, FULL, true>(a, g, node, u);
      return;
    }
  }
  const int lane = threadIdx.x & 63;
  const int h = HALF ? lane >> 5 : 0;
  const int uu = HALF ? (lane & 31) : u;
  const bool lane_on = FULL || uu < mul;
  const unsigned mask = g.mask;

  float acc[S::TOTAL];
#pragma unroll
  for (int i = 0; i < S::TOTAL; ++i) acc[i] = 0.0f;

  const int beg = uniform(a.ptr[node]), end = uniform(a.ptr[node + 1]);
  // no software pipeline (loads of edge t+1 before edge t is consumed): it costs a second register set and
  // occupancy buys more here -- measured 178 vs 186 us (l_max 2), 432 vs 438 us (l_max 3)
  for (int t = beg; t < end; t += HALF ? 2 : 1) {
    int e, s;
    bool active = lane_on;
    if constexpr (HALF) {
      const bool ev = t + h < end;
      e = a.perm[ev ? t + h : t];
      s = a.nbr[e];
      active = lane_on && ev;
    } else {
      e = uniform(a.perm[t]);
      s = uniform(a.nbr[e]);
    }
    YRegs yc;
    load_y(yc, a.sh + (int64_t)e * a.d_sh, g);
    const float* __restrict__ xr = a.x + (int64_t)s * a.d_in + g.x_off;      // wave-uniform (per lane half under HALF)
    const float* __restrict__ wr = a.w + (int64_t)e * a.W;
    float xc[D1], wc[S::NQ];
#pragma unroll
    for (int i = 0; i < D1; ++i) xc[i] = active ? (xr + i * mul)[uu] : 0.0f;
    slot_for_part<S, L1, L3MAX, PART>([&](auto qc) {
      constexpr int Q = decltype(qc)::value;
      if (FULL || (mask & (1u << Q))) wc[Q] = active ? E3K_STREAM_LOAD((wr + g.w_off[Q]) + uu) : 0.0f;
    });
    slot_for_part<S, L1, L3MAX, PART>([&](auto qc) {
      constexpr int Q = decltype(qc)::value;
      constexpr int L2 = S::L2[Q], L3 = S::L3[Q], OFF = S::OFF[Q];
      if (FULL || (mask & (1u << Q))) {
        const float wv = wc[Q] * g.coeff[Q];
        float tt[2 * L3 + 1];
        CG<L1, L2, L3>::xy(xc, yref<L2>(yc), tt);
#pragma unroll
        for (int k = 0; k < 2 * L3 + 1; ++k) acc[OFF + k] = fmaf(wv, tt[k], acc[OFF + k]);
      }
    });
  }
  if constexpr (HALF) {
    slot_for_part<S, L1, L3MAX, PART>([&](auto qc) {
      constexpr int Q = decltype(qc)::value;
      constexpr int L3 = S::L3[Q], OFF = S::OFF[Q];
      if (mask & (1u << Q)) {
#pragma unroll
        for (int k = 0; k < 2 * L3 + 1; ++k) acc[OFF + k] += __shfl_xor(acc[OFF + k], 32, 64);
      }
    });
  }
  if (lane_on && h == 0) {
    float* __restrict__ orow = a.out + (int64_t)node * a.d_mid;
    slot_for_part<S, L1, L3MAX, PART>([&](auto qc) {
      constexpr int Q = decltype(qc)::value;
      constexpr int L3 = S::L3[Q], OFF = S::OFF[Q];
      if (FULL || (mask & (1u << Q))) {
#pragma unroll
        for (int k = 0; k < 2 * L3 + 1; ++k) (orow + g.out_off[Q] + k * g.out_stride[Q])[uu] = acc[OFF + k];
      }
    });
  }
}

// ------------------------------------------------------------------------------------------
// nine (sh) or four (coef) per-lane partials -> wave totals added to a row of g_sh / g_coef by ONE wave instruction.
// A butterfly that halves the value set at every level (each lane keeps the half its partner discards) needs
// 5+3+2+1+1+1 = 13 cross-lane moves for nine values instead of 9 x 6, and leaves value `idx` on the lanes whose upper bits
// spell idx.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_add9(const float (&v9)[9], float* __restrict__ row, const int32_t* y_off, const bool store = false) {
  const int lane = threadIdx.x & 63;
  int idx = 0;
  const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8, b2 = lane & 4;
  float l5[5], l3[3], l2[2];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const float hi = i + 5 < 9 ? v9[i + 5] : 0.0f;
    const float recv = __shfl_xor(b5 ? v9[i] : hi, 32, 64);
    l5[i] = (b5 ? hi : v9[i]) + recv;
  }
  idx += b5 ? 5 : 0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float hi = i + 3 < 5 ? l5[i + 3] : 0.0f;
    const float recv = __shfl_xor(b4 ? l5[i] : hi, 16, 64);
    l3[i] = (b4 ? hi : l5[i]) + recv;
  }
  idx += b4 ? 3 : 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float hi = i + 2 < 3 ? l3[i + 2] : 0.0f;
    const float recv = __shfl_xor(b3 ? l3[i] : hi, 8, 64);
    l2[i] = (b3 ? hi : l3[i]) + recv;
  }
  idx += b3 ? 2 : 0;
  float tot = (b2 ? l2[1] : l2[0]) + __shfl_xor(b2 ? l2[0] : l2[1], 4, 64);
  idx += b2 ? 1 : 0;
  tot += __shfl_xor(tot, 2, 64);
  tot += __shfl_xor(tot, 1, 64);
  // which of the nine values this lane holds: the halving tree over (5|4) -> (3|2) -> (2|1) -> (1|1);
  // paths that step into padding carry zeros and are skipped
  // valid index sets: b5=0: counts 5 -> b4=0: 3 -> b3=0: 2 -> b2: 1|1 ; b3=1: 1 -> b2=0 only
  //                              b4=1: 2 -> b3=0: 2 -> b2: 1|1 ; b3=1: 0 (padding)
  //                   b5=1: counts 4 -> b4=0: 3 -> b3=0: 2 -> b2: 1|1 ; b3=1: 1 -> b2=0 only
  //                              b4=1: 1 -> b3=0: 1 -> b2=0 only ; b3=1: padding
  bool valid;
  if (!b4) valid = !b3 || !b2;
  else if (!b5) valid = !b3;
  else valid = !b3 && !b2;
  if (valid && (lane & 3) == 0) {
    int off = -1;
    if (idx == 0) off = y_off[0];
    else if (idx < 4) off = y_off[1] >= 0 ? y_off[1] + (idx - 1) : -1;
    else off = y_off[2] >= 0 ? y_off[2] + (idx - 4) : -1;
    if (off >= 0) {
      if (store) row[off] = tot;      // (e_store: this work item's own slice -- no other wave writes it)
      else atomicAdd(row + off, tot);
    }
  }
}
// ------------------------------------------------------------------------------------------
// backward wrt the per-edge weights (and optionally the spherical harmonics)
// ------------------------------------------------------------------------------------------
// DUAL (FULL plans, no g_sh): the weight gradient of the product's derivative along (x2, sh2) --
//   g_w[e] = coeff * <g_mid[dst], xy(x2[src], sh) + xy(x[src], sh2)> -- one of the second-order terms of force training.
template <int L1, bool WITH_SH, int L3MAX, int PART, bool FULL, bool DUAL = false, bool HALF = false>
__device__ __forceinline__ void tp_bwd_w_body(const TpArgs& a, const e3k_tp_group& g, const int node, const int u_) {
  static_assert(!DUAL || (FULL && !WITH_SH), "the dual form is built for channel-complete plans, weight gradient only");
  static_assert(!HALF || (!FULL && !WITH_SH), "two edges per wave: weight gradient only (g_sh is a reduction over the whole wave)");
  // (the buffer-addressed form of tp_fwd / tp_bwd_x was tried here too: 71 instead of 60 VGPRs, 7 instead of 8 waves per SIMD,
  //  the step 3-5 % slower -- this kernel keeps the flat row pointers; it has no scalar-register spills to begin with)
  using S = Slots<L1>;
  constexpr int D1 = 2 * L1 + 1;
  const int mul = g.mul;
  if constexpr (!FULL && !WITH_SH && !HALF) {
    if (mul <= 32) {      // two edges per wave, see tp_fwd_body (every edge writes its own g_w row: nothing to fold)
      tp_bwd_w_body<L1, WITH_SH, L3MAX, PART, FULL, DUAL, true>(a, g, node, u_);
      return;
    }
  }
  const int lane = threadIdx.x & 63;
  const int h = HALF ? lane >> 5 : 0;
  const int u = HALF ? (lane & 31) : u_;
  const bool lane_on = FULL || u < mul;
  bool active = lane_on;
  const unsigned mask = g.mask;

  // incoming gradient of this node's group outputs, resident for the whole edge walk
  float go[S::TOTAL];
  {
    const float* __restrict__ grow = a.g_out + (int64_t)node * a.d_mid;      // wave-uniform
    slot_for_part<S, L1, L3MAX, PART>([&](auto qc) {
      constexpr int Q = decltype(qc)::value;
      constexpr int L3 = S::L3[Q], OFF = S::OFF[Q];
#pragma unroll
      for (int k = 0; k < 2 * L3 + 1; ++k)
        go[OFF + k] = ((FULL || (mask & (1u << Q))) && active) ? (grow + g.out_off[Q] + k * g.out_stride[Q])[u] : 0.0f;
    });
  }
  const int beg = uniform(a.ptr[node]), end = uniform(a.ptr[node + 1]);
  // (walking edge records here -- as tp_fwd / tp_bwd_x of the packed table do -- was built and measured in round 5: 108.0 vs 107.4 us
  //  isolated, 149.5 vs 150.1 inside the step: this kernel is bound by the 0.5 GB of g_w it writes, not by its scalar chain)
  for (int t = beg; t < end; t += HALF ? 2 : 1) {
    int e, s;
    if constexpr (HALF) {
      const bool ev = t + h < end;
      e = a.perm[ev ? t + h : t];
      s = a.nbr[e];
      active = lane_on && ev;
    } else {
      e = uniform(a.perm[t]);
      s = uniform(a.nbr[e]);
    }
    YRegs yc, y2;
    load_y(yc, a.sh + (int64_t)e * a.d_sh, g);
    const float* __restrict__ xr = a.x + (int64_t)s * a.d_in + g.x_off;      // wave-uniform (per lane half under HALF)
    float xc[D1], x2c[DUAL ? D1 : 1];
#pragma unroll
    for (int i = 0; i < D1; ++i) xc[i] = active ? (xr + i * mul)[u] : 0.0f;
    if constexpr (DUAL) {
      load_y_full(y2, a.sh2 + (int64_t)e * a.d_sh);
      const float* __restrict__ x2r = a.x2 + (int64_t)s * a.d_in + g.x_off;
#pragma unroll
      for (int i = 0; i < D1; ++i) x2c[i] = (x2r + i * mul)[u];
    }
    float* __restrict__ gwr = a.g_w + (int64_t)e * a.W;
    const float* __restrict__ wr = a.w + (int64_t)e * a.W;
    YRegs gy;
    if constexpr (WITH_SH) {
      gy.y0[0] = 0.0f;
#pragma unroll
      for (int j = 0; j < 3; ++j) gy.y1[j] = 0.0f;
#pragma unroll
      for (int j = 0; j < 5; ++j) gy.y2[j] = 0.0f;
    }
    slot_for_part<S, L1, L3MAX, PART>([&](auto qc) {
      constexpr int Q = decltype(qc)::value;
      constexpr int L2 = S::L2[Q], L3 = S::L3[Q], OFF = S::OFF[Q];
      if (FULL || (mask & (1u << Q))) {
        float tt[2 * L3 + 1], gk[2 * L3 + 1];
        if constexpr (DUAL) {
          float tb[2 * L3 + 1];
          CG<L1, L2, L3>::xy(x2c, yref<L2>(yc), tt);
          CG<L1, L2, L3>::xy(xc, yref<L2>(y2), tb);
#pragma unroll
          for (int k = 0; k < 2 * L3 + 1; ++k) tt[k] += tb[k];
        } else {
          CG<L1, L2, L3>::xy(xc, yref<L2>(yc), tt);
        }
        float dot = 0.0f;
#pragma unroll
        for (int k = 0; k < 2 * L3 + 1; ++k) {
          gk[k] = go[OFF + k];
          dot = fmaf(gk[k], tt[k], dot);
        }
        if (active && a.g_w) E3K_STREAM_STORE(dot * g.coeff[Q], (gwr + g.w_off[Q]) + u);
        if constexpr (WITH_SH) {
          const float wv = active ? (wr + g.w_off[Q])[u] * g.coeff[Q] : 0.0f;
          CG<L1, L2, L3>::xg(xc, gk, wv, yref<L2>(gy));
        }
      }
    });
    if constexpr (WITH_SH) {
      float v9[9];
      v9[0] = gy.y0[0];
#pragma unroll
      for (int j = 0; j < 3; ++j) v9[1 + j] = gy.y1[j];
#pragma unroll
      for (int j = 0; j < 5; ++j) v9[4 + j] = gy.y2[j];
      wave_add9(v9, a.g_sh + (int64_t)e * a.d_sh, g.y_off);
    }
  }
}

// ------------------------------------------------------------------------------------------
// backward wrt the edge quantities with the path weights on the knot table (the FIRST backward of a force evaluation:
// d E / d pos flows through the spherical harmonics and through the radius, i.e. the interpolation weights):
//   g_sh[e, :] += d/d sh <g_mid[dst], TP(x[src], sh, w[e])>                 (nine values per edge), w[e] = sum_k coef[e,k] T[..]
//   g_r[e]     += sum_c g_w[e, c] dw/dr[e, c], dw/dr[e] = sum_k coef[e,k] D[..] (one value per edge; g_w stays in registers)
// and, when asked (g_w != NULL), the per-edge weight gradient itself.  Same walk as tp_bwd_w (per destination node, the node's
// gradient rows resident).  D is the slope of the table on the knots (the radial MLP's forward-mode derivative, evaluated
// where rounding cannot hurt it: differentiating the fp32 table itself amplifies its rounding by 1 / knot spacing).
// ------------------------------------------------------------------------------------------
template <int L1, int L3MAX, int PART, bool STREAM>
__device__ __forceinline__ void tp_bwd_e_body_full(const TpArgs& a, const e3k_tp_group& g, const int node, const int u) {
  using S = Slots<L1>;
  constexpr int D1 = 2 * L1 + 1;
  const int u4 = u * 4;
  const int xoff4 = uniform(g.x_off * 4), mul4 = uniform(g.mul * 4);
  int woff4[S::NQ];
  float cf[S::NQ];
  float go[S::TOTAL];
  {
    const float* __restrict__ grow = a.g_out + (int64_t)node * a.d_mid;      // wave-uniform
    slot_for_part<S, L1, L3MAX, PART>([&](auto qc) {
      constexpr int Q = decltype(qc)::value;
      constexpr int L3 = S::L3[Q], OFF = S::OFF[Q];
      woff4[Q] = uniform(g.w_off[Q] * 4);
      cf[Q] = g.coeff[Q];
#pragma unroll
      for (int k = 0; k < 2 * L3 + 1; ++k) go[OFF + k] = (grow + g.out_off[Q] + k * g.out_stride[Q])[u];
    });
  }
  const int row_x = a.d_in * 4, row_w = a.W * 4;
  const int y_off[3] = {0, 1, 4};
  const int beg = uniform(a.ptr[node]), end = uniform(a.ptr[node + 1]);
  for (int t = beg; t < end; ++t) {
    const int e = uniform(a.perm[t]);
    const int s = uniform(a.nbr[e]);
    YRegs yc;
    load_y_full(yc, a.sh + (int64_t)e * a.d_sh);
    const __amdgpu_buffer_rsrc_t rx = row_rsrc(a.x + (int64_t)s * a.d_in, row_x);
    float xc[D1];
#pragma unroll
    for (int i = 0; i < D1; ++i) xc[i] = buf_ld(rx, u4, xoff4 + i * mul4);
    float wv_[S::NQ], dv_[S::NQ];
    if constexpr (STREAM) {      // w[e], dw/dr[e] materialised (a.w, a.w2: [E, W])
      const __amdgpu_buffer_rsrc_t rw = row_rsrc(a.w + (int64_t)e * a.W, row_w), rd = row_rsrc(a.w2 + (int64_t)e * a.W, row_w);
      slot_for_part<S, L1, L3MAX, PART>([&](auto qc) {
        constexpr int Q = decltype(qc)::value;
        wv_[Q] = buf_ld(rw, u4, woff4[Q]);
        dv_[Q] = buf_ld(rd, u4, woff4[Q]);
      });
    } else {
      const KnotRows kr = knot_rows(a, a.coef, e, row_w);
      const KnotRows2 k2 = knot_rows2(a, e, row_w);
      slot_for_part<S, L1, L3MAX, PART>([&](auto qc) {
        constexpr int Q = decltype(qc)::value;
        wv_[Q] = knot_mix(kr.c0, kr.c1, kr.c2, kr.c3, buf_ld(kr.ra, u4, woff4[Q]), buf_ld(kr.rb, u4, woff4[Q]),
                          buf_ld(kr.rc, u4, woff4[Q]), buf_ld(kr.rd, u4, woff4[Q]));
        dv_[Q] = knot_mix(kr.c0, kr.c1, kr.c2, kr.c3, buf_ld(k2.ra, u4, woff4[Q]), buf_ld(k2.rb, u4, woff4[Q]),
                          buf_ld(k2.rc, u4, woff4[Q]), buf_ld(k2.rd, u4, woff4[Q]));
      });
    }
    YRegs gy;
    gy.y0[0] = 0.0f;
#pragma unroll
    for (int j = 0; j < 3; ++j) gy.y1[j] = 0.0f;
#pragma unroll
    for (int j = 0; j < 5; ++j) gy.y2[j] = 0.0f;
    float gr = 0.f;
    float* __restrict__ gwr = a.g_w ? a.g_w + (int64_t)e * a.W : nullptr;
    slot_for_part<S, L1, L3MAX, PART>([&](auto qc) {
      constexpr int Q = decltype(qc)::value;
      constexpr int L2 = S::L2[Q], L3 = S::L3[Q], OFF = S::OFF[Q];
      float tt[2 * L3 + 1], gk[2 * L3 + 1];
      CG<L1, L2, L3>::xy(xc, yref<L2>(yc), tt);
      float dot = 0.0f;
#pragma unroll
      for (int k = 0; k < 2 * L3 + 1; ++k) {
        gk[k] = go[OFF + k];
        dot = fmaf(gk[k], tt[k], dot);
      }
      const float gw = dot * cf[Q];
      if (gwr) E3K_STREAM_STORE(gw, (gwr + g.w_off[Q]) + u);
      gr = fmaf(gw, dv_[Q], gr);
      CG<L1, L2, L3>::xg(xc, gk, wv_[Q] * cf[Q], yref<L2>(gy));
    });
    if (a.g_sh) {
      float v9[9];
      v9[0] = gy.y0[0];
#pragma unroll
      for (int j = 0; j < 3; ++j) v9[1 + j] = gy.y1[j];
#pragma unroll
      for (int j = 0; j < 5; ++j) v9[4 + j] = gy.y2[j];
      wave_add9(v9, a.g_sh + (int64_t)e * a.d_sh, y_off, a.e_store != 0);
    }
    if (a.g_r) {
      gr = wave_sum(gr);
      if ((threadIdx.x & 63) == 0) {
        if (a.e_store) a.g_r[e] = gr;
        else atomicAdd(a.g_r + e, gr);
      }
    }
  }
}

template <int L1, int L3MAX, bool STREAM, int PART, bool FULL>
__device__ __forceinline__ void tp_bwd_e_body(const TpArgs& a, const e3k_tp_group& g, const int node, const int u) {
  static_assert(FULL, "the edge backward of force training exists for channel-complete plans only");
  tp_bwd_e_body_full<L1, L3MAX, PART, STREAM>(a, g, node, u);
}

// ------------------------------------------------------------------------------------------
// backward wrt the node features: walk the out-edges of a source node
// ------------------------------------------------------------------------------------------
// MODE 0: w streamed; 1: w interpolated from the knot table; 2 (table, DUAL): g_x = yg(sh2, g, w) + yg(sh, g, s2 * dw/dr) --
// the node-feature gradient of the product's derivative along (sh2, s2 in r) (second-order term of force training)
template <int L1, int L3MAX, int MODE, int PART>
__device__ __forceinline__ void tp_bwd_x_body_full(const TpArgs& a, const e3k_tp_group& g, const int node, const int u) {
  using S = Slots<L1>;
  constexpr int D1 = 2 * L1 + 1;
  constexpr bool TABLE = MODE == 1 || MODE == 2, DUAL = MODE == 2 || MODE == 3 || MODE == 7;     // MODE 3: the DUAL form on STREAMED w[e], dw/dr[e] rows
  // MODE 7: MODE 3 + the weight gradients that share its sums -- the DUAL one, g_w[e] = <x2, t(sh)> + <x, t(sh2)> (what tp_bwd_w_dual
  // forms in a walk of its own), and optionally the plain one, g_w2[e] = <x, t(sh)> (tp_bwd_w): the u-sweep of force training wants all three
  constexpr bool GW2 = MODE == 7;
  constexpr bool PACKED = MODE == 4 || MODE == 5;                                   // MODE 4: w from the packed table (see tp_fwd_body_full)
  // MODE 5: ... and the WEIGHT gradient of every edge beside it.  Both gradients contract the same sums t[m1] = sum CG sh[m2] g[m3]
  // of an edge and a path: g_x[src] += w t, g_w[e] = <x[src], t> -- with the source's own rows resident (2 l1 + 1 registers) the
  // second costs 2 l1 + 1 FMAs and one 256-byte store per path, against a whole second walk of the edges (tp_bwd_w_kernel: the
  // sh, x[src] and g[dst] gathers again) -- and the 0.5 GB of g_w [E, W] leave through HBM while this kernel waits on its L2 gathers
  constexpr bool GW = MODE == 5 || MODE == 6;      // (MODE 6: the same with the weights STREAMED from w[E, W] -- force training's rows)
  // MODE 8 (streamed rows w, dw/dr [E, W]; force training's first backward): MODE 6 + the EDGE gradients of tp_bwd_e in the same walk --
  // g_sh[e] += xg(x[src], g[dst], w) (nine values: butterfly + one atomic per wave), g_r[e] += <g_w[e], dw/dr[e]> (one value) -- every
  // operand of both is already in this walk's registers; g_w is stored only when asked for
  constexpr bool GE = MODE == 8;
  const int mul = g.mul;
  const int u4 = u * 4;
  int goff4[S::NQ], gstr4[S::NQ], woff4[S::NQ];
  float cf[S::NQ];
  slot_for_part<S, L1, L3MAX, PART>([&](auto qc) {
    constexpr int Q = decltype(qc)::value;
    goff4[Q] = uniform(g.out_off[Q] * 4);
    gstr4[Q] = uniform(g.out_stride[Q] * 4);
    woff4[Q] = uniform(g.w_off[Q] * 4);
    cf[Q] = g.coeff[Q];
  });
  const int row_g = a.d_mid * 4, row_w = a.W * 4;
  float gx[D1], xs[(GW || GW2 || GE) ? D1 : 1], xs2[GW2 ? D1 : 1];
#pragma unroll
  for (int i = 0; i < D1; ++i) gx[i] = 0.0f;
  if constexpr (GW || GW2 || GE) {
    const float* __restrict__ xr = a.x + (int64_t)node * a.d_in + g.x_off;      // wave-uniform
#pragma unroll
    for (int i = 0; i < D1; ++i) xs[i] = (xr + i * mul)[u];
  }
  if constexpr (GW2) {
    const float* __restrict__ xr2 = a.x2 + (int64_t)node * a.d_in + g.x_off;
#pragma unroll
    for (int i = 0; i < D1; ++i) xs2[i] = (xr2 + i * mul)[u];
  }
  const int beg = uniform(a.ptr[node]), end = uniform(a.ptr[node + 1]);
  EdgeRec cur{};
  if constexpr (PACKED) {
    if (beg < end) cur = load_rec(a.erec, beg);
  }
  for (int t = beg; t < end; ++t) {
    int e, d;
    YRegs yc, y2;
    EdgeRec nxt{};
    if constexpr (PACKED) {      // (records in the order of the source-CSR walk: nbr = the edge's destination)
      nxt = load_rec(a.erec, t + 1 < end ? t + 1 : t);
      e = cur.e;
      d = cur.nbr;
#ifdef E3K_DEBUG_KNOBS
      if (a.ablate & 8) d = node;      // timing only: every edge reads the walker's own gradient rows (an L1 hit) instead of g[dst]
#endif
      rec_y(yc, cur);
    } else {
      e = uniform(a.perm[t]);
      d = uniform(a.nbr[e]);
      load_y_full(yc, a.sh + (int64_t)e * a.d_sh);
    }
    if constexpr (DUAL) load_y_full(y2, a.sh2 + (int64_t)e * a.d_sh);
    const __amdgpu_buffer_rsrc_t rg = row_rsrc(a.g_out + (int64_t)d * a.d_mid, row_g);
    const bool gw_out = GW || GW2 || (GE && a.g_w != nullptr);
    const __amdgpu_buffer_rsrc_t rgw = row_rsrc(gw_out ? a.g_w + (int64_t)e * a.W : nullptr, gw_out ? row_w : 0);
    const bool plain2 = GW2 && a.g_w2 != nullptr;
    const __amdgpu_buffer_rsrc_t rgw2 = row_rsrc(plain2 ? a.g_w2 + (int64_t)e * a.W : nullptr, plain2 ? row_w : 0);
    float gn[S::TOTAL], wn[S::NQ], w2[DUAL ? S::NQ : 1], dvv[GE ? S::NQ : 1];
    YRegs gy;
    float gr = 0.f;
    if constexpr (GE) {
      gy.y0[0] = 0.0f;
#pragma unroll
      for (int j = 0; j < 3; ++j) gy.y1[j] = 0.0f;
#pragma unroll
      for (int j = 0; j < 5; ++j) gy.y2[j] = 0.0f;
    }
    if constexpr (PACKED) {
      const KnotPacked kp = knot_packed_rec(a, cur, row_w * 3);
      PackedRec rec[S::NQ];
      slot_for_part<S, L1, L3MAX, PART>([&](auto qc) {
        constexpr int Q = decltype(qc)::value;
        constexpr int L3 = S::L3[Q], OFF = S::OFF[Q];
#pragma unroll
        for (int k = 0; k < 2 * L3 + 1; ++k)
          gn[OFF + k] = buf_ld(rg, u4, goff4[Q] + k * gstr4[Q]);
        rec[Q] = buf_ld_rec(kp.r, u4, woff4[Q], row_w * 2, a.ablate);
      });
      __builtin_amdgcn_sched_barrier(0);
      slot_for_part<S, L1, L3MAX, PART>([&](auto qc) {
        constexpr int Q = decltype(qc)::value;
        wn[Q] = packed_mix(kp, rec[Q]);
      });
    } else if constexpr (TABLE) {
      const KnotRows kr = knot_rows(a, a.coef, e, row_w);
      float wa[S::NQ], wb[S::NQ], wcc[S::NQ], wd[S::NQ];
      slot_for_part<S, L1, L3MAX, PART>([&](auto qc) {
        constexpr int Q = decltype(qc)::value;
        constexpr int L3 = S::L3[Q], OFF = S::OFF[Q];
#pragma unroll
        for (int k = 0; k < 2 * L3 + 1; ++k)
          gn[OFF + k] = buf_ld(rg, u4, goff4[Q] + k * gstr4[Q]);
        wa[Q] = buf_ld(kr.ra, u4, woff4[Q]);
        wb[Q] = buf_ld(kr.rb, u4, woff4[Q]);
        wcc[Q] = buf_ld(kr.rc, u4, woff4[Q]);
        wd[Q] = buf_ld(kr.rd, u4, woff4[Q]);
      });
      __builtin_amdgcn_sched_barrier(0);
      slot_for_part<S, L1, L3MAX, PART>([&](auto qc) {
        constexpr int Q = decltype(qc)::value;
        wn[Q] = knot_mix(kr.c0, kr.c1, kr.c2, kr.c3, wa[Q], wb[Q], wcc[Q], wd[Q]);
      });
      if constexpr (DUAL) {
        const KnotRows2 k2 = knot_rows2(a, e, row_w);
        const float sc = sload(a.s2 + e);
        slot_for_part<S, L1, L3MAX, PART>([&](auto qc) {
          constexpr int Q = decltype(qc)::value;
          w2[Q] = sc * knot_mix(kr.c0, kr.c1, kr.c2, kr.c3, buf_ld(k2.ra, u4, woff4[Q]), buf_ld(k2.rb, u4, woff4[Q]),
                                buf_ld(k2.rc, u4, woff4[Q]), buf_ld(k2.rd, u4, woff4[Q]));
        });
      }
    } else {
      const __amdgpu_buffer_rsrc_t rw = row_rsrc(a.w + (int64_t)e * a.W, row_w);
      slot_for_part<S, L1, L3MAX, PART>([&](auto qc) {
        constexpr int Q = decltype(qc)::value;
        constexpr int L3 = S::L3[Q], OFF = S::OFF[Q];
#pragma unroll
        for (int k = 0; k < 2 * L3 + 1; ++k)
          gn[OFF + k] = buf_ld(rg, u4, goff4[Q] + k * gstr4[Q]);
        wn[Q] = DUAL ? buf_ld(rw, u4, woff4[Q]) : buf_ld_stream(rw, u4, woff4[Q]);
      });
      if constexpr (DUAL) {
        const __amdgpu_buffer_rsrc_t rd = row_rsrc(a.w2 + (int64_t)e * a.W, row_w);
        const float sc = sload(a.s2 + e);
        slot_for_part<S, L1, L3MAX, PART>([&](auto qc) {
          constexpr int Q = decltype(qc)::value;
          w2[Q] = sc * buf_ld(rd, u4, woff4[Q]);
        });
      }
      if constexpr (GE) {
        const __amdgpu_buffer_rsrc_t rd = row_rsrc(a.w2 + (int64_t)e * a.W, row_w);
        slot_for_part<S, L1, L3MAX, PART>([&](auto qc) {
          constexpr int Q = decltype(qc)::value;
          dvv[Q] = buf_ld(rd, u4, woff4[Q]);
        });
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    slot_for_part<S, L1, L3MAX, PART>([&](auto qc) {
      constexpr int Q = decltype(qc)::value;
      constexpr int L2 = S::L2[Q], L3 = S::L3[Q], OFF = S::OFF[Q];
      float gk[2 * L3 + 1];
#pragma unroll
      for (int k = 0; k < 2 * L3 + 1; ++k) gk[k] = gn[OFF + k];
      if constexpr (GE) {
        float tq[D1];
        CG<L1, L2, L3>::yt(yref<L2>(yc), gk, tq);
        const float sw = wn[Q] * cf[Q];
        float dot = 0.0f;
#pragma unroll
        for (int i = 0; i < D1; ++i) {
          gx[i] = fmaf(sw, tq[i], gx[i]);
          dot = fmaf(xs[i], tq[i], dot);
        }
        const float gwv = dot * cf[Q];
        if (gw_out) buf_st_stream(gwv, rgw, u4, woff4[Q]);
        gr = fmaf(gwv, dvv[Q], gr);
        CG<L1, L2, L3>::xg(xs, gk, sw, yref<L2>(gy));
      } else if constexpr (GW2) {
        float ta[D1], tb[D1];
        CG<L1, L2, L3>::yt(yref<L2>(y2), gk, ta);
        CG<L1, L2, L3>::yt(yref<L2>(yc), gk, tb);
        const float sa = wn[Q] * cf[Q], sb = w2[Q] * cf[Q];
        float dd = 0.0f, dp = 0.0f;
#pragma unroll
        for (int i = 0; i < D1; ++i) gx[i] = fmaf(sa, ta[i], gx[i]);      // (the order and the bits of the two CG::yg calls of MODE 3)
#pragma unroll
        for (int i = 0; i < D1; ++i) {
          gx[i] = fmaf(sb, tb[i], gx[i]);
          dd = fmaf(xs2[i], tb[i], dd);
          dd = fmaf(xs[i], ta[i], dd);
          dp = fmaf(xs[i], tb[i], dp);
        }
        buf_st_stream(dd * cf[Q], rgw, u4, woff4[Q]);
        if (plain2) buf_st_stream(dp * cf[Q], rgw2, u4, woff4[Q]);
      } else if constexpr (DUAL) {
        CG<L1, L2, L3>::yg(yref<L2>(y2), gk, wn[Q] * cf[Q], gx);
        CG<L1, L2, L3>::yg(yref<L2>(yc), gk, w2[Q] * cf[Q], gx);
      } else if constexpr (GW) {
        float tq[D1];
        CG<L1, L2, L3>::yt(yref<L2>(yc), gk, tq);
        const float sw = wn[Q] * cf[Q];
        float dot = 0.0f;
#pragma unroll
        for (int i = 0; i < D1; ++i) {
          gx[i] = fmaf(sw, tq[i], gx[i]);      // (the bits of CG::yg: MODE 4 and MODE 5 give the same g_x)
          dot = fmaf(xs[i], tq[i], dot);
        }
        buf_st_stream(dot * cf[Q], rgw, u4, woff4[Q]);
      } else {
        CG<L1, L2, L3>::yg(yref<L2>(yc), gk, wn[Q] * cf[Q], gx);
      }
    });
    if constexpr (GE) {
      const int y_off[3] = {0, 1, 4};
      if (a.g_sh) {
        float v9[9];
        v9[0] = gy.y0[0];
#pragma unroll
        for (int j = 0; j < 3; ++j) v9[1 + j] = gy.y1[j];
#pragma unroll
        for (int j = 0; j < 5; ++j) v9[4 + j] = gy.y2[j];
        wave_add9(v9, a.g_sh + (int64_t)e * a.d_sh, y_off, a.e_store != 0);
      }
      if (a.g_r) {
        gr = wave_sum(gr);
        if ((threadIdx.x & 63) == 0) {
          if (a.e_store) a.g_r[e] = gr;
          else atomicAdd(a.g_r + e, gr);
        }
      }
    }
    if constexpr (PACKED) cur = nxt;
  }
  float* __restrict__ gxr = a.g_x + (int64_t)node * a.d_in + g.x_off;
#pragma unroll
  for (int i = 0; i < D1; ++i) {
    if (PART == 2 && !a.x_shared) (gxr + i * mul)[u] = gx[i];
    else atomicAdd(gxr + i * mul + u, gx[i]);
  }
}

template <int L1, int L3MAX, int MODE, int PART, bool FULL, bool HALF = false>
__device__ __forceinline__ void tp_bwd_x_body(const TpArgs& a, const e3k_tp_group& g, const int node, const int u) {
  static_assert(FULL || MODE == 0 || MODE == 6, "the table forms exist for FULL plans only");
  if constexpr (FULL) {
    tp_bwd_x_body_full<L1, L3MAX, MODE, PART>(a, g, node, u);
    return;
  }
  constexpr bool GW = MODE == 6;      // ... with the per-edge weight gradient in the same walk (see tp_bwd_x_body_full)
  using S = Slots<L1>;
  constexpr int D1 = 2 * L1 + 1;
  const int mul = g.mul;
  if constexpr (!FULL && !HALF) {
    if (mul <= 32) {      // two edges per wave, see tp_fwd_body
      tp_bwd_x_body<L1, L3MAX, MODE, PART, FULL, true>(a, g, node, u);
      return;
    }
  }
  const int lane = threadIdx.x & 63;
  const int h = HALF ? lane >> 5 : 0;
  const int uu = HALF ? (lane & 31) : u;
  const bool lane_on = FULL || uu < mul;
  const unsigned mask = g.mask;

  float gx[D1], xs[GW ? D1 : 1];
#pragma unroll
  for (int i = 0; i < D1; ++i) gx[i] = 0.0f;
  if constexpr (GW) {
    const float* __restrict__ xr = a.x + (int64_t)node * a.d_in + g.x_off;      // wave-uniform: both lane halves walk edges of this node
#pragma unroll
    for (int i = 0; i < D1; ++i) xs[i] = lane_on ? (xr + i * mul)[uu] : 0.0f;
  }
  const int beg = uniform(a.ptr[node]), end = uniform(a.ptr[node + 1]);
  // (a one-deep software pipeline measured the same: 495 vs 498 us at l_max 2)
  for (int t = beg; t < end; t += HALF ? 2 : 1) {
    int e, d;
    bool active = lane_on;
    if constexpr (HALF) {
      const bool ev = t + h < end;
      e = a.perm[ev ? t + h : t];
      d = a.nbr[e];
      active = lane_on && ev;
    } else {
      e = uniform(a.perm[t]);
      d = uniform(a.nbr[e]);
    }
    YRegs yc;
    load_y(yc, a.sh + (int64_t)e * a.d_sh, g);
    const float* __restrict__ wr = a.w + (int64_t)e * a.W;                 // wave-uniform (per lane half under HALF)
    const float* __restrict__ grow = a.g_out + (int64_t)d * a.d_mid;
    // every row of the edge is requested before the first one is used (registers are not what limits these kernels'
    // occupancy; a load consumed right behind its issue leaves one row in flight per wave)
    float gn[S::TOTAL], wn[S::NQ];
    slot_for_part<S, L1, L3MAX, PART>([&](auto qc) {
      constexpr int Q = decltype(qc)::value;
      constexpr int L3 = S::L3[Q], OFF = S::OFF[Q];
      if (FULL || (mask & (1u << Q))) {
#pragma unroll
        for (int k = 0; k < 2 * L3 + 1; ++k) gn[OFF + k] = active ? (grow + g.out_off[Q] + k * g.out_stride[Q])[uu] : 0.0f;
        wn[Q] = active ? E3K_STREAM_LOAD((wr + g.w_off[Q]) + uu) : 0.0f;
      }
    });
    __builtin_amdgcn_sched_barrier(0);
    slot_for_part<S, L1, L3MAX, PART>([&](auto qc) {
      constexpr int Q = decltype(qc)::value;
      constexpr int L2 = S::L2[Q], L3 = S::L3[Q], OFF = S::OFF[Q];
      if (FULL || (mask & (1u << Q))) {
        float gk[2 * L3 + 1];
#pragma unroll
        for (int k = 0; k < 2 * L3 + 1; ++k) gk[k] = gn[OFF + k];
        if constexpr (GW) {
          float tq[D1];
          CG<L1, L2, L3>::yt(yref<L2>(yc), gk, tq);
          const float sw = wn[Q] * g.coeff[Q];
          float dot = 0.0f;
#pragma unroll
          for (int i = 0; i < D1; ++i) {
            gx[i] = fmaf(sw, tq[i], gx[i]);
            dot = fmaf(xs[i], tq[i], dot);
          }
          if (active) E3K_STREAM_STORE(dot * g.coeff[Q], (a.g_w + (int64_t)e * a.W + g.w_off[Q]) + uu);
        } else {
          CG<L1, L2, L3>::yg(yref<L2>(yc), gk, wn[Q] * g.coeff[Q], gx);
        }
      }
    });
  }
  if constexpr (HALF) {
#pragma unroll
    for (int i = 0; i < D1; ++i) gx[i] += __shfl_xor(gx[i], 32, 64);
  }
  if (lane_on && h == 0) {
    float* __restrict__ gxr = a.g_x + (int64_t)node * a.d_in + g.x_off;
#pragma unroll
    for (int i = 0; i < D1; ++i) {
      if (PART == 2 && !a.x_shared) (gxr + i * mul)[uu] = gx[i];
      else atomicAdd(gxr + i * mul + uu, gx[i]);   // two waves per group (a + b is order independent) or several groups on
                                                   // one input block (repeated sh degree): g_x is pre-zeroed
    }
  }
}

// ------------------------------------------------------------------------------------------
// one launch per pass: a wave looks up its work item and branches (wave-uniformly) on the input degree
// ------------------------------------------------------------------------------------------
// Work order.  0: node-major -- XCD x owns a contiguous slice of the (node, group-chunk) list, a workgroup = four consecutive items
// (mostly the groups of ONE node).  1 (table forms): group-major inside the XCD's node slice -- XCD x owns nodes [x N / 8, (x + 1) N / 8)
// and walks them group-chunk by group-chunk, a workgroup = four consecutive NODES of one group-chunk: at any time an XCD's L2 then
// holds the table columns of ONE group (2.4 of the packed table's 11.8 MB) beside that group's feature columns of its own nodes.
#define E3K_TP_PROLOGUE                                                        \
  int node_, gci_;                                                             \
  if (a.order == 0) {                                                          \
    const int b = xcd_remap(blockIdx.x, gridDim.x);                            \
    const int64_t item = (int64_t)b * 4 + (threadIdx.x >> 6);                  \
    if (item >= a.n_items) return;                                             \
    node_ = (int)(item / n_gc);                                                \
    gci_ = (int)(item % n_gc);                                                 \
  } else {                                                                     \
    const int64_t n_nodes = a.n_items / n_gc;                                  \
    const int xcd = blockIdx.x & 7;                                            \
    const int64_t n0 = xcd * n_nodes / 8, nx = (xcd + 1) * n_nodes / 8 - n0;   \
    const int64_t j = (int64_t)(blockIdx.x >> 3) * 4 + (threadIdx.x >> 6);     \
    if (j >= nx * n_gc) return;                                                \
    gci_ = (int)(j / nx);                                                      \
    node_ = (int)(n0 + j % nx);                                                \
  }                                                                            \
  const int node = uniform(node_);                                             \
  const int gci = uniform(gci_);                                               \
  const int2 gcv = gc[gci];                                                    \
  const e3k_tp_group& g = groups[uniform(gcv.x)];                              \
  const int part = uniform(gcv.y) >> 16;                                        \
  const int u = (uniform(gcv.y) & 0xffff) * 64 + (threadIdx.x & 63);

// MAXL = largest input degree of the plan: the register allocation of a kernel is the maximum over the branches of
// the degree switch, so an l_max = 2 model must not carry the l1 = 3 body (63 accumulators) it never runs.
#define E3K_TP_CASE(L, BODY, ...)                                                                    \
  if constexpr (MAXL >= L) {                                                                         \
    if (l1 == L) {                                                                                   \
      if constexpr (!SPLIT) {                                                                        \
        BODY<L, __VA_ARGS__, 2, FULL>(a, g, node, u);                                                \
      } else {                                                                                       \
        if (part == 0) BODY<L, __VA_ARGS__, 0, FULL>(a, g, node, u);                                 \
        else BODY<L, __VA_ARGS__, 1, FULL>(a, g, node, u);                                           \
      }                                                                                              \
    }                                                                                                \
  }
// SPLIT kernels (l_max = 3 plans) only contain the half-group bodies: the full bodies would set the register count
#define E3K_TP_DISPATCH(BODY, ...)                                                                   \
  const int l1 = g.l1;                                                                               \
  (void)part;                                                                                        \
  if (l1 == 0) BODY<0, __VA_ARGS__, 2, FULL>(a, g, node, u);                                         \
  E3K_TP_CASE(1, BODY, __VA_ARGS__)                                                                  \
  E3K_TP_CASE(2, BODY, __VA_ARGS__)                                                                  \
  E3K_TP_CASE(3, BODY, __VA_ARGS__)

template <int L1, int L3MAX, int PART, bool FULL>
__device__ __forceinline__ void tp_bwd_w_dual_body(const TpArgs& a, const e3k_tp_group& g, const int node, const int u) {
  tp_bwd_w_body<L1, false, L3MAX, PART, FULL, true>(a, g, node, u);
}

// MODE: 0 = per-edge weights streamed from w[E, W]; 1 = interpolated from the knot table inside the kernel; 2 = the table
// form's second-order (JVP / DUAL) variant -- FULL plans only for 1 and 2
// (packed form, l_max <= 2 plans: 67 VGPRs as the compiler first allocates them -- one step over the 64 of eight waves per SIMD; asked
//  for eight, it fits without spilling)
#ifdef E3K_DEBUG_KNOBS
#define E3K_TP_FWD_WAVES(MODE, MAXL, L3MAX) 1      // (the debug build's ablation branches do not fit 64 registers)
#else
#define E3K_TP_FWD_WAVES(MODE, MAXL, L3MAX) ((MODE == 4 && MAXL <= 2 && L3MAX <= 2) ? 8 : 1)
#endif
template <int MAXL, int L3MAX, bool SPLIT, bool FULL, int MODE = 0>
__global__ __launch_bounds__(256, E3K_TP_FWD_WAVES(MODE, MAXL, L3MAX)) void tp_fwd_kernel(TpArgs a, const e3k_tp_group* __restrict__ groups,
                                                     const int2* __restrict__ gc, int n_gc) {
  E3K_TP_PROLOGUE
  E3K_TP_DISPATCH(tp_fwd_body, L3MAX, MODE)
}

template <bool WITH_SH, int MAXL, int L3MAX, bool SPLIT, bool FULL>
__global__ __launch_bounds__(256) void tp_bwd_w_kernel(TpArgs a, const e3k_tp_group* __restrict__ groups,
                                                       const int2* __restrict__ gc, int n_gc) {
  E3K_TP_PROLOGUE
  E3K_TP_DISPATCH(tp_bwd_w_body, WITH_SH, L3MAX)
}

#ifndef E3K_TP_XW_WAVES
#define E3K_TP_XW_WAVES 8
#endif
#ifdef E3K_DEBUG_KNOBS
#define E3K_TP_BWDX_WAVES(MODE, MAXL, L3MAX) 1
#else
#define E3K_TP_BWDX_WAVES(MODE, MAXL, L3MAX) ((MODE == 5 && MAXL <= 2 && L3MAX <= 2) ? E3K_TP_XW_WAVES : 1)
#endif
template <int MAXL, int L3MAX, bool SPLIT, bool FULL, int MODE = 0>
__global__ __launch_bounds__(256, E3K_TP_BWDX_WAVES(MODE, MAXL, L3MAX)) void tp_bwd_x_kernel(TpArgs a, const e3k_tp_group* __restrict__ groups,
                                                       const int2* __restrict__ gc, int n_gc) {
  E3K_TP_PROLOGUE
  if constexpr (MODE == 8) {      // (e_store: this work item's slices of the edge-gradient partials)
    TpArgs ap = a;
    if (a.e_store) {
      if (a.g_sh) ap.g_sh = a.g_sh + (int64_t)gci * a.e_edges * a.d_sh;
      if (a.g_r) ap.g_r = a.g_r + (int64_t)gci * a.e_edges;
    }
    {
      const TpArgs& a = ap;
      E3K_TP_DISPATCH(tp_bwd_x_body, L3MAX, MODE)
    }
  } else {
    E3K_TP_DISPATCH(tp_bwd_x_body, L3MAX, MODE)
  }
}
// table-form edge backward (g_sh, g_coef, optionally g_w) and the dual weight gradient: channel-complete plans (SPLIT: the l_max 3
// plans walked by two waves per group -- streamed weights only; g_sh / g_r of the two parts meet in the same atomics)
template <int MAXL, int L3MAX, bool STREAM, bool SPLIT = false>
__global__ __launch_bounds__(256) void tp_bwd_e_kernel(TpArgs a, const e3k_tp_group* __restrict__ groups,
                                                       const int2* __restrict__ gc, int n_gc) {
  constexpr bool FULL = true;
  E3K_TP_PROLOGUE
  TpArgs ap = a;
  if (a.e_store) {
    if (a.g_sh) ap.g_sh = a.g_sh + (int64_t)gci * a.e_edges * a.d_sh;
    if (a.g_r) ap.g_r = a.g_r + (int64_t)gci * a.e_edges;
  }
  {
    const TpArgs& a = ap;
    E3K_TP_DISPATCH(tp_bwd_e_body, L3MAX, STREAM)
  }
}

// g_sh[e, :] = sum over the work items (in item order) of their stored shares; g_r likewise: the ordered combine behind an e_store launch
__global__ __launch_bounds__(256) void edge_partials_combine_kernel(const float* __restrict__ part_sh, const float* __restrict__ part_r,
                                                                    int n_items, int64_t E, int d_sh, float* __restrict__ g_sh,
                                                                    float* __restrict__ g_r) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int width = d_sh + 1;
  if (q >= E * width) return;
  const int64_t e = q / width;
  const int j = (int)(q - e * width);
  float acc = 0.f;
  if (j < d_sh) {
    if (!g_sh) return;
    for (int i = 0; i < n_items; ++i) acc += part_sh[((int64_t)i * E + e) * d_sh + j];
    g_sh[e * d_sh + j] = acc;
  } else {
    if (!g_r) return;
    for (int i = 0; i < n_items; ++i) acc += part_r[(int64_t)i * E + e];
    g_r[e] = acc;
  }
}
template <int MAXL, int L3MAX, bool SPLIT = false>
__global__ __launch_bounds__(256) void tp_bwd_w_dual_kernel(TpArgs a, const e3k_tp_group* __restrict__ groups,
                                                            const int2* __restrict__ gc, int n_gc) {
  constexpr bool FULL = true;
  E3K_TP_PROLOGUE
  E3K_TP_DISPATCH(tp_bwd_w_dual_body, L3MAX)
}
#undef E3K_TP_DISPATCH
#undef E3K_TP_CASE

}  // namespace e3k

// ------------------------------------------------------------------------------------------
// plan + C ABI
// ------------------------------------------------------------------------------------------

extern "C" void e3k_tp_limits(int* l1max, int* l2max, int* l3max) {
  if (l1max) *l1max = E3K_L1MAX;
  if (l2max) *l2max = E3K_L2MAX;
  if (l3max) *l3max = E3K_L3MAX;
}

namespace {
template <int L1>
int max_l3_of(unsigned mask) {
  int m = 0;
  for (int q = 0; q < e3k::Slots<L1>::NQ; ++q)
    if ((mask & (1u << q)) && e3k::Slots<L1>::L3[q] > m) m = e3k::Slots<L1>::L3[q];
  return m;
}
template <int L1>
void slot_counts_of(unsigned mask, int& n_acc, int& lo, int& hi) {
  using S = e3k::Slots<L1>;
  for (int q = 0; q < S::NQ; ++q) {
    if (!(mask & (1u << q))) continue;
    const int d = 2 * S::L3[q] + 1;
    n_acc += d;
    if (q < e3k::SplitAt<L1>::Q) lo += d; else hi += d;
  }
}
void plan_slot_counts(const e3k_tp_group& g, int& n_acc, int& lo, int& hi) {
  switch (g.l1) {
    case 0: slot_counts_of<0>(g.mask, n_acc, lo, hi); break;
    case 1: slot_counts_of<1>(g.mask, n_acc, lo, hi); break;
    case 2: slot_counts_of<2>(g.mask, n_acc, lo, hi); break;
    default: slot_counts_of<3>(g.mask, n_acc, lo, hi); break;
  }
}
template <int L1>
unsigned full_mask_of(int l3max) {
  unsigned m = 0;
  for (int q = 0; q < e3k::Slots<L1>::NQ; ++q)
    if (e3k::Slots<L1>::L3[q] <= l3max) m |= 1u << q;
  return m;
}
unsigned plan_full_mask(int l1, int l3max) {
  switch (l1) {
    case 0: return full_mask_of<0>(l3max);
    case 1: return full_mask_of<1>(l3max);
    case 2: return full_mask_of<2>(l3max);
    default: return full_mask_of<3>(l3max);
  }
}
int plan_max_l3(const e3k_tp_group& g) {
  switch (g.l1) {
    case 0: return max_l3_of<0>(g.mask);
    case 1: return max_l3_of<1>(g.mask);
    case 2: return max_l3_of<2>(g.mask);
    default: return max_l3_of<3>(g.mask);
  }
}
}  // namespace

extern "C" int e3k_tp_plan_create(const e3k_tp_group* groups, int32_t n_groups, int32_t d_in, int32_t d_sh,
                                  int32_t w_numel, int32_t d_mid, e3k_tp_plan** out) {
  if (!groups || n_groups <= 0 || !out) return E3K_ERR_INVALID;
  for (int i = 0; i < n_groups; ++i) {
    const e3k_tp_group& g = groups[i];
    if (g.l1 < 0 || g.mul <= 0) return E3K_ERR_INVALID;
    if (g.l1 > E3K_L1MAX) return E3K_ERR_UNSUPPORTED;
  }
  e3k_tp_plan* p = new e3k_tp_plan();
  p->n_groups = n_groups;
  p->d_in = d_in;
  p->d_sh = d_sh;
  p->w_numel = w_numel;
  p->d_mid = d_mid;
  p->d_groups = nullptr;
  p->d_gc = nullptr;
  p->n_gc = 0;
  if (hipMalloc(&p->d_groups, sizeof(e3k_tp_group) * n_groups) != hipSuccess) {
    delete p;
    return E3K_ERR_LAUNCH;
  }
  if (hipMemcpy(p->d_groups, groups, sizeof(e3k_tp_group) * n_groups, hipMemcpyHostToDevice) != hipSuccess) {
    e3k_tp_plan_destroy(p);
    return E3K_ERR_LAUNCH;
  }
  {
    // work list: (group, 64-channel chunk | part << 16).  A group whose enabled slots hold more than
    // E3K_TP_SPLIT_ACC accumulators (l_max = 3 models) is walked by two waves, one per slot part.
    E3K_KNOB_INT(split_acc, "E3K_TP_SPLIT_ACC", 24);
    int64_t cap = 0;
    for (int i = 0; i < n_groups; ++i) cap += 2 * ((groups[i].mul + 63) / 64);
    if (cap > (1 << 20)) {   // the work item index packs (chunk | part << 16): far beyond any irreps this path serves
      e3k_tp_plan_destroy(p);
      return E3K_ERR_UNSUPPORTED;
    }
    int2* host = new int2[cap ? cap : 1];
    int cnt = 0;
    bool any = false;
    for (int i = 0; i < n_groups; ++i) {
      int n_acc = 0, lo = 0, hi = 0;
      plan_slot_counts(groups[i], n_acc, lo, hi);
      if (groups[i].l1 >= 1 && n_acc > split_acc) any = true;
      if ((groups[i].mul + 63) / 64 > 0xffff) {
        delete[] host;
        e3k_tp_plan_destroy(p);
        return E3K_ERR_UNSUPPORTED;
      }
    }
    p->split = any ? 1 : 0;
    for (int i = 0; i < n_groups; ++i) {
      int n_acc = 0, lo = 0, hi = 0;
      plan_slot_counts(groups[i], n_acc, lo, hi);
      for (int c = 0; c < (groups[i].mul + 63) / 64; ++c) {
        if (any && groups[i].l1 >= 1) {   // in a split plan every l1 >= 1 group goes by parts (empty parts are skipped)
          if (lo > 0) host[cnt++] = make_int2(i, c | (0 << 16));
          if (hi > 0) host[cnt++] = make_int2(i, c | (1 << 16));
        } else {
          host[cnt++] = make_int2(i, c | (2 << 16));
        }
      }
    }
    hipError_t e1 = hipMalloc(&p->d_gc, sizeof(int2) * (cnt ? cnt : 1));
    hipError_t e2 = e1 == hipSuccess ? hipMemcpy(p->d_gc, host, sizeof(int2) * cnt, hipMemcpyHostToDevice) : e1;
    delete[] host;
    if (e2 != hipSuccess) {
      e3k_tp_plan_destroy(p);
      return E3K_ERR_LAUNCH;
    }
    p->n_gc = cnt;
    p->max_l1 = 0;
    p->max_l3 = 0;
    p->x_cols = 0;
    p->x_shared = 0;
    p->full64 = 1;
    for (int i = 0; i < n_groups; ++i) {
      if (groups[i].mul % 64) p->full64 = 0;
      const int lo_i = groups[i].x_off, hi_i = lo_i + (2 * groups[i].l1 + 1) * groups[i].mul;
      for (int j = 0; j < i; ++j) {
        const int lo_j = groups[j].x_off, hi_j = lo_j + (2 * groups[j].l1 + 1) * groups[j].mul;
        if (lo_i < hi_j && lo_j < hi_i) p->x_shared = 1;
      }
      p->x_cols += (2 * groups[i].l1 + 1) * groups[i].mul;
      p->max_l1 = groups[i].l1 > p->max_l1 ? groups[i].l1 : p->max_l1;
      p->max_l3 = plan_max_l3(groups[i]) > p->max_l3 ? plan_max_l3(groups[i]) : p->max_l3;
    }
    // ... and every slot the kernel instantiation visits is enabled in every group (see launch_all: outputs up to the
    // largest input degree when the model stops there, else up to 3)
    const int l3_inst = (p->max_l3 <= p->max_l1 && !p->split) ? p->max_l1 : 3;   // the predicate of launch_all
    for (int i = 0; i < n_groups; ++i) {
      if (groups[i].mask != plan_full_mask(groups[i].l1, l3_inst)) p->full64 = 0;
      // FULL kernels read sh[e] at fixed columns: degrees 0, 1, 2 at 0, 1, 4 of a 9-wide row
      if (d_sh != 9 || groups[i].y_off[0] != 0 || groups[i].y_off[1] != 1 || groups[i].y_off[2] != 4) p->full64 = 0;
    }
  }
  *out = p;
  return E3K_OK;
}

#ifdef E3K_DEBUG_KNOBS
extern "C" int e3k_dbg_tp_ablate(int mask) {
  e3k::g_tp_ablate_host = mask;
  return 0;
}
#endif

extern "C" int e3k_tp_bwd_x_overwrites(const e3k_tp_plan* p) {
  // every element of g_x is stored exactly once: one group per input block, all of [0, d_in) covered, single-wave groups
  return (p && !p->split && !p->x_shared && p->x_cols == p->d_in) ? 1 : 0;
}

extern "C" void e3k_tp_plan_destroy(e3k_tp_plan* p) {
  if (!p) return;
  if (p->d_groups) (void)hipFree(p->d_groups);
  if (p->d_gc) (void)hipFree(p->d_gc);
  delete p;
}

namespace {
enum TpKind { TP_FWD, TP_BWD_W, TP_BWD_W_SH, TP_BWD_X, TP_FWD_TABLE, TP_BWD_X_TABLE, TP_FWD_JVP, TP_BWD_X_DUAL, TP_BWD_E, TP_BWD_W_DUAL,
              TP_FWD_PACKED, TP_BWD_X_PACKED, TP_BWD_XW_PACKED, TP_BWD_XW, TP_BWD_XW_DUAL, TP_BWD_XE };

int launch_all(TpKind kind, const e3k::TpArgs& a, const e3k_tp_plan* p, int64_t N, hipStream_t st) {
  static_assert(E3K_L1MAX == 3, "extend the degree switch in the kernels when the CG tables grow");
  const int n_gc = p->n_gc;
  if (!n_gc || N <= 0) return E3K_OK;
  e3k::TpArgs args = a;
  args.n_items = N * n_gc;
  int64_t blocks = (args.n_items + 3) / 4;
  // (packed table, layer 3 of config_energy at 256 molecules, isolated: forward 163 -> 151 us, input gradient 221 -> 215; inside the
  //  step 124 -> 118 and 194 -> 187; the four-row form, whose 3.9 MB table stays in L2 either way, gains nothing: 187 / 188 us)
  E3K_KNOB_INT(tp_order, "E3K_TP_ORDER", 1);
  args.order = (tp_order && (kind == TP_FWD_PACKED || kind == TP_BWD_X_PACKED || kind == TP_BWD_XW_PACKED) && N >= 64) ? 1 : 0;
#ifdef E3K_DEBUG_KNOBS
  args.ablate = e3k::g_tp_ablate_host;
#endif
  if (args.order) blocks = 8 * ((((N + 7) / 8) * n_gc + 3) / 4);      // eight equal sub-grids, one per XCD (blocks b, b + 8, .. share one)
  if (blocks > 0x7fffffffLL) return E3K_ERR_INVALID;
  dim3 grid((unsigned)blocks), block(256);
  if (kind == TP_FWD_JVP || kind == TP_BWD_X_DUAL || kind == TP_BWD_E || kind == TP_BWD_W_DUAL || kind == TP_BWD_XW_DUAL || kind == TP_BWD_XE) {
    // second-order forms of force training: channel-complete plans; walked by one wave per group (the l_max <= 2 models), or by two
    // (SPLIT: l_max 3) with the weights STREAMED (w[e], dw/dr[e] materialised -- what the force block does by default)
    if (!p->full64) return E3K_ERR_UNSUPPORTED;
    const bool lo = p->max_l3 <= p->max_l1;
    const bool streamed = args.bin == nullptr;      // w[e] / dw[e] rows in a.w / a.w2 instead of the tables + per-edge knots
    if ((kind == TP_BWD_XW_DUAL || kind == TP_BWD_XE) && !streamed) return E3K_ERR_UNSUPPORTED;
    if (p->split) {
      if (!streamed) return E3K_ERR_UNSUPPORTED;
#define E3K_TP_LAUNCH_2S(ML)                                                                                                            \
  switch (kind) {                                                                                                                       \
    case TP_FWD_JVP: hipLaunchKernelGGL((e3k::tp_fwd_kernel<ML, 3, true, true, 3>), grid, block, 0, st, args, p->d_groups, p->d_gc, n_gc); break;      \
    case TP_BWD_X_DUAL: hipLaunchKernelGGL((e3k::tp_bwd_x_kernel<ML, 3, true, true, 3>), grid, block, 0, st, args, p->d_groups, p->d_gc, n_gc); break; \
    case TP_BWD_XW_DUAL: hipLaunchKernelGGL((e3k::tp_bwd_x_kernel<ML, 3, true, true, 7>), grid, block, 0, st, args, p->d_groups, p->d_gc, n_gc); break; \
    case TP_BWD_XE: hipLaunchKernelGGL((e3k::tp_bwd_x_kernel<ML, 3, true, true, 8>), grid, block, 0, st, args, p->d_groups, p->d_gc, n_gc); break; \
    case TP_BWD_E: hipLaunchKernelGGL((e3k::tp_bwd_e_kernel<ML, 3, true, true>), grid, block, 0, st, args, p->d_groups, p->d_gc, n_gc); break;         \
    default: hipLaunchKernelGGL((e3k::tp_bwd_w_dual_kernel<ML, 3, true>), grid, block, 0, st, args, p->d_groups, p->d_gc, n_gc); break;                \
  }
      switch (p->max_l1) {
        case 1: E3K_TP_LAUNCH_2S(1) break;
        case 2: E3K_TP_LAUNCH_2S(2) break;
        default: E3K_TP_LAUNCH_2S(3) break;
      }
#undef E3K_TP_LAUNCH_2S
      E3K_CHECK_LAUNCH();
      return E3K_OK;
    }
#define E3K_TP_LAUNCH_2(ML, L3)                                                                                                         \
  switch (kind) {                                                                                                                       \
    case TP_FWD_JVP:                                                                                                                    \
      if (streamed) hipLaunchKernelGGL((e3k::tp_fwd_kernel<ML, L3, false, true, 3>), grid, block, 0, st, args, p->d_groups, p->d_gc, n_gc);   \
      else hipLaunchKernelGGL((e3k::tp_fwd_kernel<ML, L3, false, true, 2>), grid, block, 0, st, args, p->d_groups, p->d_gc, n_gc);            \
      break;                                                                                                                            \
    case TP_BWD_X_DUAL:                                                                                                                 \
      if (streamed) hipLaunchKernelGGL((e3k::tp_bwd_x_kernel<ML, L3, false, true, 3>), grid, block, 0, st, args, p->d_groups, p->d_gc, n_gc); \
      else hipLaunchKernelGGL((e3k::tp_bwd_x_kernel<ML, L3, false, true, 2>), grid, block, 0, st, args, p->d_groups, p->d_gc, n_gc);          \
      break;                                                                                                                            \
    case TP_BWD_XW_DUAL:                                                                                                                \
      hipLaunchKernelGGL((e3k::tp_bwd_x_kernel<ML, L3, false, true, 7>), grid, block, 0, st, args, p->d_groups, p->d_gc, n_gc);         \
      break;                                                                                                                            \
    case TP_BWD_XE:                                                                                                                     \
      hipLaunchKernelGGL((e3k::tp_bwd_x_kernel<ML, L3, false, true, 8>), grid, block, 0, st, args, p->d_groups, p->d_gc, n_gc);         \
      break;                                                                                                                            \
    case TP_BWD_E:                                                                                                                      \
      if (streamed) hipLaunchKernelGGL((e3k::tp_bwd_e_kernel<ML, L3, true>), grid, block, 0, st, args, p->d_groups, p->d_gc, n_gc);           \
      else hipLaunchKernelGGL((e3k::tp_bwd_e_kernel<ML, L3, false>), grid, block, 0, st, args, p->d_groups, p->d_gc, n_gc);                   \
      break;                                                                                                                            \
    default: hipLaunchKernelGGL((e3k::tp_bwd_w_dual_kernel<ML, L3>), grid, block, 0, st, args, p->d_groups, p->d_gc, n_gc); break;      \
  }
    switch (p->max_l1) {
      case 0: if (lo) { E3K_TP_LAUNCH_2(0, 0) } else { E3K_TP_LAUNCH_2(0, 3) } break;
      case 1: if (lo) { E3K_TP_LAUNCH_2(1, 1) } else { E3K_TP_LAUNCH_2(1, 3) } break;
      case 2: if (lo) { E3K_TP_LAUNCH_2(2, 2) } else { E3K_TP_LAUNCH_2(2, 3) } break;
      default: E3K_TP_LAUNCH_2(3, 3) break;
    }
#undef E3K_TP_LAUNCH_2
    E3K_CHECK_LAUNCH();
    return E3K_OK;
  }
  if (kind == TP_FWD_TABLE || kind == TP_BWD_X_TABLE || kind == TP_FWD_PACKED || kind == TP_BWD_X_PACKED || kind == TP_BWD_XW_PACKED) {      // channel-complete (FULL) plans, split or not
    if (!p->full64) return E3K_ERR_UNSUPPORTED;
    const bool lo = p->max_l3 <= p->max_l1, spl = p->split != 0;
#define E3K_TP_LAUNCH_T(ML, L3, SP)                                                                                                     \
  {                                                                                                                                     \
    if (kind == TP_FWD_TABLE)                                                                                                           \
      hipLaunchKernelGGL((e3k::tp_fwd_kernel<ML, L3, SP, true, 1>), grid, block, 0, st, args, p->d_groups, p->d_gc, n_gc);              \
    else if (kind == TP_FWD_PACKED)                                                                                                     \
      hipLaunchKernelGGL((e3k::tp_fwd_kernel<ML, L3, SP, true, 4>), grid, block, 0, st, args, p->d_groups, p->d_gc, n_gc);              \
    else if (kind == TP_BWD_X_PACKED)                                                                                                   \
      hipLaunchKernelGGL((e3k::tp_bwd_x_kernel<ML, L3, SP, true, 4>), grid, block, 0, st, args, p->d_groups, p->d_gc, n_gc);            \
    else if (kind == TP_BWD_XW_PACKED)                                                                                                  \
      hipLaunchKernelGGL((e3k::tp_bwd_x_kernel<ML, L3, SP, true, 5>), grid, block, 0, st, args, p->d_groups, p->d_gc, n_gc);            \
    else                                                                                                                                \
      hipLaunchKernelGGL((e3k::tp_bwd_x_kernel<ML, L3, SP, true, 1>), grid, block, 0, st, args, p->d_groups, p->d_gc, n_gc);            \
  }
    switch (p->max_l1) {
      case 0: if (lo) E3K_TP_LAUNCH_T(0, 0, false) else E3K_TP_LAUNCH_T(0, 3, false) break;
      case 1: if (lo && !spl) E3K_TP_LAUNCH_T(1, 1, false) else if (!spl) E3K_TP_LAUNCH_T(1, 3, false) else E3K_TP_LAUNCH_T(1, 3, true) break;
      case 2: if (lo && !spl) E3K_TP_LAUNCH_T(2, 2, false) else if (!spl) E3K_TP_LAUNCH_T(2, 3, false) else E3K_TP_LAUNCH_T(2, 3, true) break;
      default: if (!spl) E3K_TP_LAUNCH_T(3, 3, false) else E3K_TP_LAUNCH_T(3, 3, true) break;
    }
#undef E3K_TP_LAUNCH_T
    E3K_CHECK_LAUNCH();
    return E3K_OK;
  }
#define E3K_TP_LAUNCH_F(ML, L3, SP, FU)                                                                                      \
  switch (kind) {                                                                                                   \
    case TP_BWD_XW:                                                                                                 \
      hipLaunchKernelGGL((e3k::tp_bwd_x_kernel<ML, L3, SP, FU, 6>), grid, block, 0, st, args, p->d_groups, p->d_gc, n_gc); \
      break;                                                                                                        \
    case TP_FWD: hipLaunchKernelGGL((e3k::tp_fwd_kernel<ML, L3, SP, FU>), grid, block, 0, st, args, p->d_groups, p->d_gc, n_gc); break; \
    case TP_BWD_W:                                                                                                  \
      hipLaunchKernelGGL((e3k::tp_bwd_w_kernel<false, ML, L3, SP, FU>), grid, block, 0, st, args, p->d_groups, p->d_gc, n_gc);  \
      break;                                                                                                        \
    case TP_BWD_W_SH:                                                                                               \
      hipLaunchKernelGGL((e3k::tp_bwd_w_kernel<true, ML, L3, SP, FU>), grid, block, 0, st, args, p->d_groups, p->d_gc, n_gc);   \
      break;                                                                                                        \
    case TP_BWD_X: hipLaunchKernelGGL((e3k::tp_bwd_x_kernel<ML, L3, SP, FU>), grid, block, 0, st, args, p->d_groups, p->d_gc, n_gc); break; \
    default: break;                                                                                                 \
  }
#define E3K_TP_LAUNCH(ML, L3, SP)                  \
  if (p->full64) { E3K_TP_LAUNCH_F(ML, L3, SP, true) } \
  else { E3K_TP_LAUNCH_F(ML, L3, SP, false) }
  // instantiations per input degree: outputs up to the same degree (l_max-limited models) or up to 3, the latter
  // also in the split form (two waves per group)
  const bool low = p->max_l3 <= p->max_l1;
  const bool sp = p->split != 0;
  switch (p->max_l1) {
    case 0: if (low) { E3K_TP_LAUNCH(0, 0, false) } else { E3K_TP_LAUNCH(0, 3, false) } break;
    case 1: if (low && !sp) { E3K_TP_LAUNCH(1, 1, false) } else if (!sp) { E3K_TP_LAUNCH(1, 3, false) } else { E3K_TP_LAUNCH(1, 3, true) } break;
    case 2: if (low && !sp) { E3K_TP_LAUNCH(2, 2, false) } else if (!sp) { E3K_TP_LAUNCH(2, 3, false) } else { E3K_TP_LAUNCH(2, 3, true) } break;
    default: if (!sp) { E3K_TP_LAUNCH(3, 3, false) } else { E3K_TP_LAUNCH(3, 3, true) } break;
  }
#undef E3K_TP_LAUNCH
#undef E3K_TP_LAUNCH_F
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}
}  // namespace

extern "C" int e3k_tp_fwd(const e3k_tp_plan* plan, const float* x, const float* sh, const float* w,
                          const int32_t* src, const int32_t* dst_ptr, const int32_t* dst_perm, int64_t N, int64_t E,
                          float* out, void* stream) {
  if (!plan || N < 0 || E < 0) return E3K_ERR_INVALID;
  if (N == 0) return E3K_OK;
  if (!x || !out || !dst_ptr || (E > 0 && (!sh || !w || !src || !dst_perm))) return E3K_ERR_INVALID;
  e3k::TpArgs a{};
  a.x = x; a.sh = sh; a.w = w; a.out = out; a.nbr = src; a.ptr = dst_ptr; a.perm = dst_perm;
  a.d_in = plan->d_in; a.d_sh = plan->d_sh; a.W = plan->w_numel; a.d_mid = plan->d_mid;
  return launch_all(TP_FWD, a, plan, N, (hipStream_t)stream);
}

extern "C" int e3k_tp_bwd_w(const e3k_tp_plan* plan, const float* x, const float* sh, const float* w,
                            const float* g_out, const int32_t* src, const int32_t* dst_ptr, const int32_t* dst_perm,
                            int64_t N, int64_t E, float* g_w, float* g_sh, void* stream) {
  if (!plan || N < 0 || E < 0) return E3K_ERR_INVALID;
  if (N == 0 || E == 0) return E3K_OK;
  if (!x || !sh || !g_out || !src || !dst_ptr || !dst_perm || (!g_w && !g_sh)) return E3K_ERR_INVALID;
  if (g_sh && !w) return E3K_ERR_INVALID;
  e3k::TpArgs a{};
  a.x = x; a.sh = sh; a.w = w; a.g_out = g_out; a.g_w = g_w; a.g_sh = g_sh;
  a.nbr = src; a.ptr = dst_ptr; a.perm = dst_perm;
  a.d_in = plan->d_in; a.d_sh = plan->d_sh; a.W = plan->w_numel; a.d_mid = plan->d_mid;
  return launch_all(g_sh ? TP_BWD_W_SH : TP_BWD_W, a, plan, N, (hipStream_t)stream);
}

extern "C" int e3k_tp_bwd_x(const e3k_tp_plan* plan, const float* sh, const float* w, const float* g_out,
                            const int32_t* dst, const int32_t* src_ptr, const int32_t* src_perm, int64_t N, int64_t E,
                            float* g_x, void* stream) {
  if (!plan || N < 0 || E < 0) return E3K_ERR_INVALID;
  if (N == 0) return E3K_OK;
  if (!g_out || !g_x || !src_ptr || (E > 0 && (!sh || !w || !dst || !src_perm))) return E3K_ERR_INVALID;
  e3k::TpArgs a{};
  a.sh = sh; a.w = w; a.g_out = g_out; a.g_x = g_x; a.nbr = dst; a.ptr = src_ptr; a.perm = src_perm;
  a.x_shared = plan->x_shared;
  a.d_in = plan->d_in; a.d_sh = plan->d_sh; a.W = plan->w_numel; a.d_mid = plan->d_mid;
  return launch_all(TP_BWD_X, a, plan, N, (hipStream_t)stream);
}

// ---- the same two passes with the path weights interpolated from the radial knot table inside the kernel ---------------
// T [K + 1, W]: the radial MLP on the knots; bin [E]: every edge's knot i (stencil rows i - 1 .. i + 2); coef [E, 4]: its four
// interpolation weights (e3k_rtable_bins).  E3K_ERR_UNSUPPORTED for plans that are not channel-complete (the caller then
// materialises w: e3k_rtable_interp_fwd).
extern "C" int e3k_tp_table_supported(const e3k_tp_plan* p) {
  return (p && p->full64) ? 1 : 0;
}
extern "C" int e3k_tp_table2_supported(const e3k_tp_plan* p) {
  return (p && p->full64 && !p->split) ? 1 : 0;
}
/* the second-order forms with w[e] / dw/dr[e] MATERIALISED (bin = coef = NULL): also the split (l_max 3) plans */
extern "C" int e3k_tp_second_order_streamed_supported(const e3k_tp_plan* p) {
  return (p && p->full64) ? 1 : 0;
}

extern "C" int e3k_tp_fwd_table(const e3k_tp_plan* plan, const float* x, const float* sh, const float* T, const int32_t* bin,
                                const float* coef, const int32_t* src, const int32_t* dst_ptr, const int32_t* dst_perm, int64_t N,
                                int64_t E, float* out, void* stream) {
  if (!plan || N < 0 || E < 0) return E3K_ERR_INVALID;
  if (N == 0) return E3K_OK;
  if (!x || !out || !dst_ptr || (E > 0 && (!sh || !T || !bin || !coef || !src || !dst_perm))) return E3K_ERR_INVALID;
  e3k::TpArgs a{};
  a.x = x; a.sh = sh; a.w = T; a.bin = bin; a.coef = coef; a.out = out; a.nbr = src; a.ptr = dst_ptr; a.perm = dst_perm;
  a.d_in = plan->d_in; a.d_sh = plan->d_sh; a.W = plan->w_numel; a.d_mid = plan->d_mid;
  return launch_all(TP_FWD_TABLE, a, plan, N, (hipStream_t)stream);
}

extern "C" int e3k_tp_bwd_x_table(const e3k_tp_plan* plan, const float* sh, const float* T, const int32_t* bin, const float* coef,
                                  const float* g_out, const int32_t* dst, const int32_t* src_ptr, const int32_t* src_perm,
                                  int64_t N, int64_t E, float* g_x, void* stream) {
  if (!plan || N < 0 || E < 0) return E3K_ERR_INVALID;
  if (N == 0) return E3K_OK;
  if (!g_out || !g_x || !src_ptr || (E > 0 && (!sh || !T || !bin || !coef || !dst || !src_perm))) return E3K_ERR_INVALID;
  e3k::TpArgs a{};
  a.sh = sh; a.w = T; a.bin = bin; a.coef = coef; a.g_out = g_out; a.g_x = g_x; a.nbr = dst; a.ptr = src_ptr; a.perm = src_perm;
  a.x_shared = plan->x_shared;
  a.d_in = plan->d_in; a.d_sh = plan->d_sh; a.W = plan->w_numel; a.d_mid = plan->d_mid;
  return launch_all(TP_BWD_X_TABLE, a, plan, N, (hipStream_t)stream);
}

// ---- ... and from the PACKED table (e3k_rtable_pack: 12 bytes per (knot, weight); same plans as the table form), walking EDGE
// RECORDS (e3k_edge_records over the destination CSR for the forward, over the source CSR for the input gradient) -----------------
extern "C" int e3k_tp_fwd_ptable(const e3k_tp_plan* plan, const float* x, const void* P, const int32_t* erec_dst,
                                 const int32_t* dst_ptr, int64_t N, int64_t E, float* out, void* stream) {
  if (!plan || N < 0 || E < 0) return E3K_ERR_INVALID;
  if (N == 0) return E3K_OK;
  if (!x || !out || !dst_ptr || (E > 0 && (!P || !erec_dst))) return E3K_ERR_INVALID;
  if ((int64_t)plan->w_numel * 12 > 0x7fffffffLL || (reinterpret_cast<uintptr_t>(erec_dst) & 63)) return E3K_ERR_UNSUPPORTED;
  e3k::TpArgs a{};
  a.x = x; a.w = static_cast<const float*>(P); a.erec = erec_dst; a.out = out; a.ptr = dst_ptr;
  a.d_in = plan->d_in; a.d_sh = plan->d_sh; a.W = plan->w_numel; a.d_mid = plan->d_mid;
  return launch_all(TP_FWD_PACKED, a, plan, N, (hipStream_t)stream);
}

extern "C" int e3k_tp_bwd_x_ptable(const e3k_tp_plan* plan, const void* P, const int32_t* erec_src, const float* g_out,
                                   const int32_t* src_ptr, int64_t N, int64_t E, float* g_x, void* stream) {
  if (!plan || N < 0 || E < 0) return E3K_ERR_INVALID;
  if (N == 0) return E3K_OK;
  if (!g_out || !g_x || !src_ptr || (E > 0 && (!P || !erec_src))) return E3K_ERR_INVALID;
  if (reinterpret_cast<uintptr_t>(erec_src) & 63) return E3K_ERR_UNSUPPORTED;
  e3k::TpArgs a{};
  a.w = static_cast<const float*>(P); a.erec = erec_src; a.g_out = g_out; a.g_x = g_x; a.ptr = src_ptr;
  a.x_shared = plan->x_shared;
  a.d_in = plan->d_in; a.d_sh = plan->d_sh; a.W = plan->w_numel; a.d_mid = plan->d_mid;
  return launch_all(TP_BWD_X_PACKED, a, plan, N, (hipStream_t)stream);
}

// ... and, in the same walk, the weight gradient of every edge: g_w [E, W] row e = dF/dw[e] (what e3k_tp_bwd_w writes; x = the
// layer's input rows [N, d_in], channel-fastest).  Every (edge, path, channel) is written exactly once: no zero-fill needed.
extern "C" int e3k_tp_bwd_xw_ptable(const e3k_tp_plan* plan, const float* x, const void* P, const int32_t* erec_src, const float* g_out,
                                    const int32_t* src_ptr, int64_t N, int64_t E, float* g_x, float* g_w, void* stream) {
  if (!plan || N < 0 || E < 0) return E3K_ERR_INVALID;
  if (N == 0) return E3K_OK;
  if (!x || !g_out || !g_x || !src_ptr || (E > 0 && (!P || !erec_src || !g_w))) return E3K_ERR_INVALID;
  if (reinterpret_cast<uintptr_t>(erec_src) & 63) return E3K_ERR_UNSUPPORTED;
  e3k::TpArgs a{};
  a.x = x; a.w = static_cast<const float*>(P); a.erec = erec_src; a.g_out = g_out; a.g_x = g_x; a.g_w = g_w; a.ptr = src_ptr;
  a.x_shared = plan->x_shared;
  a.d_in = plan->d_in; a.d_sh = plan->d_sh; a.W = plan->w_numel; a.d_mid = plan->d_mid;
  return launch_all(TP_BWD_XW_PACKED, a, plan, N, (hipStream_t)stream);
}

// ... and with the weights streamed from w [E, W] (the per-edge radial MLP's layers; force training's materialised rows): g_x AND g_w
// in the one walk of the source CSR; every plan
extern "C" int e3k_tp_bwd_xw(const e3k_tp_plan* plan, const float* x, const float* sh, const float* w, const float* g_out,
                             const int32_t* dst, const int32_t* src_ptr, const int32_t* src_perm, int64_t N, int64_t E, float* g_x,
                             float* g_w, void* stream) {
  if (!plan || N < 0 || E < 0) return E3K_ERR_INVALID;
  if (N == 0) return E3K_OK;
  if (!x || !g_out || !g_x || !src_ptr || (E > 0 && (!sh || !w || !dst || !src_perm || !g_w))) return E3K_ERR_INVALID;
  e3k::TpArgs a{};
  a.x = x; a.sh = sh; a.w = w; a.g_out = g_out; a.g_x = g_x; a.g_w = g_w; a.nbr = dst; a.ptr = src_ptr; a.perm = src_perm;
  a.x_shared = plan->x_shared;
  a.d_in = plan->d_in; a.d_sh = plan->d_sh; a.W = plan->w_numel; a.d_mid = plan->d_mid;
  return launch_all(TP_BWD_XW, a, plan, N, (hipStream_t)stream);
}

// ---- deterministic edge gradients: per-work-item partials [n_gc, E, d_sh] + [n_gc, E], combined in item order -------------------------
extern "C" int64_t e3k_tp_edge_partials_floats(const e3k_tp_plan* plan, int64_t E) {
  if (!plan || E < 0) return 0;
  return (int64_t)plan->n_gc * E * (plan->d_sh + 1);
}
namespace {
void edge_partials_args(e3k::TpArgs& a, const e3k_tp_plan* plan, int64_t E, float* partials) {
  a.e_store = 1;
  a.e_edges = E;
  if (a.g_sh) a.g_sh = partials;
  if (a.g_r) a.g_r = partials + (int64_t)plan->n_gc * E * plan->d_sh;
}
int edge_partials_combine(const e3k_tp_plan* plan, int64_t E, const float* partials, float* g_sh, float* g_r, hipStream_t st) {
  const int64_t total = E * (plan->d_sh + 1);
  hipLaunchKernelGGL(e3k::edge_partials_combine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, partials,
                     partials + (int64_t)plan->n_gc * E * plan->d_sh, plan->n_gc, E, plan->d_sh, g_sh, g_r);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}
}  // namespace

// ---- force training on the table (plans with e3k_tp_table2_supported) ------------------------------------------------------
// With F = <g, TP(x[src], sh, w(T, coef))> (linear in each of g, x, sh, T, coef):
// (w[e] = sum_k coef[e,k] T[bin[e]-1+k], dw/dr[e] = sum_k coef[e,k] D[bin[e]-1+k] with D the slope table; with bin = coef = NULL
//  the two are passed MATERIALISED instead: T = w [E, W], D = dw/dr [E, W] (e3k_rtable_interp_fwd of either table) -- at the batch
//  sizes force training runs at, five or six kernels per layer read them and a 7.7 KB row per edge is a quarter of four table rows)
//   e3k_tp_bwd_e_table   g_sh = dF/dsh, g_r = <dF/dw, dw/dr> (both ACCUMULATED with atomics: zero-fill them), optionally g_w
//   e3k_tp_fwd_jvp_table out = TP(x2, sh, w) + TP(x, sh2, w) + TP(x, sh, s2 * dw/dr)
//   e3k_tp_bwd_x_dual_table g_x = dF/dx at (sh2, w) + dF/dx at (sh, s2 * dw/dr)
//   e3k_tp_bwd_w_dual    g_w[e] = dF/dw at (x2, sh) + dF/dw at (x, sh2)       (no table involved: w is the open slot)
extern "C" int e3k_tp_bwd_e_table(const e3k_tp_plan* plan, const float* x, const float* sh, const float* T, const float* D,
                                  const int32_t* bin, const float* coef, const float* g_out, const int32_t* src, const int32_t* dst_ptr,
                                  const int32_t* dst_perm, int64_t N, int64_t E, float* g_sh, float* g_r, float* g_w, float* e_partials,
                                  void* stream) {
  if (!plan || N < 0 || E < 0) return E3K_ERR_INVALID;
  if (N == 0 || E == 0) return E3K_OK;
  if (!x || !sh || !T || !D || (bin != nullptr) != (coef != nullptr) || !g_out || !src || !dst_ptr || !dst_perm || (!g_sh && !g_r && !g_w))
    return E3K_ERR_INVALID;
  e3k::TpArgs a{};
  a.x = x; a.sh = sh; a.w = T; a.w2 = D; a.bin = bin; a.coef = coef; a.g_out = g_out; a.g_sh = g_sh; a.g_r = g_r; a.g_w = g_w;
  a.nbr = src; a.ptr = dst_ptr; a.perm = dst_perm;
  a.d_in = plan->d_in; a.d_sh = plan->d_sh; a.W = plan->w_numel; a.d_mid = plan->d_mid;
  const bool part = e_partials && (g_sh || g_r);
  if (part) edge_partials_args(a, plan, E, e_partials);
  const int rc = launch_all(TP_BWD_E, a, plan, N, (hipStream_t)stream);
  if (rc != E3K_OK || !part) return rc;
  return edge_partials_combine(plan, E, e_partials, g_sh, g_r, (hipStream_t)stream);
}

extern "C" int e3k_tp_fwd_jvp_table(const e3k_tp_plan* plan, const float* x, const float* x2, const float* sh, const float* sh2,
                                    const float* T, const float* D, const int32_t* bin, const float* coef, const float* s2,
                                    const int32_t* src, const int32_t* dst_ptr, const int32_t* dst_perm, int64_t N, int64_t E,
                                    float* out, void* stream) {
  if (!plan || N < 0 || E < 0) return E3K_ERR_INVALID;
  if (N == 0) return E3K_OK;
  if (!x || !x2 || !out || !dst_ptr || (E > 0 && (!sh || !sh2 || !T || !D || (bin != nullptr) != (coef != nullptr) || !s2 || !src || !dst_perm)))
    return E3K_ERR_INVALID;
  e3k::TpArgs a{};
  a.x = x; a.x2 = x2; a.sh = sh; a.sh2 = sh2; a.w = T; a.w2 = D; a.bin = bin; a.coef = coef; a.s2 = s2; a.out = out;
  a.nbr = src; a.ptr = dst_ptr; a.perm = dst_perm;
  a.d_in = plan->d_in; a.d_sh = plan->d_sh; a.W = plan->w_numel; a.d_mid = plan->d_mid;
  return launch_all(TP_FWD_JVP, a, plan, N, (hipStream_t)stream);
}

extern "C" int e3k_tp_bwd_x_dual_table(const e3k_tp_plan* plan, const float* sh, const float* sh2, const float* T, const float* D,
                                       const int32_t* bin, const float* coef, const float* s2, const float* g_out, const int32_t* dst,
                                       const int32_t* src_ptr, const int32_t* src_perm, int64_t N, int64_t E, float* g_x, void* stream) {
  if (!plan || N < 0 || E < 0) return E3K_ERR_INVALID;
  if (N == 0) return E3K_OK;
  if (!g_out || !g_x || !src_ptr || (E > 0 && (!sh || !sh2 || !T || !D || (bin != nullptr) != (coef != nullptr) || !s2 || !dst || !src_perm)))
    return E3K_ERR_INVALID;
  e3k::TpArgs a{};
  a.sh = sh; a.sh2 = sh2; a.w = T; a.w2 = D; a.bin = bin; a.coef = coef; a.s2 = s2; a.g_out = g_out; a.g_x = g_x;
  a.nbr = dst; a.ptr = src_ptr; a.perm = src_perm;
  a.x_shared = plan->x_shared;
  a.d_in = plan->d_in; a.d_sh = plan->d_sh; a.W = plan->w_numel; a.d_mid = plan->d_mid;
  return launch_all(TP_BWD_X_DUAL, a, plan, N, (hipStream_t)stream);
}

// the first backward of a force evaluation in ONE walk of the source CSR (streamed rows w, dw [E, W]): g_x (e3k_tp_bwd_x), g_sh and g_r
// (e3k_tp_bwd_e_table: ACCUMULATED with atomics, zero-fill them) and, when g_w != NULL, the per-edge weight gradient (e3k_tp_bwd_w)
extern "C" int e3k_tp_bwd_xe(const e3k_tp_plan* plan, const float* x, const float* sh, const float* w, const float* dw,
                             const float* g_out, const int32_t* dst, const int32_t* src_ptr, const int32_t* src_perm, int64_t N, int64_t E,
                             float* g_x, float* g_sh, float* g_r, float* g_w, float* e_partials, void* stream) {
  if (!plan || N < 0 || E < 0) return E3K_ERR_INVALID;
  if (N == 0) return E3K_OK;
  if (!x || !g_out || !g_x || !src_ptr || (E > 0 && (!sh || !w || !dw || !dst || !src_perm || (!g_sh && !g_r)))) return E3K_ERR_INVALID;
  e3k::TpArgs a{};
  a.x = x; a.sh = sh; a.w = w; a.w2 = dw; a.g_out = g_out; a.g_x = g_x; a.g_sh = g_sh; a.g_r = g_r; a.g_w = g_w;
  a.nbr = dst; a.ptr = src_ptr; a.perm = src_perm;
  a.x_shared = plan->x_shared;
  a.d_in = plan->d_in; a.d_sh = plan->d_sh; a.W = plan->w_numel; a.d_mid = plan->d_mid;
  const bool part = e_partials && E > 0;
  if (part) edge_partials_args(a, plan, E, e_partials);
  const int rc = launch_all(TP_BWD_XE, a, plan, N, (hipStream_t)stream);
  if (rc != E3K_OK || !part) return rc;
  return edge_partials_combine(plan, E, e_partials, g_sh, g_r, (hipStream_t)stream);
}

// e3k_tp_bwd_x_dual_table on STREAMED rows (w, dw [E, W]) that also writes the weight gradients sharing its sums:
//   g_w [E, W]  = dF/dw at (x2, sh) + dF/dw at (x, sh2)      (e3k_tp_bwd_w_dual)
//   g_w_plain   = dF/dw at (x, sh)                            (e3k_tp_bwd_w; may be NULL)
// one walk of the source CSR instead of three walks
extern "C" int e3k_tp_bwd_xw_dual(const e3k_tp_plan* plan, const float* x, const float* x2, const float* sh, const float* sh2,
                                  const float* w, const float* dw, const float* s2, const float* g_out, const int32_t* dst,
                                  const int32_t* src_ptr, const int32_t* src_perm, int64_t N, int64_t E, float* g_x, float* g_w,
                                  float* g_w_plain, void* stream) {
  if (!plan || N < 0 || E < 0) return E3K_ERR_INVALID;
  if (N == 0) return E3K_OK;
  if (!x || !x2 || !g_out || !g_x || !src_ptr || (E > 0 && (!sh || !sh2 || !w || !dw || !s2 || !dst || !src_perm || !g_w)))
    return E3K_ERR_INVALID;
  e3k::TpArgs a{};
  a.x = x; a.x2 = x2; a.sh = sh; a.sh2 = sh2; a.w = w; a.w2 = dw; a.s2 = s2; a.g_out = g_out; a.g_x = g_x; a.g_w = g_w; a.g_w2 = g_w_plain;
  a.nbr = dst; a.ptr = src_ptr; a.perm = src_perm;
  a.x_shared = plan->x_shared;
  a.d_in = plan->d_in; a.d_sh = plan->d_sh; a.W = plan->w_numel; a.d_mid = plan->d_mid;
  return launch_all(TP_BWD_XW_DUAL, a, plan, N, (hipStream_t)stream);
}

extern "C" int e3k_tp_bwd_w_dual(const e3k_tp_plan* plan, const float* x, const float* x2, const float* sh, const float* sh2,
                                 const float* g_out, const int32_t* src, const int32_t* dst_ptr, const int32_t* dst_perm, int64_t N,
                                 int64_t E, float* g_w, void* stream) {
  if (!plan || N < 0 || E < 0) return E3K_ERR_INVALID;
  if (N == 0 || E == 0) return E3K_OK;
  if (!x || !x2 || !sh || !sh2 || !g_out || !src || !dst_ptr || !dst_perm || !g_w) return E3K_ERR_INVALID;
  e3k::TpArgs a{};
  a.x = x; a.x2 = x2; a.sh = sh; a.sh2 = sh2; a.g_out = g_out; a.g_w = g_w; a.nbr = src; a.ptr = dst_ptr; a.perm = dst_perm;
  a.d_in = plan->d_in; a.d_sh = plan->d_sh; a.W = plan->w_numel; a.d_mid = plan->d_mid;
  return launch_all(TP_BWD_W_DUAL, a, plan, N, (hipStream_t)stream);
}
