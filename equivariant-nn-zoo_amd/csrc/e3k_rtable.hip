// Radial weights through a knot table for gfx950.
//
// Replaces, per convolution layer of the reference (paths relative to /root/reference):
//   weight = self.fc(edge_radial)        e3_layers/nn/message_passing.py:74-79,93
// when edge_radial is RadialBasisEncoding(edge_length) (e3_layers/nn/embedding.py:210-219): the per-edge path weights
// are then a smooth function of ONE scalar, w[e, :] = f(r_e) with f = fc o basis o cutoff : [0, r_max] -> R^W.
// The reference (and e3k_gemm) evaluates f per edge -- 2 * 64 * W flops per edge forward and twice that backward, the
// largest block of matrix work of a training step.  Here f is evaluated on K + 1 equidistant knots by the same MLP kernels
// and every edge interpolates the FOUR knots around it with the cubic Lagrange weights (round 4; rounds 2-3: three knots,
// quadratic, on 2 048-4 096 knots)
//     x = r / h, i = floor(x) clamped to [1, K - 2], t = x - i      (h = 2^-m: x and t are EXACT in fp32, and so are the knots k h),
//     w[e] = c0(t) T[i-1] + c1(t) T[i] + c2(t) T[i+1] + c3(t) T[i+2],
//     c0 = -t(t-1)(t-2)/6, c1 = (t+1)(t-1)(t-2)/2, c2 = -(t+1)t(t-2)/2, c3 = (t+1)t(t-1)/6.
// Error 3/128 h^4 max|f''''| on values (h^3 max|f''''| / 12 on the slope dw/dr, which force training differentiates): the
// same accuracy as the quadratic rule on 4x fewer knots, so the table the tensor-product kernels gather from
// (e3k_tp_fwd_table: 4 rows per edge) is 4 MB instead of 14 and stays in an XCD's L2.  r >= r_max maps to x = K:
// t = 2, weights (0, 0, 0, 1): the last knot, where the envelope and its first five derivatives vanish (f is constant).
//
// Every edge carries its four weights as data (coef [E, 4]): the kernels that consume a table -- interpolation, its
// transpose, the tensor-product kernels -- are BILINEAR in (coef, T).  Force training needs dw/dr: differentiating the
// weights (d coef / d r applied to T) amplifies the fp32 rounding of the table by 1 / h (measured: 7e-6 .. 4e-5 relative slope
// error at any knot count), so the slope is its own table D = dT/dr on the knots (e3k_radial_slope_*: the hidden chain
// re-evaluated in float64, differenced there, the last layer applied in fp32), interpolated with the SAME weights -- 5e-8.
//
// Edges are grouped by knot with a STABLE counting sort (ascending edge id inside a knot: the transposed interpolation
// sums in a fixed order, no atomics, bit-identical run to run): one wave ranks a chunk of 1 024 edges against its own
// histogram in LDS, one workgroup scans the (chunk x knot) counts, one thread per edge places it.  No per-row sort, so a
// knot that holds thousands of edges (real molecules: C-H 1.09 A, C-C 1.52 A cluster in a handful of bins) costs what
// any other edge costs; the transpose splits a knot's edges into segments of <= 64, one wave each, and combines the
// segments' partial sums in order.
#include "e3k_common.h"

namespace e3k {

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nt_store4(float4* p, const float4& v) {
  f32x4 t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(p));
}
__device__ __forceinline__ float4 nt_load4(const float4* p) {
#ifdef E3K_NO_NT      // (experiment builds: tools/micro/nt_policy.sh)
  return *p;
#else
  const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  return make_float4(t.x, t.y, t.z, t.w);
#endif
}

constexpr int RT_CHUNK = 1024;      // edges ranked by one wave
constexpr int RT_SEG = 64;          // edges of one knot summed by one wave of the transpose

// ---- pass 1: knot, weights, rank inside (chunk, knot), chunk histogram ------------------------------------------------
// one workgroup (4 waves) per chunk of RT_CHUNK consecutive edges; wave w ranks its quarter of the chunk against its own
// (K + 1)-entry histogram in LDS (the serial part: one ballot round per distinct knot among 64 edges), then the quarters are
// chained: a lane's rank += the edges of its knot in the waves before it
// KEYED tables (round 5): an edge embedding that is a function of the radius AND a small categorical key (a bond type, say) is
// served by n_keys tables stacked into one -- block k holds the rows of key k --: KT + 1 = n_keys (K + 1) rows in all, an edge's
// knot becomes key[e] (K + 1) + i, and every consumer (interpolation, its transpose, the tensor-product kernels) works on the
// stacked table unchanged: the stencil i - 1 .. i + 2 never leaves a block (i in [1, K - 2]) and the first and the last two knots
// of every block hold no edges, so nothing of one block's transposed sums reaches another's rows.
__global__ __launch_bounds__(256) void rtable_bins_rank_kernel(const float* __restrict__ r, const int64_t* __restrict__ key, int64_t E,
                                                               float h_inv, int32_t K, int32_t KT, int32_t* __restrict__ bin,
                                                               float* __restrict__ coef, int32_t* __restrict__ lrank,
                                                               int32_t* __restrict__ chunk_hist, int32_t* __restrict__ bad) {
  extern __shared__ int32_t hist_all[];                      // [4][KT + 1]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int32_t* hist = hist_all + wv * (KT + 1);
  const int64_t chunk = blockIdx.x;
  for (int b = threadIdx.x; b < 4 * (KT + 1); b += 256) hist_all[b] = 0;
  __syncthreads();
  constexpr int PER_WAVE = RT_CHUNK / 4 / 64;
  const int64_t base = chunk * RT_CHUNK + (int64_t)wv * (RT_CHUNK / 4);
  int my_bin[PER_WAVE], my_rank[PER_WAVE];
#pragma unroll
  for (int it = 0; it < PER_WAVE; ++it) {
    const int64_t e = base + it * 64 + lane;
    const bool live = e < E;
    int i = 1;
    if (live) {
      const float raw = r[e] * h_inv;
      float x = raw < 0.f ? 0.f : (raw > (float)K ? (float)K : raw);      // (NaN radii stay NaN: NaN weights)
      i = (int)x;
      i = i < 1 ? 1 : (i > K - 2 ? K - 2 : i);
      const float t = x - (float)i;
      const float tm1 = t - 1.f, tm2 = t - 2.f, tp1 = t + 1.f;
      float4 c;
      c.x = -t * tm1 * tm2 * (1.f / 6.f);
      c.y = tp1 * tm1 * tm2 * 0.5f;
      c.z = -tp1 * t * tm2 * 0.5f;
      c.w = tp1 * t * tm1 * (1.f / 6.f);
      if (key) {      // (a key outside [0, n_keys) would index past the stacked table: folded into block 0 -- and FLAGGED, bit 3)
        const int64_t kk = key[e];
        const int nk = (KT + 1) / (K + 1);
        const bool in_range = kk >= 0 && kk < nk;
        i += in_range ? (int)kk * (K + 1) : 0;
        if (!in_range && bad) atomicOr(bad, 8);
      }
      bin[e] = i;
      *reinterpret_cast<float4*>(coef + 4 * e) = c;
    }
    // stable rank among the edges of the same knot seen so far by this wave: the lanes of one knot are served together
    unsigned long long todo = __ballot(live);
    int rank = 0;
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int b = __shfl(i, leader, 64);
      const unsigned long long same = __ballot(live && i == b) & todo;
      const int seen = hist[b];                                // (uniform address: one LDS read, broadcast)
      if (live && i == b) rank = seen + __popcll(same & ((1ull << lane) - 1ull));
      if (lane == leader) hist[b] = seen + __popcll(same);
      todo &= ~same;
    }
    my_bin[it] = live ? i : -1;
    my_rank[it] = rank;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < PER_WAVE; ++it) {
    if (my_bin[it] < 0) continue;
    int before = 0;
    for (int w = 0; w < wv; ++w) before += hist_all[w * (KT + 1) + my_bin[it]];
    lrank[base + it * 64 + lane] = my_rank[it] + before;
  }
  for (int b = threadIdx.x; b <= KT; b += 256)
    chunk_hist[chunk * (KT + 1) + b] = hist_all[b] + hist_all[(KT + 1) + b] + hist_all[2 * (KT + 1) + b] + hist_all[3 * (KT + 1) + b];
}

// ---- pass 2: one workgroup.  chunk_hist[c][b] -> number of edges of knot b in the chunks before c (in place);
// ptr[b] = first position of knot b in knot order (ptr[K + 1] = E); seg[b] = first segment of knot b (RT_SEG edges each)
__global__ __launch_bounds__(1024) void rtable_bins_scan_kernel(int32_t* __restrict__ chunk_hist, int32_t n_chunks, int32_t K,
                                                                int32_t* __restrict__ ptr, int32_t* __restrict__ seg) {
  __shared__ int32_t wave_tot[2][16];
  __shared__ int32_t carry[2];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  if (t < 2) carry[t] = 0;
  __syncthreads();
  for (int base = 0; base <= K; base += 1024) {
    const int b = base + t;
    int32_t cnt = 0;
    if (b <= K) {
      for (int c = 0; c < n_chunks; ++c) {
        const int32_t v = chunk_hist[(int64_t)c * (K + 1) + b];
        chunk_hist[(int64_t)c * (K + 1) + b] = cnt;
        cnt += v;
      }
    }
    const int32_t sg = (cnt + RT_SEG - 1) / RT_SEG;
    int32_t x = cnt, y = sg;      // inclusive scans inside the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int32_t xx = __shfl_up(x, off, 64), yy = __shfl_up(y, off, 64);
      if (lane >= off) x += xx, y += yy;
    }
    if (lane == 63) wave_tot[0][w] = x, wave_tot[1][w] = y;
    __syncthreads();
    int32_t bx = carry[0], by = carry[1];
    for (int k = 0; k < w; ++k) bx += wave_tot[0][k], by += wave_tot[1][k];
    if (b <= K) {
      ptr[b] = bx + x - cnt;
      seg[b] = by + y - sg;
      if (b == K) ptr[K + 1] = bx + x, seg[K + 1] = by + y;
    }
    __syncthreads();
    if (t == 1023) carry[0] = bx + x, carry[1] = by + y;
    __syncthreads();
  }
}

// ---- pass 3: perm[position in knot order] = edge id
__global__ __launch_bounds__(256) void rtable_bins_place_kernel(const int32_t* __restrict__ bin, const int32_t* __restrict__ lrank,
                                                                const int32_t* __restrict__ chunk_off, const int32_t* __restrict__ ptr,
                                                                int64_t E, int32_t K, int32_t* __restrict__ perm) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  const int b = bin[e];
  perm[ptr[b] + chunk_off[(e / RT_CHUNK) * (K + 1) + b] + lrank[e]] = (int32_t)e;
}

// w[e, :] = sum_k coef[e, k] T[bin[e] - 1 + k, :];  one wave per edge, 16-byte columns.  Edges are taken in KNOT order
// (perm): the edges of a knot, handled by neighbouring waves, read the same four table rows (L1 / L2 hits).
// T2 / w2 (optional): a second table interpolated with the same weights in the same pass (force training: the slope table)
__global__ __launch_bounds__(256) void rtable_interp_fwd_kernel(const float* __restrict__ T, const int32_t* __restrict__ perm,
                                                                const int32_t* __restrict__ bin, const float* __restrict__ coef,
                                                                int64_t E, int32_t W, float* __restrict__ w,
                                                                const float* __restrict__ T2, float* __restrict__ w2) {
  const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= E) return;
  const int lane = threadIdx.x & 63;
  const int e = uniform(perm[p]);
  const int i = uniform(bin[e]);
  const float* __restrict__ cp = coef + 4 * (int64_t)e;
  const float c0 = __uint_as_float(uniform((int)__float_as_uint(cp[0]))), c1 = __uint_as_float(uniform((int)__float_as_uint(cp[1])));
  const float c2 = __uint_as_float(uniform((int)__float_as_uint(cp[2]))), c3 = __uint_as_float(uniform((int)__float_as_uint(cp[3])));
  const float4* __restrict__ a = reinterpret_cast<const float4*>(T + (int64_t)(i - 1) * W);
  const float4* __restrict__ b = reinterpret_cast<const float4*>(T + (int64_t)i * W);
  const float4* __restrict__ c = reinterpret_cast<const float4*>(T + (int64_t)(i + 1) * W);
  const float4* __restrict__ d = reinterpret_cast<const float4*>(T + (int64_t)(i + 2) * W);
  float4* __restrict__ o = reinterpret_cast<float4*>(w + (int64_t)e * W);
  for (int q = lane; q < (W >> 2); q += 64) {
    const float4 va = a[q], vb = b[q], vc = c[q], vd = d[q];
    float4 v;      // the order of e3k::knot_mix (csrc/e3k_tp.hip): the in-kernel form gives the same bits
    v.x = fmaf(c3, vd.x, fmaf(c2, vc.x, fmaf(c1, vb.x, c0 * va.x)));
    v.y = fmaf(c3, vd.y, fmaf(c2, vc.y, fmaf(c1, vb.y, c0 * va.y)));
    v.z = fmaf(c3, vd.z, fmaf(c2, vc.z, fmaf(c1, vb.z, c0 * va.z)));
    v.w = fmaf(c3, vd.w, fmaf(c2, vc.w, fmaf(c1, vb.w, c0 * va.w)));
    nt_store4(o + q, v);      // written once, read by the edge kernels later: streamed past the caches
  }
  if (T2) {
    const int64_t d2 = T2 - T;      // (same row offsets in the second table)
    float4* __restrict__ o2 = reinterpret_cast<float4*>(w2 + (int64_t)e * W);
    for (int q = lane; q < (W >> 2); q += 64) {
      const float4 va = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a) + d2)[q];
      const float4 vb = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(b) + d2)[q];
      const float4 vc = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(c) + d2)[q];
      const float4 vd = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(d) + d2)[q];
      float4 v;
      v.x = fmaf(c3, vd.x, fmaf(c2, vc.x, fmaf(c1, vb.x, c0 * va.x)));
      v.y = fmaf(c3, vd.y, fmaf(c2, vc.y, fmaf(c1, vb.y, c0 * va.y)));
      v.z = fmaf(c3, vd.z, fmaf(c2, vc.z, fmaf(c1, vb.z, c0 * va.z)));
      v.w = fmaf(c3, vd.w, fmaf(c2, vc.w, fmaf(c1, vb.w, c0 * va.w)));
      nt_store4(o2 + q, v);
    }
  }
}

// backward, pass 1: one wave per (segment of <= RT_SEG edges of ONE knot b, 256-column chunk) reads the g_w rows of its edges
// ONCE (ascending edge id) and forms their four weighted sums -- the segment's contributions to the table rows b-1 .. b+2:
// P[segment][0..3][cols].  scale [E] (optional): every edge's weights are multiplied by scale[e] (force training: the slope
// table's gradient is the transpose applied with the radius' cotangent as per-edge factor).
__global__ __launch_bounds__(256) void rtable_bwd_partial_kernel(const float* __restrict__ gw, const float* __restrict__ coef,
                                                                 const float* __restrict__ scale,
                                                                 const int32_t* __restrict__ ptr, const int32_t* __restrict__ seg,
                                                                 const int32_t* __restrict__ perm, int32_t K, int32_t W,
                                                                 int32_t n_chunks, int64_t n_seg_cap, float* __restrict__ P) {
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= n_seg_cap * n_chunks) return;
  const int s = uniform((int)(item / n_chunks)), chunk = uniform((int)(item - (int64_t)s * n_chunks));
  if (s >= uniform(seg[K + 1])) return;
  int lo = 0, hi = K + 1;                     // the knot whose segment range holds s: seg[b] <= s < seg[b + 1]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (uniform(seg[mid]) <= s) lo = mid; else hi = mid;
  }
  const int b = lo;
  const int lane = threadIdx.x & 63;
  const int col = chunk * 256 + lane * 4;
  const bool live = col < W;                  // (lanes past the last column stay: they hold edges' ids and weights for the others)
  const int beg = uniform(ptr[b]) + (s - uniform(seg[b])) * RT_SEG;
  const int end_b = uniform(ptr[b + 1]);
  const int end = beg + RT_SEG < end_b ? beg + RT_SEG : end_b;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
  // Lane l fetches edge l's id and weights ONCE (a segment has at most 64 edges); the loop then reads them across lanes and has no
  // load that depends on another: EIGHT independent g_w rows are in flight per wave.  (Round 4's loop fetched id -> weights -> row
  // per edge, four edges at a time: a chain of three latencies per batch, 4.0 TB/s.)  Sums in ascending edge order, as before.
  const int n = end - beg;
  int e_l = 0;
  float4 c_l = make_float4(0.f, 0.f, 0.f, 0.f);
  if (lane < n) {
    e_l = perm[beg + lane];
    c_l = *reinterpret_cast<const float4*>(coef + 4 * (int64_t)e_l);
    if (scale) {
      const float sc = scale[e_l];
      c_l.x *= sc; c_l.y *= sc; c_l.z *= sc; c_l.w *= sc;
    }
  }
  auto row_of = [&](int k) {
    const int e = __builtin_amdgcn_readlane(e_l, k);
    return live ? nt_load4(reinterpret_cast<const float4*>(gw + (int64_t)e * W + col)) : make_float4(0.f, 0.f, 0.f, 0.f);      // read once
  };
  auto acc = [&](int k, const float4& g) {
    const float c0 = __uint_as_float(__builtin_amdgcn_readlane((int)__float_as_uint(c_l.x), k));
    const float c1 = __uint_as_float(__builtin_amdgcn_readlane((int)__float_as_uint(c_l.y), k));
    const float c2 = __uint_as_float(__builtin_amdgcn_readlane((int)__float_as_uint(c_l.z), k));
    const float c3 = __uint_as_float(__builtin_amdgcn_readlane((int)__float_as_uint(c_l.w), k));
    a0.x = fmaf(c0, g.x, a0.x); a0.y = fmaf(c0, g.y, a0.y); a0.z = fmaf(c0, g.z, a0.z); a0.w = fmaf(c0, g.w, a0.w);
    a1.x = fmaf(c1, g.x, a1.x); a1.y = fmaf(c1, g.y, a1.y); a1.z = fmaf(c1, g.z, a1.z); a1.w = fmaf(c1, g.w, a1.w);
    a2.x = fmaf(c2, g.x, a2.x); a2.y = fmaf(c2, g.y, a2.y); a2.z = fmaf(c2, g.z, a2.z); a2.w = fmaf(c2, g.w, a2.w);
    a3.x = fmaf(c3, g.x, a3.x); a3.y = fmaf(c3, g.y, a3.y); a3.z = fmaf(c3, g.z, a3.z); a3.w = fmaf(c3, g.w, a3.w);
  };
  int k = 0;
  for (; k + 8 <= n; k += 8) {
    float4 g[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) g[u] = row_of(k + u);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc(k + u, g[u]);
  }
  if (k + 4 <= n) {
    float4 g[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) g[u] = row_of(k + u);
#pragma unroll
    for (int u = 0; u < 4; ++u) acc(k + u, g[u]);
    k += 4;
  }
  for (; k < n; ++k) {
    const float4 g = row_of(k);
    acc(k, g);
  }
  if (!live) return;
  float* row = P + (int64_t)s * 4 * W + col;
  *reinterpret_cast<float4*>(row) = a0;
  *reinterpret_cast<float4*>(row + W) = a1;
  *reinterpret_cast<float4*>(row + 2 * W) = a2;
  *reinterpret_cast<float4*>(row + 3 * W) = a3;
}

// pass 2: g_T[j] (+)= sum_k sum over the segments of knot j + 1 - k of P[segment][k]  (a fixed order: deterministic),
// one thread per 16 bytes
__global__ __launch_bounds__(256) void rtable_bwd_combine_kernel(const float* __restrict__ P, const int32_t* __restrict__ seg,
                                                                 int32_t K, int32_t W, int32_t accumulate, float* __restrict__ gT) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int w4 = W >> 2;
  if (q >= (int64_t)(K + 1) * w4) return;
  const int j = (int)(q / w4), col = (int)(q - (int64_t)j * w4) * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float4* out = reinterpret_cast<float4*>(gT + (int64_t)j * W + col);
  if (accumulate) acc = *out;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int b = j + 1 - k;
    if (b < 1 || b > K - 2) continue;       // (K = the stacked table's last row: inside it, knots that hold no edges have no segments)
    const int s0 = seg[b], s1 = seg[b + 1];
    for (int s = s0; s < s1; ++s) {
      const float4 v = *reinterpret_cast<const float4*>(P + ((int64_t)s * 4 + k) * W + col);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  *out = acc;
}

// ---- the table PACKED for the tensor-product kernels (round 5) -------------------------------------------------------------
// The in-kernel form of the tensor product (e3k_tp_fwd_table) gathers FOUR 4-byte values per (edge, weight) -- rows i-1 .. i+2 of T,
// 7.7 KB apart -- and is bound by what an L1 can fill from L2 (35 KB per edge).  The same cubic, written as
// its Taylor polynomial about the middle of the knot interval, needs its two leading coefficients in fp32 and the two small ones
// only in fp16: 12 bytes per (knot, weight) in ONE row, two loads per slot instead of four dword loads out of four rows, 23 KB per edge.
//     p_i(t) = a L0(t) + b L1(t) + c L2(t) + d L3(t),  (a, b, c, d) = T[i-1 .. i+2],  t in [0, 1) the offset inside interval i
//            = d0 + s (d1 + s (d2 + s d3)),            s = t - 1/2 in [-1/2, 1/2)
//     c0 = b, c1 = -a/3 - b/2 + c - d/6, c2 = a/2 - b + c/2, c3 = -a/6 + b/2 - c/2 + d/6          (monomials in t)
//     d0 = c0 + c1/2 + c2/4 + c3/8, d1 = c1 + c2 + 3/4 c3, d2 = c2 + 3/2 c3, d3 = c3
// formed in float64 from the fp32 rows and rounded ONCE.  Record: {d0: f32, d1: f32, (d2 * 2^10 : f16 | d3 * 2^16 : f16 << 16)};
// a table row holds its W (d0, d1) pairs first (8 W bytes), then its W f16 pairs (4 W bytes): the kernels read a slot with one
// dwordx2 and one dword load, each contiguous over the wave.  (One dwordx3 load of an interleaved record would be one instruction
// less; this compiler's __builtin_amdgcn_raw_buffer_load_b96 returns its third element as a copy of the first.)
// Sizes at 512 knots (h = 2^-7 A): d1 ~ 5e-2, d2 ~ 1e-3, d3 ~ 2e-5 of the values, so the fixed scales put d2 and d3 near 1 in
// fp16 (normal range 2^-14 .. 2^16: thirty binades either way; below it the ABSOLUTE error is < 2^-24 / scale, above it the entry
// is +-inf and the guard -- whose fourth-difference bound would have fired long before -- vetoes the table).  Rounding of the fp16
// pair (11 significant bits: relative 2^-11), after economisation (see the kernel): <= 2^-11 (|d2| / 8 + |d3| / 32): 6e-8 of the
// values at random init, beside 1e-7 of interpolation error and 6e-8 of fp32 rounding; e3k_rtable_guard bounds it (`pack_weight`).  Evaluation (same three fused
// multiply-adds in e3k_tp.hip and in rtable_interp_packed_kernel: bit-identical):
//     s = (c2 - c0 + 2 c3) - 1/2 from the edge's Lagrange weights (sum_k x_k L_k(t) = t for x = (-1, 0, 1, 2)),
//     w = fma(s, fma(s * 2^-10, fma(s * 2^-6, D3, D2), d1), d0).
// Knots outside [1, K - 2] are never an edge's knot (e3k_rtable_bins clamps): their records are zero.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
constexpr float PK_S2 = 1024.f, PK_S3 = 65536.f;      // scales of d2 and d3 in the record (powers of two: exact)

__global__ __launch_bounds__(256) void rtable_pack_kernel(const float* __restrict__ T, int32_t K, int32_t W, uint32_t* __restrict__ P) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= (int64_t)(K + 1) * W) return;
  const int i = (int)(q / W), col = (int)(q - (int64_t)i * W);
  uint32_t* row = P + 3 * (int64_t)i * W;      // [W][2] (d0, d1), then [W] f16 pairs
  if (i < 1 || i > K - 2) {
    row[2 * col] = row[2 * col + 1] = row[2 * W + col] = 0u;
    return;
  }
  const double a = T[q - W], b = T[q], c = T[q + W], d = T[q + 2 * (int64_t)W];
  const double c1 = -a / 3.0 - b / 2.0 + c - d / 6.0, c2 = a / 2.0 - b + c / 2.0, c3 = -a / 6.0 + b / 2.0 - c / 2.0 + d / 6.0;
  const double e2 = c2 + 1.5 * c3;
  f16x2 h;
  h.x = (_Float16)(float)(e2 * (double)PK_S2);
  h.y = (_Float16)(float)(c3 * (double)PK_S3);
  // what fp16 dropped, r2 s^2 + r3 s^3, is not lost but ECONOMISED into the fp32 coefficients: on |s| <= 1/2 the best constant for
  // s^2 is 1/8 and the best multiple of s for s^3 is 3/16 s (Chebyshev), leaving |r2| / 8 + |r3| / 32 instead of |r2| / 4 + |r3| / 8
  const double r2 = e2 - (double)(float)h.x / (double)PK_S2, r3 = c3 - (double)(float)h.y / (double)PK_S3;
  const float d0 = (float)(b + c1 / 2.0 + c2 / 4.0 + c3 / 8.0 + r2 / 8.0), d1 = (float)(c1 + c2 + 0.75 * c3 + 3.0 * r3 / 16.0);
  row[2 * col] = __float_as_uint(d0);
  row[2 * col + 1] = __float_as_uint(d1);
  row[2 * W + col] = __builtin_bit_cast(uint32_t, h);
}

// the tables of a radial stack (one row count, up to 16 widths) packed by ONE launch: blockIdx.y = table
struct PackMulti {
  const float* T[16];
  uint32_t* P[16];
  int32_t W[16];
};
__global__ __launch_bounds__(256) void rtable_pack_multi_kernel(PackMulti a, int32_t K) {
  const int tb = blockIdx.y;
  const int32_t W = a.W[tb];
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= (int64_t)(K + 1) * W) return;
  const float* __restrict__ T = a.T[tb];
  const int i = (int)(q / W), col = (int)(q - (int64_t)i * W);
  uint32_t* row = a.P[tb] + 3 * (int64_t)i * W;
  if (i < 1 || i > K - 2) {
    row[2 * col] = row[2 * col + 1] = row[2 * W + col] = 0u;
    return;
  }
  // (the arithmetic of rtable_pack_kernel, line for line: the two must produce the same bits)
  const double ta = T[q - W], b = T[q], c = T[q + W], d = T[q + 2 * (int64_t)W];
  const double c1 = -ta / 3.0 - b / 2.0 + c - d / 6.0, c2 = ta / 2.0 - b + c / 2.0, c3 = -ta / 6.0 + b / 2.0 - c / 2.0 + d / 6.0;
  const double e2 = c2 + 1.5 * c3;
  f16x2 h;
  h.x = (_Float16)(float)(e2 * (double)PK_S2);
  h.y = (_Float16)(float)(c3 * (double)PK_S3);
  const double r2 = e2 - (double)(float)h.x / (double)PK_S2, r3 = c3 - (double)(float)h.y / (double)PK_S3;
  const float d0 = (float)(b + c1 / 2.0 + c2 / 4.0 + c3 / 8.0 + r2 / 8.0), d1 = (float)(c1 + c2 + 0.75 * c3 + 3.0 * r3 / 16.0);
  row[2 * col] = __float_as_uint(d0);
  row[2 * col + 1] = __float_as_uint(d1);
  row[2 * W + col] = __builtin_bit_cast(uint32_t, h);
}

__device__ __forceinline__ float packed_eval(float s, float s2, float s3, float d0, float d1, uint32_t pk) {
  const f16x2 h = __builtin_bit_cast(f16x2, pk);
  return fmaf(s, fmaf(s2, fmaf(s3, (float)h.y, (float)h.x), d1), d0);
}

// w[e, :] from the packed table (the materialised counterpart of the packed in-kernel form: tests pin the two to the same bits)
__global__ __launch_bounds__(256) void rtable_interp_packed_kernel(const uint32_t* __restrict__ P, const int32_t* __restrict__ perm,
                                                                   const int32_t* __restrict__ bin, const float* __restrict__ coef,
                                                                   int64_t E, int32_t W, float* __restrict__ w) {
  const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= E) return;
  const int lane = threadIdx.x & 63;
  const int e = uniform(perm[p]);
  const int i = uniform(bin[e]);
  const float* __restrict__ cp = coef + 4 * (int64_t)e;
  const float c0 = __uint_as_float(uniform((int)__float_as_uint(cp[0]))), c2 = __uint_as_float(uniform((int)__float_as_uint(cp[2])));
  const float c3 = __uint_as_float(uniform((int)__float_as_uint(cp[3])));
  const float s = fmaf(2.f, c3, c2 - c0) - 0.5f, s2 = s * (1.f / PK_S2), s3 = s * (PK_S2 / PK_S3);
  const uint32_t* __restrict__ row = P + 3 * (int64_t)i * W;
  float* __restrict__ o = w + (int64_t)e * W;
  for (int q = lane; q < W; q += 64) o[q] = packed_eval(s, s2, s3, __uint_as_float(row[2 * q]), __uint_as_float(row[2 * q + 1]), row[2 * W + q]);
}

// ---- a-posteriori bound of the interpolation error, on the device, per weight COLUMN ---------------------------------------
// Cubic Lagrange interpolation on knots h apart is off by at most 3/128 h^4 max|f|; on the table h^4 f is the fourth
// difference, so for column c
//     err_c <= 3/128 max_i |T[i+4,c] - 4 T[i+3,c] + 6 T[i+2,c] - 4 T[i+1,c] + T[i,c]|
// (+ for the PACKED table the fp16 rounding of its two small coefficients: 2^-11 (|2nd difference| / 16 + |3rd difference| / 192))
// and TWO ratios are formed: the table-wide one, max_c err_c / max|T| (what the forward's parity feels: every column's error against
// the scale of the weights it is summed with), and the per-column one, max_c err_c / max(max_i |T[i,c]|, floor * max|T|) -- a column's
// error against its OWN scale (a column a thousand times smaller than the largest must not hide a thousand times the relative
// error behind it), floored at `floor` of the table's largest entry (a column that contributes nothing has no relative error
// worth a veto).  Random combinations of the hidden units that cancel their smooth part have 20-40 x the typical relative
// curvature (measured at random init, 2-3 k columns: per-column 1-4e-6 where the table-wide ratio is 1.1-1.8e-7), so the two have
// tolerances of their own; the launch reports est = max(table-wide, col_weight * per-column) with col_weight = the tolerances'
// ratio, to be compared with the table-wide tolerance.  One launch for up to 16 tables (the
// radial stack's layers): a workgroup = 64 columns x 4 row quarters; the LAST workgroup of a table (ticket counter in the
// table's state) reduces the per-column maxima to the estimate.  state [4]: [0] running maximum of the estimate since the host
// last reset it (what a replayed HIP graph leaves behind: the captured step never re-enters Python), [1] the estimate of this
// launch, [2] unused, [3] the per-column ratio of this launch (un-weighted).  A non-finite table entry
// gives +inf.
struct GuardArgs {
  const float* T[16];
  float* state[16];
  float* scratch[16];     // [2 W]: column maxima of |T| and of the error bound
  int32_t W[16];
  int32_t rows;
  float floor_rel, c4, col_weight, pack_weight;      // pack_weight: 1 when the kernels read the PACKED table (fp16 rounding of d2, d3), else 0
};

__device__ __forceinline__ void atomic_max_pos(float* p, float v) {      // v >= 0 (or +inf): the bit patterns order like the values
  atomicMax(reinterpret_cast<unsigned int*>(p), __float_as_uint(v));
}

constexpr int RT_GUARD_PARTS = 8;      // row ranges per table column = 4 waves x this many workgroups (e3k.h: scratch size)
// pass 1: per (table, 64-column chunk, row range) the column maxima of |T| and of the error bound -> scratch [PARTS][2][W].
// (Round 5 first had ONE kernel whose last workgroup per table -- found with a ticket counter behind __threadfence() -- did pass 2:
//  97 us for five tables of 641 x 1 920.  A device-scope fence on this multi-XCD part writes an L2's dirty lines back, and the tables
//  had just been written: 1 200 workgroups each paid for it.  Two launches, no fence: the kernel boundary publishes the scratch.)
__global__ __launch_bounds__(256) void rtable_guard_kernel(GuardArgs a) {
  __shared__ float s_col[4][64], s_d4[4][64];
  const int tb = blockIdx.y;
  const int W = a.W[tb];
  const int n_chunks = (W + 63) / 64;
  if ((int)blockIdx.x >= n_chunks) return;
  const float* __restrict__ T = a.T[tb];
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  const int rows = a.rows;
  // row range (blockIdx.z, q) of 4 RT_GUARD_PARTS covers fourth differences starting at rows [lo, hi): reads rows lo .. hi + 3
  const int n_d4 = rows - 4;
  const int per = (n_d4 + 4 * RT_GUARD_PARTS - 1) / (4 * RT_GUARD_PARTS);
  const int rq = blockIdx.z * 4 + q;
  const int lo = rq * per < n_d4 ? rq * per : n_d4, hi = (lo + per < n_d4) ? lo + per : n_d4;
  float cmax = 0.f, dmax = 0.f;
  bool bad = false;
  if (col < W && lo < hi) {
    float v0 = T[(int64_t)lo * W + col], v1 = T[(int64_t)(lo + 1) * W + col], v2 = T[(int64_t)(lo + 2) * W + col],
          v3 = T[(int64_t)(lo + 3) * W + col];
    cmax = fmaxf(fmaxf(fabsf(v0), fabsf(v1)), fmaxf(fabsf(v2), fabsf(v3)));
    bad = !(fabsf(v0) < INFINITY) || !(fabsf(v1) < INFINITY) || !(fabsf(v2) < INFINITY) || !(fabsf(v3) < INFINITY);
    for (int i = lo; i < hi; ++i) {
      const float v4 = T[(int64_t)(i + 4) * W + col];
      bad = bad || !(fabsf(v4) < INFINITY);
      cmax = fmaxf(cmax, fabsf(v4));
      const float d4 = (v4 + v0) - 4.f * (v3 + v1) + 6.f * v2;
      // packed table: d2 ~ (second difference) / 2 and d3 ~ (third difference) / 6 are stored in fp16 (relative rounding 2^-11); what
      // is dropped enters economised: |r2| / 8 + |r3| / 32 (e3k::rtable_pack_kernel)
      const float d2 = (v1 + v3) - 2.f * v2, d3 = (v3 - v0) - 3.f * (v2 - v1);
      dmax = fmaxf(dmax, a.c4 * fabsf(d4) + a.pack_weight * (1.f / 2048.f) * (fabsf(d2) * (1.f / 16.f) + fabsf(d3) * (1.f / 192.f)));
      v0 = v1; v1 = v2; v2 = v3; v3 = v4;
    }
  }
  s_col[q][lane] = bad ? INFINITY : cmax;
  s_d4[q][lane] = bad ? INFINITY : dmax;
  __syncthreads();
  if (q == 0 && col < W) {
    float* __restrict__ mine = a.scratch[tb] + (int64_t)blockIdx.z * 2 * W;      // [RT_GUARD_PARTS][2][W]
    mine[col] = fmaxf(fmaxf(s_col[0][lane], s_col[1][lane]), fmaxf(s_col[2][lane], s_col[3][lane]));
    mine[W + col] = fmaxf(fmaxf(s_d4[0][lane], s_d4[1][lane]), fmaxf(s_d4[2][lane], s_d4[3][lane]));
  }
}

// pass 2: one workgroup per table: fold the row ranges per column, the table's largest entry, then the worst column
__global__ __launch_bounds__(256) void rtable_guard_reduce_kernel(GuardArgs a) {
  __shared__ float s_red[4], s_red2[4];
  const int tb = blockIdx.x;
  const int W = a.W[tb];
  const float* __restrict__ scratch = a.scratch[tb];
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  float g = 0.f;
  auto fold = [&](int c, float& cm, float& dm) {
    cm = 0.f, dm = 0.f;
#pragma unroll
    for (int z = 0; z < RT_GUARD_PARTS; ++z) {
      cm = fmaxf(cm, scratch[(int64_t)z * 2 * W + c]);
      dm = fmaxf(dm, scratch[(int64_t)z * 2 * W + W + c]);
    }
  };
  for (int c = threadIdx.x; c < W; c += 256) {
    float cm, dm;
    fold(c, cm, dm);
    g = fmaxf(g, cm);
  }
  g = wave_max_f(g);
  if (lane == 0) s_red[q] = g;
  __syncthreads();
  g = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
  const float fl = fmaxf(a.floor_rel * g, 1e-30f), gs = fmaxf(g, 1e-30f);
  float est_c = 0.f, est_g = 0.f;
  for (int c = threadIdx.x; c < W; c += 256) {      // (folded again rather than kept: no limit on the width, the scratch is an L2 hit)
    float cm, dm;
    fold(c, cm, dm);
    const bool fin = dm < INFINITY && cm < INFINITY;
    est_c = fmaxf(est_c, fin ? dm / fmaxf(cm, fl) : INFINITY);
    est_g = fmaxf(est_g, fin ? dm / gs : INFINITY);
  }
  est_c = wave_max_f(est_c);
  est_g = wave_max_f(est_g);
  __syncthreads();
  if (lane == 0) s_red[q] = est_c, s_red2[q] = est_g;
  __syncthreads();
  if (threadIdx.x == 0) {
    est_c = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    est_g = fmaxf(fmaxf(s_red2[0], s_red2[1]), fmaxf(s_red2[2], s_red2[3]));
    const float est = fmaxf(est_g, a.col_weight * est_c);
    a.state[tb][1] = est;
    a.state[tb][3] = est_c;
    atomic_max_pos(a.state[tb], est);
  }
}

}  // namespace e3k

extern "C" int e3k_rtable_guard(const float* const* tables, float* const* states, float* const* scratch, const int32_t* widths,
                                int32_t n, int32_t rows, float floor_rel, float col_weight, int32_t packed, void* stream) {
  if (!tables || !states || !scratch || !widths || n <= 0 || n > 16 || rows < 0 || !(floor_rel >= 0.f) || !(col_weight >= 0.f))
    return E3K_ERR_INVALID;
  if (rows < 5) return E3K_OK;      // no fourth difference to look at
  e3k::GuardArgs a{};
  int wmax = 0;
  for (int i = 0; i < n; ++i) {
    if (!tables[i] || !states[i] || !scratch[i] || widths[i] <= 0) return E3K_ERR_INVALID;
    a.T[i] = tables[i]; a.state[i] = states[i]; a.scratch[i] = scratch[i]; a.W[i] = widths[i];
    wmax = widths[i] > wmax ? widths[i] : wmax;
  }
  a.rows = rows; a.floor_rel = floor_rel; a.c4 = 3.0f / 128.0f; a.col_weight = col_weight; a.pack_weight = packed ? 1.f : 0.f;
  hipLaunchKernelGGL(e3k::rtable_guard_kernel, dim3((unsigned)((wmax + 63) / 64), (unsigned)n, e3k::RT_GUARD_PARTS), dim3(256), 0,
                     (hipStream_t)stream, a);
  hipLaunchKernelGGL(e3k::rtable_guard_reduce_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, a);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int e3k_rtable_pack(const float* T, int32_t K, int32_t W, void* P, void* stream) {
  if (K < 4 || W <= 0) return E3K_ERR_INVALID;
  if (!T || !P) return E3K_ERR_INVALID;
  const int64_t q = (int64_t)(K + 1) * W;
  hipLaunchKernelGGL(e3k::rtable_pack_kernel, dim3((unsigned)((q + 255) / 256)), dim3(256), 0, (hipStream_t)stream, T, K, W,
                     static_cast<uint32_t*>(P));
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int e3k_rtable_pack_multi(const float* const* T, int32_t K, const int32_t* W, void* const* P, int32_t n, void* stream) {
  if (K < 4 || n < 0 || n > 16) return E3K_ERR_INVALID;
  if (n == 0) return E3K_OK;
  if (!T || !W || !P) return E3K_ERR_INVALID;
  e3k::PackMulti a{};
  int32_t wmax = 0;
  for (int i = 0; i < n; ++i) {
    if (!T[i] || !P[i] || W[i] <= 0) return E3K_ERR_INVALID;
    a.T[i] = T[i];
    a.P[i] = static_cast<uint32_t*>(P[i]);
    a.W[i] = W[i];
    wmax = W[i] > wmax ? W[i] : wmax;
  }
  const int64_t q = (int64_t)(K + 1) * wmax;
  hipLaunchKernelGGL(e3k::rtable_pack_multi_kernel, dim3((unsigned)((q + 255) / 256), (unsigned)n), dim3(256), 0, (hipStream_t)stream, a, K);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int e3k_rtable_interp_packed(const void* P, const int32_t* bin_perm, const int32_t* bin, const float* coef, int64_t E,
                                        int32_t K, int32_t W, float* w, void* stream) {
  if (E < 0 || K < 4 || W <= 0) return E3K_ERR_INVALID;
  if (E == 0) return E3K_OK;
  if (!P || !bin_perm || !bin || !coef || !w) return E3K_ERR_INVALID;
  hipLaunchKernelGGL(e3k::rtable_interp_packed_kernel, dim3((unsigned)((E + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     static_cast<const uint32_t*>(P), bin_perm, bin, coef, E, W, w);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int64_t e3k_rtable_bins_workspace_ints(int64_t E, int32_t K) {
  const int64_t n_chunks = (E + e3k::RT_CHUNK - 1) / e3k::RT_CHUNK;
  return E + n_chunks * ((int64_t)K + 1);      // [rank inside (chunk, knot) | chunk x knot counts -> offsets]
}

static int rtable_bins_impl(const float* r, const int64_t* key, int32_t n_keys, int64_t E, float h_inv, int32_t K, int32_t* bin,
                            float* coef, int32_t* bin_ptr, int32_t* bin_seg, int32_t* bin_perm, int32_t* workspace, int32_t* bad_flag,
                            void* stream) {
  // h_inv = 1 / knot spacing; a power of two makes x = r * h_inv and the offset t = x - floor(x) exact (the caller's choice:
  // backend/radial_table.py lays its tables out that way)
  if (E < 0 || K < 4 || !(h_inv > 0.f) || n_keys < 1) return E3K_ERR_INVALID;
  const int64_t KT64 = (int64_t)n_keys * (K + 1) - 1;      // last row of the stacked table
  if (KT64 > 4000 || E >= 0x7fffffffLL) return E3K_ERR_UNSUPPORTED;       // 4 x (KT + 1) int32 of LDS per workgroup
  const int32_t KT = (int32_t)KT64;
  if (!bin_ptr || !bin_seg) return E3K_ERR_INVALID;
  if (E > 0 && (!r || !bin || !coef || !bin_perm || !workspace)) return E3K_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const int64_t n_chunks = (E + e3k::RT_CHUNK - 1) / e3k::RT_CHUNK;
  int32_t* lrank = workspace;
  int32_t* chunk_hist = workspace + E;
  if (E > 0)
    hipLaunchKernelGGL(e3k::rtable_bins_rank_kernel, dim3((unsigned)n_chunks), dim3(256), sizeof(int32_t) * 4 * (KT + 1), st, r, key, E,
                       h_inv, K, KT, bin, coef, lrank, chunk_hist, bad_flag);
  hipLaunchKernelGGL(e3k::rtable_bins_scan_kernel, dim3(1), dim3(1024), 0, st, chunk_hist, (int32_t)n_chunks, KT, bin_ptr, bin_seg);
  if (E > 0)
    hipLaunchKernelGGL(e3k::rtable_bins_place_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, st, bin, lrank, chunk_hist,
                       bin_ptr, E, KT, bin_perm);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int e3k_rtable_bins(const float* r, int64_t E, float h_inv, int32_t K, int32_t* bin, float* coef,
                               int32_t* bin_ptr, int32_t* bin_seg, int32_t* bin_perm, int32_t* workspace, void* stream) {
  return rtable_bins_impl(r, nullptr, 1, E, h_inv, K, bin, coef, bin_ptr, bin_seg, bin_perm, workspace, nullptr, stream);
}

// n_keys tables of K + 1 rows stacked: bin[e] = key[e] (K + 1) + i; bin_ptr / bin_seg [n_keys (K + 1) + 1]; workspace:
// e3k_rtable_bins_workspace_ints(E, n_keys (K + 1) - 1)
extern "C" int e3k_rtable_bins_keyed(const float* r, const int64_t* key, int32_t n_keys, int64_t E, float h_inv, int32_t K,
                                     int32_t* bin, float* coef, int32_t* bin_ptr, int32_t* bin_seg, int32_t* bin_perm,
                                     int32_t* workspace, int32_t* bad_flag, void* stream) {
  if (E > 0 && !key) return E3K_ERR_INVALID;
  return rtable_bins_impl(r, key, n_keys, E, h_inv, K, bin, coef, bin_ptr, bin_seg, bin_perm, workspace, bad_flag, stream);
}

extern "C" int e3k_rtable_interp_fwd(const float* T, const int32_t* bin_perm, const int32_t* bin, const float* coef, int64_t E,
                                     int32_t K, int32_t W, float* w, void* stream) {
  if (E < 0 || K < 4 || W <= 0) return E3K_ERR_INVALID;
  if (W % 4) return E3K_ERR_UNSUPPORTED;
  if (E == 0) return E3K_OK;
  if (!T || !bin_perm || !bin || !coef || !w) return E3K_ERR_INVALID;
  hipLaunchKernelGGL(e3k::rtable_interp_fwd_kernel, dim3((unsigned)((E + 3) / 4)), dim3(256), 0, (hipStream_t)stream, T, bin_perm,
                     bin, coef, E, W, w, (const float*)nullptr, (float*)nullptr);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

// two tables (same shape) through the same weights in one pass: w = I(coef) T, w2 = I(coef) T2
extern "C" int e3k_rtable_interp_fwd2(const float* T, const float* T2, const int32_t* bin_perm, const int32_t* bin, const float* coef,
                                      int64_t E, int32_t K, int32_t W, float* w, float* w2, void* stream) {
  if (E < 0 || K < 4 || W <= 0) return E3K_ERR_INVALID;
  if (W % 4) return E3K_ERR_UNSUPPORTED;
  if (E == 0) return E3K_OK;
  if (!T || !T2 || !bin_perm || !bin || !coef || !w || !w2) return E3K_ERR_INVALID;
  hipLaunchKernelGGL(e3k::rtable_interp_fwd_kernel, dim3((unsigned)((E + 3) / 4)), dim3(256), 0, (hipStream_t)stream, T, bin_perm,
                     bin, coef, E, W, w, T2, w2);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

static inline int64_t rtable_seg_cap(int64_t E, int32_t K) { return E / e3k::RT_SEG + (int64_t)K + 2; }

extern "C" int64_t e3k_rtable_bwd_workspace_floats(int64_t E, int32_t K, int32_t W) { return rtable_seg_cap(E, K) * 4 * W; }

extern "C" int e3k_rtable_interp_bwd(const float* g_w, const float* coef, const float* scale,
                                     const int32_t* bin_ptr, const int32_t* bin_seg, const int32_t* bin_perm, int64_t E, int32_t K,
                                     int32_t W, float* workspace, float* g_T, int32_t accumulate, void* stream) {
  if (E < 0 || K < 4 || W <= 0) return E3K_ERR_INVALID;
  if (W % 4) return E3K_ERR_UNSUPPORTED;
  if (!g_T || !bin_ptr || !bin_seg || !workspace || (E > 0 && (!g_w || !coef || !bin_perm))) return E3K_ERR_INVALID;
  const int n_chunks = (W + 255) / 256;
  const int64_t cap = rtable_seg_cap(E, K);
  const int64_t items = cap * n_chunks;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(e3k::rtable_bwd_partial_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, g_w, coef, scale,
                     bin_ptr, bin_seg, bin_perm, K, W, n_chunks, cap, workspace);
  const int64_t q = (int64_t)(K + 1) * (W / 4);
  hipLaunchKernelGGL(e3k::rtable_bwd_combine_kernel, dim3((unsigned)((q + 255) / 256)), dim3(256), 0, st, workspace, bin_seg, K, W,
                     accumulate, g_T);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}
