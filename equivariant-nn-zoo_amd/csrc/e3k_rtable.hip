// Radial weights through a knot table for gfx950.
//
// Replaces, per convolution layer of the reference (paths relative to /root/reference):
//   weight = self.fc(edge_radial)        e3_layers/nn/message_passing.py:74-79,93
// when edge_radial is RadialBasisEncoding(edge_length) (e3_layers/nn/embedding.py:210-219): the per-edge path weights
// are then a smooth function of ONE scalar, w[e, :] = f(r_e) with f = fc o basis o cutoff : [0, r_max] -> R^W.
// The reference (and e3k_gemm) evaluates f per edge -- 2 * 64 * W flops per edge forward and twice that backward, the
// largest block of matrix work of a training step.  Here f is evaluated on K + 1 equidistant knots (K = 4096: 17x fewer
// rows than a 256-molecule batch has edges) by the same MLP kernels, and every edge interpolates the three knots around
// it with the quadratic Lagrange weights
//     i = round(r / h), t = r / h - i,   w[e] = t(t-1)/2 T[i-1] + (1 - t^2) T[i] + t(t+1)/2 T[i+1].
// f is smooth (sines over r times a polynomial envelope through an MLP with ssp activations): the interpolation error is
// ~h^3 f''' / 16 -- measured 5e-9 relative for K = 4096, below the 2.5e-8 rounding error of evaluating f in fp32
// (tests/test_gpu_ops.py::test_radial_table_matches_the_per_edge_mlp).  r >= r_max maps to the last knot, where the
// envelope (and every derivative up to the 5th) is zero: f is constant there, exactly.
// The backward is the transpose: g_T[j] = sum over the edges whose stencil holds knot j of their weight times g_w[e] --
// edges are sorted by centre knot once per batch (the CSR build of e3k_graph.hip with the knot as "node"), so row j is
// a deterministic ordered sum over three consecutive bins; no atomics.  The MLP's own backward then runs on K + 1 rows.
#include "e3k_common.h"

namespace e3k {

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nt_store4(float4* p, const float4& v) {
  f32x4 t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(p));
}
__device__ __forceinline__ float4 nt_load4(const float4* p) {
  const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  return make_float4(t.x, t.y, t.z, t.w);
}

// centre knot and offset of every edge
__global__ __launch_bounds__(256) void rtable_bin_kernel(const float* __restrict__ r, int64_t E, float h_inv, int32_t K,
                                                         int64_t* __restrict__ bin2, float* __restrict__ t_out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  float x = r[e] * h_inv;
  x = x < 0.f ? 0.f : (x > (float)K ? (float)K : x);     // (NaN radii fall through to knot 1 with t = NaN: the weights are NaN)
  int i = (int)(x + 0.5f);
  i = i < 1 ? 1 : (i > K - 1 ? K - 1 : i);
  bin2[e] = i;            // [2, E] int64 "edge_index" (both rows the knot): the CSR builder groups edges by centre knot
  bin2[E + e] = i;
  t_out[e] = x - (float)i;
}

// w[e, :] = c_-1 T[i-1, :] + c_0 T[i, :] + c_+1 T[i+1, :];  one wave per edge, 16-byte columns.  Edges are taken in
// KNOT order (perm): the ~17 edges of a knot, handled by neighbouring waves, read the same three table rows, which
// then come from L1/L2 instead of 23 KB per edge from the Infinity Cache (measured 255 -> see DESIGN.md us per layer).
__global__ __launch_bounds__(256) void rtable_interp_fwd_kernel(const float* __restrict__ T, const int32_t* __restrict__ perm,
                                                                const int32_t* __restrict__ bin, const float* __restrict__ tt,
                                                                int64_t E, int32_t W, float* __restrict__ w) {
  const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= E) return;
  const int lane = threadIdx.x & 63;
  const int e = uniform(perm[p]);
  const int i = uniform(bin[e]);
  const float t = __uint_as_float(uniform((int)__float_as_uint(tt[e])));
  const float cm = 0.5f * t * (t - 1.f), c0 = 1.f - t * t, cp = 0.5f * t * (t + 1.f);
  const float4* __restrict__ a = reinterpret_cast<const float4*>(T + (int64_t)(i - 1) * W);
  const float4* __restrict__ b = reinterpret_cast<const float4*>(T + (int64_t)i * W);
  const float4* __restrict__ c = reinterpret_cast<const float4*>(T + (int64_t)(i + 1) * W);
  float4* __restrict__ o = reinterpret_cast<float4*>(w + (int64_t)e * W);
  for (int q = lane; q < (W >> 2); q += 64) {
    const float4 va = a[q], vb = b[q], vc = c[q];
    float4 v;
    v.x = fmaf(cp, vc.x, fmaf(c0, vb.x, cm * va.x));
    v.y = fmaf(cp, vc.y, fmaf(c0, vb.y, cm * va.y));
    v.z = fmaf(cp, vc.z, fmaf(c0, vb.z, cm * va.z));
    v.w = fmaf(cp, vc.w, fmaf(c0, vb.w, cm * va.w));
    nt_store4(o + q, v);      // written once, read by the edge kernels later: streamed past the caches
  }
}

// backward, pass 1: one wave per (knot bin b, 256-column chunk) reads the g_w rows of the bin's edges ONCE (ascending edge
// id) and forms their three weighted sums -- the contributions of bin b to the table rows b-1, b, b+1:  P[b][0..2][cols]
__global__ __launch_bounds__(256) void rtable_bwd_partial_kernel(const float* __restrict__ gw, const int32_t* __restrict__ ptr,
                                                                 const int32_t* __restrict__ perm, const float* __restrict__ tt,
                                                                 int32_t K, int32_t W, int32_t n_chunks, float* __restrict__ P) {
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= (int64_t)(K + 1) * n_chunks) return;
  const int b = (int)(item / n_chunks), chunk = (int)(item - (int64_t)b * n_chunks);
  const int lane = threadIdx.x & 63;
  const int col = chunk * 256 + lane * 4;
  if (col >= W) return;
  float4 am = make_float4(0.f, 0.f, 0.f, 0.f), a0 = am, ap = am;
  const int beg = uniform(ptr[b]), end = uniform(ptr[b + 1]);
  for (int p = beg; p < end; ++p) {
    const int e = uniform(perm[p]);
    const float t = __uint_as_float(uniform((int)__float_as_uint(tt[e])));
    const float cm = 0.5f * t * (t - 1.f), c0 = 1.f - t * t, cp = 0.5f * t * (t + 1.f);
    const float4 g = nt_load4(reinterpret_cast<const float4*>(gw + (int64_t)e * W + col));      // read once
    am.x = fmaf(cm, g.x, am.x); am.y = fmaf(cm, g.y, am.y); am.z = fmaf(cm, g.z, am.z); am.w = fmaf(cm, g.w, am.w);
    a0.x = fmaf(c0, g.x, a0.x); a0.y = fmaf(c0, g.y, a0.y); a0.z = fmaf(c0, g.z, a0.z); a0.w = fmaf(c0, g.w, a0.w);
    ap.x = fmaf(cp, g.x, ap.x); ap.y = fmaf(cp, g.y, ap.y); ap.z = fmaf(cp, g.z, ap.z); ap.w = fmaf(cp, g.w, ap.w);
  }
  float* row = P + (int64_t)b * 3 * W + col;
  *reinterpret_cast<float4*>(row) = am;
  *reinterpret_cast<float4*>(row + W) = a0;
  *reinterpret_cast<float4*>(row + 2 * W) = ap;
}

// pass 2: g_T[j] = P[j+1][0] + P[j][1] + P[j-1][2]  (a fixed order: deterministic), one thread per 16 bytes
__global__ __launch_bounds__(256) void rtable_bwd_combine_kernel(const float* __restrict__ P, int32_t K, int32_t W,
                                                                 float* __restrict__ gT) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int w4 = W >> 2;
  if (q >= (int64_t)(K + 1) * w4) return;
  const int j = (int)(q / w4), col = (int)(q - (int64_t)j * w4) * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  auto add = [&](int b, int slot) {
    if (b < 1 || b > K - 1) return;       // only these bins hold edges
    const float4 v = *reinterpret_cast<const float4*>(P + ((int64_t)b * 3 + slot) * W + col);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  };
  add(j + 1, 0);
  add(j, 1);
  add(j - 1, 2);
  *reinterpret_cast<float4*>(gT + (int64_t)j * W + col) = acc;
}

}  // namespace e3k

extern "C" int e3k_rtable_bin(const float* r, int64_t E, float r_max, int32_t K, int64_t* bin2, float* t, void* stream) {
  if (E < 0 || K < 4 || !(r_max > 0.f)) return E3K_ERR_INVALID;
  if (E == 0) return E3K_OK;
  if (!r || !bin2 || !t) return E3K_ERR_INVALID;
  hipLaunchKernelGGL(e3k::rtable_bin_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, (hipStream_t)stream, r, E,
                     (float)K / r_max, K, bin2, t);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int e3k_rtable_interp_fwd(const float* T, const int32_t* bin_perm, const int32_t* bin, const float* t, int64_t E,
                                     int32_t K, int32_t W, float* w, void* stream) {
  if (E < 0 || K < 4 || W <= 0) return E3K_ERR_INVALID;
  if (W % 4) return E3K_ERR_UNSUPPORTED;
  if (E == 0) return E3K_OK;
  if (!T || !bin_perm || !bin || !t || !w) return E3K_ERR_INVALID;
  hipLaunchKernelGGL(e3k::rtable_interp_fwd_kernel, dim3((unsigned)((E + 3) / 4)), dim3(256), 0, (hipStream_t)stream, T, bin_perm,
                     bin, t, E, W, w);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int64_t e3k_rtable_bwd_workspace_floats(int32_t K, int32_t W) { return (int64_t)(K + 1) * 3 * W; }

// second pass alone (the first one done elsewhere: e3k_tp_bwd_table_partial)
extern "C" int e3k_rtable_bwd_combine(const float* P, int32_t K, int32_t W, float* g_T, void* stream) {
  if (K < 4 || W <= 0 || !P || !g_T) return E3K_ERR_INVALID;
  if (W % 4) return E3K_ERR_UNSUPPORTED;
  const int64_t q = (int64_t)(K + 1) * (W / 4);
  hipLaunchKernelGGL(e3k::rtable_bwd_combine_kernel, dim3((unsigned)((q + 255) / 256)), dim3(256), 0, (hipStream_t)stream, P, K, W, g_T);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int e3k_rtable_interp_bwd(const float* g_w, const int32_t* bin_ptr, const int32_t* bin_perm, const float* t,
                                     int64_t E, int32_t K, int32_t W, float* workspace, float* g_T, void* stream) {
  if (E < 0 || K < 4 || W <= 0) return E3K_ERR_INVALID;
  if (W % 4) return E3K_ERR_UNSUPPORTED;
  if (!g_T || !bin_ptr || !workspace || (E > 0 && (!g_w || !bin_perm || !t))) return E3K_ERR_INVALID;
  const int n_chunks = (W + 255) / 256;
  const int64_t items = (int64_t)(K + 1) * n_chunks;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(e3k::rtable_bwd_partial_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, g_w, bin_ptr, bin_perm,
                     t, K, W, n_chunks, workspace);
  const int64_t q = (int64_t)(K + 1) * (W / 4);
  hipLaunchKernelGGL(e3k::rtable_bwd_combine_kernel, dim3((unsigned)((q + 255) / 256)), dim3(256), 0, st, workspace, K, W, g_T);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}
