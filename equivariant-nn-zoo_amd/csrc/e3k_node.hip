// Node-side elementwise kernels for gfx950: activations, layout change, Gate, per-block RMS
// normalisation, sorted-segment sum.
//
// Replaces (paths relative to /root/reference):
//   normalize2mom(act) inside FullyConnectedNet  e3_layers/nn/message_passing.py:74-79
//   e3nn.nn.Gate                                 e3_layers/nn/message_passing.py:195-205,249
//   LayerNormalization                           e3_layers/nn/pointwise.py:32-51
//   Pooling (scatter over _node_segment)         e3_layers/nn/output.py:66-74
#include "e3k_common.h"
#include "e3k_act.h"

namespace e3k {

__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, int64_t n, int act, float cst,
                                                       float* __restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    y[i] = cst * act_f(act, x[i]);
}
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                       int64_t n, int act, float cst, float* __restrict__ gx) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    gx[i] = gy[i] * cst * act_df(act, x[i]);
}

// backward of act_bwd: gx = gy*cst*act'(x) with cotangent gh  ->  g_gy = gh*cst*act'(x), g_x = gh*gy*cst*act''(x)
__global__ __launch_bounds__(256) void act_bwd2_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                        const float* __restrict__ gh, int64_t n, int act, float cst,
                                                        float* __restrict__ g_gy, float* __restrict__ g_x) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float xv = x[i], h = gh[i] * cst;
    if (g_gy) g_gy[i] = h * act_df(act, xv);
    if (g_x) g_x[i] = h * gy[i] * act_d2f(act, xv);
  }
}

__global__ __launch_bounds__(256) void act_bwd_out_kernel(const float* __restrict__ y, const float* __restrict__ gy,
                                                           int64_t n, float cst, float* __restrict__ gx) {
  // ssp: y = cst*(softplus(x) - ln 2)  =>  cst*sigmoid(x) = cst*(1 - 0.5*exp(-y/cst))
  const float inv = 1.0f / cst;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    gx[i] = gy[i] * cst * (1.0f - 0.5f * expf(-y[i] * inv));
}

// ---------------------------------------------------------------------------------------
// relayout
// ---------------------------------------------------------------------------------------
constexpr int MAXBLK = 16;
struct BlockArgs {
  int n;
  e3k_block b[MAXBLK];
};

// Column-fixed threads: the output column -> source column map is the same for every row, so a thread resolves its
// block once and then walks rows (grid.y strides).  The first version recomputed a 64-bit division and the block search
// per element: 54 us for a [4.6 k, 1152] tensor that moves 43 MB.
__global__ __launch_bounds__(256) void relayout_kernel(const float* __restrict__ x, int64_t rows, int row_dim,
                                                        BlockArgs ba, int to_cf, float* __restrict__ y) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= row_dim) return;
  int src = c;
  for (int k = 0; k < ba.n; ++k) {
    const e3k_block& b = ba.b[k];
    const int rel = c - b.off;
    if (rel >= 0 && rel < b.mul * b.dim) {
      if (to_cf) {  // out [dim][mul] <- in [mul][dim]
        const int m = rel / b.mul, u = rel - m * b.mul;
        src = b.off + u * b.dim + m;
      } else {      // out [mul][dim] <- in [dim][mul]
        const int u = rel / b.dim, m = rel - u * b.dim;
        src = b.off + m * b.mul + u;
      }
      break;
    }
  }
  const float* __restrict__ xp = x + src;
  float* __restrict__ yp = y + c;
  const int64_t step = gridDim.y;
  int64_t r = blockIdx.y;
  for (; r + 3 * step < rows; r += 4 * step) {
    const float v0 = xp[r * row_dim], v1 = xp[(r + step) * row_dim], v2 = xp[(r + 2 * step) * row_dim],
                v3 = xp[(r + 3 * step) * row_dim];
    yp[r * row_dim] = v0;
    yp[(r + step) * row_dim] = v1;
    yp[(r + 2 * step) * row_dim] = v2;
    yp[(r + 3 * step) * row_dim] = v3;
  }
  for (; r < rows; r += step) yp[r * row_dim] = xp[r * row_dim];
}

// ---------------------------------------------------------------------------------------
// Gate
// ---------------------------------------------------------------------------------------
struct GateArgs {
  int n;
  int out_cf;   // layout of the gate OUTPUT row (y, g_y): 0 = e3nn ([mul][2l+1] blocks), 1 = channel-fastest ([2l+1][mul]):
                // consecutive MessagePassing layers hand their features over in cf, no relayout in between
  e3k_gate_seg s[MAXBLK];
};
// position of gated element (channel u, component m) of segment s in an output row, and its inverse
__device__ __forceinline__ int gate_out_idx(const e3k_gate_seg& s, int u, int m, int cf) {
  return s.out_off + (cf ? m * s.mul + u : u * s.dim + m);
}
__device__ __forceinline__ void gate_out_um(const e3k_gate_seg& s, int rel, int cf, int& u, int& m) {
  if (cf) { m = rel / s.mul; u = rel - m * s.mul; }
  else { u = rel / s.dim; m = rel - u * s.dim; }
}

// (column-fixed threads, as in relayout_kernel: the segment of a column is resolved once, rows are walked after)
__global__ __launch_bounds__(256) void gate_fwd_kernel(const float* __restrict__ x, int64_t rows, int in_dim,
                                                        int out_dim, GateArgs ga, float* __restrict__ y) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= out_dim) return;
  int mode = -1, xi = 0, gi = 0, act = 0;
  float cst = 0.f;
  for (int k = 0; k < ga.n; ++k) {
    const e3k_gate_seg& s = ga.s[k];
    const int rel = c - s.out_off;
    if (rel >= 0 && rel < s.mul * s.dim) {
      mode = s.kind, act = s.act, cst = s.cst;
      if (s.kind == 0) {
        xi = s.in_off + rel;
      } else {
        int u, m;
        gate_out_um(s, rel, ga.out_cf, u, m);
        xi = s.in_off + m * s.mul + u;
        gi = s.gate_off + u;
      }
      break;
    }
  }
  auto eval = [&](int64_t r) -> float {
    const float* xr = x + r * in_dim;
    float v = 0.f;
    if (mode == 0) v = cst * act_f(act, xr[xi]);
    else if (mode > 0) v = xr[xi] * (cst * act_f(act, xr[gi]));
    return v;
  };
  int64_t r = blockIdx.y;
  const int64_t step = gridDim.y;
  for (; r + 3 * step < rows; r += 4 * step) {      // four rows in flight per trip
    const float v0 = eval(r), v1 = eval(r + step), v2 = eval(r + 2 * step), v3 = eval(r + 3 * step);
    y[r * out_dim + c] = v0;
    y[(r + step) * out_dim + c] = v1;
    y[(r + 2 * step) * out_dim + c] = v2;
    y[(r + 3 * step) * out_dim + c] = v3;
  }
  for (; r < rows; r += step) y[r * out_dim + c] = eval(r);
}

__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                        const float* __restrict__ gy2, int64_t rows, int in_dim, int out_dim,
                                                        GateArgs ga, float* __restrict__ gx) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= in_dim) return;
  // 0: activated scalar, 1: gate scalar (dot over the gated block), 2: gated element, -1: feeds nothing
  int mode = -1, go = 0, gstep = 1, xi = 0, dim = 0, mul = 0, act = 0;
  float cst = 0.f;
  for (int k = 0; k < ga.n; ++k) {
    const e3k_gate_seg& s = ga.s[k];
    if (s.kind == 0) {
      const int rel = c - s.in_off;
      if (rel >= 0 && rel < s.mul) {
        mode = 0, go = s.out_off + rel, act = s.act, cst = s.cst;
        break;
      }
    } else {
      const int relg = c - s.gate_off;
      if (relg >= 0 && relg < s.mul) {
        mode = 1, go = gate_out_idx(s, relg, 0, ga.out_cf), gstep = ga.out_cf ? s.mul : 1, xi = s.in_off + relg, dim = s.dim,
        mul = s.mul, act = s.act, cst = s.cst;
        break;
      }
      const int rel = c - s.in_off;
      if (rel >= 0 && rel < s.mul * s.dim) {
        const int m = rel / s.mul, u = rel - m * s.mul;
        mode = 2, go = gate_out_idx(s, u, m, ga.out_cf), xi = s.gate_off + u, act = s.act, cst = s.cst;
        break;
      }
    }
  }
  auto eval = [&](int64_t r) -> float {
    const float* xr = x + r * in_dim;
    const float* gr = gy + r * out_dim;
    const float* gr2 = gy2 ? gy2 + r * out_dim : nullptr;      // second addend of the incoming gradient (same layout)
#define G(i) (gr2 ? gr[i] + gr2[i] : gr[i])
    float v = 0.f;
    if (mode == 0) {
      v = G(go) * cst * act_df(act, xr[c]);
    } else if (mode == 1) {
      float dot = 0.f;
      for (int m = 0; m < dim; ++m) dot = fmaf(G(go + m * gstep), xr[xi + m * mul], dot);
      v = dot * cst * act_df(act, xr[c]);
    } else if (mode == 2) {
      v = G(go) * (cst * act_f(act, xr[xi]));
    }
#undef G
    return v;
  };
  // four rows per trip: the loads of the four are in flight together (a one-row loop is a chain of load latencies: measured
  // 60-88 us per layer at 256 molecules for 73 MB of traffic)
  int64_t r = blockIdx.y;
  const int64_t step = gridDim.y;
  for (; r + 3 * step < rows; r += 4 * step) {
    const float v0 = eval(r), v1 = eval(r + step), v2 = eval(r + 2 * step), v3 = eval(r + 3 * step);
    gx[r * in_dim + c] = v0;
    gx[(r + step) * in_dim + c] = v1;
    gx[(r + 2 * step) * in_dim + c] = v2;
    gx[(r + 3 * step) * in_dim + c] = v3;
  }
  for (; r < rows; r += step) gx[r * in_dim + c] = eval(r);
}

// the same for channel-fastest outputs whose segments sit on 16-byte boundaries (what consecutive MessagePassing layers hand
// over): a thread owns FOUR consecutive input columns (one segment, one mode: segment bounds are multiples of four), every access
// is a float4, a wave walks its own (few) rows.  Same arithmetic per element, same bits.
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__global__ __launch_bounds__(256) void gate_bwd4_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                         const float* __restrict__ gy2, int64_t rows, int in_dim, int out_dim,
                                                         GateArgs ga, float* __restrict__ gx) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + lane) * 4;
  if (c >= in_dim) return;
  int mode = -1, go = 0, xi = 0, dim = 0, mul = 0, act = 0;
  float cst = 0.f;
  for (int k = 0; k < ga.n; ++k) {
    const e3k_gate_seg& s = ga.s[k];
    if (s.kind == 0) {
      const int rel = c - s.in_off;
      if (rel >= 0 && rel < s.mul) {
        mode = 0, go = s.out_off + rel, act = s.act, cst = s.cst;
        break;
      }
    } else {
      const int relg = c - s.gate_off;
      if (relg >= 0 && relg < s.mul) {
        mode = 1, go = s.out_off + relg, xi = s.in_off + relg, dim = s.dim, mul = s.mul, act = s.act, cst = s.cst;
        break;
      }
      const int rel = c - s.in_off;
      if (rel >= 0 && rel < s.mul * s.dim) {
        const int m = rel / s.mul, u = rel - m * s.mul;
        mode = 2, go = s.out_off + m * s.mul + u, xi = s.gate_off + u, act = s.act, cst = s.cst;
        break;
      }
    }
  }
  auto eval = [&](int64_t r) -> float4 {
    const float* xr = x + r * in_dim;
    const float* gr = gy + r * out_dim;
    const float* gr2 = gy2 ? gy2 + r * out_dim : nullptr;
    auto G = [&](int i) -> float4 {
      float4 a = ld4(gr + i);
      if (gr2) {
        const float4 b = ld4(gr2 + i);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
      }
      return a;
    };
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (mode == 0) {
      const float4 g = G(go), xc = ld4(xr + c);
      v.x = g.x * cst * act_df(act, xc.x); v.y = g.y * cst * act_df(act, xc.y);
      v.z = g.z * cst * act_df(act, xc.z); v.w = g.w * cst * act_df(act, xc.w);
    } else if (mode == 1) {
      float4 dot = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int m = 0; m < dim; ++m) {
        const float4 g = G(go + m * mul), xv = ld4(xr + xi + m * mul);
        dot.x = fmaf(g.x, xv.x, dot.x); dot.y = fmaf(g.y, xv.y, dot.y); dot.z = fmaf(g.z, xv.z, dot.z); dot.w = fmaf(g.w, xv.w, dot.w);
      }
      const float4 xc = ld4(xr + c);
      v.x = dot.x * cst * act_df(act, xc.x); v.y = dot.y * cst * act_df(act, xc.y);
      v.z = dot.z * cst * act_df(act, xc.z); v.w = dot.w * cst * act_df(act, xc.w);
    } else if (mode == 2) {
      const float4 g = G(go), xg = ld4(xr + xi);
      v.x = g.x * (cst * act_f(act, xg.x)); v.y = g.y * (cst * act_f(act, xg.y));
      v.z = g.z * (cst * act_f(act, xg.z)); v.w = g.w * (cst * act_f(act, xg.w));
    }
    return v;
  };
  int64_t r = (int64_t)blockIdx.y * 4 + w;
  const int64_t step = (int64_t)gridDim.y * 4;
  for (; r + 3 * step < rows; r += 4 * step) {
    const float4 v0 = eval(r), v1 = eval(r + step), v2 = eval(r + 2 * step), v3 = eval(r + 3 * step);
    *reinterpret_cast<float4*>(gx + r * in_dim + c) = v0;
    *reinterpret_cast<float4*>(gx + (r + step) * in_dim + c) = v1;
    *reinterpret_cast<float4*>(gx + (r + 2 * step) * in_dim + c) = v2;
    *reinterpret_cast<float4*>(gx + (r + 3 * step) * in_dim + c) = v3;
  }
  for (; r < rows; r += step) *reinterpret_cast<float4*>(gx + r * in_dim + c) = eval(r);
}

// backward of gate_bwd (cotangent gh on gx): g_gy = (d y / d x) gh  — one thread per OUTPUT element
__global__ __launch_bounds__(256) void gate_bwd2_gy_kernel(const float* __restrict__ x, const float* __restrict__ gh,
                                                            int64_t rows, int in_dim, int out_dim, GateArgs ga,
                                                            float* __restrict__ g_gy) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= out_dim) return;
  for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) {
    const int64_t i = r * out_dim + c;
    const float* xr = x + r * in_dim;
    const float* hr = gh + r * in_dim;
    float v = 0.f;
    for (int k = 0; k < ga.n; ++k) {
      const e3k_gate_seg& s = ga.s[k];
      const int rel = c - s.out_off;
      if (rel >= 0 && rel < s.mul * s.dim) {
        if (s.kind == 0) {
          v = hr[s.in_off + rel] * s.cst * act_df(s.act, xr[s.in_off + rel]);
        } else {
          int u, m;
          gate_out_um(s, rel, ga.out_cf, u, m);
          const float gt = xr[s.gate_off + u];
          const int xi = s.in_off + m * s.mul + u;
          v = hr[xi] * (s.cst * act_f(s.act, gt)) + hr[s.gate_off + u] * xr[xi] * (s.cst * act_df(s.act, gt));
        }
        break;
      }
    }
    g_gy[i] = v;
  }
}

// ... and g_x = d/dx <gh, gate_bwd(x, gy)> — one thread per INPUT element
__global__ __launch_bounds__(256) void gate_bwd2_x_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                           const float* __restrict__ gh, int64_t rows, int in_dim,
                                                           int out_dim, GateArgs ga, float* __restrict__ g_x) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= in_dim) return;
  for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) {
    const int64_t i = r * in_dim + c;
    const float* xr = x + r * in_dim;
    const float* hr = gh + r * in_dim;
    const float* gr = gy + r * out_dim;
    float v = 0.f;
    for (int k = 0; k < ga.n; ++k) {
      const e3k_gate_seg& s = ga.s[k];
      if (s.kind == 0) {
        const int rel = c - s.in_off;
        if (rel >= 0 && rel < s.mul) {
          v = hr[c] * gr[s.out_off + rel] * s.cst * act_d2f(s.act, xr[c]);
          break;
        }
      } else {
        const int relg = c - s.gate_off;
        if (relg >= 0 && relg < s.mul) {
          float dgh = 0.f, dgx = 0.f;   // sum_m gy*gh_gated, sum_m gy*x_gated
          for (int m = 0; m < s.dim; ++m) {
            const float g = gr[gate_out_idx(s, relg, m, ga.out_cf)];
            dgh = fmaf(g, hr[s.in_off + m * s.mul + relg], dgh);
            dgx = fmaf(g, xr[s.in_off + m * s.mul + relg], dgx);
          }
          v = s.cst * (dgh * act_df(s.act, xr[c]) + hr[c] * dgx * act_d2f(s.act, xr[c]));
          break;
        }
        const int rel = c - s.in_off;
        if (rel >= 0 && rel < s.mul * s.dim) {
          const int m = rel / s.mul, u = rel - m * s.mul;
          v = hr[s.gate_off + u] * gr[gate_out_idx(s, u, m, ga.out_cf)] * (s.cst * act_df(s.act, xr[s.gate_off + u]));
          break;
        }
      }
    }
    g_x[i] = v;
  }
}

// ---------------------------------------------------------------------------------------
// keyed self-connection weights:  M[t, (j,u,w)] = sum_v a[t,v] W_j[u,v,w]
// (W_j stored [U][V][Wout] as e3nn's 'uvw' weights; one thread per column (j,u,w), coalesced along w)
// ---------------------------------------------------------------------------------------
constexpr int KW_MAXI = 16, KW_MAXV = 32, KW_MAXK = 1 << 22;
struct KwArgs {
  int n, K, V;
  int64_t ld_m, total;
  e3k_kw_instr ins[KW_MAXI];
};

__device__ __forceinline__ const float* kw_locate(const KwArgs& ka, int64_t c, const float* W, int& wout) {
  int j = 0;
  for (int i = 1; i < ka.n; ++i)
    if (c >= ka.ins[i].m_off) j = i;
  const e3k_kw_instr& in = ka.ins[j];
  const int local = (int)(c - in.m_off);
  const int u = local / in.w_out, w = local - u * in.w_out;
  wout = in.w_out;
  return W + in.w_off + (int64_t)u * ka.V * in.w_out + w;
}

// grid.y tiles the keys: a handful of species keys is one tile; the un-keyed self-connection uses the same kernels with
// one key per node (thousands of keys).  The tile is a loop bound only (the attribute rows come through the scalar cache),
// so the launcher picks it: 64 keys, or 512 when the columns alone fill the chip -- the protein net's 80 keys are then ONE
// tile and MODE 1 stores its sums instead of adding two tiles' with 77 M atomics.
constexpr int KW_KT = 64, KW_KT_WIDE = 512;
static inline unsigned kw_tiles(int n_keys, int64_t col_blocks) {
  const int kt = col_blocks >= 1024 ? KW_KT_WIDE : KW_KT;
  return (unsigned)((n_keys + kt - 1) / kt);
}
template <int MODE>   // 0: M = a . W    1: gW (+)= a^T . gM    (MODE 1: `M` is gM, `Wout_` is gW)
__device__ __forceinline__ void kw_body(const float* __restrict__ a, const float* __restrict__ W, const KwArgs& ka, const int K,
                                        float* __restrict__ M, float* __restrict__ Wout_, int accumulate) {
  // The attribute rows a[t, :] are the same for every lane: they are read through the scalar cache (uniform address, s_load)
  // and enter the FMAs as SGPR operands.  (Staged in LDS before: 32 broadcast ds_read_b32 per key and thread made both
  // modes LDS-bound -- 198 us per launch for the 8 layers of the protein net, 80 keys x 32 attributes.)
  const int tile = uniform((K + (int)gridDim.y - 1) / (int)gridDim.y);
  const int t0 = uniform((int)blockIdx.y * tile);
  const int kt = (K - t0 < tile) ? K - t0 : tile;
  if (kt <= 0) return;
  const int V = uniform(ka.V);
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= ka.total) return;
  int wout;
  const float* base = kw_locate(ka, c, W, wout);
  if constexpr (MODE == 0) {
    float wv[KW_MAXV];
#pragma unroll
    for (int v = 0; v < KW_MAXV; ++v) wv[v] = v < V ? base[(int64_t)v * wout] : 0.f;
    for (int t = 0; t < kt; ++t) {
      const float* __restrict__ ar = a + (int64_t)(t0 + t) * V;
      float acc = 0.f;
#pragma unroll
      for (int v = 0; v < KW_MAXV; ++v)
        if (v < V) acc = fmaf(ar[v], wv[v], acc);
      M[(int64_t)(t0 + t) * ka.ld_m + c] = acc;
    }
  } else {
    float gv[KW_MAXV];
#pragma unroll
    for (int v = 0; v < KW_MAXV; ++v) gv[v] = 0.f;
    for (int t = 0; t < kt; ++t) {
      const float* __restrict__ ar = a + (int64_t)(t0 + t) * V;
      const float g = M[(int64_t)(t0 + t) * ka.ld_m + c];
#pragma unroll
      for (int v = 0; v < KW_MAXV; ++v)
        if (v < V) gv[v] = fmaf(ar[v], g, gv[v]);
    }
    float* dst = Wout_ + (base - W);
    if (gridDim.y == 1) {
#pragma unroll
      for (int v = 0; v < KW_MAXV; ++v)
        if (v < V) dst[(int64_t)v * wout] = accumulate ? dst[(int64_t)v * wout] + gv[v] : gv[v];
    } else {   // several key tiles add into the same element (the launcher zero-fills when not accumulating)
#pragma unroll
      for (int v = 0; v < KW_MAXV; ++v)
        if (v < V) atomicAdd(dst + (int64_t)v * wout, gv[v]);
    }
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void keyed_weights_kernel(const float* __restrict__ a, const float* __restrict__ W,
                                                             KwArgs ka, float* __restrict__ M, float* __restrict__ Wout_,
                                                             int accumulate) {
  kw_body<MODE>(a, W, ka, ka.K, M, Wout_, accumulate);
}

// the same for the self-connections of SEVERAL layers that share the attribute rows `a` (grid.z = layer): their instruction
// tables live in device memory (e3k_kw_args_create, once per layer), the per-call pointers in the kernel arguments
constexpr int KW_MAXL = 8;
struct KwMulti {
  int n, K;
  const KwArgs* ka[KW_MAXL];
  const float* W[KW_MAXL];
  float* M[KW_MAXL];       // MODE 0: out; MODE 1 / bwd_a: gM
  float* gW[KW_MAXL];
  int acc[KW_MAXL];
  int64_t ws_off[KW_MAXL]; // bwd_a: first workspace row of the layer
};
template <int MODE>
__global__ __launch_bounds__(256) void keyed_weights_multi_kernel(const float* __restrict__ a, KwMulti m) {
  const int z = blockIdx.z;
  if (MODE == 1 && !m.gW[z]) return;
  kw_body<MODE>(a, m.W[z], *m.ka[z], m.K, m.M[z], m.gW[z], m.acc[z]);
}

// ga[t,v] += sum_c gM[t,c] W[c,v]  (c over all ~1e5 columns): a [K x C] x [C x V] product with a tiny output.
// Stage 1: a block stages 256 columns of gM ([K][256]) and of W ([V][256]) in LDS with coalesced loads; thread
// (t, v) then owns one output and walks the 256 columns with 16-byte LDS reads; the block's K*V partials go to a
// workspace row.  Stage 2: one block sums the workspace rows into ga (no same-address atomics: they serialised).
constexpr int KWA_COLS = 256, KWA_KT = 16, KWA_RANGE = 128;
// rows of the staged W tile are padded by one 16-byte slot: the lanes of a ds_read_b128 group read the same columns of 16
// different rows (v), and with a 1 024-byte row stride all of them would sit on the same four banks (16-way conflict)
constexpr int KWA_WLD = KWA_COLS + 4;
__device__ __forceinline__ void kw_bwd_a_body(const float* __restrict__ gM, const float* __restrict__ W, const KwArgs& ka_,
                                              const int K, float* __restrict__ ws, float (*gs)[KWA_COLS], float (*wsm)[KWA_WLD]) {
  struct { int K, V; int64_t ld_m, total; } ka{K, ka_.V, ka_.ld_m, ka_.total};
  const int t_id = threadIdx.x;
  const int64_t c = (int64_t)blockIdx.x * KWA_COLS + t_id;
  const bool ok = c < ka.total;
  {
    int wout = 1;
    const float* base = ok ? kw_locate(ka_, c, W, wout) : W;
    for (int v = 0; v < ka.V; ++v) wsm[v][t_id] = ok ? base[(int64_t)v * wout] : 0.f;
  }
  float* out = ws + (int64_t)blockIdx.x * ka.K * ka.V;
  // grid.y splits the keys into ranges of KWA_RANGE (outputs of different keys are distinct: no conflicts)
  const int t_begin = blockIdx.y * KWA_RANGE;
  const int t_end = (t_begin + KWA_RANGE < ka.K) ? t_begin + KWA_RANGE : ka.K;
  for (int t0 = t_begin; t0 < t_end; t0 += KWA_KT) {
    const int kt = (t_end - t0 < KWA_KT) ? t_end - t0 : KWA_KT;
    __syncthreads();   // previous chunk's readers are done (and wsm is complete on the first trip)
    for (int t = 0; t < kt; ++t) gs[t][t_id] = ok ? gM[(int64_t)(t0 + t) * ka.ld_m + c] : 0.f;
    __syncthreads();
    for (int o = t_id; o < kt * ka.V; o += 256) {
      const int t = o / ka.V, v = o - t * ka.V;
      const float4* g4 = reinterpret_cast<const float4*>(gs[t]);
      const float4* w4 = reinterpret_cast<const float4*>(wsm[v]);
      float acc = 0.f;
#pragma unroll 8
      for (int q = 0; q < KWA_COLS / 4; ++q) {
        const float4 g = g4[q], w = w4[q];
        acc = fmaf(g.x, w.x, acc);
        acc = fmaf(g.y, w.y, acc);
        acc = fmaf(g.z, w.z, acc);
        acc = fmaf(g.w, w.w, acc);
      }
      out[(t0 + t) * ka.V + v] = acc;
    }
  }
}

__global__ __launch_bounds__(256) void keyed_weights_bwd_a_kernel(const float* __restrict__ gM, const float* __restrict__ W,
                                                                   KwArgs ka, float* __restrict__ ws) {
  __shared__ __attribute__((aligned(16))) float gs[KWA_KT][KWA_COLS];
  __shared__ __attribute__((aligned(16))) float wsm[KW_MAXV][KWA_WLD];
  kw_bwd_a_body(gM, W, ka, ka.K, ws, gs, wsm);
}

__global__ __launch_bounds__(256) void keyed_weights_bwd_a_multi_kernel(KwMulti m, float* __restrict__ ws) {
  __shared__ __attribute__((aligned(16))) float gs[KWA_KT][KWA_COLS];
  __shared__ __attribute__((aligned(16))) float wsm[KW_MAXV][KWA_WLD];
  const int z = blockIdx.z;
  const KwArgs& ka = *m.ka[z];
  if ((int64_t)blockIdx.x * KWA_COLS >= ka.total) return;      // (block-uniform: this layer has fewer column blocks)
  kw_bwd_a_body(m.M[z], m.W[z], ka, m.K, ws + m.ws_off[z] * m.K * ka.V, gs, wsm);
}

// one block per 4 outputs: each wave sums its output's column of the workspace with 64 rows in flight
__global__ __launch_bounds__(256) void keyed_weights_bwd_a_reduce_kernel(const float* __restrict__ ws, int n_rows, int n,
                                                                          float* __restrict__ ga) {
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (o >= n) return;
  float acc = 0.f;
  for (int r = lane; r < n_rows; r += 64) acc += ws[(int64_t)r * n + o];
  acc = wave_sum(acc);
  if (lane == 0) ga[o] += acc;
}

// ---------------------------------------------------------------------------------------
// NormActivation (e3nn.nn.NormActivation, e3_layers/nn/message_passing.py:212-219): per irrep channel
//   n2 = max(sum_m x_m^2, eps^2), n = sqrt(n2), y_m = x_m * act(n) / n        (normalize = 1; else y_m = x_m act(n2'))
// input channel-fastest [2l+1][mul], output e3nn layout [mul][2l+1] at the same block offsets (as the Gate kernel)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void normact_scale(int act, float eps2, int normalize, float n2, float& s, float& ds_over_n) {
  // s = scaling; ds_over_n = (d s / d n) / n  (0 where the clamp is active), so that d s / d x_m = ds_over_n * x_m
  if (eps2 > 0.f) {
    const bool clamped = n2 < eps2;
    const float n = sqrtf(clamped ? eps2 : n2);
    const float a = act_f(act, n), da = act_df(act, n);
    if (normalize) {
      s = a / n;
      ds_over_n = clamped ? 0.f : (da * n - a) / (n * n * n);
    } else {
      s = a;
      ds_over_n = clamped ? 0.f : da / n;
    }
  } else {   // no epsilon: the argument of the nonlinearity is the SQUARED norm (o3.Norm(squared=True)), no division
    s = act_f(act, n2);
    ds_over_n = 2.0f * act_df(act, n2);   // d s / d x_m = act'(n2) * 2 x_m
  }
}

__global__ __launch_bounds__(256) void normact_fwd_kernel(const float* __restrict__ x, int64_t rows, int row_dim,
                                                           BlockArgs ba, int act, float eps2, int normalize,
                                                           float* __restrict__ y) {
  const int64_t total = rows * row_dim;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / row_dim;
    const int c = (int)(i - r * row_dim);
    const float* xr = x + r * row_dim;
    float v = 0.f;
    for (int k = 0; k < ba.n; ++k) {
      const e3k_block& b = ba.b[k];
      const int rel = c - b.off;
      if (rel >= 0 && rel < b.mul * b.dim) {
        const int u = rel / b.dim, m = rel - u * b.dim;   // output element (u, m)
        float n2 = 0.f;
        for (int q = 0; q < b.dim; ++q) {
          const float t = xr[b.off + q * b.mul + u];
          n2 = fmaf(t, t, n2);
        }
        float s, d;
        normact_scale(act, eps2, normalize, n2, s, d);
        v = s * xr[b.off + m * b.mul + u];
        break;
      }
    }
    y[i] = v;
  }
}

// s, D = (d s / d x_m) / x_m and D2 = (d D / d x_m) / x_m of the channel's scaling (second derivatives: force training through a
// 'norm' nonlinearity): with q = sum_m x_m^2,  y_m = x_m s,  g_x_m = g_y_m s + x_m D (g_y . x)
__device__ __forceinline__ void normact_scale2(int act, float eps2, int normalize, float n2, float& s, float& d, float& d2) {
  if (eps2 > 0.f) {
    const bool clamped = n2 < eps2;
    const float n = sqrtf(clamped ? eps2 : n2);
    const float a = act_f(act, n), a1 = act_df(act, n), a2 = act_d2f(act, n);
    float s1, s2;
    if (normalize) {
      s = a / n;
      s1 = (a1 * n - a) / (n * n);
      s2 = (a2 * n * n - 2.0f * a1 * n + 2.0f * a) / (n * n * n);
    } else {
      s = a;
      s1 = a1;
      s2 = a2;
    }
    d = clamped ? 0.f : s1 / n;
    d2 = clamped ? 0.f : (s2 * n - s1) / (n * n * n);
  } else {   // no epsilon: the argument of the nonlinearity is the squared norm
    s = act_f(act, n2);
    d = 2.0f * act_df(act, n2);
    d2 = 4.0f * act_d2f(act, n2);
  }
}

// backward of normact_bwd (cotangent h on g_x, cf layout like x):  F = sum_m h_m g_x_m = s (h . g_y) + D (h . x)(g_y . x)
//   g_gy_m = s h_m + D (h . x) x_m                                                        (e3nn layout, like g_y)
//   g_x_m  = D x_m (h . g_y) + D2 x_m (h . x)(g_y . x) + D (h_m (g_y . x) + (h . x) g_y_m)      (cf layout)
__global__ __launch_bounds__(256) void normact_bwd2_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                            const float* __restrict__ h, int64_t rows, int row_dim, BlockArgs ba,
                                                            int act, float eps2, int normalize, float* __restrict__ g_gy,
                                                            float* __restrict__ g_x) {
  const int64_t total = rows * row_dim;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / row_dim;
    const int c = (int)(i - r * row_dim);
    const float* xr = x + r * row_dim;
    const float* gr = gy + r * row_dim;
    const float* hr = h + r * row_dim;
    for (int k = 0; k < ba.n; ++k) {
      const e3k_block& b = ba.b[k];
      const int rel = c - b.off;
      if (rel >= 0 && rel < b.mul * b.dim) {
        const int m = rel / b.mul, u = rel - m * b.mul;   // input element (m, u), channel-fastest
        float n2 = 0.f, gx_ = 0.f, hx = 0.f, hg = 0.f;
        for (int q = 0; q < b.dim; ++q) {
          const float t = xr[b.off + q * b.mul + u], g = gr[b.off + u * b.dim + q], hh = hr[b.off + q * b.mul + u];
          n2 = fmaf(t, t, n2);
          gx_ = fmaf(g, t, gx_);
          hx = fmaf(hh, t, hx);
          hg = fmaf(hh, g, hg);
        }
        float s, d, d2;
        normact_scale2(act, eps2, normalize, n2, s, d, d2);
        const float xm = xr[c], hm = hr[c], gm = gr[b.off + u * b.dim + m];
        if (g_gy) g_gy[r * row_dim + b.off + u * b.dim + m] = fmaf(s, hm, d * hx * xm);
        if (g_x) g_x[i] = d * xm * hg + d2 * xm * hx * gx_ + d * (hm * gx_ + hx * gm);
        break;
      }
    }
  }
}

__global__ __launch_bounds__(256) void normact_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                           int64_t rows, int row_dim, BlockArgs ba, int act, float eps2,
                                                           int normalize, float* __restrict__ gx) {
  const int64_t total = rows * row_dim;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / row_dim;
    const int c = (int)(i - r * row_dim);
    const float* xr = x + r * row_dim;
    const float* gr = gy + r * row_dim;
    float v = 0.f;
    for (int k = 0; k < ba.n; ++k) {
      const e3k_block& b = ba.b[k];
      const int rel = c - b.off;
      if (rel >= 0 && rel < b.mul * b.dim) {
        const int m = rel / b.mul, u = rel - m * b.mul;   // input element (m, u), channel-fastest
        float n2 = 0.f, dot = 0.f;
        for (int q = 0; q < b.dim; ++q) {
          const float t = xr[b.off + q * b.mul + u];
          n2 = fmaf(t, t, n2);
          dot = fmaf(gr[b.off + u * b.dim + q], t, dot);
        }
        float s, d;
        normact_scale(act, eps2, normalize, n2, s, d);
        v = fmaf(s, gr[b.off + u * b.dim + m], d * dot * xr[c]);
        break;
      }
    }
    gx[i] = v;
  }
}

// ---------------------------------------------------------------------------------------
// per-block RMS normalisation: one wave per row
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, int64_t rows, int row_dim,
                                                             BlockArgs ba, const float* __restrict__ stdv,
                                                             float* __restrict__ y, float* __restrict__ inv_norm) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  for (int k = 0; k < ba.n; ++k) {
    const e3k_block& b = ba.b[k];
    const int len = b.mul * b.dim;
    const float* xb = x + r * row_dim + b.off;
    float ss = 0.f;
    for (int j = lane; j < len; j += 64) ss = fmaf(xb[j], xb[j], ss);
    ss = wave_sum(ss);
    const float inv = 1.0f / sqrtf(ss / (float)b.mul + 1e-6f);
    if (lane == 0) inv_norm[r * ba.n + k] = inv;
    const float sc = inv * stdv[k];
    float* yb = y + r * row_dim + b.off;
    for (int j = lane; j < len; j += 64) yb[j] = xb[j] * sc;
  }
}

__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                             const float* __restrict__ inv_norm, int64_t rows,
                                                             int row_dim, BlockArgs ba, const float* __restrict__ stdv,
                                                             float* __restrict__ gx, float* __restrict__ g_std) {
  // one wave per row, rows grid-strided; the gradient of the per-block scale is summed per wave in registers, the
  // block's four waves meet in LDS and issue one atomic per irrep block (one per ROW serialised on n_blocks addresses)
  __shared__ float part[4][MAXBLK];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float gs[MAXBLK];
#pragma unroll
  for (int k = 0; k < MAXBLK; ++k) gs[k] = 0.f;
  for (int64_t r = (int64_t)blockIdx.x * 4 + wv; r < rows; r += (int64_t)gridDim.x * 4) {
#pragma unroll
    for (int k = 0; k < MAXBLK; ++k) {
      if (k < ba.n) {
        const e3k_block& b = ba.b[k];
        const int len = b.mul * b.dim;
        const float* xb = x + r * row_dim + b.off;
        const float* gb = gy + r * row_dim + b.off;
        float dot = 0.f;
        for (int j = lane; j < len; j += 64) dot = fmaf(gb[j], xb[j], dot);
        dot = wave_sum(dot);
        const float inv = inv_norm[r * ba.n + k], sc = stdv[k];
        // y = s * x * inv, inv = (sum x^2 / mul + eps)^-1/2  =>  dx = s*inv*(g - x * dot * inv^2 / mul)
        const float coef = dot * inv * inv / (float)b.mul;
        float* gxb = gx + r * row_dim + b.off;
        for (int j = lane; j < len; j += 64) gxb[j] = sc * inv * (gb[j] - xb[j] * coef);
        gs[k] += dot * inv;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < MAXBLK; ++k)
    if (k < ba.n && lane == 0) part[wv][k] = gs[k];
  __syncthreads();
  if ((int)threadIdx.x < ba.n)
    atomicAdd(g_std + threadIdx.x, part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// Double backward of the normalisation (force training through LayerNormalization, e3_layers/nn/pointwise.py:32-51 under
// GradientOutput's create_graph=True): cotangents h (of g_x) and hs (of g_std, may be NULL) in, gradients of
//   S = <h, g_x(x, g, s)> + <hs, g_std(x, g)>      with   g_x_i = s v g_i - s v^3 D x_i / m,   g_std = sum_rows D v,
//   v = (Q / m + eps)^-1/2,  Q = sum x^2,  D = sum g x,  A = sum h g,  B = sum h x   (sums over the irrep block of one row)
// out:  dS/dg_j = s v h_j - s v^3 B x_j / m + hs v x_j
//       dS/dx_j = -s A v^3 x_j / m + 3 s v^5 D B x_j / m^2 - s v^3 (g_j B + D h_j) / m + hs (g_j v - D v^3 x_j / m)
//       dS/ds   = sum_rows (v A - v^3 D B / m)
__global__ __launch_bounds__(256) void layernorm_bwd2_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                              const float* __restrict__ h, const float* __restrict__ hs,
                                                              const float* __restrict__ inv_norm, int64_t rows, int row_dim,
                                                              BlockArgs ba, const float* __restrict__ stdv, float* __restrict__ g_g,
                                                              float* __restrict__ g_x, float* __restrict__ g_std) {
  __shared__ float part[4][MAXBLK];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float gs[MAXBLK];
#pragma unroll
  for (int k = 0; k < MAXBLK; ++k) gs[k] = 0.f;
  for (int64_t r = (int64_t)blockIdx.x * 4 + wv; r < rows; r += (int64_t)gridDim.x * 4) {
#pragma unroll
    for (int k = 0; k < MAXBLK; ++k) {
      if (k < ba.n) {
        const e3k_block& b = ba.b[k];
        const int len = b.mul * b.dim;
        const float* xb = x + r * row_dim + b.off;
        const float* gb = g + r * row_dim + b.off;
        const float* hb = h + r * row_dim + b.off;
        float A = 0.f, B = 0.f, D = 0.f;
        for (int j = lane; j < len; j += 64) {
          A = fmaf(hb[j], gb[j], A);
          B = fmaf(hb[j], xb[j], B);
          D = fmaf(gb[j], xb[j], D);
        }
        A = wave_sum(A);
        B = wave_sum(B);
        D = wave_sum(D);
        const float v = inv_norm[r * ba.n + k], sc = stdv[k], m = (float)b.mul;
        const float v3 = v * v * v, hsk = hs ? hs[k] : 0.f;
        const float cg_x = -sc * v3 * B / m + hsk * v;                                             // dS/dg: coefficient of x_j
        const float cx_x = -sc * A * v3 / m + 3.f * sc * v3 * v * v * D * B / (m * m) - hsk * D * v3 / m;   // dS/dx: of x_j
        const float cx_g = -sc * v3 * B / m + hsk * v, cx_h = -sc * v3 * D / m;                    //        of g_j and h_j
        for (int j = lane; j < len; j += 64) {
          if (g_g) g_g[r * row_dim + b.off + j] = fmaf(sc * v, hb[j], cg_x * xb[j]);
          if (g_x) g_x[r * row_dim + b.off + j] = fmaf(cx_x, xb[j], fmaf(cx_g, gb[j], cx_h * hb[j]));
        }
        gs[k] += v * A - v3 * D * B / m;
      }
    }
  }
  if (g_std) {
#pragma unroll
    for (int k = 0; k < MAXBLK; ++k)
      if (k < ba.n && lane == 0) part[wv][k] = gs[k];
    __syncthreads();
    if ((int)threadIdx.x < ba.n)
      atomicAdd(g_std + threadIdx.x, part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
  }
}

// ---------------------------------------------------------------------------------------
// sorted segment sum
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void segment_sum_kernel(const float* __restrict__ x, const int32_t* __restrict__ ptr,
                                                           int64_t n_seg, int dim, int mean, float* __restrict__ out) {
  const int64_t total = n_seg * dim;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t s = i / dim;
    const int c = (int)(i - s * dim);
    const int beg = ptr[s], end = ptr[s + 1];
    float acc = 0.f;
    for (int r = beg; r < end; ++r) acc += x[(int64_t)r * dim + c];
    if (mean) acc /= (float)((end - beg) > 1 ? (end - beg) : 1);
    out[i] = acc;
  }
}

// segment sizes -> row pointers: ptr[0] = 0, ptr[s + 1] = n[0] + .. + n[s]  (one workgroup: chunks of 256 counts, a running carry;
// replaces zeros + cumsum + a dtype conversion + a slice copy in front of every Pooling)
__global__ __launch_bounds__(256) void counts_to_ptr_kernel(const int64_t* __restrict__ n, int G, int32_t* __restrict__ ptr) {
  __shared__ int part[256];
  __shared__ int carry;
  const int t = threadIdx.x;
  if (t == 0) carry = 0, ptr[0] = 0;
  __syncthreads();
  for (int base = 0; base < G; base += 256) {
    const int v = base + t < G ? (int)n[base + t] : 0;
    part[t] = v;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {      // inclusive Hillis-Steele scan of the chunk
      const int add = t >= off ? part[t - off] : 0;
      __syncthreads();
      part[t] += add;
      __syncthreads();
    }
    if (base + t < G) ptr[base + t + 1] = carry + part[t];
    __syncthreads();
    if (t == 255) carry += part[255];
    __syncthreads();
  }
}

// one-hot rows of a type index: out[r, t] = (idx[r] == t) -- replaces zeros + scatter + a dtype conversion (OneHotEncoding)
// (an index outside [0, T) leaves an all-zero row -- the reference's torch one_hot raises on it: here bit 2 of the caller's error
//  flag is set, ORed in and never cleared by this kernel, so the flag may be a persistent one that several checks share)
__global__ __launch_bounds__(256) void onehot_kernel(const int64_t* __restrict__ idx, int64_t total, int T, float* __restrict__ out,
                                                     int32_t* __restrict__ bad) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / T;
    const int64_t v = idx[r];
    const int t = (int)(i - r * T);
    out[i] = v == (int64_t)t ? 1.f : 0.f;
    if (bad && t == 0 && (v < 0 || v >= T)) atomicOr(bad, 4);
  }
}

// the value of a device error flag to (pinned, device-visible) host memory and the flag cleared, in ONE launch and in stream
// order: a flag that several replays fold their index checks into is handed over exactly once per bad batch -- a copy followed
// by a clear issued when the HOST gets round to reading the copy loses whatever was flagged in between
__global__ void flag_fetch_clear_kernel(int32_t* __restrict__ flag, int32_t* __restrict__ host) {
  const int32_t v = atomicExch(flag, 0);
  __hip_atomic_store(host, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// column-fixed kernels: grid.x covers the columns, grid.y strides the rows (about 8 k workgroups in all)
inline dim3 grid_cols(int cols, int64_t rows) {
  const int gx = (cols + 255) / 256;
  int64_t gy = 8192 / (gx > 0 ? gx : 1);
  if (gy > rows) gy = rows;
  if (gy < 1) gy = 1;
  return dim3((unsigned)gx, (unsigned)gy);
}

// the same with fewer workgroups: threads that walk several rows, four at a time (gate kernels; measured at 4 608 rows:
// forward 18.7 / 17.0 / 13.6 / 15.8 us with 8 k / 4 k / 2 k / 1 k workgroups, backward 31.3 / 31.0 / 36.0 / 58.0 us)
inline dim3 grid_cols_deep(int cols, int64_t rows, int target) {
  const int gx = (cols + 255) / 256;
  int64_t gy = target / (gx > 0 ? gx : 1);
  if (gy > rows) gy = rows;
  if (gy < 1) gy = 1;
  return dim3((unsigned)gx, (unsigned)gy);
}

inline unsigned grid_for(int64_t n) {
  int64_t g = (n + 255) / 256;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace e3k

extern "C" const char* e3k_strerror(int code) {
  switch (code) {
    case E3K_OK: return "ok";
    case E3K_ERR_INVALID: return "invalid argument or unsupported shape";
    case E3K_ERR_LAUNCH: return "HIP launch/runtime error";
    case E3K_ERR_UNSUPPORTED: return "degree or size beyond the compiled tables";
    default: return "unknown e3k error";
  }
}

extern "C" int e3k_version(void) { return 100; }

extern "C" int e3k_act_fwd(const float* x, int64_t n, int32_t act, float cst, float* y, void* stream) {
  if (n < 0 || act < 0 || act > 5) return E3K_ERR_INVALID;
  if (n == 0) return E3K_OK;
  if (!x || !y) return E3K_ERR_INVALID;
  hipLaunchKernelGGL(e3k::act_fwd_kernel, dim3(e3k::grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, n, act, cst, y);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int e3k_act_bwd(const float* x, const float* g_y, int64_t n, int32_t act, float cst, float* g_x,
                           void* stream) {
  if (n < 0 || act < 0 || act > 5) return E3K_ERR_INVALID;
  if (n == 0) return E3K_OK;
  if (!x || !g_y || !g_x) return E3K_ERR_INVALID;
  hipLaunchKernelGGL(e3k::act_bwd_kernel, dim3(e3k::grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, g_y, n, act, cst,
                     g_x);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int e3k_act_bwd_from_output(const float* y, const float* g_y, int64_t n, int32_t act, float cst,
                                       float* g_x, void* stream) {
  if (n < 0 || act != 1 || !(cst > 0.f)) return E3K_ERR_INVALID;
  if (n == 0) return E3K_OK;
  if (!y || !g_y || !g_x) return E3K_ERR_INVALID;
  hipLaunchKernelGGL(e3k::act_bwd_out_kernel, dim3(e3k::grid_for(n)), dim3(256), 0, (hipStream_t)stream, y, g_y, n, cst,
                     g_x);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int e3k_act_bwd2(const float* x, const float* g_y, const float* g_hat, int64_t n, int32_t act, float cst,
                            float* g_gy, float* g_x, void* stream) {
  if (n < 0 || act < 0 || act > 5) return E3K_ERR_INVALID;
  if (n == 0) return E3K_OK;
  if (!x || !g_hat || (!g_gy && !g_x) || (g_x && !g_y)) return E3K_ERR_INVALID;
  hipLaunchKernelGGL(e3k::act_bwd2_kernel, dim3(e3k::grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, g_y, g_hat, n,
                     act, cst, g_gy, g_x);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

namespace {
int make_blocks(const e3k_block* blocks, int32_t n_blocks, int32_t row_dim, e3k::BlockArgs& ba) {
  if (n_blocks < 0 || n_blocks > e3k::MAXBLK || (n_blocks && !blocks)) return E3K_ERR_INVALID;
  ba.n = n_blocks;
  for (int i = 0; i < n_blocks; ++i) {
    if (blocks[i].off < 0 || blocks[i].mul <= 0 || blocks[i].dim <= 0 ||
        blocks[i].off + blocks[i].mul * blocks[i].dim > row_dim)
      return E3K_ERR_INVALID;
    ba.b[i] = blocks[i];
  }
  return E3K_OK;
}
int make_gate(const e3k_gate_seg* segs, int32_t n_segs, e3k::GateArgs& ga) {
  if (n_segs <= 0 || n_segs > e3k::MAXBLK || !segs) return E3K_ERR_INVALID;
  ga.n = n_segs;
  for (int i = 0; i < n_segs; ++i) {
    if (segs[i].mul <= 0 || segs[i].dim <= 0 || segs[i].act < 0 || segs[i].act > 5) return E3K_ERR_INVALID;
    ga.s[i] = segs[i];
  }
  return E3K_OK;
}
}  // namespace

/* blocks / segs are HOST arrays (copied into the kernel arguments). */
extern "C" int e3k_relayout(const float* x, int64_t rows, int32_t row_dim, const e3k_block* blocks, int32_t n_blocks,
                            int32_t to_cf, float* y, void* stream) {
  e3k::BlockArgs ba{};
  const int rc = make_blocks(blocks, n_blocks, row_dim, ba);
  if (rc != E3K_OK) return rc;
  if (rows < 0 || row_dim <= 0) return E3K_ERR_INVALID;
  if (rows == 0) return E3K_OK;
  if (!x || !y) return E3K_ERR_INVALID;
  hipLaunchKernelGGL(e3k::relayout_kernel, e3k::grid_cols(row_dim, rows), dim3(256), 0, (hipStream_t)stream, x,
                     rows, row_dim, ba, to_cf, y);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int e3k_gate_fwd(const float* x, int64_t rows, int32_t in_dim, int32_t out_dim, const e3k_gate_seg* segs,
                            int32_t n_segs, int32_t out_cf, float* y, void* stream) {
  e3k::GateArgs ga{};
  const int rc = make_gate(segs, n_segs, ga);
  if (rc != E3K_OK) return rc;
  ga.out_cf = out_cf ? 1 : 0;
  if (rows < 0 || in_dim <= 0 || out_dim <= 0) return E3K_ERR_INVALID;
  if (rows == 0) return E3K_OK;
  if (!x || !y) return E3K_ERR_INVALID;
  hipLaunchKernelGGL(e3k::gate_fwd_kernel, e3k::grid_cols_deep(out_dim, rows, 2048), dim3(256), 0, (hipStream_t)stream, x,
                     rows, in_dim, out_dim, ga, y);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int e3k_gate_bwd(const float* x, const float* g_y, const float* g_y2, int64_t rows, int32_t in_dim,
                            int32_t out_dim, const e3k_gate_seg* segs, int32_t n_segs, int32_t out_cf, float* g_x,
                            void* stream) {
  e3k::GateArgs ga{};
  const int rc = make_gate(segs, n_segs, ga);
  if (rc != E3K_OK) return rc;
  ga.out_cf = out_cf ? 1 : 0;
  if (rows < 0 || in_dim <= 0 || out_dim <= 0) return E3K_ERR_INVALID;
  if (rows == 0) return E3K_OK;
  if (!x || !g_y || !g_x) return E3K_ERR_INVALID;
  // rows per wave of the float4 form (isolated, 4 608 rows: scalar form 30.9 us; 1 / 2 / 4 / 8 rows per wave 26.4 / 23.2 / 25.3 / 31.8)
  E3K_KNOB_INT(gate4, "E3K_GATE4", 2);
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  bool vec = gate4 > 0 && out_cf && in_dim % 4 == 0 && out_dim % 4 == 0 && al16(x) && al16(g_y) && al16(g_x) && (!g_y2 || al16(g_y2));
  for (int k = 0; k < n_segs && vec; ++k)
    vec = segs[k].in_off % 4 == 0 && segs[k].out_off % 4 == 0 && segs[k].mul % 4 == 0 && (segs[k].kind == 0 || segs[k].gate_off % 4 == 0);
  if (vec) {
    const int gx = (in_dim / 4 + 63) / 64;
    int64_t gy = (rows + 4 * gate4 - 1) / (4 * gate4);      // gate4 rows per wave
    if (gy < 1) gy = 1;
    hipLaunchKernelGGL(e3k::gate_bwd4_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, (hipStream_t)stream, x, g_y, g_y2, rows,
                       in_dim, out_dim, ga, g_x);
    E3K_CHECK_LAUNCH();
    return E3K_OK;
  }
  hipLaunchKernelGGL(e3k::gate_bwd_kernel, e3k::grid_cols_deep(in_dim, rows, 4096), dim3(256), 0, (hipStream_t)stream, x, g_y,
                     g_y2, rows, in_dim, out_dim, ga, g_x);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int e3k_gate_bwd2(const float* x, const float* g_y, const float* g_hat, int64_t rows, int32_t in_dim,
                             int32_t out_dim, const e3k_gate_seg* segs, int32_t n_segs, int32_t out_cf, float* g_gy,
                             float* g_x, void* stream) {
  e3k::GateArgs ga{};
  const int rc = make_gate(segs, n_segs, ga);
  if (rc != E3K_OK) return rc;
  ga.out_cf = out_cf ? 1 : 0;
  if (rows < 0 || in_dim <= 0 || out_dim <= 0) return E3K_ERR_INVALID;
  if (rows == 0) return E3K_OK;
  if (!x || !g_hat || (!g_gy && !g_x) || (g_x && !g_y)) return E3K_ERR_INVALID;
  if (g_gy) {
    hipLaunchKernelGGL(e3k::gate_bwd2_gy_kernel, e3k::grid_cols(out_dim, rows), dim3(256), 0, (hipStream_t)stream,
                       x, g_hat, rows, in_dim, out_dim, ga, g_gy);
    E3K_CHECK_LAUNCH();
  }
  if (g_x) {
    hipLaunchKernelGGL(e3k::gate_bwd2_x_kernel, e3k::grid_cols(in_dim, rows), dim3(256), 0, (hipStream_t)stream, x,
                       g_y, g_hat, rows, in_dim, out_dim, ga, g_x);
    E3K_CHECK_LAUNCH();
  }
  return E3K_OK;
}

namespace {
int make_kw(const e3k_kw_instr* instr, int32_t n_instr, int32_t n_keys, int32_t V, int64_t ld_m, e3k::KwArgs& ka) {
  if (!instr || n_instr <= 0 || n_instr > e3k::KW_MAXI || n_keys <= 0 || n_keys > e3k::KW_MAXK || V <= 0 ||
      V > e3k::KW_MAXV)
    return E3K_ERR_UNSUPPORTED;
  ka.n = n_instr;
  ka.K = n_keys;
  ka.V = V;
  ka.ld_m = ld_m;
  int64_t pos = 0;
  for (int i = 0; i < n_instr; ++i) {
    if (instr[i].u <= 0 || instr[i].w_out <= 0 || instr[i].m_off != pos) return E3K_ERR_INVALID;   // columns are packed
    ka.ins[i] = instr[i];
    pos += (int64_t)instr[i].u * instr[i].w_out;
  }
  ka.total = pos;
  if (ld_m < pos) return E3K_ERR_INVALID;
  return E3K_OK;
}
}  // namespace

extern "C" int e3k_keyed_weights_fwd(const float* a, const float* W, const e3k_kw_instr* instr, int32_t n_instr,
                                     int32_t n_keys, int32_t V, int64_t ld_m, float* M, void* stream) {
  e3k::KwArgs ka{};
  const int rc = make_kw(instr, n_instr, n_keys, V, ld_m, ka);
  if (rc != E3K_OK) return rc;
  if (!a || !W || !M) return E3K_ERR_INVALID;
  hipLaunchKernelGGL(e3k::keyed_weights_kernel<0>,
                     dim3((unsigned)((ka.total + 255) / 256), e3k::kw_tiles(n_keys, (ka.total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, a, W, ka, M, (float*)nullptr, 0);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int64_t e3k_keyed_weights_bwd_workspace(const e3k_kw_instr* instr, int32_t n_instr, int32_t n_keys, int32_t V) {
  if (!instr || n_instr <= 0 || n_keys <= 0 || V <= 0) return 0;
  int64_t total = 0;
  for (int i = 0; i < n_instr; ++i) total += (int64_t)instr[i].u * instr[i].w_out;
  return ((total + e3k::KWA_COLS - 1) / e3k::KWA_COLS) * n_keys * V;
}

extern "C" int e3k_keyed_weights_bwd(const float* a, const float* W, const float* g_M, const e3k_kw_instr* instr,
                                     int32_t n_instr, int32_t n_keys, int32_t V, int64_t ld_m, float* g_a, float* g_W,
                                     int32_t accumulate_w, float* workspace, void* stream) {
  e3k::KwArgs ka{};
  const int rc = make_kw(instr, n_instr, n_keys, V, ld_m, ka);
  if (rc != E3K_OK) return rc;
  if (!a || !W || !g_M || (!g_a && !g_W)) return E3K_ERR_INVALID;
  if (g_W) {
    const unsigned tiles = e3k::kw_tiles(n_keys, (ka.total + 255) / 256);
    if (tiles > 1 && !accumulate_w) {   // the key tiles add with atomics: start from zero
      for (int i = 0; i < n_instr; ++i)
        if (e3k::zero_fill(g_W + instr[i].w_off, sizeof(float) * (size_t)instr[i].u * V * instr[i].w_out, (hipStream_t)stream))
          return E3K_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(e3k::keyed_weights_kernel<1>, dim3((unsigned)((ka.total + 255) / 256), tiles), dim3(256), 0,
                       (hipStream_t)stream, a, W, ka, const_cast<float*>(g_M), g_W, accumulate_w);
    E3K_CHECK_LAUNCH();
  }
  if (g_a) {
    if (!workspace) return E3K_ERR_INVALID;
    const int64_t blocks = (ka.total + e3k::KWA_COLS - 1) / e3k::KWA_COLS;
    hipLaunchKernelGGL(e3k::keyed_weights_bwd_a_kernel,
                       dim3((unsigned)blocks, (unsigned)((n_keys + e3k::KWA_RANGE - 1) / e3k::KWA_RANGE)), dim3(256), 0,
                       (hipStream_t)stream, g_M, W, ka, workspace);
    E3K_CHECK_LAUNCH();
    const int n = n_keys * V;
    hipLaunchKernelGGL(e3k::keyed_weights_bwd_a_reduce_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, workspace, (int)blocks, n, g_a);
    E3K_CHECK_LAUNCH();
  }
  return E3K_OK;
}

// ---- the keyed weights of several layers that share the attribute rows, one launch per pass ----------------------------
struct e3k_kw_args {
  e3k::KwArgs host;
  e3k::KwArgs* dev;
};

extern "C" int e3k_kw_args_create(const e3k_kw_instr* instr, int32_t n_instr, int32_t V, int64_t ld_m, e3k_kw_args** out) {
  if (!out) return E3K_ERR_INVALID;
  e3k_kw_args* o = new (std::nothrow) e3k_kw_args();
  if (!o) return E3K_ERR_LAUNCH;
  const int rc = make_kw(instr, n_instr, 1, V, ld_m, o->host);
  if (rc != E3K_OK) {
    delete o;
    return rc;
  }
  o->host.K = 0;      // (the number of keys is a per-call argument)
  if (hipMalloc(&o->dev, sizeof(e3k::KwArgs)) != hipSuccess ||
      hipMemcpy(o->dev, &o->host, sizeof(e3k::KwArgs), hipMemcpyHostToDevice) != hipSuccess) {
    if (o->dev) (void)hipFree(o->dev);
    delete o;
    return E3K_ERR_LAUNCH;
  }
  *out = o;
  return E3K_OK;
}

extern "C" void e3k_kw_args_destroy(e3k_kw_args* o) {
  if (!o) return;
  if (o->dev) (void)hipFree(o->dev);
  delete o;
}

namespace {
int fill_multi(const e3k_kw_multi_item* items, int32_t n, int32_t n_keys, e3k::KwMulti& m, int64_t& max_total, int64_t& ws_rows) {
  if (!items || n <= 0 || n > e3k::KW_MAXL || n_keys <= 0 || n_keys > e3k::KW_MAXK) return E3K_ERR_INVALID;
  m.n = n;
  m.K = n_keys;
  max_total = 0;
  ws_rows = 0;
  for (int i = 0; i < n; ++i) {
    const e3k_kw_multi_item& it = items[i];
    if (!it.args || !it.W || !it.M) return E3K_ERR_INVALID;
    if (it.args->host.V != items[0].args->host.V) return E3K_ERR_UNSUPPORTED;      // one attribute width for all
    m.ka[i] = it.args->dev;
    m.W[i] = it.W;
    m.M[i] = it.M;
    m.gW[i] = it.g_W;
    m.acc[i] = it.accumulate_w;
    m.ws_off[i] = ws_rows;
    const int64_t total = it.args->host.total;
    if (total > max_total) max_total = total;
    ws_rows += (total + e3k::KWA_COLS - 1) / e3k::KWA_COLS;
  }
  return E3K_OK;
}
}  // namespace

extern "C" int e3k_keyed_weights_fwd_multi(const e3k_kw_multi_item* items, int32_t n, const float* a, int32_t n_keys, void* stream) {
  e3k::KwMulti m{};
  int64_t max_total, ws_rows;
  const int rc = fill_multi(items, n, n_keys, m, max_total, ws_rows);
  if (rc != E3K_OK) return rc;
  if (!a) return E3K_ERR_INVALID;
  hipLaunchKernelGGL(e3k::keyed_weights_multi_kernel<0>,
                     dim3((unsigned)((max_total + 255) / 256), e3k::kw_tiles(n_keys, (max_total + 255) / 256 * n), (unsigned)n),
                     dim3(256), 0, (hipStream_t)stream, a, m);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int64_t e3k_keyed_weights_bwd_multi_workspace(const e3k_kw_multi_item* items, int32_t n, int32_t n_keys) {
  e3k::KwMulti m{};
  int64_t max_total, ws_rows;
  if (fill_multi(items, n, n_keys, m, max_total, ws_rows) != E3K_OK) return 0;
  return ws_rows * n_keys * items[0].args->host.V;
}

/* items[i].M = g_M of layer i; g_W written / accumulated per item (NULL: skipped); g_a [n_keys, V] ACCUMULATED over all the
 * layers (NULL: skipped; needs `workspace`). */
extern "C" int e3k_keyed_weights_bwd_multi(const e3k_kw_multi_item* items, int32_t n, const float* a, int32_t n_keys, float* g_a,
                                           float* workspace, void* stream) {
  e3k::KwMulti m{};
  int64_t max_total, ws_rows;
  const int rc = fill_multi(items, n, n_keys, m, max_total, ws_rows);
  if (rc != E3K_OK) return rc;
  if (!a) return E3K_ERR_INVALID;
  const int V = items[0].args->host.V;
  const unsigned tiles = e3k::kw_tiles(n_keys, (max_total + 255) / 256 * n);
  bool any_w = false;
  for (int i = 0; i < n; ++i) {
    if (!items[i].g_W) continue;
    any_w = true;
    if (tiles > 1 && !items[i].accumulate_w) {   // the key tiles add with atomics: start from zero
      const e3k::KwArgs& ka = items[i].args->host;
      for (int j = 0; j < ka.n; ++j)
        if (e3k::zero_fill(items[i].g_W + ka.ins[j].w_off, sizeof(float) * (size_t)ka.ins[j].u * V * ka.ins[j].w_out, (hipStream_t)stream))
          return E3K_ERR_LAUNCH;
    }
  }
  if (any_w) {
    hipLaunchKernelGGL(e3k::keyed_weights_multi_kernel<1>, dim3((unsigned)((max_total + 255) / 256), tiles, (unsigned)n), dim3(256), 0,
                       (hipStream_t)stream, a, m);
    E3K_CHECK_LAUNCH();
  }
  if (g_a) {
    if (!workspace) return E3K_ERR_INVALID;
    hipLaunchKernelGGL(e3k::keyed_weights_bwd_a_multi_kernel,
                       dim3((unsigned)((max_total + e3k::KWA_COLS - 1) / e3k::KWA_COLS),
                            (unsigned)((n_keys + e3k::KWA_RANGE - 1) / e3k::KWA_RANGE), (unsigned)n),
                       dim3(256), 0, (hipStream_t)stream, m, workspace);
    E3K_CHECK_LAUNCH();
    const int nn = n_keys * V;
    hipLaunchKernelGGL(e3k::keyed_weights_bwd_a_reduce_kernel, dim3((unsigned)((nn + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       workspace, (int)ws_rows, nn, g_a);
    E3K_CHECK_LAUNCH();
  }
  return E3K_OK;
}

extern "C" int e3k_norm_act_fwd(const float* x, int64_t rows, int32_t row_dim, const e3k_block* blocks, int32_t n_blocks,
                                int32_t act, float epsilon, int32_t normalize, float* y, void* stream) {
  e3k::BlockArgs ba{};
  const int rc = make_blocks(blocks, n_blocks, row_dim, ba);
  if (rc != E3K_OK) return rc;
  if (rows < 0 || n_blocks == 0 || act < 0 || act > 5 || epsilon < 0.f) return E3K_ERR_INVALID;
  if (rows == 0) return E3K_OK;
  if (!x || !y) return E3K_ERR_INVALID;
  hipLaunchKernelGGL(e3k::normact_fwd_kernel, dim3(e3k::grid_for(rows * row_dim)), dim3(256), 0, (hipStream_t)stream, x,
                     rows, row_dim, ba, act, epsilon * epsilon, normalize, y);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int e3k_norm_act_bwd(const float* x, const float* g_y, int64_t rows, int32_t row_dim, const e3k_block* blocks,
                                int32_t n_blocks, int32_t act, float epsilon, int32_t normalize, float* g_x,
                                void* stream) {
  e3k::BlockArgs ba{};
  const int rc = make_blocks(blocks, n_blocks, row_dim, ba);
  if (rc != E3K_OK) return rc;
  if (rows < 0 || n_blocks == 0 || act < 0 || act > 5 || epsilon < 0.f) return E3K_ERR_INVALID;
  if (rows == 0) return E3K_OK;
  if (!x || !g_y || !g_x) return E3K_ERR_INVALID;
  hipLaunchKernelGGL(e3k::normact_bwd_kernel, dim3(e3k::grid_for(rows * row_dim)), dim3(256), 0, (hipStream_t)stream, x,
                     g_y, rows, row_dim, ba, act, epsilon * epsilon, normalize, g_x);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int e3k_norm_act_bwd2(const float* x, const float* g_y, const float* h, int64_t rows, int32_t row_dim,
                                 const e3k_block* blocks, int32_t n_blocks, int32_t act, float epsilon, int32_t normalize,
                                 float* g_gy, float* g_x, void* stream) {
  e3k::BlockArgs ba{};
  const int rc = make_blocks(blocks, n_blocks, row_dim, ba);
  if (rc != E3K_OK) return rc;
  if (rows < 0 || n_blocks == 0 || act < 0 || act > 5 || epsilon < 0.f) return E3K_ERR_INVALID;
  if (rows == 0) return E3K_OK;
  if (!x || !g_y || !h || (!g_gy && !g_x)) return E3K_ERR_INVALID;
  hipLaunchKernelGGL(e3k::normact_bwd2_kernel, dim3(e3k::grid_for(rows * row_dim)), dim3(256), 0, (hipStream_t)stream, x, g_y, h,
                     rows, row_dim, ba, act, epsilon * epsilon, normalize, g_gy, g_x);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int e3k_layernorm_fwd(const float* x, int64_t rows, int32_t row_dim, const e3k_block* blocks,
                                 int32_t n_blocks, const float* std, float* y, float* inv_norm, void* stream) {
  e3k::BlockArgs ba{};
  const int rc = make_blocks(blocks, n_blocks, row_dim, ba);
  if (rc != E3K_OK) return rc;
  if (rows < 0 || n_blocks == 0) return E3K_ERR_INVALID;
  if (rows == 0) return E3K_OK;
  if (!x || !std || !y || !inv_norm) return E3K_ERR_INVALID;
  hipLaunchKernelGGL(e3k::layernorm_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x,
                     rows, row_dim, ba, std, y, inv_norm);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int e3k_layernorm_bwd(const float* x, const float* g_y, const float* inv_norm, int64_t rows,
                                 int32_t row_dim, const e3k_block* blocks, int32_t n_blocks, const float* std,
                                 float* g_x, float* g_std, void* stream) {
  e3k::BlockArgs ba{};
  const int rc = make_blocks(blocks, n_blocks, row_dim, ba);
  if (rc != E3K_OK) return rc;
  if (rows < 0 || n_blocks == 0) return E3K_ERR_INVALID;
  if (rows == 0) return E3K_OK;
  if (!x || !g_y || !inv_norm || !std || !g_x || !g_std) return E3K_ERR_INVALID;
  int64_t lblocks = (rows + 3) / 4;
  if (lblocks > 512) lblocks = 512;
  hipLaunchKernelGGL(e3k::layernorm_bwd_kernel, dim3((unsigned)lblocks), dim3(256), 0, (hipStream_t)stream, x, g_y,
                     inv_norm, rows, row_dim, ba, std, g_x, g_std);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int e3k_layernorm_bwd2(const float* x, const float* g_y, const float* h, const float* h_std, const float* inv_norm,
                                  int64_t rows, int32_t row_dim, const e3k_block* blocks, int32_t n_blocks, const float* std,
                                  float* g_gy, float* g_x, float* g_std, void* stream) {
  e3k::BlockArgs ba{};
  const int rc = make_blocks(blocks, n_blocks, row_dim, ba);
  if (rc != E3K_OK) return rc;
  if (rows < 0 || n_blocks == 0) return E3K_ERR_INVALID;
  if (rows == 0) return E3K_OK;
  if (!x || !g_y || !h || !inv_norm || !std || (!g_gy && !g_x && !g_std)) return E3K_ERR_INVALID;
  int64_t lblocks = (rows + 3) / 4;
  if (lblocks > 512) lblocks = 512;
  hipLaunchKernelGGL(e3k::layernorm_bwd2_kernel, dim3((unsigned)lblocks), dim3(256), 0, (hipStream_t)stream, x, g_y, h, h_std,
                     inv_norm, rows, row_dim, ba, std, g_gy, g_x, g_std);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int e3k_counts_to_ptr(const int64_t* counts, int32_t n_seg, int32_t* ptr, void* stream) {
  if (n_seg < 0 || !ptr || (n_seg > 0 && !counts)) return E3K_ERR_INVALID;
  hipLaunchKernelGGL(e3k::counts_to_ptr_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, counts, n_seg, ptr);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int e3k_onehot(const int64_t* idx, int64_t rows, int32_t num_types, float* out, int32_t* bad_flag, void* stream) {
  if (rows < 0 || num_types <= 0) return E3K_ERR_INVALID;
  if (rows == 0) return E3K_OK;
  if (!idx || !out) return E3K_ERR_INVALID;
  hipLaunchKernelGGL(e3k::onehot_kernel, dim3(e3k::grid_for(rows * num_types)), dim3(256), 0, (hipStream_t)stream, idx,
                     rows * num_types, num_types, out, bad_flag);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int e3k_flag_fetch_clear(int32_t* flag, int32_t* host_out, void* stream) {
  if (!flag || !host_out) return E3K_ERR_INVALID;
  hipLaunchKernelGGL(e3k::flag_fetch_clear_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, flag, host_out);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int e3k_segment_sum(const float* x, const int32_t* ptr, int64_t n_seg, int32_t dim, int32_t mean,
                               float* out, void* stream) {
  if (n_seg < 0 || dim <= 0) return E3K_ERR_INVALID;
  if (n_seg == 0) return E3K_OK;
  if (!x || !ptr || !out) return E3K_ERR_INVALID;
  hipLaunchKernelGGL(e3k::segment_sum_kernel, dim3(e3k::grid_for(n_seg * dim)), dim3(256), 0, (hipStream_t)stream, x,
                     ptr, n_seg, dim, mean, out);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}
