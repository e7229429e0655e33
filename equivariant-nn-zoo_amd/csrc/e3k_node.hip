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

__global__ __launch_bounds__(256) void relayout_kernel(const float* __restrict__ x, int64_t rows, int row_dim,
                                                        BlockArgs ba, int to_cf, float* __restrict__ y) {
  const int64_t total = rows * row_dim;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / row_dim;
    const int c = (int)(i - r * row_dim);
    // i indexes the OUTPUT element; find its block
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
    y[i] = x[r * row_dim + src];
  }
}

// ---------------------------------------------------------------------------------------
// Gate
// ---------------------------------------------------------------------------------------
struct GateArgs {
  int n;
  e3k_gate_seg s[MAXBLK];
};

__global__ __launch_bounds__(256) void gate_fwd_kernel(const float* __restrict__ x, int64_t rows, int in_dim,
                                                        int out_dim, GateArgs ga, float* __restrict__ y) {
  const int64_t total = rows * out_dim;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / out_dim;
    const int c = (int)(i - r * out_dim);
    const float* xr = x + r * in_dim;
    float v = 0.f;
    for (int k = 0; k < ga.n; ++k) {
      const e3k_gate_seg& s = ga.s[k];
      const int rel = c - s.out_off;
      if (rel >= 0 && rel < s.mul * s.dim) {
        if (s.kind == 0) {
          v = s.cst * act_f(s.act, xr[s.in_off + rel]);
        } else {
          const int u = rel / s.dim, m = rel - u * s.dim;
          v = xr[s.in_off + m * s.mul + u] * (s.cst * act_f(s.act, xr[s.gate_off + u]));
        }
        break;
      }
    }
    y[i] = v;
  }
}

__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                        int64_t rows, int in_dim, int out_dim, GateArgs ga,
                                                        float* __restrict__ gx) {
  const int64_t total = rows * in_dim;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / in_dim;
    const int c = (int)(i - r * in_dim);
    const float* xr = x + r * in_dim;
    const float* gr = gy + r * out_dim;
    float v = 0.f;
    for (int k = 0; k < ga.n; ++k) {
      const e3k_gate_seg& s = ga.s[k];
      if (s.kind == 0) {
        const int rel = c - s.in_off;
        if (rel >= 0 && rel < s.mul) {
          v = gr[s.out_off + rel] * s.cst * act_df(s.act, xr[c]);
          break;
        }
      } else {
        const int relg = c - s.gate_off;
        if (relg >= 0 && relg < s.mul) {
          float dot = 0.f;
          for (int m = 0; m < s.dim; ++m) dot = fmaf(gr[s.out_off + relg * s.dim + m], xr[s.in_off + m * s.mul + relg], dot);
          v = dot * s.cst * act_df(s.act, xr[c]);
          break;
        }
        const int rel = c - s.in_off;
        if (rel >= 0 && rel < s.mul * s.dim) {
          const int m = rel / s.mul, u = rel - m * s.mul;
          v = gr[s.out_off + u * s.dim + m] * (s.cst * act_f(s.act, xr[s.gate_off + u]));
          break;
        }
      }
    }
    gx[i] = v;
  }
}

// backward of gate_bwd (cotangent gh on gx): g_gy = (d y / d x) gh  — one thread per OUTPUT element
__global__ __launch_bounds__(256) void gate_bwd2_gy_kernel(const float* __restrict__ x, const float* __restrict__ gh,
                                                            int64_t rows, int in_dim, int out_dim, GateArgs ga,
                                                            float* __restrict__ g_gy) {
  const int64_t total = rows * out_dim;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / out_dim;
    const int c = (int)(i - r * out_dim);
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
          const int u = rel / s.dim, m = rel - u * s.dim;
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
  const int64_t total = rows * in_dim;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / in_dim;
    const int c = (int)(i - r * in_dim);
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
            const float g = gr[s.out_off + relg * s.dim + m];
            dgh = fmaf(g, hr[s.in_off + m * s.mul + relg], dgh);
            dgx = fmaf(g, xr[s.in_off + m * s.mul + relg], dgx);
          }
          v = s.cst * (dgh * act_df(s.act, xr[c]) + hr[c] * dgx * act_d2f(s.act, xr[c]));
          break;
        }
        const int rel = c - s.in_off;
        if (rel >= 0 && rel < s.mul * s.dim) {
          const int m = rel / s.mul, u = rel - m * s.mul;
          v = hr[s.gate_off + u] * gr[s.out_off + u * s.dim + m] * (s.cst * act_df(s.act, xr[s.gate_off + u]));
          break;
        }
      }
    }
    g_x[i] = v;
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
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  for (int k = 0; k < ba.n; ++k) {
    const e3k_block& b = ba.b[k];
    const int len = b.mul * b.dim;
    const float* xb = x + r * row_dim + b.off;
    const float* gb = gy + r * row_dim + b.off;
    float dot = 0.f;
    for (int j = lane; j < len; j += 64) dot = fmaf(gb[j], xb[j], dot);
    dot = wave_sum(dot);
    const float inv = inv_norm[r * ba.n + k], s = stdv[k];
    // y = s * x * inv, inv = (sum x^2 / mul + eps)^-1/2  =>  dx = s*inv*(g - x * dot * inv^2 / mul)
    const float coef = dot * inv * inv / (float)b.mul;
    float* gxb = gx + r * row_dim + b.off;
    for (int j = lane; j < len; j += 64) gxb[j] = s * inv * (gb[j] - xb[j] * coef);
    if (lane == 0) atomicAdd(g_std + k, dot * inv);
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
  hipLaunchKernelGGL(e3k::relayout_kernel, dim3(e3k::grid_for(rows * row_dim)), dim3(256), 0, (hipStream_t)stream, x,
                     rows, row_dim, ba, to_cf, y);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int e3k_gate_fwd(const float* x, int64_t rows, int32_t in_dim, int32_t out_dim, const e3k_gate_seg* segs,
                            int32_t n_segs, float* y, void* stream) {
  e3k::GateArgs ga{};
  const int rc = make_gate(segs, n_segs, ga);
  if (rc != E3K_OK) return rc;
  if (rows < 0 || in_dim <= 0 || out_dim <= 0) return E3K_ERR_INVALID;
  if (rows == 0) return E3K_OK;
  if (!x || !y) return E3K_ERR_INVALID;
  hipLaunchKernelGGL(e3k::gate_fwd_kernel, dim3(e3k::grid_for(rows * out_dim)), dim3(256), 0, (hipStream_t)stream, x,
                     rows, in_dim, out_dim, ga, y);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int e3k_gate_bwd(const float* x, const float* g_y, int64_t rows, int32_t in_dim, int32_t out_dim,
                            const e3k_gate_seg* segs, int32_t n_segs, float* g_x, void* stream) {
  e3k::GateArgs ga{};
  const int rc = make_gate(segs, n_segs, ga);
  if (rc != E3K_OK) return rc;
  if (rows < 0 || in_dim <= 0 || out_dim <= 0) return E3K_ERR_INVALID;
  if (rows == 0) return E3K_OK;
  if (!x || !g_y || !g_x) return E3K_ERR_INVALID;
  hipLaunchKernelGGL(e3k::gate_bwd_kernel, dim3(e3k::grid_for(rows * in_dim)), dim3(256), 0, (hipStream_t)stream, x, g_y,
                     rows, in_dim, out_dim, ga, g_x);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int e3k_gate_bwd2(const float* x, const float* g_y, const float* g_hat, int64_t rows, int32_t in_dim,
                             int32_t out_dim, const e3k_gate_seg* segs, int32_t n_segs, float* g_gy, float* g_x,
                             void* stream) {
  e3k::GateArgs ga{};
  const int rc = make_gate(segs, n_segs, ga);
  if (rc != E3K_OK) return rc;
  if (rows < 0 || in_dim <= 0 || out_dim <= 0) return E3K_ERR_INVALID;
  if (rows == 0) return E3K_OK;
  if (!x || !g_hat || (!g_gy && !g_x) || (g_x && !g_y)) return E3K_ERR_INVALID;
  if (g_gy) {
    hipLaunchKernelGGL(e3k::gate_bwd2_gy_kernel, dim3(e3k::grid_for(rows * out_dim)), dim3(256), 0, (hipStream_t)stream,
                       x, g_hat, rows, in_dim, out_dim, ga, g_gy);
    E3K_CHECK_LAUNCH();
  }
  if (g_x) {
    hipLaunchKernelGGL(e3k::gate_bwd2_x_kernel, dim3(e3k::grid_for(rows * in_dim)), dim3(256), 0, (hipStream_t)stream, x,
                       g_y, g_hat, rows, in_dim, out_dim, ga, g_x);
    E3K_CHECK_LAUNCH();
  }
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
  hipLaunchKernelGGL(e3k::layernorm_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x,
                     g_y, inv_norm, rows, row_dim, ba, std, g_x, g_std);
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
