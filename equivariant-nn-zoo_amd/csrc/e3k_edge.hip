// Edge geometry kernels for gfx950: displacement vectors, real spherical harmonics (l <= 3),
// Bessel radial basis x cutoff envelope — forward and backward.
//
// Replaces (paths relative to /root/reference):
//   computeEdgeVector                         e3_layers/data/compute_edge.py:13-36
//   SphericalEncoding -> o3.SphericalHarmonics e3_layers/nn/embedding.py:163-178   
//   RadialBasisEncoding = BesselBasis * cutoff e3_layers/nn/embedding.py:114-127,31-40,26-29,210-219
//
// All of these are O(E) elementwise streams (<= 100 B per edge); one lane per edge.
#include "e3k_common.h"

namespace e3k {

// ---------------------------------------------------------------------------------------
// edge vectors
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void edge_vector_fwd_kernel(const float* __restrict__ pos,
                                                               const int32_t* __restrict__ src,
                                                               const int32_t* __restrict__ dst, int64_t E,
                                                               float* __restrict__ vec, float* __restrict__ len) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  const int s = src[e], d = dst[e];
  const float vx = pos[3 * (int64_t)d + 0] - pos[3 * (int64_t)s + 0];
  const float vy = pos[3 * (int64_t)d + 1] - pos[3 * (int64_t)s + 1];
  const float vz = pos[3 * (int64_t)d + 2] - pos[3 * (int64_t)s + 2];
  vec[3 * e + 0] = vx;
  vec[3 * e + 1] = vy;
  vec[3 * e + 2] = vz;
  if (len) len[e] = sqrtf(vx * vx + vy * vy + vz * vz);
}

__global__ __launch_bounds__(256) void edge_vector_bwd_kernel(const float* __restrict__ g_vec,
                                                               const float* __restrict__ g_len,
                                                               const float* __restrict__ vec,
                                                               const float* __restrict__ len,
                                                               const int32_t* __restrict__ dst_ptr,
                                                               const int32_t* __restrict__ dst_perm,
                                                               const int32_t* __restrict__ src_ptr,
                                                               const int32_t* __restrict__ src_perm, int64_t N,
                                                               float* __restrict__ g_pos) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  float ax = 0.f, ay = 0.f, az = 0.f;
  auto edge_grad = [&](int e, float sign) {
    float gx = g_vec ? g_vec[3 * (int64_t)e + 0] : 0.f;
    float gy = g_vec ? g_vec[3 * (int64_t)e + 1] : 0.f;
    float gz = g_vec ? g_vec[3 * (int64_t)e + 2] : 0.f;
    if (g_len) {
      const float l = len[e];
      const float f = l > 0.f ? g_len[e] / l : 0.f;
      gx = fmaf(f, vec[3 * (int64_t)e + 0], gx);
      gy = fmaf(f, vec[3 * (int64_t)e + 1], gy);
      gz = fmaf(f, vec[3 * (int64_t)e + 2], gz);
    }
    ax = fmaf(sign, gx, ax);
    ay = fmaf(sign, gy, ay);
    az = fmaf(sign, gz, az);
  };
  for (int t = dst_ptr[n]; t < dst_ptr[n + 1]; ++t) edge_grad(dst_perm[t], 1.f);
  for (int t = src_ptr[n]; t < src_ptr[n + 1]; ++t) edge_grad(src_perm[t], -1.f);
  g_pos[3 * n + 0] = ax;
  g_pos[3 * n + 1] = ay;
  g_pos[3 * n + 2] = az;
}

// ---------------------------------------------------------------------------------------
// spherical harmonics, component normalisation, polar axis y
// forward-mode duals give value and d/dx, d/dy, d/dz in one evaluation
// ---------------------------------------------------------------------------------------
struct D3 {
  float v, dx, dy, dz;
};
__device__ __forceinline__ D3 operator*(const D3& a, const D3& b) {
  return {a.v * b.v, fmaf(a.v, b.dx, a.dx * b.v), fmaf(a.v, b.dy, a.dy * b.v), fmaf(a.v, b.dz, a.dz * b.v)};
}
__device__ __forceinline__ D3 operator+(const D3& a, const D3& b) { return {a.v + b.v, a.dx + b.dx, a.dy + b.dy, a.dz + b.dz}; }
__device__ __forceinline__ D3 operator-(const D3& a, const D3& b) { return {a.v - b.v, a.dx - b.dx, a.dy - b.dy, a.dz - b.dz}; }
__device__ __forceinline__ D3 operator*(float s, const D3& a) { return {s * a.v, s * a.dx, s * a.dy, s * a.dz}; }

struct ShArgs {
  int n_ls;
  int ls[8];
  int normalize, normalization;
};

__device__ __forceinline__ float sh_norm_factor(int l, int normalization) {
  if (normalization == 1) return 0.28209479177387814f;          // 1/sqrt(4 pi)
  if (normalization == 2) return rsqrtf((float)(2 * l + 1));    // 'norm'
  return 1.0f;
}

// First-order dual number (value, derivative along ONE direction): the double-backward kernels run the
// same formulas on Du instead of float, which yields Jacobian-vector and Hessian-vector products.
struct Du {
  float v, e;
  __device__ __forceinline__ Du(float a = 0.f, float b = 0.f) : v(a), e(b) {}
};
__device__ __forceinline__ Du operator+(const Du& a, const Du& b) { return {a.v + b.v, a.e + b.e}; }
__device__ __forceinline__ Du operator-(const Du& a, const Du& b) { return {a.v - b.v, a.e - b.e}; }
__device__ __forceinline__ Du operator*(const Du& a, const Du& b) { return {a.v * b.v, fmaf(a.v, b.e, a.e * b.v)}; }
__device__ __forceinline__ Du operator*(float s, const Du& a) { return {s * a.v, s * a.e}; }
__device__ __forceinline__ Du operator/(const Du& a, const Du& b) {
  const float q = a.v / b.v;
  return {q, (a.e - q * b.e) / b.v};
}
__device__ __forceinline__ Du du_sqrt(const Du& a) {
  const float s = sqrtf(a.v);
  return {s, s > 0.f ? 0.5f * a.e / s : 0.f};
}
__device__ __forceinline__ Du du_sin(const Du& a) { return {sinf(a.v), cosf(a.v) * a.e}; }
__device__ __forceinline__ Du du_cos(const Du& a) { return {cosf(a.v), -sinf(a.v) * a.e}; }
__device__ __forceinline__ Du du_pow(const Du& a, float p) {   // a^p, a >= 0
  const float pm1 = (a.v != 0.f) ? powf(a.v, p - 1.0f) : 0.f;
  return {pm1 * a.v, p * pm1 * a.e};
}

// value + gradient whose four entries are themselves duals
struct D3u {
  Du v, dx, dy, dz;
};
__device__ __forceinline__ D3u operator*(const D3u& a, const D3u& b) {
  return {a.v * b.v, a.v * b.dx + a.dx * b.v, a.v * b.dy + a.dy * b.v, a.v * b.dz + a.dz * b.v};
}
__device__ __forceinline__ D3u operator+(const D3u& a, const D3u& b) { return {a.v + b.v, a.dx + b.dx, a.dy + b.dy, a.dz + b.dz}; }
__device__ __forceinline__ D3u operator-(const D3u& a, const D3u& b) { return {a.v - b.v, a.dx - b.dx, a.dy - b.dy, a.dz - b.dz}; }
__device__ __forceinline__ D3u operator*(float s, const D3u& a) { return {s * a.v, s * a.dx, s * a.dy, s * a.dz}; }

__device__ __forceinline__ void set_one(D3& o) { o = {1.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ void set_one(D3u& o) { o = {Du(1.f), Du(), Du(), Du()}; }

// evaluates degree l into out[0..2l]; returns count   (D3 = float duals, D3u = second-order)
template <class D3>
__device__ __forceinline__ int sh_eval(int l, const D3& x, const D3& y, const D3& z, D3* out) {
  const float s3 = 1.7320508075688772f, s5 = 2.23606797749979f, s15 = 3.872983346207417f, s7 = 2.6457513110645907f;
  if (l == 0) {
    set_one(out[0]);
    return 1;
  }
  if (l == 1) {
    out[0] = s3 * x;
    out[1] = s3 * y;
    out[2] = s3 * z;
    return 3;
  }
  const D3 x2 = x * x, y2 = y * y, z2 = z * z;
  const D3 x2z2 = x2 + z2;
  D3 q[5];
  q[0] = s15 * (x * z);
  q[1] = s15 * (x * y);
  q[2] = s5 * (y2 - 0.5f * x2z2);
  q[3] = s15 * (y * z);
  q[4] = (0.5f * s15) * (z2 - x2);
  if (l == 2) {
#pragma unroll
    for (int i = 0; i < 5; ++i) out[i] = q[i];
    return 5;
  }
  // l == 3
  const float a = 1.0801234497346435f;  // sqrt(42)/6
  const float b = 1.6201851746019651f;  // sqrt(168)/8
  const D3 f = 4.0f * y2 - x2z2;
  out[0] = a * (q[0] * z + q[4] * x);
  out[1] = s7 * (q[0] * y);
  out[2] = b * (f * x);
  out[3] = (0.5f * s7) * (y * (2.0f * y2 - 3.0f * x2z2));
  out[4] = b * (z * f);
  out[5] = s7 * (q[4] * y);
  out[6] = a * (q[4] * z - q[0] * x);
  return 7;
}

template <bool BWD>
__global__ __launch_bounds__(256) void sph_harm_kernel(const float* __restrict__ vec, const float* __restrict__ g_sh,
                                                        int64_t E, ShArgs sa, int dim, float* __restrict__ sh,
                                                        float* __restrict__ g_vec) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  const float vx = vec[3 * e], vy = vec[3 * e + 1], vz = vec[3 * e + 2];
  float ux = vx, uy = vy, uz = vz, inv = 1.f;
  if (sa.normalize) {
    const float r = sqrtf(vx * vx + vy * vy + vz * vz);
    inv = 1.0f / fmaxf(r, 1e-12f);
    ux *= inv;
    uy *= inv;
    uz *= inv;
  }
  const D3 x{ux, 1.f, 0.f, 0.f}, y{uy, 0.f, 1.f, 0.f}, z{uz, 0.f, 0.f, 1.f};
  int off = 0;
  float gx = 0.f, gy = 0.f, gz = 0.f;
  for (int i = 0; i < sa.n_ls; ++i) {
    const int l = sa.ls[i];
    D3 out[7];
    const int cnt = sh_eval(l, x, y, z, out);
    const float nf = sh_norm_factor(l, sa.normalization);
    for (int m = 0; m < cnt; ++m) {
      if constexpr (!BWD) {
        sh[e * dim + off + m] = nf * out[m].v;
      } else {
        const float g = nf * g_sh[e * dim + off + m];
        gx = fmaf(g, out[m].dx, gx);
        gy = fmaf(g, out[m].dy, gy);
        gz = fmaf(g, out[m].dz, gz);
      }
    }
    off += cnt;
  }
  if constexpr (BWD) {
    if (sa.normalize) {
      // u = v / |v|:  dL/dv = (g - u (u . g)) / |v|
      const float dot = gx * ux + gy * uy + gz * uz;
      gx = (gx - ux * dot) * inv;
      gy = (gy - uy * dot) * inv;
      gz = (gz - uz * dot) * inv;
    }
    g_vec[3 * e] = gx;
    g_vec[3 * e + 1] = gy;
    g_vec[3 * e + 2] = gz;
  }
}

// Backward of sph_harm_kernel<true>: g_vec = f(vec, g_sh) received the cotangent t = g_hat [E,3].
//   g_gsh[e,k] = d/d eps  sh_k(vec + eps t)                    (Jacobian-vector product)
//   g_vec2[e]  = d/d eps  f(vec + eps t, g_sh)                 (Hessian-vector product; the Hessian is symmetric)
// Both fall out of ONE evaluation of the forward+backward formulas on dual numbers.
__global__ __launch_bounds__(256) void sph_harm_bwd2_kernel(const float* __restrict__ vec, const float* __restrict__ g_sh,
                                                             const float* __restrict__ g_hat, int64_t E, ShArgs sa,
                                                             int dim, float* __restrict__ g_gsh,
                                                             float* __restrict__ g_vec2) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  const Du vx(vec[3 * e], g_hat[3 * e]), vy(vec[3 * e + 1], g_hat[3 * e + 1]), vz(vec[3 * e + 2], g_hat[3 * e + 2]);
  Du ux = vx, uy = vy, uz = vz, inv(1.f);
  if (sa.normalize) {
    const Du r = du_sqrt(vx * vx + vy * vy + vz * vz);
    inv = (r.v > 1e-12f) ? Du(1.f) / r : Du(1e12f);
    ux = ux * inv;
    uy = uy * inv;
    uz = uz * inv;
  }
  const D3u x{ux, Du(1.f), Du(), Du()}, y{uy, Du(), Du(1.f), Du()}, z{uz, Du(), Du(), Du(1.f)};
  int off = 0;
  Du gx, gy, gz;
  for (int i = 0; i < sa.n_ls; ++i) {
    const int l = sa.ls[i];
    D3u out[7];
    const int cnt = sh_eval(l, x, y, z, out);
    const float nf = sh_norm_factor(l, sa.normalization);
    for (int m = 0; m < cnt; ++m) {
      if (g_gsh) g_gsh[e * dim + off + m] = nf * out[m].v.e;
      if (g_vec2) {
        const float g = nf * g_sh[e * dim + off + m];
        gx = gx + g * out[m].dx;
        gy = gy + g * out[m].dy;
        gz = gz + g * out[m].dz;
      }
    }
    off += cnt;
  }
  if (g_vec2) {
    if (sa.normalize) {
      const Du dot = gx * ux + gy * uy + gz * uz;
      gx = (gx - ux * dot) * inv;
      gy = (gy - uy * dot) * inv;
      gz = (gz - uz * dot) * inv;
    }
    g_vec2[3 * e] = gx.e;
    g_vec2[3 * e + 1] = gy.e;
    g_vec2[3 * e + 2] = gz.e;
  }
}

// ---------------------------------------------------------------------------------------
// radial basis
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void cutoff_eval(float r, float r_max, float p, int kind, float& c, float& dc) {
  const float x = r / r_max;
  if (kind == 1) {
    // (x-1)^2 (x+1)^2 = (x^2-1)^2 on |x| < 1
    if (fabsf(x) < 1.0f) {
      const float q = x * x - 1.0f;
      c = q * q;
      dc = 4.0f * q * x / r_max;
    } else {
      c = 0.f;
      dc = 0.f;
    }
    return;
  }
  if (x < 1.0f) {
    const float xp = powf(x, p);
    const float c0 = (p + 1.0f) * (p + 2.0f) * 0.5f, c1 = p * (p + 2.0f), c2 = p * (p + 1.0f) * 0.5f;
    c = 1.0f - c0 * xp + c1 * xp * x - c2 * xp * x * x;
    const float xpm1 = (x != 0.0f) ? xp / x : 0.0f;  // x^(p-1)
    dc = (-c0 * p * xpm1 + c1 * (p + 1.0f) * xp - c2 * (p + 2.0f) * xp * x) / r_max;
  } else {
    c = 0.f;
    dc = 0.f;
  }
}

constexpr int RB_MAXB = 64;

__global__ __launch_bounds__(256) void radial_fwd_kernel(const float* __restrict__ r, int64_t E,
                                                          const float* __restrict__ bw, int nb, float r_max,
                                                          float r_min, float p, int one_over_r, int kind,
                                                          float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  const float rv = r[e];
  const float delta = r_max - r_min, pref = 2.0f / delta;
  float c, dc;
  cutoff_eval(rv, r_max, p, kind, c, dc);
  const float scale = one_over_r ? pref * c / rv : pref * c;
  for (int n = 0; n < nb; ++n) out[e * nb + n] = sinf(bw[n] * rv / delta) * scale;
}

template <int MAXB>   // compile-time bound of the (unrolled) basis loop: 8 / 16 / 32 / 64
__global__ __launch_bounds__(256) void radial_bwd_kernel(const float* __restrict__ r, const float* __restrict__ g_out,
                                                          int64_t E, const float* __restrict__ bw, int nb, float r_max,
                                                          float r_min, float p, int one_over_r, int kind,
                                                          float* __restrict__ g_r, float* __restrict__ g_w) {
  // grid-stride over edges; the frequency gradients are summed per thread first, so the whole launch
  // issues (waves x n_basis) atomics instead of (E / 64 x n_basis)
  const float delta = r_max - r_min, pref = 2.0f / delta;
  float acc[MAXB];
#pragma unroll
  for (int n = 0; n < MAXB; ++n) acc[n] = 0.f;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < E; e += (int64_t)gridDim.x * 256) {
    const float rv = r[e];
    float c, dc;
    cutoff_eval(rv, r_max, p, kind, c, dc);
    const float inv_r = one_over_r ? 1.0f / rv : 1.0f;
    float gr = 0.f;
#pragma unroll
    for (int n = 0; n < MAXB; ++n) {
      if (n < nb) {
        const float w = bw[n];
        float sn, cs;
        sincosf(w * rv / delta, &sn, &cs);
        const float g = g_out[e * nb + n];
        // out = pref * sin(arg) * inv_r * c
        float dbasis = pref * cs * (w / delta) * inv_r;
        if (one_over_r) dbasis -= pref * sn * inv_r * inv_r;
        gr = fmaf(g, dbasis * c + pref * sn * inv_r * dc, gr);
        acc[n] = fmaf(g, pref * cs * (rv / delta) * inv_r * c, acc[n]);
      }
    }
    if (g_r) g_r[e] = gr;
  }
  if (g_w) {
    // the block's four waves meet in LDS: one atomic per basis function and block (same-address atomics from
    // every wave of the launch serialise)
    __shared__ float part[4][MAXB];
#pragma unroll
    for (int n = 0; n < MAXB; ++n) {
      if (n < nb) {
        const float tot = wave_sum(acc[n]);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6][n] = tot;
      }
    }
    __syncthreads();
    if ((int)threadIdx.x < nb) atomicAdd(g_w + threadIdx.x, part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
  }
}

// cutoff envelope on duals (same two kinds as cutoff_eval); c only — its derivative comes from the dual part
__device__ __forceinline__ void cutoff_du(const Du& r, float r_max, float p, int kind, Du& c, Du& dc) {
  const Du x = (1.0f / r_max) * r;
  if (kind == 1) {
    if (fabsf(x.v) < 1.0f) {
      const Du q = x * x - Du(1.0f);
      c = q * q;
      dc = (4.0f / r_max) * (q * x);
    } else {
      c = Du();
      dc = Du();
    }
    return;
  }
  if (x.v < 1.0f) {
    const Du xp = du_pow(x, p), xpm1 = du_pow(x, p - 1.0f);
    const float c0 = (p + 1.0f) * (p + 2.0f) * 0.5f, c1 = p * (p + 2.0f), c2 = p * (p + 1.0f) * 0.5f;
    c = Du(1.0f) - c0 * xp + c1 * (xp * x) - c2 * (xp * x * x);
    dc = (1.0f / r_max) * ((-c0 * p) * xpm1 + (c1 * (p + 1.0f)) * xp - (c2 * (p + 2.0f)) * (xp * x));
  } else {
    c = Du();
    dc = Du();
  }
}

// Backward of radial_bwd_kernel: (g_r, g_w) = f(r, w, g_out) received cotangents (hat_r [E], hat_w [nb]).
//   g_gout[e,n] = d/d eps out_n(r_e + eps hat_r_e, w_n + eps hat_w_n)
//   g_r2, g_w2  = d/d eps f(r + eps hat_r, w + eps hat_w, g_out)
__global__ __launch_bounds__(256) void radial_bwd2_kernel(const float* __restrict__ r, const float* __restrict__ g_out,
                                                           const float* __restrict__ hat_r,
                                                           const float* __restrict__ hat_w, int64_t E,
                                                           const float* __restrict__ bw, int nb, float r_max,
                                                           float r_min, float p, int one_over_r, int kind,
                                                           float* __restrict__ g_gout, float* __restrict__ g_r2,
                                                           float* __restrict__ g_w2) {
  const float delta = r_max - r_min, pref = 2.0f / delta;
  float acc[RB_MAXB];
#pragma unroll
  for (int n = 0; n < RB_MAXB; ++n) acc[n] = 0.f;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < E; e += (int64_t)gridDim.x * 256) {
    const Du rv(r[e], hat_r ? hat_r[e] : 0.f);
    Du c, dc;
    cutoff_du(rv, r_max, p, kind, c, dc);
    const Du inv_r = one_over_r ? Du(1.0f) / rv : Du(1.0f);
    float gr = 0.f;
#pragma unroll
    for (int n = 0; n < RB_MAXB; ++n) {
      if (n < nb) {
        const Du w(bw[n], hat_w ? hat_w[n] : 0.f);
        const Du arg = (1.0f / delta) * (w * rv);
        const Du sn = du_sin(arg), cs = du_cos(arg);
        const Du out = pref * (sn * inv_r * c);
        if (g_gout) g_gout[e * nb + n] = out.e;
        const float g = g_out[e * nb + n];
        Du dbasis = (pref / delta) * (cs * w * inv_r);
        if (one_over_r) dbasis = dbasis - pref * (sn * inv_r * inv_r);
        const Du d_r = dbasis * c + pref * (sn * inv_r * dc);
        const Du d_w = (pref / delta) * (cs * rv * inv_r * c);
        gr = fmaf(g, d_r.e, gr);
        acc[n] = fmaf(g, d_w.e, acc[n]);
      }
    }
    if (g_r2) g_r2[e] = gr;
  }
  if (g_w2) {
#pragma unroll
    for (int n = 0; n < RB_MAXB; ++n) {
      if (n < nb) {
        const float tot = wave_sum(acc[n]);
        if ((threadIdx.x & 63) == 0) atomicAdd(g_w2 + n, tot);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// radius graph (computeEdgeIndex, e3_layers/data/compute_edge.py:38-113) on the device
// One wave per source node i walks its own graph's nodes 64 at a time; a ballot compacts the kept
// pairs in ascending j, so the edge list comes out in the reference's order — graphs concatenated,
// (i, j) lexicographic inside a graph — without a sort.  Two passes (count, exclusive scan on the host
// side with torch.cumsum, fill).  Distance test exactly as the reference states it: fp32
// sqrt(dx^2 + dy^2 + dz^2) < r_max, strict, no fused multiply-add.
// ---------------------------------------------------------------------------------------
template <bool FILL>
__global__ __launch_bounds__(256) void radius_graph_kernel(const float* __restrict__ pos,
                                                            const int32_t* __restrict__ g_start,
                                                            const int32_t* __restrict__ g_end, int64_t N, float r_max,
                                                            const int32_t* __restrict__ old_ptr,
                                                            const int32_t* __restrict__ old_dst,
                                                            int32_t* __restrict__ counts,
                                                            const int64_t* __restrict__ offsets,
                                                            int64_t* __restrict__ edge_index, int64_t E) {
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= N) return;
  const int lane = threadIdx.x & 63;
  const int beg = g_start[i], end = g_end[i];
  const float px = pos[3 * i], py = pos[3 * i + 1], pz = pos[3 * i + 2];
  const int ob = old_ptr ? old_ptr[i] : 0, oe = old_ptr ? old_ptr[i + 1] : 0;
  const int64_t off = FILL ? offsets[i] : 0;
  int cnt = 0;
  for (int j0 = beg; j0 < end; j0 += 64) {
    const int j = j0 + lane;
    bool keep = false;
    if (j < end) {
      if (j != i) {
        const float dx = __fsub_rn(px, pos[3 * (int64_t)j]);
        const float dy = __fsub_rn(py, pos[3 * (int64_t)j + 1]);
        const float dz = __fsub_rn(pz, pos[3 * (int64_t)j + 2]);
        const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        keep = __fsqrt_rn(d2) < r_max;
      }
      if (!keep && oe > ob) {   // pre-existing edges stay (binary search in i's sorted old neighbours)
        int lo = ob, hi = oe;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (old_dst[mid] < j) lo = mid + 1; else hi = mid;
        }
        keep = lo < oe && old_dst[lo] == j;
      }
    }
    const unsigned long long mask = __ballot(keep);
    if constexpr (FILL) {
      if (keep) {
        const int64_t at = off + cnt + __popcll(mask & ((1ull << lane) - 1ull));
        edge_index[at] = i;
        edge_index[E + at] = j;
      }
    }
    cnt += __popcll(mask);
  }
  if constexpr (!FILL) {
    if (lane == 0) counts[i] = cnt;
  }
}

}  // namespace e3k

extern "C" int e3k_edge_vector_fwd(const float* pos, const int32_t* src, const int32_t* dst, int64_t E,
                                   float* edge_vec, float* edge_len, void* stream) {
  if (E < 0) return E3K_ERR_INVALID;
  if (E == 0) return E3K_OK;
  if (!pos || !src || !dst || !edge_vec) return E3K_ERR_INVALID;
  hipLaunchKernelGGL(e3k::edge_vector_fwd_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     pos, src, dst, E, edge_vec, edge_len);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int e3k_edge_vector_bwd(const float* g_vec, const float* g_len, const float* edge_vec,
                                   const float* edge_len, const int32_t* dst_ptr, const int32_t* dst_perm,
                                   const int32_t* src_ptr, const int32_t* src_perm, int64_t N, float* g_pos,
                                   void* stream) {
  if (N < 0) return E3K_ERR_INVALID;
  if (N == 0) return E3K_OK;
  if (!dst_ptr || !src_ptr || !g_pos || (!g_vec && !g_len)) return E3K_ERR_INVALID;
  if (g_len && (!edge_vec || !edge_len)) return E3K_ERR_INVALID;
  hipLaunchKernelGGL(e3k::edge_vector_bwd_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     g_vec, g_len, edge_vec, edge_len, dst_ptr, dst_perm, src_ptr, src_perm, N, g_pos);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

namespace {
int make_sh_args(const int32_t* ls, int32_t n_ls, int32_t normalize, int32_t normalization, e3k::ShArgs& sa, int& dim) {
  if (!ls || n_ls <= 0 || n_ls > 8) return E3K_ERR_INVALID;
  if (normalization < 0 || normalization > 2) return E3K_ERR_INVALID;
  dim = 0;
  sa.n_ls = n_ls;
  for (int i = 0; i < n_ls; ++i) {
    if (ls[i] < 0) return E3K_ERR_INVALID;
    if (ls[i] > 3) return E3K_ERR_UNSUPPORTED;
    sa.ls[i] = ls[i];
    dim += 2 * ls[i] + 1;
  }
  sa.normalize = normalize;
  sa.normalization = normalization;
  return E3K_OK;
}
}  // namespace

/* `ls` is a HOST array (copied into the kernel arguments). */
extern "C" int e3k_sph_harm_fwd(const float* vec, int64_t E, const int32_t* ls, int32_t n_ls, int32_t normalize,
                                int32_t normalization, float* sh, void* stream) {
  e3k::ShArgs sa{};
  int dim = 0;
  const int rc = make_sh_args(ls, n_ls, normalize, normalization, sa, dim);
  if (rc != E3K_OK) return rc;
  if (E < 0) return E3K_ERR_INVALID;
  if (E == 0) return E3K_OK;
  if (!vec || !sh) return E3K_ERR_INVALID;
  hipLaunchKernelGGL(e3k::sph_harm_kernel<false>, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     vec, (const float*)nullptr, E, sa, dim, sh, (float*)nullptr);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int e3k_sph_harm_bwd(const float* vec, const float* g_sh, int64_t E, const int32_t* ls, int32_t n_ls,
                                int32_t normalize, int32_t normalization, float* g_vec, void* stream) {
  e3k::ShArgs sa{};
  int dim = 0;
  const int rc = make_sh_args(ls, n_ls, normalize, normalization, sa, dim);
  if (rc != E3K_OK) return rc;
  if (E < 0) return E3K_ERR_INVALID;
  if (E == 0) return E3K_OK;
  if (!vec || !g_sh || !g_vec) return E3K_ERR_INVALID;
  hipLaunchKernelGGL(e3k::sph_harm_kernel<true>, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     vec, g_sh, E, sa, dim, (float*)nullptr, g_vec);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int e3k_sph_harm_bwd2(const float* vec, const float* g_sh, const float* g_hat, int64_t E, const int32_t* ls,
                                 int32_t n_ls, int32_t normalize, int32_t normalization, float* g_gsh, float* g_vec,
                                 void* stream) {
  e3k::ShArgs sa{};
  int dim = 0;
  const int rc = make_sh_args(ls, n_ls, normalize, normalization, sa, dim);
  if (rc != E3K_OK) return rc;
  if (E < 0) return E3K_ERR_INVALID;
  if (E == 0) return E3K_OK;
  if (!vec || !g_hat || (!g_gsh && !g_vec) || (g_vec && !g_sh)) return E3K_ERR_INVALID;
  hipLaunchKernelGGL(e3k::sph_harm_bwd2_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, (hipStream_t)stream, vec,
                     g_sh, g_hat, E, sa, dim, g_gsh, g_vec);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int e3k_radial_basis_fwd(const float* r, int64_t E, const float* bessel_w, int32_t n_basis, float r_max,
                                    float r_min, float p, int32_t one_over_r, int32_t cutoff_kind, float* out,
                                    void* stream) {
  if (E < 0 || n_basis <= 0 || n_basis > e3k::RB_MAXB || !(r_max > r_min)) return E3K_ERR_INVALID;
  if (cutoff_kind < 0 || cutoff_kind > 1) return E3K_ERR_INVALID;
  if (E == 0) return E3K_OK;
  if (!r || !bessel_w || !out) return E3K_ERR_INVALID;
  hipLaunchKernelGGL(e3k::radial_fwd_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, (hipStream_t)stream, r, E,
                     bessel_w, n_basis, r_max, r_min, p, one_over_r, cutoff_kind, out);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int e3k_radial_basis_bwd(const float* r, const float* g_out, int64_t E, const float* bessel_w,
                                    int32_t n_basis, float r_max, float r_min, float p, int32_t one_over_r,
                                    int32_t cutoff_kind, float* g_r, float* g_w, void* stream) {
  if (E < 0 || n_basis <= 0 || n_basis > e3k::RB_MAXB || !(r_max > r_min)) return E3K_ERR_INVALID;
  if (cutoff_kind < 0 || cutoff_kind > 1) return E3K_ERR_INVALID;
  if (E == 0) return E3K_OK;
  if (!r || !g_out || !bessel_w || (!g_r && !g_w)) return E3K_ERR_INVALID;
  int64_t blocks = (E + 255) / 256;
  if (blocks > 256) blocks = 256;
#define E3K_RB_LAUNCH(MB)                                                                                            \
  hipLaunchKernelGGL(e3k::radial_bwd_kernel<MB>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, r, g_out, E, \
                     bessel_w, n_basis, r_max, r_min, p, one_over_r, cutoff_kind, g_r, g_w)
  if (n_basis <= 8) E3K_RB_LAUNCH(8);
  else if (n_basis <= 16) E3K_RB_LAUNCH(16);
  else if (n_basis <= 32) E3K_RB_LAUNCH(32);
  else E3K_RB_LAUNCH(64);
#undef E3K_RB_LAUNCH
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int e3k_radial_basis_bwd2(const float* r, const float* g_out, const float* hat_r, const float* hat_w,
                                     int64_t E, const float* bessel_w, int32_t n_basis, float r_max, float r_min,
                                     float p, int32_t one_over_r, int32_t cutoff_kind, float* g_gout, float* g_r,
                                     float* g_w, void* stream) {
  if (E < 0 || n_basis <= 0 || n_basis > e3k::RB_MAXB || !(r_max > r_min)) return E3K_ERR_INVALID;
  if (cutoff_kind < 0 || cutoff_kind > 1) return E3K_ERR_INVALID;
  if (E == 0) return E3K_OK;
  if (!r || !g_out || !bessel_w || (!hat_r && !hat_w) || (!g_gout && !g_r && !g_w)) return E3K_ERR_INVALID;
  int64_t blocks = (E + 255) / 256;
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(e3k::radial_bwd2_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, r, g_out, hat_r,
                     hat_w, E, bessel_w, n_basis, r_max, r_min, p, one_over_r, cutoff_kind, g_gout, g_r, g_w);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int e3k_radius_graph_count(const float* pos, const int32_t* graph_start, const int32_t* graph_end, int64_t N,
                                      float r_max, const int32_t* old_ptr, const int32_t* old_dst, int32_t* counts,
                                      void* stream) {
  if (N < 0) return E3K_ERR_INVALID;
  if (N == 0) return E3K_OK;
  if (!pos || !graph_start || !graph_end || !counts || (old_ptr && !old_dst)) return E3K_ERR_INVALID;
  hipLaunchKernelGGL(e3k::radius_graph_kernel<false>, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, pos,
                     graph_start, graph_end, N, r_max, old_ptr, old_dst, counts, (const int64_t*)nullptr,
                     (int64_t*)nullptr, (int64_t)0);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int e3k_radius_graph_fill(const float* pos, const int32_t* graph_start, const int32_t* graph_end, int64_t N,
                                     float r_max, const int32_t* old_ptr, const int32_t* old_dst, const int64_t* offsets,
                                     int64_t E, int64_t* edge_index, void* stream) {
  if (N < 0 || E < 0) return E3K_ERR_INVALID;
  if (N == 0 || E == 0) return E3K_OK;
  if (!pos || !graph_start || !graph_end || !offsets || !edge_index || (old_ptr && !old_dst)) return E3K_ERR_INVALID;
  hipLaunchKernelGGL(e3k::radius_graph_kernel<true>, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, pos,
                     graph_start, graph_end, N, r_max, old_ptr, old_dst, (int32_t*)nullptr, offsets, edge_index, E);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}
