// The SLOPE of the radial knot table for gfx950: D [K + 1, W] = d/dr fc(basis(r)) on the knots.
//
// Why it exists (reference: forces are -dE/dpos by autograd, e3_layers/nn/output.py:31-53; the per-edge path weights
// weight = self.fc(edge_radial), nn/message_passing.py:93, then depend on pos through the edge length): with the weights
// interpolated from a table T on knots h apart, dw/dr could be had by differentiating the interpolation weights -- but T is
// stored in fp32, and any difference quotient of fp32 samples h apart carries their rounding times 1 / h: measured 7e-6
// (K = 768) to 4e-5 (K = 4096) relative slope error, at or beyond the 1e-5 force tolerance, at EVERY knot count.  So the slope
// is tabulated as a function in its own right and interpolated with the same four weights (measured 5e-8 .. 3e-7):
//   H'(r_k) = d/dr of the radial MLP's last hidden activation, by FORWARD-MODE differentiation of the hidden chain
//             (basis -> L x (linear, act)), carried out in float64 per knot (K + 1 rows x 64 units per net: nothing);
//   D       = H' W_last / sqrt(h_in): the MLP's LAST layer applied to H' by the ordinary fp32 MFMA GEMM.
// D is a function of the parameters.  Its gradient: g_H' = g_D W_last^T and g_W_last += H'^T g_D are the ordinary GEMMs;
// the hidden weights' and the Bessel frequencies' share is the reverse sweep of the tangent chain (second derivatives of
// the activation), again per knot in float64 (slope_tangent_bwd_kernel) -- a first version pushed S^T g_H' (S = a
// difference stencil over the knots) through the fp32 chain backward and lost 2e-4 on the Bessel frequencies to
// cancellation: a rough g_H' differenced over neighbouring knots is 400x larger than the gradient it sums to.
#include "e3k_common.h"

namespace e3k {

constexpr int SL_ROWS = 2;       // knot rows per workgroup (R / 2 workgroups of 64 threads per net: the chain is latency, not work)
constexpr int SL_MAXH = 64;
constexpr int SL_MAXL = 4;

__device__ __forceinline__ double sig64(double x) { return 1.0 / (1.0 + exp(-x)); }
// act, act', act'' (ids of e3k_act.h: 0 identity, 1 ssp, 2 silu, 4 tanh)
__device__ __forceinline__ void act64(int id, double x, double& f, double& d1, double& d2) {
  switch (id) {
    case 1: {
      const double s = sig64(x);
      f = fmax(x, 0.0) + log1p(exp(-fabs(x))) - 0.69314718055994530942;
      d1 = s;
      d2 = s * (1.0 - s);
      return;
    }
    case 2: {
      const double s = sig64(x);
      f = x * s;
      d1 = s * (1.0 + x * (1.0 - s));
      d2 = s * (1.0 - s) * (2.0 + x * (1.0 - 2.0 * s));
      return;
    }
    case 4: {
      const double t = tanh(x);
      f = t;
      d1 = 1.0 - t * t;
      d2 = -2.0 * t * d1;
      return;
    }
    default: f = x; d1 = 1.0; d2 = 0.0; return;
  }
}

struct BasisPar {
  double r_max, r_min, p;
  int one_over_r, kind;
};
// b_n(r) = A(r) sin(w r / delta) (nn/embedding.py:114-127 x :31-40, as csrc/e3k_edge.hip radial_fwd_kernel evaluates it in fp32):
// value, d/dr, and the derivatives of both w.r.t. the frequency w
__device__ __forceinline__ void basis64(double r, double w, const BasisPar& bp, double& b, double& db, double& b_w, double& db_w) {
  const double delta = bp.r_max - bp.r_min, pref = 2.0 / delta, x = r / bp.r_max, p = bp.p;
  double c = 0.0, dc = 0.0;
  if (bp.kind == 1) {
    if (fabs(x) < 1.0) {
      const double q = x * x - 1.0;
      c = q * q;
      dc = 4.0 * q * x / bp.r_max;
    }
  } else if (x < 1.0) {
    const double xp = pow(x, p), c0 = (p + 1.0) * (p + 2.0) * 0.5, c1 = p * (p + 2.0), c2 = p * (p + 1.0) * 0.5;
    c = 1.0 - c0 * xp + c1 * xp * x - c2 * xp * x * x;
    const double xpm1 = x != 0.0 ? xp / x : 0.0;
    dc = (-c0 * p * xpm1 + c1 * (p + 1.0) * xp - c2 * (p + 2.0) * xp * x) / bp.r_max;
  }
  double A, dA;
  if (bp.one_over_r) {
    A = pref * c / r;
    dA = pref * (dc / r - c / (r * r));
  } else {
    A = pref * c;
    dA = pref * dc;
  }
  const double th = w * r / delta, sn = sin(th), cs = cos(th);
  b = A * sn;
  db = dA * sn + A * (w / delta) * cs;
  b_w = A * (r / delta) * cs;
  db_w = dA * (r / delta) * cs + A * (cs / delta - (w / delta) * (r / delta) * sn);
}

// workspace of one net: per hidden layer l [a_l | a'_l] ([R, k_l] each) then [gz_l | gz'_l] ([R, H] each); behind the last layer the
// Bessel terms [R, k0]
__host__ __device__ inline int64_t ws_level_offset(int l, int64_t R, int k0, int H) {
  int64_t off = 0;
  for (int i = 0; i < l; ++i) off += 2 * R * (i == 0 ? k0 : H) + 2 * R * H;
  return off;
}
__host__ __device__ inline int64_t slope_ws_doubles(int n_hidden, int64_t R, int k0, int H) {
  return ws_level_offset(n_hidden, R, k0, H) + R * k0;
}

struct SlopeNet {
  const float* w[SL_MAXL];      // hidden weights [k_l, H] fp32
  float* hp;                    // forward: [R, H] out (H' in fp32)
  const float* g_hp;            // backward: [R, H] in
  double* ws;                   // backward: per-row operands of the weight-gradient sums (slope_ws_doubles per net)
  float* g_w[SL_MAXL];          // backward: [k_l, H] fp32, ADDED to (NULL = not wanted)
};
struct SlopeBatch {
  SlopeNet net[16];
  float alpha[SL_MAXL];
};

// value and tangent of the chain for SL_ROWS rows, level by level: a[l], da[l] = activations entering layer l (level 0 = the basis)
// one workgroup = SL_ROWS knot rows of one net; thread j = hidden unit j (H = blockDim.x = 32 or 64)
template <bool BWD>
__global__ void slope_tangent_kernel(SlopeBatch b, const float* __restrict__ knots, int R, const float* __restrict__ bessel_w, int k0,
                                     int H, int n_hidden, BasisPar bp, int act, float cst_f) {
  __shared__ double a[SL_MAXL][SL_ROWS][SL_MAXH], da[SL_MAXL][SL_ROWS][SL_MAXH];      // inputs of layer l (l = 0: the basis)
  __shared__ double ga[SL_ROWS][SL_MAXH], gda[SL_ROWS][SL_MAXH];                        // backward: cotangents of the current level
  const SlopeNet& net = b.net[blockIdx.y];
  const int j = threadIdx.x;
  const int row0 = blockIdx.x * SL_ROWS;
  const double cst = (double)cst_f;
  double bw_[SL_ROWS], dbw_[SL_ROWS];      // (thread j < k0: d b_j / d w_j and d b'_j / d w_j of its rows)
  for (int i = 0; i < SL_ROWS; ++i) {
    const int row = row0 + i < R ? row0 + i : R - 1;
    bw_[i] = dbw_[i] = 0.0;
    if (j < k0) {
      double v, dv;
      basis64((double)knots[row], (double)bessel_w[j], bp, v, dv, bw_[i], dbw_[i]);
      a[0][i][j] = v;
      da[0][i][j] = dv;
    }
  }
  __syncthreads();
  // ---- forward: value and tangent
  double out_a[SL_ROWS], out_da[SL_ROWS];
  int kin = k0;
  for (int l = 0; l < n_hidden; ++l) {
    const float* __restrict__ w = net.w[l];
    const double al = (double)b.alpha[l];
    double z[SL_ROWS], dz[SL_ROWS];
#pragma unroll
    for (int i = 0; i < SL_ROWS; ++i) z[i] = dz[i] = 0.0;
    for (int k = 0; k < kin; ++k) {
      const double wk = (double)w[(int64_t)k * H + j];
#pragma unroll
      for (int i = 0; i < SL_ROWS; ++i) {
        z[i] = fma(a[l][i][k], wk, z[i]);
        dz[i] = fma(da[l][i][k], wk, dz[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < SL_ROWS; ++i) {
      double f, d1, d2;
      act64(act, al * z[i], f, d1, d2);
      out_a[i] = cst * f;
      out_da[i] = cst * d1 * al * dz[i];
    }
    if (l + 1 < n_hidden) {
#pragma unroll
      for (int i = 0; i < SL_ROWS; ++i) {
        a[l + 1][i][j] = out_a[i];
        da[l + 1][i][j] = out_da[i];
      }
      __syncthreads();
    }
    kin = H;
  }
  if constexpr (!BWD) {
    for (int i = 0; i < SL_ROWS; ++i)
      if (row0 + i < R) net.hp[(int64_t)(row0 + i) * H + j] = (float)out_da[i];
    return;
  } else {
    // ---- backward of <g_hp, tangent of the last level>: cotangents (ga, gda) of (value, tangent) of the level below, layer by layer
    double g_v[SL_ROWS], g_t[SL_ROWS];      // cotangent of this thread's unit: of its value a_L[j] and of its tangent a'_L[j]
    for (int i = 0; i < SL_ROWS; ++i) {
      g_v[i] = 0.0;
      g_t[i] = row0 + i < R ? (double)net.g_hp[(int64_t)(row0 + i) * H + j] : 0.0;
    }
    for (int l = n_hidden - 1; l >= 0; --l) {
      const float* __restrict__ w = net.w[l];
      const double al = (double)b.alpha[l];
      const int kl = l == 0 ? k0 : H;
      // recompute this layer's pre-activation and its tangent from the saved level
      double z[SL_ROWS], dz[SL_ROWS];
#pragma unroll
      for (int i = 0; i < SL_ROWS; ++i) z[i] = dz[i] = 0.0;
      for (int k = 0; k < kl; ++k) {
        const double wk = (double)w[(int64_t)k * H + j];
#pragma unroll
        for (int i = 0; i < SL_ROWS; ++i) {
          z[i] = fma(a[l][i][k], wk, z[i]);
          dz[i] = fma(da[l][i][k], wk, dz[i]);
        }
      }
      // a_out = cst f(al z), a'_out = cst f'(al z) al z'  ->  cotangents of z and z'
      double gz[SL_ROWS], gdz[SL_ROWS];
#pragma unroll
      for (int i = 0; i < SL_ROWS; ++i) {
        double f, d1, d2;
        act64(act, al * z[i], f, d1, d2);
        gdz[i] = g_t[i] * cst * d1 * al;
        gz[i] = (g_v[i] * cst * d1 + g_t[i] * cst * d2 * al * dz[i]) * al;
      }
      // weight gradient: g_W[k, j] = sum_rows a[l][row][k] gz[row][j] + da[l][row][k] gdz[row][j] -- a sum over ALL knot rows: the
      // per-row operands go to the workspace, slope_wgrad_kernel sums them (a fixed order, no atomics)
      {
        double* wl = net.ws + ws_level_offset(l, R, k0, H);
        for (int i = 0; i < SL_ROWS; ++i) {
          const int row = row0 + i;
          if (row >= R) break;
          if (j < kl) {
            wl[(int64_t)row * kl + j] = a[l][i][j];
            wl[(int64_t)R * kl + (int64_t)row * kl + j] = da[l][i][j];
          }
          wl[(int64_t)2 * R * kl + (int64_t)row * H + j] = gz[i];
          wl[(int64_t)2 * R * kl + (int64_t)R * H + (int64_t)row * H + j] = gdz[i];
        }
      }
      // cotangents of the level below: ga[row][k] = sum_j gz[row][j] W[k, j] (a reduction over the threads: through LDS)
      __syncthreads();                   // (everybody is done reading ga / gda of the level above)
#pragma unroll
      for (int i = 0; i < SL_ROWS; ++i) {
        ga[i][j] = gz[i];
        gda[i][j] = gdz[i];
      }
      __syncthreads();
      if (j < kl) {
#pragma unroll
        for (int i = 0; i < SL_ROWS; ++i) g_v[i] = g_t[i] = 0.0;
        for (int u = 0; u < H; ++u) {
          const double wk = (double)w[(int64_t)j * H + u];      // W[k = j, unit u]
#pragma unroll
          for (int i = 0; i < SL_ROWS; ++i) {
            g_v[i] = fma(ga[i][u], wk, g_v[i]);
            g_t[i] = fma(gda[i][u], wk, g_t[i]);
          }
        }
      }
    }
    // level 0 = the basis: the Bessel frequencies' per-row terms
    if (j < k0) {
      double* wb = net.ws + ws_level_offset(n_hidden, R, k0, H);
      for (int i = 0; i < SL_ROWS; ++i)
        if (row0 + i < R) wb[(int64_t)(row0 + i) * k0 + j] = g_v[i] * bw_[i] + g_t[i] * dbw_[i];
    }
  }
}

// the sums over the knot rows: grid (tiles of [k_l, H], net, level); level n_hidden = the Bessel frequencies (summed over the nets)
__global__ __launch_bounds__(256) void slope_wgrad_kernel(SlopeBatch b, int R, int k0, int H, int n_hidden, float* __restrict__ g_bessel) {
  const SlopeNet& net = b.net[blockIdx.y];
  const int l = blockIdx.z;
  if (l == n_hidden) {      // the Bessel frequencies: one WAVE per frequency, its lanes stride the rows of all the nets (fixed order)
    // (frequencies strided over the grid: k0 may be up to H = 64 while gridDim.x * 4 = 32 -- ADVICE r4: with one frequency per wave
    //  and no stride the slope table's share of g_bessel[32:] was silently dropped)
    const int lane = threadIdx.x & 63;
    if (!g_bessel || blockIdx.y != 0) return;
    for (int f = blockIdx.x * 4 + (threadIdx.x >> 6); f < k0; f += (int)gridDim.x * 4) {
      double s = 0.0;
      for (int nn = 0; nn < (int)gridDim.y; ++nn) {
        const double* wb = b.net[nn].ws + ws_level_offset(n_hidden, R, k0, H);
        for (int row = lane; row < R; row += 64) s += wb[(int64_t)row * k0 + f];
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
      if (lane == 0) g_bessel[f] += (float)s;
    }
    return;
  }
  // hidden layer l: g_W[k, j] = sum_rows a[row][k] gz[row][j] + a'[row][k] gz'[row][j] -- a [kl x R] x [R x H] product pair in
  // float64: 4 x 4 outputs per thread, 16-row tiles through LDS; the knot rows are split over gridDim.x workgroups whose partial
  // sums meet in the fp32 gradient buffer through atomics (as the weight-gradient GEMMs of every Linear do: e3k_gemm_wgrad)
  const int kl = l == 0 ? k0 : H;
  if (!net.g_w[l]) return;
  const int per = ((R + (int)gridDim.x - 1) / (int)gridDim.x + 15) / 16 * 16;
  const int r_beg = blockIdx.x * per, r_end = r_beg + per < R ? r_beg + per : R;
  if (r_beg >= R) return;
  __shared__ double sA[16][SL_MAXH], sdA[16][SL_MAXH], sG[16][SL_MAXH], sdG[16][SL_MAXH];
  const double* wl = net.ws + ws_level_offset(l, R, k0, H);
  const double* pa = wl;
  const double* pda = wl + (int64_t)R * kl;
  const double* pg = wl + (int64_t)2 * R * kl;
  const double* pdg = pg + (int64_t)R * H;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;      // outputs k = 4 ty .. 4 ty + 3, j = 4 tx .. 4 tx + 3
  double acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[i][q] = 0.0;
  for (int row0 = r_beg; row0 < r_end; row0 += 16) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = threadIdx.x + 256 * q, r = e >> 6, c = e & 63, row = row0 + r;
      const bool in = row < r_end;
      sA[r][c] = (in && c < kl) ? pa[(int64_t)row * kl + c] : 0.0;
      sdA[r][c] = (in && c < kl) ? pda[(int64_t)row * kl + c] : 0.0;
      sG[r][c] = (in && c < H) ? pg[(int64_t)row * H + c] : 0.0;
      sdG[r][c] = (in && c < H) ? pdg[(int64_t)row * H + c] : 0.0;
    }
    __syncthreads();
#pragma unroll 4
    for (int r = 0; r < 16; ++r) {
      double a4[4], d4[4], g4[4], h4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a4[i] = sA[r][4 * ty + i];
        d4[i] = sdA[r][4 * ty + i];
        g4[i] = sG[r][4 * tx + i];
        h4[i] = sdG[r][4 * tx + i];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[i][q] = fma(a4[i], g4[q], fma(d4[i], h4[q], acc[i][q]));
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = 4 * ty + i, j = 4 * tx + q;
      if (k < kl && j < H) atomicAdd(net.g_w[l] + (int64_t)k * H + j, (float)acc[i][q]);
    }
}

}  // namespace e3k

namespace {
int check_shape(int32_t n_nets, int32_t n_hidden, int64_t R, int32_t k0, int32_t H, int32_t act, int32_t cutoff_kind, float r_max,
                float r_min) {
  if (n_nets <= 0 || n_nets > 16 || n_hidden < 1 || n_hidden > e3k::SL_MAXL || R <= 0) return E3K_ERR_INVALID;
  if ((H != 32 && H != 64) || k0 <= 0 || k0 > H || (act != 1 && act != 2 && act != 4 && act != 0)) return E3K_ERR_UNSUPPORTED;
  if (cutoff_kind < 0 || cutoff_kind > 1 || !(r_max > r_min)) return E3K_ERR_INVALID;
  return E3K_OK;
}
}  // namespace

// H'_i [R, H] (fp32) for n_nets hidden chains on the same R radii; w_hidden[i * 4 + l]: layer l of net i, [k_l, H]
extern "C" int e3k_slope_tangent_fwd(const float* const* w_hidden, int32_t n_nets, int32_t n_hidden, const float* alphas,
                                     const float* knots, int64_t R, const float* bessel_w, int32_t k0, int32_t H, float r_max,
                                     float r_min, float p, int32_t one_over_r, int32_t cutoff_kind, int32_t act, float cst,
                                     float* const* hp, void* stream) {
  const int rc = check_shape(n_nets, n_hidden, R, k0, H, act, cutoff_kind, r_max, r_min);
  if (rc != E3K_OK) return rc;
  if (!w_hidden || !alphas || !knots || !bessel_w || !hp) return E3K_ERR_INVALID;
  e3k::SlopeBatch b{};
  for (int i = 0; i < n_nets; ++i) {
    for (int l = 0; l < n_hidden; ++l) {
      if (!w_hidden[i * 4 + l]) return E3K_ERR_INVALID;
      b.net[i].w[l] = w_hidden[i * 4 + l];
    }
    if (!hp[i]) return E3K_ERR_INVALID;
    b.net[i].hp = hp[i];
  }
  for (int l = 0; l < n_hidden; ++l) b.alpha[l] = alphas[l];
  const e3k::BasisPar bp{(double)r_max, (double)r_min, (double)p, one_over_r, cutoff_kind};
  dim3 grid((unsigned)((R + e3k::SL_ROWS - 1) / e3k::SL_ROWS), (unsigned)n_nets);
  hipLaunchKernelGGL(e3k::slope_tangent_kernel<false>, grid, dim3(H), 0, (hipStream_t)stream, b, knots, (int)R, bessel_w, k0, H, n_hidden,
                     bp, act, cst);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

/* The reverse sweep: from g_hp[i] [R, H] (the gradient of H'_i) to the hidden weights' and the Bessel frequencies' gradients.
 * acc: float64 scratch of e3k_slope_tangent_bwd_scratch(n_nets, n_hidden, k0, H, R) doubles;
 * g_hidden[i * 4 + l] (fp32 [k_l, H], ADDED to; NULL = not wanted), g_bessel (fp32 [k0], ADDED to, summed over the nets; NULL).
 * Two launches: the chain per knot row (value, tangent, then their cotangents: per-row operands into the scratch), then the sums
 * over the rows (float64 partial sums of eight row ranges, combined by fp32 atomics as every weight-gradient GEMM here does). */
extern "C" int64_t e3k_slope_tangent_bwd_scratch(int32_t n_nets, int32_t n_hidden, int32_t k0, int32_t H, int64_t R) {
  return (int64_t)n_nets * e3k::slope_ws_doubles(n_hidden, R, k0, H);
}

extern "C" int e3k_slope_tangent_bwd(const float* const* w_hidden, int32_t n_nets, int32_t n_hidden, const float* alphas,
                                     const float* knots, int64_t R, const float* bessel_w, int32_t k0, int32_t H, float r_max,
                                     float r_min, float p, int32_t one_over_r, int32_t cutoff_kind, int32_t act, float cst,
                                     const float* const* g_hp, double* acc, float* const* g_hidden, float* g_bessel, void* stream) {
  const int rc = check_shape(n_nets, n_hidden, R, k0, H, act, cutoff_kind, r_max, r_min);
  if (rc != E3K_OK) return rc;
  if (!w_hidden || !alphas || !knots || !bessel_w || !g_hp || !acc || !g_hidden) return E3K_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const int64_t per_net = e3k::slope_ws_doubles(n_hidden, R, k0, H);
  e3k::SlopeBatch b{};
  for (int i = 0; i < n_nets; ++i) {
    for (int l = 0; l < n_hidden; ++l) {
      if (!w_hidden[i * 4 + l]) return E3K_ERR_INVALID;
      b.net[i].w[l] = w_hidden[i * 4 + l];
      b.net[i].g_w[l] = g_hidden[i * 4 + l];
    }
    if (!g_hp[i]) return E3K_ERR_INVALID;
    b.net[i].g_hp = g_hp[i];
    b.net[i].ws = acc + (int64_t)i * per_net;
  }
  for (int l = 0; l < n_hidden; ++l) b.alpha[l] = alphas[l];
  const e3k::BasisPar bp{(double)r_max, (double)r_min, (double)p, one_over_r, cutoff_kind};
  dim3 grid((unsigned)((R + e3k::SL_ROWS - 1) / e3k::SL_ROWS), (unsigned)n_nets);
  hipLaunchKernelGGL(e3k::slope_tangent_kernel<true>, grid, dim3(H), 0, st, b, knots, (int)R, bessel_w, k0, H, n_hidden, bp, act, cst);
  dim3 grid2(8, (unsigned)n_nets, (unsigned)(n_hidden + 1));      // (x: row splits of the hidden levels; the Bessel level uses the first two)
  hipLaunchKernelGGL(e3k::slope_wgrad_kernel, grid2, dim3(256), 0, st, b, (int)R, k0, H, n_hidden, g_bessel);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}
