// Un-keyed self-connection through per-node contracted weights (gfx950).
//
// Replaces o3.FullyConnectedTensorProduct(in, node_attrs, out) with GENERAL scalar node attributes
// (e3_layers/nn/message_passing.py:81-87,100: the diffusion configs, whose attrs carry the time embedding):
//     out[n, k, w] = alpha * sum_{u,v} W[u,v,w] x[n,k,u] a[n,v]
// The direct form is a GEMM over K = U*V per row (n, k).  Contracting the attributes first,
//     M[n, u, w] = sum_v a[n,v] W[u,v,w]          (e3k_keyed_weights_fwd with one key per node)
//     out[n, k, w] = alpha * sum_u x[n,k,u] M[n,u,w]
// costs N*V*U*W + N*(2l+1)*U*W multiply-adds instead of N*(2l+1)*U*V*W: (2l+1)-fold fewer for the l > 0 blocks, and
// every derivative has the same two-stage shape.  The second stage is a batch of (2l+1) x U x W products, one per
// node — far below an MFMA tile, and bound by streaming M once: a wave owns (node, instruction), lanes run over w.
//
//   rowmat_fwd : out = x . M[n]
//   rowmat_bwd : dM[n] = x^T . dOut   (lanes over w, stored as it is produced)
//                dX    = dOut . M[n]^T (M tile transposed through LDS, lanes over u)
#include "e3k_common.h"

namespace e3k {

// value of lane `src` (wave-uniform index) in every lane: v_readlane_b32 instead of the LDS-routed ds_bpermute of __shfl
__device__ __forceinline__ float lane_bcast(float v, int src) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

constexpr int RM_MAXI = 16, RM_MAXD = 7;
struct RowmatArgs {
  int n_instr;
  int items_per_node;            // fwd: sum_j ceil(w_out_j / 64); bwd: n_instr
  int item_start[RM_MAXI + 1];   // fwd: prefix of the w-chunk counts
  e3k_rowmat_instr ins[RM_MAXI];
  int64_t d_in, d_out, ld_m;
};

__global__ __launch_bounds__(256) void rowmat_fwd_kernel(const float* __restrict__ x, const float* __restrict__ M,
                                                          RowmatArgs ra, int64_t rows, float* __restrict__ y) {
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= rows * ra.items_per_node) return;
  const int lane = threadIdx.x & 63;
  const int64_t n = item / ra.items_per_node;
  const int q = uniform((int)(item - n * ra.items_per_node));
  int j = 0;
  for (int i = 1; i < ra.n_instr; ++i)
    if (q >= ra.item_start[i]) j = i;
  j = uniform(j);
  const e3k_rowmat_instr in = ra.ins[j];
  const int w = (q - ra.item_start[j]) * 64 + lane;
  const bool active = w < in.w_out;
  // the node's input block, lane = u (cf layout: [dim][U]); broadcast per u with readlane below
  float xk[RM_MAXD];
  const float* __restrict__ xr = x + n * ra.d_in + in.in_off;
#pragma unroll
  for (int k = 0; k < RM_MAXD; ++k) xk[k] = (k < in.dim && lane < in.u) ? xr[k * in.u + lane] : 0.f;
  float acc[RM_MAXD];
#pragma unroll
  for (int k = 0; k < RM_MAXD; ++k) acc[k] = 0.f;
  const float* __restrict__ mr = M + n * ra.ld_m + in.m_off + w;
  for (int u0 = 0; u0 < in.u; u0 += 64) {   // U > 64: the block is re-read per 64-channel slab
    if (u0 > 0) {
#pragma unroll
      for (int k = 0; k < RM_MAXD; ++k) xk[k] = (k < in.dim && u0 + lane < in.u) ? xr[k * in.u + u0 + lane] : 0.f;
    }
    const int ucount = in.u - u0 < 64 ? in.u - u0 : 64;
#pragma unroll 4
    for (int uu = 0; uu < ucount; ++uu) {
      const float m = active ? mr[(int64_t)(u0 + uu) * in.w_out] : 0.f;
#pragma unroll
      for (int k = 0; k < RM_MAXD; ++k)
        if (k < in.dim) acc[k] = fmaf(lane_bcast(xk[k], uu), m, acc[k]);
    }
  }
  if (active) {
    float* __restrict__ yr = y + n * ra.d_out + in.out_off + w;
#pragma unroll
    for (int k = 0; k < RM_MAXD; ++k)
      if (k < in.dim) yr[k * in.w_out] = in.alpha * acc[k];
  }
}

// a wave only exchanges data with itself through its own LDS slice: LDS requests of one wave are served in issue order, so
// a compiler-level fence is all the "barrier" it needs (waves of a block run different instructions: no __syncthreads)
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// one wave per (node, instruction); U <= 64
constexpr int RM_LDT = 65;
__global__ __launch_bounds__(256) void rowmat_bwd_kernel(const float* __restrict__ x, const float* __restrict__ M,
                                                          const float* __restrict__ gy, RowmatArgs ra, int64_t rows,
                                                          float* __restrict__ gx, float* __restrict__ gM) {
  __shared__ float tile[4][64 * RM_LDT];       // M[u][w-chunk], row stride 65: transposed reads are conflict-free
  __shared__ float gs[4][RM_MAXD * 64];        // dOut[k][w-chunk]
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int64_t item = (int64_t)blockIdx.x * 4 + wid;
  const bool valid = item < rows * ra.n_instr;
  if (!valid) return;
  const int64_t n = item / ra.n_instr;
  const int j = uniform((int)(item - n * ra.n_instr));
  const e3k_rowmat_instr in = ra.ins[j];
  float xk[RM_MAXD], ax[RM_MAXD];
  const float* __restrict__ xr = x + n * ra.d_in + in.in_off;
#pragma unroll
  for (int k = 0; k < RM_MAXD; ++k) {
    xk[k] = (k < in.dim && lane < in.u) ? xr[k * in.u + lane] : 0.f;
    ax[k] = 0.f;
  }
  const float* __restrict__ gr = gy + n * ra.d_out + in.out_off;
  const float* __restrict__ mr = M + n * ra.ld_m + in.m_off;
  float* __restrict__ gmr = gM ? gM + n * ra.ld_m + in.m_off : nullptr;
  for (int w0 = 0; w0 < in.w_out; w0 += 64) {
    const int w = w0 + lane;
    const bool active = w < in.w_out;
    float gk[RM_MAXD];
#pragma unroll
    for (int k = 0; k < RM_MAXD; ++k) {
      gk[k] = (k < in.dim && active) ? gr[k * in.w_out + w] : 0.f;
      if (k < in.dim) gs[wid][k * 64 + lane] = gk[k];
    }
    for (int u = 0; u < in.u; ++u) {
      const float m = active ? mr[(int64_t)u * in.w_out + w] : 0.f;
      tile[wid][u * RM_LDT + lane] = m;
      if (gmr) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < RM_MAXD; ++k)
          if (k < in.dim) v = fmaf(lane_bcast(xk[k], u), gk[k], v);
        if (active && valid) gmr[(int64_t)u * in.w_out + w] = in.alpha * v;
      }
    }
    wave_lds_sync();
    if (gx && lane < in.u) {
      const float* __restrict__ trow = &tile[wid][lane * RM_LDT];
      const int wcount = in.w_out - w0 < 64 ? in.w_out - w0 : 64;
      for (int ww = 0; ww < wcount; ++ww) {
        const float m = trow[ww];
#pragma unroll
        for (int k = 0; k < RM_MAXD; ++k)
          if (k < in.dim) ax[k] = fmaf(gs[wid][k * 64 + ww], m, ax[k]);
      }
    }
    wave_lds_sync();
  }
  if (gx && valid && lane < in.u) {
    float* __restrict__ gxr = gx + n * ra.d_in + in.in_off + lane;
#pragma unroll
    for (int k = 0; k < RM_MAXD; ++k)
      if (k < in.dim) gxr[k * in.u] = in.x_accumulate ? gxr[k * in.u] + in.alpha * ax[k] : in.alpha * ax[k];
  }
}

}  // namespace e3k

namespace {
int make_rowmat(const e3k_rowmat_instr* instr, int32_t n_instr, int64_t d_in, int64_t d_out, int64_t ld_m, bool bwd,
                e3k::RowmatArgs& ra) {
  if (!instr || n_instr <= 0 || n_instr > e3k::RM_MAXI || d_in <= 0 || d_out <= 0 || ld_m <= 0) return E3K_ERR_INVALID;
  ra.n_instr = n_instr;
  ra.d_in = d_in;
  ra.d_out = d_out;
  ra.ld_m = ld_m;
  int pos = 0;
  for (int i = 0; i < n_instr; ++i) {
    const e3k_rowmat_instr& in = instr[i];
    if (in.u <= 0 || in.w_out <= 0 || in.dim <= 0 || in.in_off < 0 || in.out_off < 0 || in.m_off < 0) return E3K_ERR_INVALID;
    if (in.dim > e3k::RM_MAXD) return E3K_ERR_UNSUPPORTED;
    if (bwd && in.u > 64) return E3K_ERR_UNSUPPORTED;
    if (in.in_off + (int64_t)in.dim * in.u > d_in || in.out_off + (int64_t)in.dim * in.w_out > d_out ||
        in.m_off + (int64_t)in.u * in.w_out > ld_m)
      return E3K_ERR_INVALID;
    ra.ins[i] = in;
    ra.item_start[i] = pos;
    pos += bwd ? 1 : (in.w_out + 63) / 64;
  }
  ra.item_start[n_instr] = pos;
  ra.items_per_node = pos;
  return E3K_OK;
}
}  // namespace

extern "C" int e3k_rowmat_fwd(const float* x, const float* M, const e3k_rowmat_instr* instr, int32_t n_instr, int64_t rows,
                              int64_t d_in, int64_t d_out, int64_t ld_m, float* y, void* stream) {
  e3k::RowmatArgs ra{};
  const int rc = make_rowmat(instr, n_instr, d_in, d_out, ld_m, false, ra);
  if (rc != E3K_OK) return rc;
  if (rows < 0) return E3K_ERR_INVALID;
  if (rows == 0) return E3K_OK;
  if (!x || !M || !y) return E3K_ERR_INVALID;
  const int64_t items = rows * ra.items_per_node;
  hipLaunchKernelGGL(e3k::rowmat_fwd_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, M, ra,
                     rows, y);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int e3k_rowmat_bwd(const float* x, const float* M, const float* g_y, const e3k_rowmat_instr* instr,
                              int32_t n_instr, int64_t rows, int64_t d_in, int64_t d_out, int64_t ld_m, float* g_x,
                              float* g_M, void* stream) {
  e3k::RowmatArgs ra{};
  const int rc = make_rowmat(instr, n_instr, d_in, d_out, ld_m, true, ra);
  if (rc != E3K_OK) return rc;
  if (rows < 0) return E3K_ERR_INVALID;
  if (rows == 0) return E3K_OK;
  if (!x || !M || !g_y || (!g_x && !g_M)) return E3K_ERR_INVALID;
  const int64_t items = rows * n_instr;
  hipLaunchKernelGGL(e3k::rowmat_bwd_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, M, g_y,
                     ra, rows, g_x, g_M);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}
