// Fused hidden chain of the radial MLP for gfx950:  x[E,k0] -> (linear -> normalised activation) x L -> h[E,H]
// in ONE launch forward and ONE launch backward.
//
// Replaces (paths relative to /root/reference) the hidden layers of
//   e3nn.nn.FullyConnectedNet([n_radial, H, ..., H, weight_numel], act)   e3_layers/nn/message_passing.py:74-79,93
// (the last, linear layer H -> weight_numel stays a GEMM: e3k_gemm small-K kernel).  Unfused, every hidden layer is
// a [E,64]x[64,64] GEMM + an activation pass forward and dgrad + wgrad + activation-backward passes backward:
// ~20 launches per convolution that each stream [E,H] through HBM twice.  Here a 64-edge tile stays in LDS through
// the whole chain; only the pre-activations z_l (needed by the backward) and the final h touch HBM, once.
//
// Matrix work on v_mfma_f32_32x32x2_f32 (exact fp32): block = 4 waves = 2x2 tiles of 32x32 over a 64-row x H tile.
// Backward: per tile   gz_l = g (.) cst act'(z_l);  gW_l += alpha_l prev_l^T gz_l (accumulated in registers over all the
// tiles of a persistent block, one atomic add per element at the end);  g <- alpha_l gz_l W_l^T;  prev_l = cst act(z_{l-1})
// is formed on the fly from the saved pre-activation when the MFMA operand is read.
#include <cstdlib>

#include "e3k_act.h"
#include "e3k_common.h"

namespace e3k {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MLP_MAXL = 4;
constexpr int MLP_BM = 64;     // rows (edges) per tile
constexpr int MLP_LD = 68;     // LDS row stride of the [64][<=64] tiles (16-byte aligned rows)

struct MlpArgs {
  const float* x;
  int64_t E;
  int k0, h, n_layers, act;
  float cst;
  const float* w[MLP_MAXL];
  float alpha[MLP_MAXL];
  float* z[MLP_MAXL];   // forward: pre-activations out (NULL: not kept); backward: in
  float* out;           // forward: h_{L-1}
  const float* g;       // backward: gradient wrt out
  float* gw[MLP_MAXL];  // backward: accumulated with atomics (NULL: not needed)
  float* gx;            // backward: [E,k0] or NULL
};

// several MLPs over the SAME input rows in one launch (the radial MLPs of all the layers of a network read one edge
// embedding): blockIdx.y picks the net
constexpr int MLP_MAXNETS = 8;
struct MlpBatch {
  int n;
  MlpArgs a[MLP_MAXNETS];
};

__device__ __forceinline__ int acc_row(int i, int lane) { return (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5); }

// rows x cols tile, global row-major (ld = cols) -> LDS (row stride MLP_LD); rows beyond E and columns beyond cols
// up to cols_pad are zero-filled
__device__ __forceinline__ void load_tile(const float* __restrict__ src, int64_t row0, int64_t E, int cols, int cols_pad,
                                          float* dst) {
  const int t = threadIdx.x;
  if ((cols & 3) == 0 && cols_pad == cols) {
    const int c4n = cols >> 2;
    for (int idx = t; idx < MLP_BM * c4n; idx += 256) {
      const int r = idx / c4n, c = (idx - r * c4n) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row0 + r < E) v = *reinterpret_cast<const float4*>(src + (row0 + r) * cols + c);
      *reinterpret_cast<float4*>(dst + r * MLP_LD + c) = v;
    }
  } else {
    for (int idx = t; idx < MLP_BM * cols_pad; idx += 256) {
      const int r = idx / cols_pad, c = idx - r * cols_pad;
      dst[r * MLP_LD + c] = (row0 + r < E && c < cols) ? src[(row0 + r) * cols + c] : 0.f;
    }
  }
}

__global__ __launch_bounds__(256) void mlp_hidden_fwd_kernel(const MlpBatch mb) {
  const MlpArgs& a = mb.a[blockIdx.y];
  __shared__ __attribute__((aligned(16))) float As[MLP_BM * MLP_LD];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wm = w & 1, wn = w >> 1;
  const int h = a.h, kp0 = (a.k0 + 1) & ~1;
  const bool col_ok = wn * 32 < h;
  const int64_t row0 = (int64_t)blockIdx.x * MLP_BM;
  load_tile(a.x, row0, a.E, a.k0, kp0, As);
  for (int l = 0; l < a.n_layers; ++l) {
    const int k_real = l == 0 ? a.k0 : h, K = l == 0 ? kp0 : h;
    const int KH = K >> 1;
    // B operand straight from global memory (the weights are L2 resident) in MFMA operand order, with the
    // contraction index permuted so that lane half hh owns k = hh*K/2 + s: no weight tile in LDS, no staging pass
    float bw[32];
    {
      const float* wl = a.w[l] + wn * 32 + (lane & 31);
      const int kb = (lane >> 5) * KH;
#pragma unroll
      for (int sI = 0; sI < 32; ++sI) bw[sI] = (col_ok && sI < KH && kb + sI < k_real) ? wl[(kb + sI) * h] : 0.f;
    }
    __syncthreads();   // the A tile (input or previous activations) is complete
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    if (col_ok) {
      const float* ap = As + (wm * 32 + (lane & 31)) * MLP_LD + (lane >> 5) * KH;
#pragma unroll
      for (int sI = 0; sI < 32; ++sI)
        if (sI < KH) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[sI], bw[sI], acc, 0, 0, 0);
    }
    __syncthreads();   // every wave is done with this layer's operands
    if (col_ok) {
      const float al = a.alpha[l];
#pragma unroll
      for (int i = 0; i < 16; ++i) As[(wm * 32 + acc_row(i, lane)) * MLP_LD + wn * 32 + (lane & 31)] = al * acc[i];
    }
    __syncthreads();
    // row-major pass over the tile: keep z_l, activate in place, emit the last layer's activations
    const int c4n = h >> 2, hs = h == 64 ? 4 : 3;
    const bool last = l == a.n_layers - 1;
    for (int idx = t; idx < MLP_BM * c4n; idx += 256) {
      const int r = idx >> hs, c = (idx & (c4n - 1)) * 4;
      float4 v = *reinterpret_cast<float4*>(As + r * MLP_LD + c);
      const bool ok = row0 + r < a.E;
      if (ok && a.z[l]) *reinterpret_cast<float4*>(a.z[l] + (row0 + r) * h + c) = v;
      v.x = a.cst * act_f(a.act, v.x);
      v.y = a.cst * act_f(a.act, v.y);
      v.z = a.cst * act_f(a.act, v.z);
      v.w = a.cst * act_f(a.act, v.w);
      if (last) {
        if (ok) *reinterpret_cast<float4*>(a.out + (row0 + r) * h + c) = v;
      } else {
        *reinterpret_cast<float4*>(As + r * MLP_LD + c) = v;
      }
    }
    __syncthreads();
  }
}

template <int NL>   // number of hidden layers: one register-resident weight-gradient tile each
__global__ __launch_bounds__(256, 3) void mlp_hidden_bwd_kernel(const MlpBatch mb, int n_tiles) {
  const MlpArgs& a = mb.a[blockIdx.y];
  __shared__ __attribute__((aligned(16))) float Gs[MLP_BM * MLP_LD];   // gradient wrt h_l, then gz_l
  __shared__ __attribute__((aligned(16))) float Ds[MLP_BM * MLP_LD];   // cst act'(z_l)
  __shared__ __attribute__((aligned(16))) float Hs[MLP_BM * MLP_LD];   // layer input: h_{l-1} = cst act(z_{l-1}) or x
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wm = w & 1, wn = w >> 1;   // dgrad / wgrad tile coordinates (rows or k block, column block)
  const int h = a.h, kp0 = (a.k0 + 1) & ~1;
  const int hs = h == 64 ? 4 : 3;      // log2(h / 4): float4 columns per row
  const int c4n = h >> 2;
  constexpr int L = NL;

  f32x16 accw[NL];
#pragma unroll
  for (int l = 0; l < NL; ++l)
#pragma unroll
    for (int i = 0; i < 16; ++i) accw[l][i] = 0.f;

  // every elementwise pass maps thread t to the same tile elements, so a thread only ever re-reads what it wrote
  // itself: the passes need no barrier between them
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t row0 = (int64_t)tile * MLP_BM;
    for (int idx = t; idx < MLP_BM * c4n; idx += 256) {
      const int r = idx >> hs, c = (idx & (c4n - 1)) * 4;
      float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f), z4 = g4;
      if (row0 + r < a.E) {
        g4 = *reinterpret_cast<const float4*>(a.g + (row0 + r) * h + c);
        z4 = *reinterpret_cast<const float4*>(a.z[L - 1] + (row0 + r) * h + c);
      }
      *reinterpret_cast<float4*>(Gs + r * MLP_LD + c) = g4;
      *reinterpret_cast<float4*>(Ds + r * MLP_LD + c) =
          make_float4(a.cst * act_df(a.act, z4.x), a.cst * act_df(a.act, z4.y), a.cst * act_df(a.act, z4.z), a.cst * act_df(a.act, z4.w));
    }
#pragma unroll
    for (int li = 0; li < NL; ++li) {
      const int l = L - 1 - li;
      const int k_real = l == 0 ? a.k0 : h, K = l == 0 ? kp0 : h;   // input width of layer l
      // gz = g (.) cst act'(z_l); then this layer's input and the next iteration's derivative factors
      const float* zprev = l > 0 ? a.z[l - 1] : nullptr;
      for (int idx = t; idx < MLP_BM * c4n; idx += 256) {
        const int r = idx >> hs, c = (idx & (c4n - 1)) * 4;
        float4 g4 = *reinterpret_cast<float4*>(Gs + r * MLP_LD + c);
        const float4 d4 = *reinterpret_cast<const float4*>(Ds + r * MLP_LD + c);
        g4.x *= d4.x; g4.y *= d4.y; g4.z *= d4.z; g4.w *= d4.w;
        *reinterpret_cast<float4*>(Gs + r * MLP_LD + c) = g4;
        if (zprev) {
          float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
          if (row0 + r < a.E) z4 = *reinterpret_cast<const float4*>(zprev + (row0 + r) * h + c);
          *reinterpret_cast<float4*>(Hs + r * MLP_LD + c) =
              make_float4(a.cst * act_f(a.act, z4.x), a.cst * act_f(a.act, z4.y), a.cst * act_f(a.act, z4.z), a.cst * act_f(a.act, z4.w));
          *reinterpret_cast<float4*>(Ds + r * MLP_LD + c) =
              make_float4(a.cst * act_df(a.act, z4.x), a.cst * act_df(a.act, z4.y), a.cst * act_df(a.act, z4.z), a.cst * act_df(a.act, z4.w));
        }
      }
      if (l == 0) load_tile(a.x, row0, a.E, a.k0, kp0, Hs);
      // dgrad B operand W_l[k][n] straight from global memory: lane (k, hh) owns n = hh*h/2 + s, a contiguous run
      const bool need_d = l > 0 || a.gx;
      const bool d_ok = need_d && wn * 32 < K;
      const int NH = h >> 1;
      float bd[32];
      {
        const int krow = wn * 32 + (lane & 31);
        const bool kin = d_ok && krow < k_real;
        const float* wl = a.w[l] + krow * h + (lane >> 5) * NH;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (kin && 4 * q < NH) v = *reinterpret_cast<const float4*>(wl + 4 * q);
          bd[4 * q] = v.x; bd[4 * q + 1] = v.y; bd[4 * q + 2] = v.z; bd[4 * q + 3] = v.w;
        }
      }
      __syncthreads();
      // wgrad: gW_l[k, n] += sum_rows input[row, k] gz[row, n]; wave tile (k block wm, n block wn)
      if (a.gw[l] && wm * 32 < K && wn * 32 < h) {
        const float* pp = Hs + (lane >> 5) * MLP_LD + wm * 32 + (lane & 31);
        const float* gp = Gs + (lane >> 5) * MLP_LD + wn * 32 + (lane & 31);
        const bool kin = wm * 32 + (lane & 31) < K;
#pragma unroll 8
        for (int rr = 0; rr < MLP_BM; rr += 2)
          accw[li] = __builtin_amdgcn_mfma_f32_32x32x2f32(kin ? pp[rr * MLP_LD] : 0.f, gp[rr * MLP_LD], accw[li], 0, 0, 0);
      }
      // dgrad: g_prev[row, k] = alpha_l sum_n gz[row, n] W_l[k, n]; wave tile (row block wm, k block wn)
      f32x16 accd;
#pragma unroll
      for (int i = 0; i < 16; ++i) accd[i] = 0.f;
      if (d_ok) {
        const float* gp = Gs + (wm * 32 + (lane & 31)) * MLP_LD + (lane >> 5) * NH;
#pragma unroll
        for (int sI = 0; sI < 32; ++sI)
          if (sI < NH) accd = __builtin_amdgcn_mfma_f32_32x32x2f32(gp[sI], bd[sI], accd, 0, 0, 0);
      }
      __syncthreads();   // everyone is done reading Gs / Hs
      if (d_ok) {
        const float al = a.alpha[l];
#pragma unroll
        for (int i = 0; i < 16; ++i) Gs[(wm * 32 + acc_row(i, lane)) * MLP_LD + wn * 32 + (lane & 31)] = al * accd[i];
      }
      __syncthreads();   // Gs now holds the gradient wrt this layer's input
    }
    if (a.gx) {
      for (int idx = t; idx < MLP_BM * a.k0; idx += 256) {
        const int r = idx / a.k0, c = idx - r * a.k0;
        if (row0 + r < a.E) a.gx[(row0 + r) * a.k0 + c] = Gs[r * MLP_LD + c];
      }
      __syncthreads();   // the next tile overwrites Gs
    }
  }
  // one atomic add per weight element and block
#pragma unroll
  for (int li = 0; li < NL; ++li) {
    const int l = L - 1 - li;
    const int k_real = l == 0 ? a.k0 : h;
    if (!a.gw[l] || wn * 32 >= h) continue;
    const int n = wn * 32 + (lane & 31);
    const float al = a.alpha[l];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int k = wm * 32 + acc_row(i, lane);
      if (k < k_real) atomicAdd(a.gw[l] + k * h + n, al * accw[li][i]);
    }
  }
}

}  // namespace e3k

namespace {
int fill_args(e3k::MlpArgs& a, const float* x, int64_t E, int32_t k0, int32_t h, int32_t n_layers,
              const float* const* weights, const float* alphas, int32_t act, float cst) {
  if (E < 0 || k0 <= 0 || k0 > 64 || (h != 32 && h != 64) || n_layers < 1 || n_layers > e3k::MLP_MAXL) return E3K_ERR_UNSUPPORTED;
  if (act < 0 || act > 5 || !weights || !alphas || (E > 0 && !x)) return E3K_ERR_INVALID;
  a.x = x;
  a.E = E;
  a.k0 = k0;
  a.h = h;
  a.n_layers = n_layers;
  a.act = act;
  a.cst = cst;
  for (int l = 0; l < n_layers; ++l) {
    if (!weights[l]) return E3K_ERR_INVALID;
    a.w[l] = weights[l];
    a.alpha[l] = alphas[l];
  }
  return E3K_OK;
}
}  // namespace

namespace {
int launch_fwd(const e3k::MlpBatch& mb, int64_t E, hipStream_t st) {
  const int64_t tiles = (E + e3k::MLP_BM - 1) / e3k::MLP_BM;
  if (tiles > 0x7fffffffLL) return E3K_ERR_INVALID;
  hipLaunchKernelGGL(e3k::mlp_hidden_fwd_kernel, dim3((unsigned)tiles, (unsigned)mb.n), dim3(256), 0, st, mb);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

int launch_bwd(const e3k::MlpBatch& mb, int64_t E, int n_layers, hipStream_t st) {
  const int64_t tiles = (E + e3k::MLP_BM - 1) / e3k::MLP_BM;
  if (tiles > 0x7fffffffLL) return E3K_ERR_INVALID;
  E3K_KNOB_INT(max_blocks, "E3K_MLP_BLOCKS", 768);
  int64_t blocks = max_blocks / mb.n;      // persistent: two to three workgroups per CU share the weight-gradient atomics
  if (blocks < 1) blocks = 1;
  if (tiles < blocks) blocks = tiles;
  const dim3 grid((unsigned)blocks, (unsigned)mb.n);
  switch (n_layers) {
    case 1: hipLaunchKernelGGL(e3k::mlp_hidden_bwd_kernel<1>, grid, dim3(256), 0, st, mb, (int)tiles); break;
    case 2: hipLaunchKernelGGL(e3k::mlp_hidden_bwd_kernel<2>, grid, dim3(256), 0, st, mb, (int)tiles); break;
    case 3: hipLaunchKernelGGL(e3k::mlp_hidden_bwd_kernel<3>, grid, dim3(256), 0, st, mb, (int)tiles); break;
    default: hipLaunchKernelGGL(e3k::mlp_hidden_bwd_kernel<4>, grid, dim3(256), 0, st, mb, (int)tiles); break;
  }
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}
}  // namespace

extern "C" int e3k_mlp_hidden_fwd(const float* x, int64_t E, int32_t k0, int32_t h, int32_t n_layers,
                                  const float* const* weights, const float* alphas, int32_t act, float cst,
                                  float* const* z, float* out, void* stream) {
  e3k::MlpBatch mb{};
  mb.n = 1;
  const int rc = fill_args(mb.a[0], x, E, k0, h, n_layers, weights, alphas, act, cst);
  if (rc != E3K_OK) return rc;
  if (E == 0) return E3K_OK;
  if (!out) return E3K_ERR_INVALID;
  for (int l = 0; l < n_layers; ++l) mb.a[0].z[l] = z ? z[l] : nullptr;
  mb.a[0].out = out;
  return launch_fwd(mb, E, (hipStream_t)stream);
}

extern "C" int e3k_mlp_hidden_bwd(const float* x, int64_t E, int32_t k0, int32_t h, int32_t n_layers,
                                  const float* const* weights, const float* alphas, int32_t act, float cst,
                                  const float* const* z, const float* g_out, float* const* g_weights, float* g_x,
                                  void* stream) {
  e3k::MlpBatch mb{};
  mb.n = 1;
  e3k::MlpArgs& a = mb.a[0];
  const int rc = fill_args(a, x, E, k0, h, n_layers, weights, alphas, act, cst);
  if (rc != E3K_OK) return rc;
  if (E == 0) return E3K_OK;
  if (!z || !g_out) return E3K_ERR_INVALID;
  for (int l = 0; l < n_layers; ++l) {
    if (!z[l]) return E3K_ERR_INVALID;
    a.z[l] = const_cast<float*>(z[l]);
    a.gw[l] = g_weights ? g_weights[l] : nullptr;
  }
  a.g = g_out;
  a.gx = g_x;
  return launch_bwd(mb, E, n_layers, (hipStream_t)stream);
}

// n_nets MLPs of one shape over the same input rows (e3k_mlp_net: the per-net pointers), one launch
extern "C" int e3k_mlp_hidden_fwd_multi(const e3k_mlp_net* nets, int32_t n_nets, const float* x, int64_t E, int32_t k0, int32_t h,
                                        int32_t n_layers, const float* alphas, int32_t act, float cst, void* stream) {
  if (!nets || n_nets <= 0) return E3K_ERR_INVALID;
  for (int base = 0; base < n_nets; base += e3k::MLP_MAXNETS) {
    e3k::MlpBatch mb{};
    mb.n = n_nets - base < e3k::MLP_MAXNETS ? n_nets - base : e3k::MLP_MAXNETS;
    for (int i = 0; i < mb.n; ++i) {
      const e3k_mlp_net& nt = nets[base + i];
      const int rc = fill_args(mb.a[i], x, E, k0, h, n_layers, nt.weights, alphas, act, cst);
      if (rc != E3K_OK) return rc;
      if (E > 0 && !nt.out) return E3K_ERR_INVALID;
      for (int l = 0; l < n_layers; ++l) mb.a[i].z[l] = nt.z[l];
      mb.a[i].out = nt.out;
    }
    if (E == 0) return E3K_OK;
    const int rc = launch_fwd(mb, E, (hipStream_t)stream);
    if (rc != E3K_OK) return rc;
  }
  return E3K_OK;
}

extern "C" int e3k_mlp_hidden_bwd_multi(const e3k_mlp_net* nets, int32_t n_nets, const float* x, int64_t E, int32_t k0, int32_t h,
                                        int32_t n_layers, const float* alphas, int32_t act, float cst, void* stream) {
  if (!nets || n_nets <= 0) return E3K_ERR_INVALID;
  for (int base = 0; base < n_nets; base += e3k::MLP_MAXNETS) {
    e3k::MlpBatch mb{};
    mb.n = n_nets - base < e3k::MLP_MAXNETS ? n_nets - base : e3k::MLP_MAXNETS;
    for (int i = 0; i < mb.n; ++i) {
      const e3k_mlp_net& nt = nets[base + i];
      e3k::MlpArgs& a = mb.a[i];
      const int rc = fill_args(a, x, E, k0, h, n_layers, nt.weights, alphas, act, cst);
      if (rc != E3K_OK) return rc;
      if (E > 0 && !nt.g_out) return E3K_ERR_INVALID;
      for (int l = 0; l < n_layers; ++l) {
        if (E > 0 && !nt.z[l]) return E3K_ERR_INVALID;
        a.z[l] = nt.z[l];
        a.gw[l] = nt.g_weights[l];
      }
      a.g = nt.g_out;
      a.gx = nt.g_x;
    }
    if (E == 0) return E3K_OK;
    const int rc = launch_bwd(mb, E, n_layers, (hipStream_t)stream);
    if (rc != E3K_OK) return rc;
  }
  return E3K_OK;
}
