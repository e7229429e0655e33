// Training-step plumbing on the flat parameter vector for gfx950: gradient-norm clipping, Adam and the
// exponential moving average of the weights in ONE pass over HBM.
//
// Replaces (paths relative to /root/reference):
//   torch.nn.utils.clip_grad_norm_ + optim.step() + ema.update()     e3_layers/run/trainer.py:374-386
//   the same sequence with the non-finite-gradient skip              e3_layers/run/sde_utils.py:233-248
//   (torch.optim.Adam defaults, amsgrad off; torch_ema.ExponentialMovingAverage with use_num_updates)
//
// All step-dependent scalars (bias corrections, effective EMA decay, clip coefficient, skip flag) live in a
// small DEVICE state block updated by a one-thread "tick" kernel, so a captured HIP graph of the whole
// training step replays correctly (no host-side step counter baked into kernel arguments).
// HBM-bound: per parameter 4 streams read (p, g, m, v [+ ema]) and 3 written (p, m, v [+ ema]) = 28-36 B.
#include "e3k_common.h"

namespace e3k {

// state (16 floats, zero-initialised by the caller): [0] optimizer steps taken, [1] 1-beta1^t, [2] 1-beta2^t,
// [4] gradient scale (clip coefficient), [5] sum g^2 accumulator (reset by the tick), [6] skip flag (non-finite
// gradient), [7] last total gradient norm, [8] number of EMA updates (the reference updates the EMA on skipped
// steps too), [9] effective EMA decay of this update
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ state) {
  float acc = 0.f;
  const int64_t n4 = n >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 v = g4[i];
    acc = fmaf(v.x, v.x, acc);
    acc = fmaf(v.y, v.y, acc);
    acc = fmaf(v.z, v.z, acc);
    acc = fmaf(v.w, v.w, acc);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float v = g[(n4 << 2) + threadIdx.x];
    acc = fmaf(v, v, acc);
  }
  // one atomic per workgroup (same-address atomics serialise: 8 k of them cost 100 us)
  __shared__ float part[4];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(state + 5, (part[0] + part[1]) + (part[2] + part[3]));
}

__global__ void tick_kernel(float* __restrict__ state, float beta1, float beta2, float max_norm, int have_norm,
                            int skip_nonfinite) {
  const float sumsq = state[5];
  const bool bad = have_norm && !isfinite(sumsq);
  const bool skip = skip_nonfinite && bad;
  state[6] = skip ? 1.f : 0.f;
  state[5] = 0.f;
  float t = state[0];
  if (!skip) t += 1.f;
  state[0] = t;
  state[1] = (float)(1.0 - pow((double)beta1, (double)fmaxf(t, 1.f)));
  state[2] = (float)(1.0 - pow((double)beta2, (double)fmaxf(t, 1.f)));
  const float norm = have_norm ? sqrtf(sumsq) : 0.f;
  state[7] = norm;
  float scale = 1.f;
  if (max_norm > 0.f && have_norm && !bad) {
    const float c = max_norm / (norm + 1e-6f);
    scale = c < 1.f ? c : 1.f;
  }
  state[4] = scale;
}

// both ticks in ONE launch (round 6: a step without a gradient norm).  Folding them into the update kernel itself was built and
// measured: every workgroup has to have read the old state before anybody writes the new one, i.e. a ticket -- 2 048 atomics on
// one address took the update from 24 to 43-46 us (profiles/r06_trace_graph_energy.txt of that build); one tiny launch it is.
__global__ void tick_both_kernel(float* __restrict__ state, float beta1, float beta2, float ema_decay, int use_num_updates) {
  const float t = state[0] + 1.f;
  state[6] = 0.f;
  state[5] = 0.f;
  state[0] = t;
  state[1] = (float)(1.0 - pow((double)beta1, (double)fmaxf(t, 1.f)));
  state[2] = (float)(1.0 - pow((double)beta2, (double)fmaxf(t, 1.f)));
  state[7] = 0.f;
  state[4] = 1.f;
  float* ema_state = state + 8;
  const float k = ema_state[0] + 1.f;
  ema_state[0] = k;
  float d = ema_decay;
  if (use_num_updates) {
    const float alt = (1.f + k) / (10.f + k);
    d = alt < d ? alt : d;
  }
  ema_state[1] = d;
}

__global__ void ema_tick_kernel(float* __restrict__ ema_state, float ema_decay, int use_num_updates) {
  // ema_state[0] number of updates so far, [1] effective decay of THIS update
  const float k = ema_state[0] + 1.f;
  ema_state[0] = k;
  float d = ema_decay;
  if (use_num_updates) {
    const float alt = (1.f + k) / (10.f + k);
    d = alt < d ? alt : d;
  }
  ema_state[1] = d;
}

template <bool EMA>
__global__ __launch_bounds__(256) void adam_ema_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v,
                                                        float* __restrict__ ema, int64_t n, float lr, float beta1,
                                                        float beta2, float eps, float wd,
                                                        const float* __restrict__ state,
                                                        const float* __restrict__ ema_state) {
  const float bc1 = state[1], bc2 = state[2], gscale = state[4];
  const bool skip = state[6] != 0.f;
  const float step_size = lr / bc1, inv_sqrt_bc2 = rsqrtf(bc2);
  const float one_minus_decay = EMA ? 1.0f - ema_state[1] : 0.f;
  auto upd = [&](float& pv, float gv, float& mv, float& vv, float& ev) {
    if (!skip) {
      gv *= gscale;
      if (wd != 0.f) gv = fmaf(wd, pv, gv);
      mv = fmaf(beta1, mv, (1.0f - beta1) * gv);
      vv = fmaf(beta2, vv, (1.0f - beta2) * gv * gv);
      const float denom = sqrtf(vv) * inv_sqrt_bc2 + eps;
      pv -= step_size * (mv / denom);
    }
    if constexpr (EMA) ev -= one_minus_decay * (ev - pv);
  };
  const int64_t n4 = n >> 2;
  float4* p4 = reinterpret_cast<float4*>(p);
  const float4* g4 = reinterpret_cast<const float4*>(g);
  float4* m4 = reinterpret_cast<float4*>(m);
  float4* v4 = reinterpret_cast<float4*>(v);
  float4* e4 = reinterpret_cast<float4*>(ema);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 pv = p4[i], mv = m4[i], vv = v4[i], ev = EMA ? e4[i] : float4{0.f, 0.f, 0.f, 0.f};
    const float4 gv = g4[i];
    upd(pv.x, gv.x, mv.x, vv.x, ev.x);
    upd(pv.y, gv.y, mv.y, vv.y, ev.y);
    upd(pv.z, gv.z, mv.z, vv.z, ev.z);
    upd(pv.w, gv.w, mv.w, vv.w, ev.w);
    if (!skip) {
      p4[i] = pv;
      m4[i] = mv;
      v4[i] = vv;
    }
    if constexpr (EMA) e4[i] = ev;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (n4 << 2) + threadIdx.x;
    float pv = p[i], mv = m[i], vv = v[i], ev = EMA ? ema[i] : 0.f;
    upd(pv, g[i], mv, vv, ev);
    if (!skip) {
      p[i] = pv;
      m[i] = mv;
      v[i] = vv;
    }
    if constexpr (EMA) ema[i] = ev;
  }
}

}  // namespace e3k

// loss = scale * sum_i w_i (pred_i - target_i)^2 and its gradient 2 scale w_i (pred_i - target_i) in ONE launch (w == NULL: 1 / n, a
// mean): the step's loss on a few hundred graph energies (or a few thousand force components) is launch latency, not arithmetic --
// as torch ops it is a subtraction, a power, one or two products, a reduction and their five or six backward kernels.  One
// workgroup, a fixed summation order (bit-reproducible).
namespace e3k {
__global__ __launch_bounds__(1024) void sq_error_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                        const float* __restrict__ w, int w_group, int64_t n, float scale,
                                                        float* __restrict__ loss, float* __restrict__ grad) {
  __shared__ float part[16];
  const float wu = 1.0f / (float)n;
  float acc = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    const float d = pred[i] - target[i], wi = w ? w[i / w_group] : wu;
    acc = fmaf(wi * d, d, acc);
    grad[i] = 2.0f * scale * wi * d;
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int k = 0; k < 16; ++k) s += part[k];
    loss[0] = scale * s;
  }
}
}  // namespace e3k

extern "C" int e3k_sq_error(const float* pred, const float* target, const float* weight, int32_t w_group, int64_t n, float scale,
                            float* loss, float* grad, void* stream) {
  if (n <= 0 || !pred || !target || !loss || !grad || (weight && w_group < 1)) return E3K_ERR_INVALID;
  hipLaunchKernelGGL(e3k::sq_error_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, pred, target, weight, weight ? w_group : 1, n,
                     scale, loss, grad);
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}

extern "C" int e3k_adam_ema_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* ema,
                                 int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                                 float ema_decay, int32_t ema_use_num_updates, float max_grad_norm,
                                 int32_t skip_nonfinite, float* state, void* stream) {
  if (n < 0 || !(lr >= 0.f) || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f) || !(eps >= 0.f))
    return E3K_ERR_INVALID;
  if (ema && !(ema_decay >= 0.f && ema_decay <= 1.f)) return E3K_ERR_INVALID;
  if (n == 0) return E3K_OK;
  if (!param || !grad || !exp_avg || !exp_avg_sq || !state) return E3K_ERR_INVALID;
  if ((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(exp_avg) |
       reinterpret_cast<uintptr_t>(exp_avg_sq) | reinterpret_cast<uintptr_t>(ema)) & 15)
    return E3K_ERR_INVALID;   // flat buffers are 16-byte aligned (float4 streams)
  hipStream_t st = (hipStream_t)stream;
  int64_t blocks = ((n >> 2) + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  const int have_norm = (max_grad_norm > 0.f || skip_nonfinite) ? 1 : 0;
  if (have_norm) {
    hipLaunchKernelGGL(e3k::sumsq_kernel, dim3((unsigned)(blocks < 512 ? blocks : 512)), dim3(256), 0, st, grad, n, state);
    E3K_CHECK_LAUNCH();
  }
  if (ema && !have_norm) {      // (the same values tick_kernel + ema_tick_kernel leave behind, one launch)
    hipLaunchKernelGGL(e3k::tick_both_kernel, dim3(1), dim3(1), 0, st, state, beta1, beta2, ema_decay, ema_use_num_updates);
    E3K_CHECK_LAUNCH();
  } else {
    hipLaunchKernelGGL(e3k::tick_kernel, dim3(1), dim3(1), 0, st, state, beta1, beta2, max_grad_norm, have_norm, skip_nonfinite);
    E3K_CHECK_LAUNCH();
    if (ema) {
      hipLaunchKernelGGL(e3k::ema_tick_kernel, dim3(1), dim3(1), 0, st, state + 8, ema_decay, ema_use_num_updates);
      E3K_CHECK_LAUNCH();
    }
  }
  if (ema) {
    hipLaunchKernelGGL(e3k::adam_ema_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, param, grad, exp_avg,
                       exp_avg_sq, ema, n, lr, beta1, beta2, eps, weight_decay, state, state + 8);
  } else {
    hipLaunchKernelGGL(e3k::adam_ema_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, param, grad, exp_avg,
                       exp_avg_sq, (float*)nullptr, n, lr, beta1, beta2, eps, weight_decay, state,
                       (const float*)nullptr);
  }
  E3K_CHECK_LAUNCH();
  return E3K_OK;
}
