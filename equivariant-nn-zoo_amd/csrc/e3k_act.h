// Activation functions shared by the node kernels and the fused radial MLP (gfx950).
//   normalize2mom(act) inside FullyConnectedNet / Gate   e3_layers/utils/utils.py:64-84, SURVEY.md A.5
#pragma once
#include <hip/hip_runtime.h>

namespace e3k {

// activation ids: 0 identity, 1 ssp, 2 silu, 3 tanhlu, 4 tanh, 5 abs  (e3_layers/utils/utils.py:64-84)
// Hardware transcendentals (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1-2 ulp): the precise library expf / log1pf cost
// ~100 VALU instructions per softplus and made the activation passes VALU-bound, not HBM-bound.
__device__ __forceinline__ float sigmoidf_(float x) { return __frcp_rn(1.0f + __expf(-x)); }
// log(1 + e^-|x|): absolute error <= 1 ulp of 1.0 (6e-8), far inside the 1e-5 parity budget of outputs that are O(1)
__device__ __forceinline__ float softplus_tail(float ax) { return __logf(1.0f + __expf(-ax)); }

__device__ __forceinline__ float act_f(int id, float x) {
  switch (id) {
    case 1: return fmaxf(x, 0.f) + softplus_tail(fabsf(x)) - 0.6931471805599453f;
    case 2: return x * sigmoidf_(x);
    case 3: return tanhf(x) * fabsf(x);
    case 4: return tanhf(x);
    case 5: return fabsf(x);
    default: return x;
  }
}
__device__ __forceinline__ float act_df(int id, float x) {
  switch (id) {
    case 1: return sigmoidf_(x);
    case 2: {
      const float s = sigmoidf_(x);
      return s * (1.0f + x * (1.0f - s));
    }
    case 3: {
      const float th = tanhf(x);
      const float sg = (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f);
      return (1.0f - th * th) * fabsf(x) + th * sg;
    }
    case 4: {
      const float th = tanhf(x);
      return 1.0f - th * th;
    }
    case 5: return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f);
    default: return 1.0f;
  }
}

// second derivative (double backward: force training differentiates the backward pass once more)
__device__ __forceinline__ float act_d2f(int id, float x) {
  switch (id) {
    case 1: {
      const float s = sigmoidf_(x);
      return s * (1.0f - s);
    }
    case 2: {
      const float s = sigmoidf_(x);
      return s * (1.0f - s) * (2.0f + x * (1.0f - 2.0f * s));
    }
    case 3: {
      const float th = tanhf(x), sech2 = 1.0f - th * th;
      const float sg = (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f);
      return 2.0f * sech2 * (sg - th * fabsf(x));
    }
    case 4: {
      const float th = tanhf(x);
      return -2.0f * th * (1.0f - th * th);
    }
    default: return 0.0f;
  }
}

}  // namespace e3k
