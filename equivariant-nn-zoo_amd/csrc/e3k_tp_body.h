// Pieces of the tensor-product kernels (e3k_tp.hip): compile-time loops over the (l2, l3) slots of an input degree, the
// spherical-harmonics registers, the plan object.
#pragma once
#include <type_traits>

#include "e3k_common.h"
#include "e3k_cg_gen.h"

static_assert(E3K_MAXQ == E3K_TP_MAXQ, "e3k.h and e3k_cg_gen.h disagree on the slot count");
static_assert(E3K_L2MAX == 2, "YRegs below is written for sh degrees 0..2");

namespace e3k {

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

struct YRegs {
  float y0[1];
  float y1[3];
  float y2[5];
};
template <int L2>
__device__ __forceinline__ auto& yref(YRegs& y) {
  if constexpr (L2 == 0) return y.y0;
  else if constexpr (L2 == 1) return y.y1;
  else return y.y2;
}

struct TpArgs {
  const float* x;      // [N, d_in]  cf
  const float* sh;     // [E, d_sh]
  const float* w;      // [E, W]
  const float* g_out;  // [N, d_mid] cf (backward)
  float* out;          // [N, d_mid] cf (forward)
  float* g_w;          // [E, W]
  float* g_w2;         // [E, W], optional second weight-gradient output (tp_bwd_x MODE 7: the plain one beside the dual one)
  float* g_sh;         // [E, d_sh]
  float* g_x;          // [N, d_in]
  const int32_t* nbr;  // src[e] (fwd, bwd_w) or dst[e] (bwd_x)
  const int32_t* ptr;  // CSR row pointers [N+1]
  const int32_t* perm; // CSR edge ids [E]
  const int32_t* bin;  // TABLE kernels: knot i of every edge [E] (stencil rows i-1 .. i+2); w is then the knot table [K + 1, W]
  const float* coef;   // TABLE kernels: the four interpolation weights of every edge [E, 4] (e3k_rtable_bins)
  // force training on the table: the slope table D [K + 1, W] (d T / d r on the knots, interpolated with the SAME weights:
  // dw/dr[e] = sum_k coef[e, k] D[bin[e] - 1 + k]) and the second operand set of the JVP / DUAL forms (the double backward:
  // the product rule over (x, sh, r))
  const float* w2;     // D
  const float* x2;     // [N, d_in] cf
  const float* sh2;    // [E, d_sh]
  const float* s2;     // [E]: the radius' partner (a cotangent): the third term's weights are s2[e] * dw/dr[e]
  float* g_r;          // [E]: gradient w.r.t. the radius (tp_bwd_e), accumulated with one atomic per wave and edge
  const int32_t* erec; // FULL kernels, optional: the walk's edge records [E, 16] (e3k_edge_records) -- perm, nbr, bin, coef, sh are then unused
  int32_t d_in, d_sh, W, d_mid;
  int32_t x_shared;    // bwd_x: some input block is read by more than one group => accumulate g_x with atomics
  int32_t ablate;      // debug build only: timing-only ablation mask (0 in the product library)
  int32_t order;       // work order of the launch (E3K_TP_PROLOGUE): 0 node-major, 1 group-major inside an XCD's node slice
  int32_t e_store;     // edge gradients (g_sh, g_r): 0 = float atomics into [E, .]; 1 = every work item STORES its share into its own
                       // slice of [n_gc, E, .] (g_sh / g_r point at the slices' base, e_edges = E): summed in a fixed order afterwards
  int64_t e_edges;
  int64_t n_items;
};

struct GroupRegs {
  int x_off, mul;
  unsigned mask;
  int y_off[3];
};

__device__ __forceinline__ void load_y(YRegs& y, const float* __restrict__ yr, const e3k_tp_group& g) {
  // wave-uniform addresses: these become scalar loads
  if (g.y_off[0] >= 0) y.y0[0] = yr[g.y_off[0]];
  if (g.y_off[1] >= 0) {
#pragma unroll
    for (int j = 0; j < 3; ++j) y.y1[j] = yr[g.y_off[1] + j];
  }
  if (g.y_off[2] >= 0) {
#pragma unroll
    for (int j = 0; j < 5; ++j) y.y2[j] = yr[g.y_off[2] + j];
  }
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
// visits the path slots whose output degree the plan can contain: slots with l3 > L3MAX are compiled out, so an
// l_max = 2 model carries no accumulators for l3 = 3 outputs (14 of the 36 registers of an l1 = 2 group)
// PART splits a group's slots between two waves (0: slots below the split point, 1: the rest, 2: all of them): at
// l_max = 3 a group carries 27-36 accumulators; halving them takes the kernels from 3-4 to 6-7 waves per SIMD at the
// price of gathering x[src] twice (an L2 hit).  Split points: l1 = 1 -> slot 4, l1 = 2 -> slot 4, l1 = 3 -> slot 3.
template <int L1> struct SplitAt { static constexpr int Q = L1 == 3 ? 3 : 4; };
template <class S, int L1, int L3MAX, int PART, class F>
__device__ __forceinline__ void slot_for_part(F&& f) {
  static_for<0, S::NQ>([&](auto qc) {
    constexpr int Q = decltype(qc)::value;
    constexpr bool in_part = PART == 2 || (PART == 0 ? Q < SplitAt<L1>::Q : Q >= SplitAt<L1>::Q);
    if constexpr (S::L3[Q] <= L3MAX && in_part) f(qc);
  });
}

}  // namespace e3k

struct e3k_tp_plan {
  int32_t n_groups, d_in, d_sh, w_numel, d_mid;
  e3k_tp_group* d_groups;
  int2* d_gc;     // (group, 64-channel chunk) work list, all degrees
  int32_t n_gc;
  int32_t max_l1; // largest input degree among the groups (selects the kernel instantiation)
  int32_t max_l3; // largest output degree any group's mask enables
  int32_t split;  // 1: groups with l1 >= 1 are walked by two waves (slot parts 0 / 1)
  int32_t x_cols; // input columns owned by some group: sum over groups of (2 l1 + 1) * mul
  int32_t full64;   // 1: every group has a multiple of 64 channels (the kernels drop their idle-lane handling)
  int32_t x_shared; // 1: two groups read overlapping input columns (an sh degree that repeats opens a second group)
};
