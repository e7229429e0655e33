// A convolution layer as ONE call: FactorizedConvolution + Gate forward and backward
// (reference: e3_layers/nn/message_passing.py:91-124 forward, :249 gate; their autograd backward).
//
// This file launches nothing of its own except two row gather / scatter kernels: it is the HOST side of a layer in native
// code.  The kernels are the library's (e3k_tp, e3k_gemm, e3k_node, e3k_mlp, e3k_rtable); what used to sit between them was
// Python -- one ctypes call, a descriptor rebase, a stream switch and an output allocation per launch: 0.22 ms forward
// and 0.30 ms backward per layer, ~2.6 ms of a training step that takes 3.4 ms at 32 molecules (host-bound up to 128
// molecules per GPU: every batch size the reference ships) and as long as the GPU time at 256.  Here the whole sequence
// is issued back to back on the caller's streams:
//
//   forward   side : radial MLP hidden chain -> last layer (-> table interpolation)          -> w [E, W]
//             side2: per-key self-connection weights M, keyed self-connection GEMM           -> conv [N, d_conv]
//             main : (relayout) -> linear_1 -> [wait side] tensor product + reduce -> [wait side2]
//                    conv += scale * Linear(mid) -> gate                                     -> y
//             side : (look-ahead) the NEXT layer's radial branch behind this layer's tensor product
//   backward  main : gate' -> {trailing Linear dgrad, self-connection dgrad} -> tp_bwd_x -> tp_bwd_w -> linear_1 dgrad
//             side3: {trailing Linear wgrad, self-connection wgrad} behind gate'; linear_1 wgrad behind tp_bwd_x
//             side2: per-key weight gradient -> flat weight gradient + attribute gradient
//             side : (table^T) -> last-layer wgrad / dgrad -> hidden chain backward
//
// Cross-stream edges are hipEvents owned by the layer object (no timing, re-recorded every call).  Without `fork` all
// streams are the same and the events are skipped.  GEMMs that share an operand go out in one e3k_gemm_multi call.
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "e3k_common.h"

namespace e3k {

// a_rep[k, :] = attrs[reps[k], :]
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, const int64_t* __restrict__ rows, int n_rows,
                                                          int width, float* __restrict__ dst) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_rows * width) return;
  const int r = i / width, c = i - r * width;
  dst[i] = src[rows[r] * width + c];
}

// g_attrs[reps[k], :] += ga[k, :]   (g_attrs zero-filled before; keys with no rows carry ga = 0 and reps = 0: harmless adds)
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ src, const int64_t* __restrict__ rows,
                                                           const int32_t* __restrict__ bounds, int n_rows, int width,
                                                           float* __restrict__ dst) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_rows * width) return;
  const int r = i / width, c = i - r * width;
  if (bounds[2 * r + 1] > 0) atomicAdd(dst + rows[r] * width + c, src[i]);   // distinct representatives: no contention
}

}  // namespace e3k

// per-kernel timing (bench.py's roofline block): event pairs around the edge kernels, on the stream that runs them
constexpr int PROF_KINDS = 6;   // E3K_PROF_* in e3k.h
struct ProfRing {
  std::vector<hipEvent_t> beg, end;
  std::vector<int64_t> n, e;
  int count = 0;
};

struct e3k_layer {
  e3k_layer_desc d;
  std::vector<e3k_gemm_problem> sets[13];
  int rounds[13];
  int round_start[13][5];
  std::vector<e3k_gate_seg> gate;
  std::vector<e3k_block> in_blocks;
  std::vector<e3k_kw_instr> kw;
  e3k_kw_args* kwa = nullptr;      // the same table in device memory (the multi-layer keyed-weight launches)
  hipEvent_t ev[8];
  int n_ev;
  mutable ProfRing prof[PROF_KINDS];
  int prof_cap = 0;
  unsigned prof_mask = ~0u;      // kinds that record event pairs while prof_cap > 0
};

namespace {
struct Timed {   // records an event pair around one launch when the layer is being profiled
  const e3k_layer* L;
  int kind, slot;
  hipStream_t st;
  Timed(const e3k_layer* L_, int kind_, void* st_, int64_t n, int64_t e) : L(L_), kind(kind_), slot(-1), st((hipStream_t)st_) {
    if (L->prof_cap <= 0 || !((L->prof_mask >> kind) & 1u)) return;
    ProfRing& r = L->prof[kind];
    if (r.count >= L->prof_cap) return;
    slot = r.count++;
    r.n[slot] = n;
    r.e[slot] = e;
    (void)hipEventRecord(r.beg[slot], st);
  }
  ~Timed() {
    if (slot >= 0) (void)hipEventRecord(L->prof[kind].end[slot], st);
  }
};
}  // namespace

namespace {

// timing-only ablation (tools/ablate.sh): E3K_ABLATE is a bitmask of launches to SKIP -- the results are wrong with any
// bit set; it answers "what would the step gain if this kernel family were free" before anyone optimises it.
// 1: keyed-weight kernels (fwd + bwd), 2: weight-gradient GEMMs, 4: gate fwd + bwd, 8: table interpolation fwd + bwd,
// 16: radial MLP hidden chain + last layer (fwd + bwd), 32: tp_bwd_x, 64: input-gradient GEMMs
// (debug build only -- see e3k_common.h: in the product library ABLATE is the constant 0 and every `ABLATE & bit` below folds away)
E3K_KNOB_INT(ABLATE, "E3K_ABLATE", 0);
// where the backward's side work runs (re-measured in round 4 after the GEMM kernels got shorter, tools/ab_bench.py, 2-3
// interleaved rounds per variant): tp_bwd_w on the radial stream + the weight gradients beside tp_bwd_x (1 / 0) against
// tp_bwd_w on the main stream + the weight gradients behind tp_bwd_x (0 / 1, round 3's choice): 256 molecules eager 4.815 vs
// 4.846 ms, l_max 3 7.32 vs 7.51, config_diffusion_CA 9.12 vs 9.60, config_diffusion and the replayed steps within noise
E3K_KNOB_INT(BWDW_SIDE, "E3K_BWDW_SIDE", 1);
E3K_KNOB_INT(WGRAD_LATE, "E3K_WGRAD_LATE", 0);
// the tensor-product kernels interpolate the path weights from the knot table themselves (no w[E, W])
static inline bool in_kernel_table(const e3k_layer_desc& d, const e3k_layer_radial& r) { return r.use_table && r.in_kernel; }

enum SetId { LIN1_FWD, LIN1_DGRAD, LIN1_DGRAD_ACC, LIN1_WGRAD, POST_FWD, POST_DGRAD, POST_WGRAD, SC_FWD, SC_DGRAD, SC_WGRAD,
             LAST_FWD, LAST_DGRAD, LAST_WGRAD, N_SETS };

#define E3K_TRY(call)              \
  do {                             \
    const int rc_ = (call);        \
    if (rc_ != E3K_OK) return rc_; \
  } while (0)

// template sets that go out together: round r of every set in one e3k_gemm_multi call, the rounds in order
// (eight sets per call since round 6: the last layers of ALL five radial MLPs of a stack in one call -- with four the fifth layer's
//  went out as launches of its own: 2 x 20 us of split-K for what rides in the first launch for 5)
constexpr int SEG_MAX = 8;
struct Seg {
  e3k_gemm_segment s[SEG_MAX];
  const e3k_layer* layer[SEG_MAX];
  int id[SEG_MAX];
  int n = 0;
  void add(const e3k_layer* L, SetId set, const void* a, const void* b, void* c, int64_t rows) {
    e3k_gemm_segment sg{};
    sg.a_base = a; sg.b_base = b; sg.c_base = c;
    sg.M1 = rows;
    layer[n] = L;
    id[n] = set;
    s[n++] = sg;
  }
  void add_keyed(const e3k_layer* L, SetId set, const void* a, const void* b, void* c, int64_t rows, const int32_t* perm,
                 const int32_t* bounds, int32_t n_keys) {
    add(L, set, a, b, c, rows);
    e3k_gemm_segment& sg = s[n - 1];
    sg.n_keys = n_keys; sg.perm = perm; sg.groups_dev = bounds; sg.b_key_stride = L->d.ld_m;
  }
  int run(int wgrad, void* st) {
    for (int r = 0;; ++r) {
      e3k_gemm_segment round[SEG_MAX];
      int m = 0;
      for (int i = 0; i < n; ++i) {
        const e3k_layer* L = layer[i];
        if (r >= L->rounds[id[i]]) continue;
        e3k_gemm_segment sg = s[i];
        const int beg = L->round_start[id[i]][r], end = L->round_start[id[i]][r + 1];
        sg.templates = L->sets[id[i]].data() + beg;
        sg.n_templates = end - beg;
        round[m++] = sg;
      }
      if (!m) return E3K_OK;
      const int rc = e3k_gemm_multi(round, m, wgrad, st);
      if (rc != E3K_OK) return rc;
    }
  }
};

// consumer waits for everything enqueued on producer so far
int edge(const e3k_layer* L, int slot, void* producer, void* consumer) {
  if (producer == consumer) return E3K_OK;
  hipEvent_t ev = L->ev[slot];
  if (hipEventRecord(ev, (hipStream_t)producer) != hipSuccess) return E3K_ERR_LAUNCH;
  if (hipStreamWaitEvent((hipStream_t)consumer, ev, 0) != hipSuccess) return E3K_ERR_LAUNCH;
  return E3K_OK;
}

// the in-kernel table in its packed form (12-byte records, e3k_rtable_pack): behind whatever wrote T, on the radial stream
int pack_table(const e3k_layer* L, const e3k_layer_radial& r, void* st) {
  if (!in_kernel_table(L->d, r) || !r.P || r.packed) return E3K_OK;      // (packed: e3k_radial_stack_fwd did it for all its layers at once)
  return e3k_rtable_pack(r.T, r.knots, L->d.W, r.P, st);
}

int radial_fwd(const e3k_layer* L, const e3k_layer_radial& r, void* st) {
  const e3k_layer_desc& d = L->d;
  if (r.R == 0 || r.E == 0) return E3K_OK;
  if (r.have_rows) {      // the stack computed T (table) or w (per edge) already
    if (r.use_table && !(ABLATE & 8) && !in_kernel_table(d, r)) {
      Timed t(L, E3K_PROF_RTABLE_FWD, st, r.R, r.E);
      E3K_TRY(e3k_rtable_interp_fwd(r.T, r.bin_perm, r.bin, r.bin_coef, r.E, r.knots, d.W, r.w, st));
    }
    return pack_table(L, r, st);
  }
  float* zs[4] = {r.z[0], r.z[1], r.z[2], r.z[3]};
  if (!(ABLATE & 16))
    E3K_TRY(e3k_mlp_hidden_fwd(r.radial, r.R, d.k0, d.h, d.n_hidden, r.w_hidden, d.alphas, d.act, d.cst, r.keep ? zs : nullptr, r.h, st));
  if (!(ABLATE & 16)) {
    Timed t(L, E3K_PROF_RADIAL_LAST_FWD, st, r.R, r.E);
    Seg g;
    g.add(L, LAST_FWD, r.h, r.w_last, r.use_table ? r.T : r.w, r.R);
    E3K_TRY(g.run(0, st));
  }
  if (r.use_table && !(ABLATE & 8) && !in_kernel_table(d, r)) {
    Timed t(L, E3K_PROF_RTABLE_FWD, st, r.R, r.E);
    E3K_TRY(e3k_rtable_interp_fwd(r.T, r.bin_perm, r.bin, r.bin_coef, r.E, r.knots, d.W, r.w, st));
  }
  return pack_table(L, r, st);
}

}  // namespace

extern "C" int e3k_layer_create(const e3k_layer_desc* desc, e3k_layer** out) {
  if (!desc || !out || !desc->tp) return E3K_ERR_INVALID;
  if (desc->n_hidden < 1 || desc->n_hidden > 4) return E3K_ERR_UNSUPPORTED;
  e3k_layer* L = new (std::nothrow) e3k_layer();
  if (!L) return E3K_ERR_LAUNCH;
  L->d = *desc;
  const e3k_gemm_set* src[N_SETS] = {&desc->lin1_fwd, &desc->lin1_dgrad, &desc->lin1_dgrad_acc, &desc->lin1_wgrad, &desc->post_fwd,
                                     &desc->post_dgrad, &desc->post_wgrad, &desc->sc_fwd, &desc->sc_dgrad, &desc->sc_wgrad,
                                     &desc->last_fwd, &desc->last_dgrad, &desc->last_wgrad};
  for (int i = 0; i < N_SETS; ++i) {
    const e3k_gemm_set& gs = *src[i];
    bool ok = gs.n >= 0 && (gs.n == 0 || gs.p) && gs.n_rounds >= 0 && gs.n_rounds <= 4 && (gs.n == 0 || gs.n_rounds >= 1);
    for (int r = 0; ok && r < gs.n_rounds; ++r) ok = gs.round_start[r] <= gs.round_start[r + 1];
    ok = ok && (gs.n_rounds == 0 || (gs.round_start[0] == 0 && gs.round_start[gs.n_rounds] == gs.n));
    if (!ok) {
      delete L;
      return E3K_ERR_INVALID;
    }
    L->sets[i].assign(gs.p, gs.p + gs.n);
    L->rounds[i] = gs.n_rounds;
    for (int r = 0; r < 5; ++r) L->round_start[i][r] = gs.round_start[r];
  }
  if (desc->gate) L->gate.assign(desc->gate, desc->gate + desc->n_gate);
  if (desc->in_blocks) L->in_blocks.assign(desc->in_blocks, desc->in_blocks + desc->n_in_blocks);
  if (desc->kw) L->kw.assign(desc->kw, desc->kw + desc->n_kw);
  L->n_ev = 0;
  if (!L->kw.empty() && e3k_kw_args_create(L->kw.data(), (int32_t)L->kw.size(), desc->V, desc->ld_m, &L->kwa) != E3K_OK) {
    delete L;
    return E3K_ERR_INVALID;
  }
  for (int i = 0; i < 8; ++i) {
    if (hipEventCreateWithFlags(&L->ev[i], hipEventDisableTiming) != hipSuccess) {
      e3k_layer_destroy(L);
      return E3K_ERR_LAUNCH;
    }
    L->n_ev = i + 1;
  }
  *out = L;
  return E3K_OK;
}

static void prof_free(e3k_layer* L) {
  for (int k = 0; k < PROF_KINDS; ++k) {
    for (hipEvent_t ev : L->prof[k].beg) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : L->prof[k].end) (void)hipEventDestroy(ev);
    L->prof[k] = ProfRing();
  }
  L->prof_cap = 0;
}

extern "C" void e3k_layer_destroy(e3k_layer* L) {
  if (!L) return;
  prof_free(L);
  for (int i = 0; i < L->n_ev; ++i) (void)hipEventDestroy(L->ev[i]);
  e3k_kw_args_destroy(L->kwa);
  delete L;
}

extern "C" int e3k_layer_profile_mask(e3k_layer* L, int32_t capacity, uint32_t kinds) {
  const int rc = e3k_layer_profile(L, capacity);
  if (rc == E3K_OK) L->prof_mask = kinds;
  return rc;
}

extern "C" int e3k_layer_profile(e3k_layer* L, int32_t capacity) {
  if (!L || capacity < 0) return E3K_ERR_INVALID;
  prof_free(L);
  L->prof_mask = ~0u;
  for (int k = 0; k < PROF_KINDS && capacity > 0; ++k) {
    ProfRing& r = L->prof[k];
    r.beg.resize(capacity);
    r.end.resize(capacity);
    r.n.assign(capacity, 0);
    r.e.assign(capacity, 0);
    for (int i = 0; i < capacity; ++i)
      if (hipEventCreate(&r.beg[i]) != hipSuccess || hipEventCreate(&r.end[i]) != hipSuccess) return E3K_ERR_LAUNCH;
  }
  L->prof_cap = capacity;
  return E3K_OK;
}

extern "C" int e3k_layer_profile_read(e3k_layer* L, int32_t kind, float* ms, int64_t* n, int64_t* e, int32_t cap) {
  if (!L || kind < 0 || kind >= PROF_KINDS || cap < 0) return E3K_ERR_INVALID;
  const ProfRing& r = L->prof[kind];
  const int cnt = r.count < cap ? r.count : cap;
  for (int i = 0; i < cnt; ++i) {
    if (hipEventSynchronize(r.end[i]) != hipSuccess) return E3K_ERR_LAUNCH;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.beg[i], r.end[i]) != hipSuccess) return E3K_ERR_LAUNCH;
    ms[i] = t;
    n[i] = r.n[i];
    e[i] = r.e[i];
  }
  return cnt;
}

extern "C" int e3k_layer_fwd(const e3k_layer* L, const e3k_layer_fwd_args* a) {
  if (!L || !a || a->N < 0 || a->E < 0) return E3K_ERR_INVALID;
  const e3k_layer_desc& d = L->d;
  const bool has_sc = !L->sets[SC_FWD].empty();
  void* main = a->main;
  void* side = a->fork ? a->side : main;
  void* side2 = a->fork ? a->side2 : main;
  if (a->N == 0) return E3K_OK;
  if (!a->x || !a->sh || !a->x1 || !a->mid || !a->conv || !a->y) return E3K_ERR_INVALID;
  if (in_kernel_table(d, a->rad) ? (!a->rad.T || !e3k_tp_table_supported(d.tp)) : !a->rad.w) return E3K_ERR_INVALID;
  // --- radial branch (unless the previous layer's look-ahead already issued it)
  if (!a->has_w && !(a->rad.have_rows && !a->rad.use_table)) {
    E3K_TRY(edge(L, 0, main, side));
    E3K_TRY(radial_fwd(L, a->rad, side));
  }
  // --- node side
  const float* x_cf = a->x;
  if (!a->in_cf && !L->in_blocks.empty()) {
    E3K_TRY(e3k_relayout(a->x, a->N, d.d_in, L->in_blocks.data(), (int32_t)L->in_blocks.size(), 1, a->x_cf, main));
    x_cf = a->x_cf;
  }
  if (has_sc) {
    if (!a->m || !a->perm || !a->bounds || a->n_keys <= 0) return E3K_ERR_INVALID;
    if (!a->have_m && (!a->node_attrs || !a->a_rep || !a->reps || !a->w_sc)) return E3K_ERR_INVALID;
    E3K_TRY(edge(L, 1, main, side2));        // x_cf (and the attributes) are ready
    if (!a->have_m) {      // (have_m: the per-key weights M came from e3k_kw_stack_fwd, issued on this stream before)
      const int tot = a->n_keys * d.V;
      hipLaunchKernelGGL(e3k::gather_rows_kernel, dim3((tot + 255) / 256), dim3(256), 0, (hipStream_t)side2, a->node_attrs, a->reps,
                         a->n_keys, d.V, a->a_rep);
      if (!(ABLATE & 1))
        E3K_TRY(e3k_keyed_weights_fwd(a->a_rep, a->w_sc, L->kw.data(), (int32_t)L->kw.size(), a->n_keys, d.V, d.ld_m, a->m, side2));
    }
    if (!d.sc_out_covered && e3k::zero_fill(a->conv, sizeof(float) * a->N * d.d_conv, (hipStream_t)side2))
      return E3K_ERR_LAUNCH;
    Seg g;
    g.add_keyed(L, SC_FWD, x_cf, a->m, a->conv, a->N, a->perm, a->bounds, a->n_keys);
    if (side2 == main) g.add(L, LIN1_FWD, x_cf, a->w_lin1, a->x1, a->N);      // one stream: both readers of x_cf in one launch
    E3K_TRY(g.run(0, side2));
    if (side2 != main) {
      Seg g1;
      g1.add(L, LIN1_FWD, x_cf, a->w_lin1, a->x1, a->N);
      E3K_TRY(g1.run(0, main));
    }
  } else {
    if (!d.post_out_covered && e3k::zero_fill(a->conv, sizeof(float) * a->N * d.d_conv, (hipStream_t)main))
      return E3K_ERR_LAUNCH;
    Seg g1;
    g1.add(L, LIN1_FWD, x_cf, a->w_lin1, a->x1, a->N);
    E3K_TRY(g1.run(0, main));
  }
  E3K_TRY(edge(L, 2, side, main));           // the per-edge weights
  {
    Timed t(L, E3K_PROF_TP_FWD, main, a->N, a->E);
    if (in_kernel_table(d, a->rad) && a->rad.P)
      E3K_TRY(e3k_tp_fwd_ptable(d.tp, a->x1, a->rad.P, a->rad.erec_dst, a->dst_ptr, a->N, a->E, a->mid, main));
    else if (in_kernel_table(d, a->rad))
      E3K_TRY(e3k_tp_fwd_table(d.tp, a->x1, a->sh, a->rad.T, a->rad.bin, a->rad.bin_coef, a->src, a->dst_ptr, a->dst_perm, a->N, a->E,
                               a->mid, main));
    else
      E3K_TRY(e3k_tp_fwd(d.tp, a->x1, a->sh, a->rad.w, a->src, a->dst_ptr, a->dst_perm, a->N, a->E, a->mid, main));
  }
  if (a->next && a->next_rad && side != main) {
    E3K_TRY(edge(L, 3, main, side));         // behind this layer's tensor product: both are HBM streams
    E3K_TRY(radial_fwd(a->next, *a->next_rad, side));
  }
  if (has_sc) E3K_TRY(edge(L, 4, side2, main));
  {
    Seg g;
    g.add(L, POST_FWD, a->mid, a->w_post, a->conv, a->N);
    E3K_TRY(g.run(0, main));
  }
  if (!(ABLATE & 4))
    E3K_TRY(e3k_gate_fwd(a->conv, a->N, d.d_conv, d.d_out, L->gate.data(), (int32_t)L->gate.size(), a->out_cf, a->y, main));
  if (hipGetLastError() != hipSuccess) return E3K_ERR_LAUNCH;
  return E3K_OK;
}

extern "C" int e3k_layer_bwd(const e3k_layer* L, const e3k_layer_bwd_args* a) {
  if (!L || !a || a->N < 0 || a->E < 0) return E3K_ERR_INVALID;
  const e3k_layer_desc& d = L->d;
  const bool has_sc = !L->sets[SC_FWD].empty();
  if (a->N == 0) return E3K_OK;
  void* main = a->main;
  void* side = a->fork ? a->side : main;
  void* side2 = a->fork ? a->side2 : main;
  void* side3 = a->fork ? a->side3 : main;
  const e3k_layer_radial& r = a->rad;
  const bool need_last = a->gb_last != nullptr;
  bool need_hidden = false;
  for (int i = 0; i < d.n_hidden; ++i) need_hidden = need_hidden || a->gb_hidden[i];
  const bool need_radial_side = need_last || need_hidden || a->need_radial || (r.have_rows && (a->g_w || a->g_T));
  const bool need_post = a->gb_post != nullptr, need_lin1 = a->gb_lin1 != nullptr;
  // have_m: the gradient of the per-key weights, gm, is this layer's OUTPUT (e3k_kw_stack_bwd turns the gm of all the
  // layers into weight and attribute gradients in one pass, on the self-connection stream)
  const bool want_sc = has_sc && (a->have_m ? a->gm != nullptr : (a->gb_sc || a->need_attrs));
  const bool need_x1 = a->need_x || need_lin1;
  if (!a->gy || !a->g_conv || !a->g_mid || !a->conv) return E3K_ERR_INVALID;

  // gate' -> gradient of the convolution output; both readers of it in one call
  if (!(ABLATE & 4))
    E3K_TRY(e3k_gate_bwd(a->conv, a->gy, nullptr, a->N, d.d_conv, d.d_out, L->gate.data(), (int32_t)L->gate.size(), a->out_cf,
                         a->g_conv, main));
  if (!d.post_in_covered && e3k::zero_fill(a->g_mid, sizeof(float) * a->N * d.d_mid, (hipStream_t)main))
    return E3K_ERR_LAUNCH;
  {
    Seg g;
    g.add(L, POST_DGRAD, a->g_conv, a->w_post, a->g_mid, a->N);
    if (a->need_x) {
      if (!a->g_xcf) return E3K_ERR_INVALID;
      const bool covered = has_sc ? d.sc_in_covered : d.lin1_in_covered;
      if (!covered && e3k::zero_fill(a->g_xcf, sizeof(float) * a->N * d.d_in, (hipStream_t)main)) return E3K_ERR_LAUNCH;
      if (has_sc) g.add_keyed(L, SC_DGRAD, a->g_conv, a->m, a->g_xcf, a->N, a->perm, a->bounds, a->n_keys);
    }
    if (!(ABLATE & 64)) E3K_TRY(g.run(0, main));
  }
  // weight gradients that only need g_conv: off the critical path (forked) or together with linear_1's below
  auto keyed_weight_grads = [&]() -> int {
    E3K_TRY(edge(L, 1, side3, side2));
    if (a->have_m) return E3K_OK;
    if (a->need_attrs && e3k::zero_fill(a->ga, sizeof(float) * a->n_keys * d.V, (hipStream_t)side2)) return E3K_ERR_LAUNCH;
    if (!(ABLATE & 1))
      E3K_TRY(e3k_keyed_weights_bwd(a->a_rep, a->w_sc, a->gm, L->kw.data(), (int32_t)L->kw.size(), a->n_keys, d.V, d.ld_m,
                                    a->need_attrs ? a->ga : nullptr, a->gb_sc, a->acc_sc, a->kw_ws, side2));
    if (a->need_attrs) {
      if (e3k::zero_fill(a->g_attrs, sizeof(float) * a->N * d.V, (hipStream_t)side2)) return E3K_ERR_LAUNCH;
      const int tot = a->n_keys * d.V;
      hipLaunchKernelGGL(e3k::scatter_rows_kernel, dim3((tot + 255) / 256), dim3(256), 0, (hipStream_t)side2, a->ga, a->reps, a->bounds,
                         a->n_keys, d.V, a->g_attrs);
    }
    return E3K_OK;
  };
  auto weight_grads = [&](bool with_g_conv, bool with_lin1, void* st) -> int {
    Seg g;
    if (with_g_conv && need_post) g.add(L, POST_WGRAD, a->mid, a->gb_post, const_cast<float*>(a->g_conv), a->N);
    if (with_g_conv && want_sc) {
      // (have_m == 2: gm arrives zero-filled -- e3k_kw_stack's caller fills the gM of all its layers with one launch)
      if (a->have_m != 2 && e3k::zero_fill(a->gm, sizeof(float) * a->n_keys * d.ld_m, (hipStream_t)st)) return E3K_ERR_LAUNCH;
      g.add_keyed(L, SC_WGRAD, a->x_cf, a->gm, const_cast<float*>(a->g_conv), a->N, a->perm, a->bounds, a->n_keys);
    }
    if (with_lin1 && need_lin1) g.add(L, LIN1_WGRAD, a->x_cf, a->gb_lin1, a->g_x1, a->N);
    if (ABLATE & 2) return E3K_OK;
    return g.run(1, st);
  };
  // WGRAD_LATE: the weight-gradient GEMMs start BEHIND tp_bwd_x (all three Linears in one call) instead of beside it: the
  // gather-bound tp_bwd_x then runs alone, the MFMA-bound weight gradients beside the HBM-bound tp_bwd_w (256 molecules
  // 5.34 -> 5.30 ms, 192: 4.51 -> 4.47, one launch less per layer in round 3; with round 4's kernels the early start wins
  // again -- see the knobs' comment at the top -- and is the default)
  const bool wgrad_late = WGRAD_LATE && side3 != main && need_x1;
  if (side3 != main && (need_post || want_sc) && !wgrad_late) {
    E3K_TRY(edge(L, 0, main, side3));
    E3K_TRY(weight_grads(true, false, side3));
    if (want_sc) E3K_TRY(keyed_weight_grads());
  }
  // tensor product
  // packed table + both gradients wanted: ONE walk of the source CSR forms g_x1 and g_w [E, W] (csrc/e3k_tp.hip, MODE 5): the
  // weight-gradient pass re-gathered sh, x[src] and g_mid[dst] of every edge for a dot product with sums the input gradient
  // already holds (layer 3 of config_energy at 256 molecules: 213 + 104 us -> 258 us isolated; 217 + 169 -> 261 in the replayed step)
  const bool fused_xw = a->fuse_xw && need_x1 && in_kernel_table(d, r) && r.P && need_radial_side && a->E > 0 && a->g_w && a->x1 && !(ABLATE & 32);
  // ... the same for layers whose weights are streamed from w [E, W] (per-edge radial MLP, 32-channel plans), where the replayed /
  // one-stream step gains what the packed layers gain; with the backward forked over streams the separate weight-gradient pass
  // already runs beside the main stream's GEMMs, and stays
  const bool fused_xw_s = a->fuse_xw && !fused_xw && need_x1 && !in_kernel_table(d, r) && need_radial_side && a->E > 0 && a->g_w && a->x1 &&
                          r.w && side == main && !(ABLATE & 32);
  if (need_x1) {
    if (!a->g_x1) return E3K_ERR_INVALID;
    if (!d.tp_bwd_x_overwrites && e3k::zero_fill(a->g_x1, sizeof(float) * a->N * d.d_x1, (hipStream_t)main))
      return E3K_ERR_LAUNCH;
    Timed t(L, E3K_PROF_TP_BWD_X, main, a->N, a->E);
    if (ABLATE & 32) {
    } else if (fused_xw) {      // ... and the per-edge weight gradient in the same walk: no tp_bwd_w pass below
      E3K_TRY(e3k_tp_bwd_xw_ptable(d.tp, a->x1, r.P, r.erec_src, a->g_mid, a->src_ptr, a->N, a->E, a->g_x1, a->g_w, main));
    } else if (fused_xw_s) {
      E3K_TRY(e3k_tp_bwd_xw(d.tp, a->x1, a->sh, r.w, a->g_mid, a->dst, a->src_ptr, a->src_perm, a->N, a->E, a->g_x1, a->g_w, main));
    } else if (in_kernel_table(d, r) && r.P) {
      E3K_TRY(e3k_tp_bwd_x_ptable(d.tp, r.P, r.erec_src, a->g_mid, a->src_ptr, a->N, a->E, a->g_x1, main));
    } else if (in_kernel_table(d, r)) {
      E3K_TRY(e3k_tp_bwd_x_table(d.tp, a->sh, r.T, r.bin, r.bin_coef, a->g_mid, a->dst, a->src_ptr, a->src_perm, a->N, a->E, a->g_x1, main));
    } else {
      E3K_TRY(e3k_tp_bwd_x(d.tp, a->sh, r.w, a->g_mid, a->dst, a->src_ptr, a->src_perm, a->N, a->E, a->g_x1, main));
    }
  }
  if (wgrad_late && (need_post || want_sc || need_lin1)) {
    E3K_TRY(edge(L, 0, main, side3));
    E3K_TRY(weight_grads(true, true, side3));
    if (want_sc) E3K_TRY(keyed_weight_grads());
  }
  if (need_radial_side && a->E > 0) {
    if (!a->g_w) return E3K_ERR_INVALID;
    const float* g_rows = a->g_w;                 // gradient of the MLP's output rows: per edge, or per knot behind the table
    {
    // the weight-gradient pass: on the radial stream BEHIND tp_bwd_x (both are memory streams: side by side they only
    // stretch each other), where it runs beside the GEMMs that follow on the main stream (this layer's linear_1 dgrad, the
    // previous layer's gate' and post-TP dgrad) -- nothing on the main stream waits for it
    void* wst = (BWDW_SIDE && side != main && !fused_xw) ? side : main;
    if (wst != main) E3K_TRY(edge(L, 2, main, side));
    if (!fused_xw && !fused_xw_s) {
      Timed t(L, E3K_PROF_TP_BWD_W, wst, a->N, a->E);
      E3K_TRY(e3k_tp_bwd_w(d.tp, a->x1, a->sh, r.w, a->g_mid, a->src, a->dst_ptr, a->dst_perm, a->N, a->E, a->g_w, nullptr, wst));
    }
    if (wst == main) E3K_TRY(edge(L, 2, main, side));
    if (r.use_table) {
      Timed t(L, E3K_PROF_RTABLE_BWD, side, r.R, r.E);
      if (!(ABLATE & 8))
        E3K_TRY(e3k_rtable_interp_bwd(a->g_w, r.bin_coef, nullptr, r.bin_ptr, r.bin_seg, r.bin_perm, r.E, r.knots, d.W, a->table_ws,
                                      a->g_T, 0, side));
      g_rows = a->g_T;
    }
    }
    if (need_last && !(ABLATE & 16) && !r.have_rows) {
      Seg g;
      g.add(L, LAST_WGRAD, r.h, a->gb_last, const_cast<float*>(g_rows), r.R);
      E3K_TRY(g.run(1, side));
    }
    if ((need_hidden || a->need_radial) && !(ABLATE & 16) && !r.have_rows) {
      Seg g;
      g.add(L, LAST_DGRAD, g_rows, r.w_last, a->g_h, r.R);
      E3K_TRY(g.run(0, side));
      float* gws[4] = {a->gb_hidden[0], a->gb_hidden[1], a->gb_hidden[2], a->gb_hidden[3]};
      const float* zs[4] = {r.z[0], r.z[1], r.z[2], r.z[3]};
      E3K_TRY(e3k_mlp_hidden_bwd(r.radial, r.R, d.k0, d.h, d.n_hidden, r.w_hidden, d.alphas, d.act, d.cst, zs, a->g_h, gws,
                                 a->need_radial ? a->g_radial : nullptr, side));
    }
  }
  // linear_1's input gradient on top of the self-connection's
  if (a->need_x) {
    Seg g;
    g.add(L, has_sc ? LIN1_DGRAD_ACC : LIN1_DGRAD, a->g_x1, a->w_lin1, a->g_xcf, a->N);
    if (!(ABLATE & 64)) E3K_TRY(g.run(0, main));
    if (!a->in_cf && !L->in_blocks.empty()) {
      if (!a->g_x) return E3K_ERR_INVALID;
      E3K_TRY(e3k_relayout(a->g_xcf, a->N, d.d_in, L->in_blocks.data(), (int32_t)L->in_blocks.size(), 0, a->g_x, main));
    }
  }
  if (wgrad_late) {
  } else if (side3 != main) {
    if (need_lin1) {
      E3K_TRY(edge(L, 3, main, side3));
      E3K_TRY(weight_grads(false, true, side3));
    }
  } else if (need_post || need_lin1 || want_sc) {
    E3K_TRY(weight_grads(true, true, main));
    if (want_sc) E3K_TRY(keyed_weight_grads());
  }
  if (hipGetLastError() != hipSuccess) return E3K_ERR_LAUNCH;
  return E3K_OK;
}

// ---- the radial MLPs of several layers, batched ------------------------------------------------------------------------
extern "C" int e3k_radial_stack_fwd(const e3k_layer* const* layers, const e3k_layer_radial* rads, int32_t n, void* stream) {
  if (!layers || !rads || n <= 0 || n > 16) return E3K_ERR_INVALID;
  const e3k_layer_desc& d0 = layers[0]->d;
  e3k_mlp_net nets[16];
  for (int i = 0; i < n; ++i) {
    const e3k_layer_desc& d = layers[i]->d;
    const e3k_layer_radial& r = rads[i];
    if (d.k0 != d0.k0 || d.h != d0.h || d.n_hidden != d0.n_hidden || d.act != d0.act || d.cst != d0.cst) return E3K_ERR_UNSUPPORTED;
    for (int l = 0; l < d.n_hidden; ++l)
      if (d.alphas[l] != d0.alphas[l]) return E3K_ERR_UNSUPPORTED;
    if (r.R != rads[0].R || r.radial != rads[0].radial) return E3K_ERR_INVALID;
    e3k_mlp_net nt{};
    for (int l = 0; l < 4; ++l) {
      nt.weights[l] = r.w_hidden[l];
      nt.z[l] = r.keep ? r.z[l] : nullptr;
    }
    nt.out = r.h;
    nets[i] = nt;
  }
  if (rads[0].R == 0) return E3K_OK;
  if (!(ABLATE & 16)) {
    E3K_TRY(e3k_mlp_hidden_fwd_multi(nets, n, rads[0].radial, rads[0].R, d0.k0, d0.h, d0.n_hidden, d0.alphas, d0.act, d0.cst, stream));
    for (int base = 0; base < n; base += SEG_MAX) {      // last layers: one e3k_gemm_multi call per SEG_MAX layers
      Seg g;
      for (int i = base; i < n && i < base + SEG_MAX; ++i) {
        const e3k_layer_radial& r = rads[i];
        g.add(layers[i], LAST_FWD, r.h, r.w_last, r.use_table ? r.T : r.w, r.R);
      }
      Timed t(layers[base], E3K_PROF_RADIAL_LAST_FWD, stream, rads[0].R, rads[0].E);
      E3K_TRY(g.run(0, stream));
    }
    // the tables the tensor-product kernels will read packed (rads with in_kernel and P): all of them in one launch, behind the
    // last layers (round 6: five launches of 5-8 us in front of the five layers' forward otherwise)
    const float* tabs[16];
    void* packs[16];
    int32_t widths[16];
    int np = 0;
    for (int i = 0; i < n; ++i)
      if (rads[i].use_table && rads[i].in_kernel && rads[i].P && rads[i].T) {
        tabs[np] = rads[i].T;
        packs[np] = rads[i].P;
        widths[np++] = layers[i]->d.W;
      }
    if (np) E3K_TRY(e3k_rtable_pack_multi(tabs, (int32_t)(rads[0].R - 1), widths, packs, np, stream));
  }
  if (hipGetLastError() != hipSuccess) return E3K_ERR_LAUNCH;
  return E3K_OK;
}

extern "C" int e3k_radial_slope_fwd(const e3k_layer* const* layers, const e3k_layer_radial* rads, int32_t n, const e3k_slope_ctx* sl,
                                    float* const* hp, float* const* D, void* stream) {
  if (!layers || !rads || n <= 0 || n > 16 || !sl || !sl->knots || !sl->bessel_w || !hp || !D) return E3K_ERR_INVALID;
  const e3k_layer_desc& d0 = layers[0]->d;
  const int64_t R = rads[0].R;
  if (R <= 0) return E3K_OK;
  const float* wh[16 * 4] = {nullptr};
  for (int i = 0; i < n; ++i) {
    const e3k_layer_desc& d = layers[i]->d;
    if (d.k0 != d0.k0 || d.h != d0.h || d.n_hidden != d0.n_hidden || d.act != d0.act || d.cst != d0.cst) return E3K_ERR_UNSUPPORTED;
    for (int l = 0; l < d.n_hidden; ++l) {
      if (d.alphas[l] != d0.alphas[l]) return E3K_ERR_UNSUPPORTED;
      wh[i * 4 + l] = rads[i].w_hidden[l];
    }
    if (rads[i].R != R || !hp[i] || !D[i]) return E3K_ERR_INVALID;
  }
  E3K_TRY(e3k_slope_tangent_fwd(wh, n, d0.n_hidden, d0.alphas, sl->knots, R, sl->bessel_w, d0.k0, d0.h, sl->r_max, sl->r_min, sl->p,
                                sl->one_over_r, sl->cutoff_kind, d0.act, d0.cst, hp, stream));
  for (int base = 0; base < n; base += SEG_MAX) {      // D_l = H'_l W_last_l: the layers' last-layer GEMMs, SEG_MAX per call
    Seg g;
    for (int i = base; i < n && i < base + SEG_MAX; ++i) g.add(layers[i], LAST_FWD, hp[i], rads[i].w_last, D[i], R);
    E3K_TRY(g.run(0, stream));
  }
  if (hipGetLastError() != hipSuccess) return E3K_ERR_LAUNCH;
  return E3K_OK;
}

extern "C" int e3k_radial_stack_bwd(const e3k_layer* const* layers, const e3k_radial_stack_item* items, int32_t n,
                                    const e3k_slope_ctx* sl, void* stream) {
  if (!layers || !items || n <= 0 || n > 16) return E3K_ERR_INVALID;
  const e3k_layer_desc& d0 = layers[0]->d;
  const int64_t R = items[0].rad.R;
  if (R == 0) return E3K_OK;
  if (ABLATE & 16) return E3K_OK;
  e3k_mlp_net nets[16];
  int n_nets = 0;
  for (int base = 0; base < n; base += SEG_MAX) {        // weight gradients of the last layers
    Seg g;
    for (int i = base; i < n && i < base + SEG_MAX; ++i)
      if (items[i].gb_last && items[i].g_rows)
        g.add(layers[i], LAST_WGRAD, items[i].rad.h, items[i].gb_last, const_cast<float*>(items[i].g_rows), R);
    E3K_TRY(g.run(1, stream));
    Seg gs;                                        // ... and through the slope table: gb_last += H'^T g_D
    for (int i = base; i < n && i < base + SEG_MAX; ++i)
      if (items[i].gb_last && items[i].g_slope && items[i].hp)
        gs.add(layers[i], LAST_WGRAD, items[i].hp, items[i].gb_last, const_cast<float*>(items[i].g_slope), R);
    E3K_TRY(gs.run(1, stream));
  }
  const float* sl_w[16 * 4] = {nullptr};      // the nets whose slope table received a gradient: the float64 reverse sweep
  float* sl_g[16 * 4] = {nullptr};
  const float* sl_ghp[16];
  int n_adj = 0;
  for (int base = 0; base < n; base += SEG_MAX) {        // their input gradients, then the hidden chains
    Seg g, gsl;
    for (int i = base; i < n && i < base + SEG_MAX; ++i) {
      const e3k_radial_stack_item& it = items[i];
      bool need_hidden = it.g_radial != nullptr;
      for (int l = 0; l < d0.n_hidden; ++l) need_hidden = need_hidden || it.gb_hidden[l];
      if (!need_hidden || !it.g_rows) continue;
      if (!it.g_h) return E3K_ERR_INVALID;
      g.add(layers[i], LAST_DGRAD, it.g_rows, it.rad.w_last, it.g_h, R);
      if (it.g_slope) {
        if (!it.g_hp || !sl) return E3K_ERR_INVALID;
        for (int l = 0; l < 4; ++l) {
          sl_w[n_adj * 4 + l] = it.rad.w_hidden[l];
          sl_g[n_adj * 4 + l] = it.gb_hidden[l];
        }
        sl_ghp[n_adj++] = it.g_hp;
        gsl.add(layers[i], LAST_DGRAD, it.g_slope, it.rad.w_last, it.g_hp, R);
      }
      e3k_mlp_net nt{};
      for (int l = 0; l < 4; ++l) {
        nt.weights[l] = it.rad.w_hidden[l];
        nt.z[l] = it.rad.z[l];
        nt.g_weights[l] = it.gb_hidden[l];
      }
      nt.g_out = it.g_h;
      nt.g_x = it.g_radial;
      nets[n_nets++] = nt;
    }
    E3K_TRY(g.run(0, stream));
    E3K_TRY(gsl.run(0, stream));
  }
  if (n_adj) {    // the hidden weights' and Bessel frequencies' share of the slope tables' gradient (added into the same buffers)
    if (!sl->acc || !sl->knots || !sl->bessel_w) return E3K_ERR_INVALID;
    E3K_TRY(e3k_slope_tangent_bwd(sl_w, n_adj, d0.n_hidden, d0.alphas, sl->knots, R, sl->bessel_w, d0.k0, d0.h, sl->r_max, sl->r_min,
                                  sl->p, sl->one_over_r, sl->cutoff_kind, d0.act, d0.cst, sl_ghp, sl->acc, sl_g, sl->g_bessel, stream));
  }
  if (n_nets)
    E3K_TRY(e3k_mlp_hidden_bwd_multi(nets, n_nets, items[0].rad.radial, R, d0.k0, d0.h, d0.n_hidden, d0.alphas, d0.act, d0.cst, stream));
  if (hipGetLastError() != hipSuccess) return E3K_ERR_LAUNCH;
  return E3K_OK;
}

// ---- the per-key self-connection weights of several layers, batched ---------------------------------------------------
extern "C" int e3k_kw_stack_fwd(const e3k_layer* const* layers, const e3k_kw_stack_item* items, int32_t n, const float* node_attrs,
                                const int64_t* reps, int32_t n_keys, float* a_rep, void* stream) {
  if (!layers || !items || n <= 0 || n > 8 || !node_attrs || !reps || !a_rep || n_keys <= 0) return E3K_ERR_INVALID;
  const int V = layers[0]->d.V;
  e3k_kw_multi_item mi[8];
  for (int i = 0; i < n; ++i) {
    if (!layers[i]->kwa || layers[i]->d.V != V) return E3K_ERR_UNSUPPORTED;
    mi[i] = e3k_kw_multi_item{layers[i]->kwa, items[i].w_sc, items[i].m, nullptr, 0, 0};
  }
  const int tot = n_keys * V;
  hipLaunchKernelGGL(e3k::gather_rows_kernel, dim3((tot + 255) / 256), dim3(256), 0, (hipStream_t)stream, node_attrs, reps, n_keys, V,
                     a_rep);
  if (!(ABLATE & 1)) E3K_TRY(e3k_keyed_weights_fwd_multi(mi, n, a_rep, n_keys, stream));
  if (hipGetLastError() != hipSuccess) return E3K_ERR_LAUNCH;
  return E3K_OK;
}

extern "C" int64_t e3k_kw_stack_bwd_workspace(const e3k_layer* const* layers, int32_t n, int32_t n_keys) {
  if (!layers || n <= 0 || n > 8) return 0;
  e3k_kw_multi_item mi[8];
  float dummy = 0.f;
  for (int i = 0; i < n; ++i) {
    if (!layers[i]->kwa) return 0;
    mi[i] = e3k_kw_multi_item{layers[i]->kwa, &dummy, &dummy, nullptr, 0, 0};
  }
  return e3k_keyed_weights_bwd_multi_workspace(mi, n, n_keys);
}

/* items[i].m = g_M of layer i (written by e3k_layer_bwd with have_m), items[i].gb_sc / acc_sc = its weight-gradient buffer;
 * ga [n_keys, V] scratch, g_attrs [N, V] out (both NULL: the attributes need no gradient). */
extern "C" int e3k_kw_stack_bwd(const e3k_layer* const* layers, const e3k_kw_stack_item* items, int32_t n, const float* a_rep,
                                const int64_t* reps, const int32_t* bounds, int64_t N, int32_t n_keys, float* ga, float* g_attrs,
                                float* workspace, void* stream) {
  if (!layers || !items || n <= 0 || n > 8 || !a_rep || n_keys <= 0 || N < 0) return E3K_ERR_INVALID;
  if ((ga != nullptr) != (g_attrs != nullptr) || (g_attrs && (!reps || !bounds))) return E3K_ERR_INVALID;
  const int V = layers[0]->d.V;
  e3k_kw_multi_item mi[8];
  for (int i = 0; i < n; ++i) {
    if (!layers[i]->kwa || layers[i]->d.V != V) return E3K_ERR_UNSUPPORTED;
    mi[i] = e3k_kw_multi_item{layers[i]->kwa, items[i].w_sc, items[i].m, items[i].gb_sc, items[i].acc_sc, 0};
  }
  hipStream_t st = (hipStream_t)stream;
  // (ga directly in front of g_attrs in one allocation -- what the Python side hands over --: ONE fill for both)
  const bool together = ga && g_attrs == ga + (int64_t)n_keys * V;
  if (ga && e3k::zero_fill(ga, sizeof(float) * ((int64_t)n_keys * V + (together ? N * V : 0)), st)) return E3K_ERR_LAUNCH;
  if (!(ABLATE & 1)) E3K_TRY(e3k_keyed_weights_bwd_multi(mi, n, a_rep, n_keys, ga, workspace, stream));
  if (g_attrs) {
    if (!together && e3k::zero_fill(g_attrs, sizeof(float) * N * V, st)) return E3K_ERR_LAUNCH;
    const int tot = n_keys * V;
    hipLaunchKernelGGL(e3k::scatter_rows_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, ga, reps, bounds, n_keys, V, g_attrs);
  }
  if (hipGetLastError() != hipSuccess) return E3K_ERR_LAUNCH;
  return E3K_OK;
}
