/*
 * e3k.h — C ABI of libe3k.so, the MI355X (gfx950) kernels behind the e3_layers tensor-product
 * message-passing path.
 *
 * The reference (20171130/Equivariant-NN-Zoo) has no native plugin interface: its hot path is
 * Python that calls e3nn / torch_runstats operators.  Each entry point below
 * therefore cites the *operator call site* it replaces (paths relative to /root/reference).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (PyTorch on the Python side); the
 *    library allocates nothing except the opaque plans created by e3k_tp_plan_create;
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises;
 *  - return value: 0 = ok, <0 = error (see e3k_strerror); no exceptions cross the boundary;
 *  - floating tensors are fp32, node/edge ids are int32 inside the library (the int64
 *    `edge_index` of the reference's Batch is narrowed once per batch by the host shim);
 *  - feature layouts: "e3nn" = each `mul x l` block stored [mul][2l+1] (reference layout,
 *    README.md:108-110); "cf" = channel-fastest, block stored [2l+1][mul] (internal layout of
 *    the fused convolution: 64 channels = 64 lanes read 256 contiguous bytes).
 */
#ifndef E3K_H
#define E3K_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define E3K_OK 0
#define E3K_ERR_INVALID (-1)     /* bad argument / unsupported shape */
#define E3K_ERR_LAUNCH (-2)      /* hip launch or runtime error */
#define E3K_ERR_UNSUPPORTED (-3) /* degree beyond the generated CG tables */

#define E3K_TP_MAXQ 8 /* max (l2,l3) slots per input degree — must equal E3K_MAXQ of e3k_cg_gen.h */

const char* e3k_strerror(int code);
int e3k_version(void);
/* compile-time limits of the generated Clebsch-Gordan tables: l1max, l2max, l3max */
void e3k_tp_limits(int* l1max, int* l2max, int* l3max);

/* ------------------------------------------------------------------------------------------
 * Grouped strided GEMM on the f32 MFMA pipe (v_mfma_f32_32x32x2_f32).
 * Replaces: o3.Linear (nn/message_passing.py:58,102; nn/pointwise.py:18,87,142),
 *           e3nn.nn.FullyConnectedNet layers (nn/message_passing.py:74,93),
 *           o3.FullyConnectedTensorProduct with scalar second operand (nn/message_passing.py:83,100).
 *
 * One problem computes, for rows (r1, r2), r1 < M1, r2 < M2:
 *     C[r1, r2, n] = alpha * sum_k Aeff[r1, r2, k] * B[k, n]  (+ C if accumulate) (+ bias[n])
 * with   A addr = A + r1*a_r1 + r2*a_r2 + k*a_k,  B addr = B + k*b_k + n*b_n,
 *        C addr = C + r1*c_r1 + r2*c_r2 + n*c_n   (all strides in elements).
 * Outer mode (V > 0): K = U*V and Aeff[r1,r2,u*V+v] = A[r1,r2,u] * A2[r1*a2_r1 + v]
 * (the x (x) node_attrs operand of the self-connection, never materialised).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const float* A;
  const float* A2;
  const float* B;
  float* C;
  const float* bias;
  const int32_t* row_index; /* optional: r1 := row_index[r1] for the A, A2 and C rows (grouped GEMM over a
                               gathered subset of nodes); M1 = length of the list */
  const int32_t* group_dev; /* optional, with row_index: DEVICE pair {start, count}; the list is
                               row_index[start .. start+count) and M1 is only an upper bound of count
                               (grids are sized from M1, surplus workgroups exit) -- no host sync needed */
  int32_t M1, M2, N, K;
  int32_t V;
  int32_t accumulate;
  int64_t a_r1, a_r2, a_k;
  int64_t a2_r1;
  int64_t b_k, b_n;
  int64_t c_r1, c_r2, c_n;
  float alpha;
  int32_t act;   /* 0 none; 1: C = act_cst * ssp(alpha*A.B + bias) fused into the epilogue (radial MLP layers) */
  float act_cst; /* second-moment normalisation constant of the activation */
  int32_t chain; /* K-chain (e3k_gemm, e3k_gemm_multi; forward / input-gradient problems only): the `chain` problems that FOLLOW
                    this one in the array continue its K loop into the same accumulators,
                        C = alpha_0 A_0.B_0 + alpha_1 A_1.B_1 + ...  (+ C if the head accumulates) (+ the head's bias, activation),
                    in ONE pass over C instead of one accumulating launch per term -- e3nn's o3.Linear input gradient of an irrep
                    that feeds several outputs (e3_layers/nn/pointwise.py:87-92: `0e` feeds the scalars and the gates).  A follower
                    repeats the head's M1, M2, N, C, c_* strides, row_index / group_dev (keyed: the same groups) and has chain = 0,
                    V = 0, no bias; it brings A, B, K, their strides and alpha.  Every link runs on the plain kernel. */
} e3k_gemm_problem;

int e3k_gemm(const e3k_gemm_problem* problems, int n_problems, void* stream);

/* Weight gradient ("TN"):  B[k, n] += alpha * sum_{r1,r2} Aeff[r1,r2,k] * C[r1,r2,n]
 * (C is read as the incoming gradient, B is accumulated with fp32 atomics; the caller zeroes B). */
int e3k_gemm_wgrad(const e3k_gemm_problem* problems, int n_problems, void* stream);

/* Cached-descriptor launch: `templates` is an array the host built once per (layer, pass); its pointer fields hold
 * BYTE OFFSETS instead of addresses -- A, B, C relative to a_base, b_base, c_base; A2 and bias hold offset + 1
 * (0 = absent) relative to a2_base, bias_base -- and M1 is overwritten by the argument.  wgrad != 0 runs
 * e3k_gemm_wgrad semantics.  Same kernels, same results as e3k_gemm / e3k_gemm_wgrad on the resolved problems. */
int e3k_gemm_rebased(const e3k_gemm_problem* templates, int n_templates, const void* a_base, const void* a2_base,
                     const void* b_base, void* c_base, const void* bias_base, int64_t M1, int32_t wgrad,
                     void* stream);

/* Several descriptor arrays in ONE call (as few launches as the kernel kinds allow; up to 20 problems per launch, keyed
 * and plain problems mixed): what a convolution layer issues together -- linear_1 with the keyed self-connection
 * (e3_layers/nn/message_passing.py:100,102: both read the node features), the input gradients of the trailing Linear
 * and of the self-connection (both read the gradient of the convolution output), the three weight gradients.
 * A segment is a template array as for e3k_gemm_rebased (byte offsets in the pointer fields, M1 >= 0 overwrites the
 * templates' row count; M1 < 0: the templates carry addresses and row counts) and, with n_keys > 0, the key expansion
 * of e3k_gemm_grouped (row_index = perm, group_dev = groups_dev + 2 t, B += t * b_key_stride for key t).
 * At most 64 problems per call. */
typedef struct {
  const e3k_gemm_problem* templates;
  int32_t n_templates;
  int32_t n_keys;            /* 0: plain problems */
  const void* a_base;
  const void* a2_base;
  const void* b_base;
  void* c_base;
  const void* bias_base;
  int64_t M1;
  const int32_t* perm;
  const int32_t* groups_dev;
  int64_t b_key_stride;
} e3k_gemm_segment;
int e3k_gemm_multi(const e3k_gemm_segment* segments, int32_t n_segments, int32_t wgrad, void* stream);

/* Grouped launch over key groups: every template problem is expanded into n_keys problems, one per
 * key t, with  B += t * b_key_stride,  row_index = perm,  group_dev = groups_dev + 2*t  and M1 kept as the
 * (host-known) upper bound of the group size.  wgrad != 0 runs e3k_gemm_wgrad semantics (B accumulated).
 * Used by the keyed self-connection: rows = nodes sorted by the key of their attribute row. */
int e3k_gemm_grouped(const e3k_gemm_problem* templates, int n_templates, const int32_t* perm,
                     const int32_t* groups_dev, int32_t n_keys, int64_t b_key_stride, int32_t wgrad, void* stream);

/* e3k_gemm_grouped over cached descriptors: A, B, C of the templates hold byte offsets relative to the bases (as in
 * e3k_gemm_rebased; no A2 / bias in keyed problems), M1 is overwritten. */
int e3k_gemm_grouped_rebased(const e3k_gemm_problem* templates, int n_templates, const void* a_base,
                             const void* b_base, void* c_base, int64_t M1, const int32_t* perm,
                             const int32_t* groups_dev, int32_t n_keys, int64_t b_key_stride, int32_t wgrad,
                             void* stream);

/* Column sums: out[n] += sum_r G[r*ld + n]   (bias gradients) */
int e3k_colsum(const float* G, int64_t rows, int32_t cols, int64_t ld, float* out, void* stream);

/* Self-connection backward helper: H[(r1,r2),(u,v)] = dC . W^T was produced by e3k_gemm; this
 * reduces it to  dX[r1,r2,u] (+)= sum_v A2[r1,v] H[..,(u,v)]  and  dA2[r1,v] += sum_{r2,u} X[r1,r2,u] H[..].
 * X/dX addressed as x + r1*x_r1 + r2*x_r2 + u (cf layout). */
int e3k_fctp_reduce_bwd(const float* H, const float* X, const float* A2, int32_t M1, int32_t M2, int32_t U,
                        int32_t V, int64_t x_r1, int64_t x_r2, int64_t a2_r1, float* dX, int32_t dx_accumulate,
                        float* dA2, void* stream);

/* ------------------------------------------------------------------------------------------
 * Edge geometry.  Replaces computeEdgeVector (data/compute_edge.py:13-36),
 * o3.SphericalHarmonics via SphericalEncoding (nn/embedding.py:163-178) and
 * BesselBasis x cutoff via RadialBasisEncoding (nn/embedding.py:114-127,31-40,210-219).
 * ------------------------------------------------------------------------------------------ */
int e3k_edge_vector_fwd(const float* pos, const int32_t* src, const int32_t* dst, int64_t E, float* edge_vec,
                        float* edge_len, void* stream);
/* g_pos[n] = sum_{e: dst=n} gv[e] - sum_{e: src=n} gv[e], gv = g_vec + g_len * vec/len; CSR by dst and by src */
int e3k_edge_vector_bwd(const float* g_vec, const float* g_len, const float* edge_vec, const float* edge_len,
                        const int32_t* dst_ptr, const int32_t* dst_perm, const int32_t* src_ptr,
                        const int32_t* src_perm, int64_t N, float* g_pos, void* stream);

/* lmask: bit l set => degree l present (l <= 3), output blocks in ascending-l order of the set
 * bits, repeated per `ls` entry; normalization: 0 component, 1 integral, 2 norm. */
int e3k_sph_harm_fwd(const float* vec, int64_t E, const int32_t* ls, int32_t n_ls, int32_t normalize,
                     int32_t normalization, float* sh, void* stream);
int e3k_sph_harm_bwd(const float* vec, const float* g_sh, int64_t E, const int32_t* ls, int32_t n_ls,
                     int32_t normalize, int32_t normalization, float* g_vec, void* stream);
/* double backward: g_hat [E,3] is the cotangent of e3k_sph_harm_bwd's g_vec.
 * g_gsh [E,dim] = J(vec) g_hat;  g_vec [E,3] = d/dvec <g_hat, sph_harm_bwd(vec, g_sh)>  (either may be NULL). */
int e3k_sph_harm_bwd2(const float* vec, const float* g_sh, const float* g_hat, int64_t E, const int32_t* ls,
                      int32_t n_ls, int32_t normalize, int32_t normalization, float* g_gsh, float* g_vec,
                      void* stream);

/* cutoff_kind: 0 polynomial (p), 1 symmetric (x^2-1)^2 */
int e3k_radial_basis_fwd(const float* r, int64_t E, const float* bessel_w, int32_t n_basis, float r_max, float r_min,
                         float p, int32_t one_over_r, int32_t cutoff_kind, float* out, void* stream);
int e3k_radial_basis_bwd(const float* r, const float* g_out, int64_t E, const float* bessel_w, int32_t n_basis,
                         float r_max, float r_min, float p, int32_t one_over_r, int32_t cutoff_kind, float* g_r,
                         float* g_w, void* stream);
/* double backward: hat_r [E] / hat_w [n_basis] are the cotangents of e3k_radial_basis_bwd's g_r / g_w (either
 * may be NULL = zero).  g_gout [E,n_basis] = d out along (hat_r, hat_w); g_r [E] written, g_w [n_basis]
 * ACCUMULATED (caller zeroes): the second derivatives contracted with g_out.  Outputs may be NULL. */
int e3k_radial_basis_bwd2(const float* r, const float* g_out, const float* hat_r, const float* hat_w, int64_t E,
                          const float* bessel_w, int32_t n_basis, float r_max, float r_min, float p,
                          int32_t one_over_r, int32_t cutoff_kind, float* g_gout, float* g_r, float* g_w,
                          void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused 'uvu' tensor product + destination reduce.
 * Replaces TensorProductExpansion.tp (nn/pointwise.py:78-85,94-99) applied to x[edge_src]
 * (nn/message_passing.py:104-106) followed by scatter over edge_dst (:109).  The trailing
 * per-edge o3.Linear of the reference (nn/pointwise.py:87-92,99) commutes with the sum and is
 * applied on nodes by e3k_gemm.
 *
 * A group = one input irrep block (degree l1, `mul` channels at x_off in the cf row) and up
 * to E3K_TP_MAXQ paths, at most one per (l2, l3) slot; slot q is the q-th valid (l2,l3) pair
 * for l1 in the order of e3k_cg_gen.h (l2 ascending, then l3 ascending).
 *   out[n, out_off[q] + k*out_stride[q] + u] =
 *       sum_{e: dst(e)=n} coeff[q] * w[e, w_off[q]+u] * sum_ij C_ijk x[src(e), x_off + i*mul + u] sh[e, y_off[l2]+j]
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t l1, x_off, mul;
  uint32_t mask; /* bit q set => slot q present */
  int32_t y_off[4];
  int32_t w_off[E3K_TP_MAXQ];
  int32_t out_off[E3K_TP_MAXQ];
  int32_t out_stride[E3K_TP_MAXQ];
  float coeff[E3K_TP_MAXQ];
} e3k_tp_group;

typedef struct e3k_tp_plan e3k_tp_plan;

int e3k_tp_plan_create(const e3k_tp_group* groups, int32_t n_groups, int32_t d_in, int32_t d_sh, int32_t w_numel,
                       int32_t d_mid, e3k_tp_plan** plan);
void e3k_tp_plan_destroy(e3k_tp_plan* plan);

/* x [N,d_in] cf, sh [E,d_sh], w [E,w_numel]; src [E]; CSR by destination (dst_ptr [N+1],
 * dst_perm [E] = edge ids grouped by destination, ascending inside a group); out [N,d_mid]. */
int e3k_tp_fwd(const e3k_tp_plan* plan, const float* x, const float* sh, const float* w, const int32_t* src,
               const int32_t* dst_ptr, const int32_t* dst_perm, int64_t N, int64_t E, float* out, void* stream);
/* g_w [E,w_numel] written (may be NULL when only g_sh is wanted); g_sh [E,d_sh] accumulated with atomics when
 * non-null (needs w). */
int e3k_tp_bwd_w(const e3k_tp_plan* plan, const float* x, const float* sh, const float* w, const float* g_out,
                 const int32_t* src, const int32_t* dst_ptr, const int32_t* dst_perm, int64_t N, int64_t E,
                 float* g_w, float* g_sh, void* stream);
/* g_x [N,d_in]: the CALLER ZERO-FILLS it; plans whose groups are walked by two waves (l_max = 3: more than 24
 * accumulators per group) add their two partial sums with atomics (order independent), the others store.
 * CSR by source (src_ptr, src_perm), dst [E]. */
/* 1 when e3k_tp_bwd_x stores every element of g_x (no zero-fill needed): single-wave groups that tile [0, d_in). */
int e3k_tp_bwd_x_overwrites(const e3k_tp_plan* plan);

/* The forward and the node-feature backward with the per-edge path weights interpolated INSIDE the kernel from the radial
 * knot table (the weights `self.fc(edge_radial)` of nn/message_passing.py:93, never materialised as [E, W]):
 * T [K + 1, W] = the radial MLP on the knots, bin [E] = each edge's knot i (stencil rows i - 1 .. i + 2), coef [E, 4] = its
 * four interpolation weights (e3k_rtable_bins).
 * Same results as e3k_rtable_interp_fwd followed by e3k_tp_fwd / e3k_tp_bwd_x (same interpolation arithmetic).
 * e3k_tp_table_supported: 1 when the plan has this form (channel-complete plans: every group a multiple of 64 channels with
 * all its degree slots present -- the inner layers of every shipped model), else the two entry points return
 * E3K_ERR_UNSUPPORTED. */
int e3k_tp_table_supported(const e3k_tp_plan* plan);
int e3k_tp_fwd_table(const e3k_tp_plan* plan, const float* x, const float* sh, const float* T, const int32_t* bin,
                     const float* coef, const int32_t* src, const int32_t* dst_ptr, const int32_t* dst_perm, int64_t N, int64_t E,
                     float* out, void* stream);
int e3k_tp_bwd_x_table(const e3k_tp_plan* plan, const float* sh, const float* T, const int32_t* bin, const float* coef,
                       const float* g_out, const int32_t* dst, const int32_t* src_ptr, const int32_t* src_perm, int64_t N,
                       int64_t E, float* g_x, void* stream);
/* ... and from the PACKED table P [K + 1, 3 W] (e3k_rtable_pack) -- same plans, same results up to the fp16 rounding stated there --
 * walking EDGE RECORDS (e3k_edge_records): erec_dst over the destination CSR (dst_perm, src), erec_src over the source CSR
 * (src_perm, dst); each [E, 16] int32, 64-byte aligned. */
int e3k_tp_fwd_ptable(const e3k_tp_plan* plan, const float* x, const void* P, const int32_t* erec_dst, const int32_t* dst_ptr,
                      int64_t N, int64_t E, float* out, void* stream);
int e3k_tp_bwd_x_ptable(const e3k_tp_plan* plan, const void* P, const int32_t* erec_src, const float* g_out, const int32_t* src_ptr,
                        int64_t N, int64_t E, float* g_x, void* stream);
/* ... the input gradient AND every edge's weight gradient in the one walk (replaces e3k_tp_bwd_x_ptable + e3k_tp_bwd_w of a layer's
 * backward, nn/message_passing.py:93,104-109 under autograd): x [N, d_in] channel-fastest = the layer's tensor-product input rows,
 * g_w [E, W] row e = d F / d w[e] (written once each: no zero-fill).  g_x carries the bits of e3k_tp_bwd_x_ptable. */
int e3k_tp_bwd_xw_ptable(const e3k_tp_plan* plan, const float* x, const void* P, const int32_t* erec_src, const float* g_out,
                         const int32_t* src_ptr, int64_t N, int64_t E, float* g_x, float* g_w, void* stream);
/* ... the same pair of gradients with the weights streamed from w [E, W] (every plan): replaces e3k_tp_bwd_x + e3k_tp_bwd_w where
 * both are wanted (layers whose radial MLP runs per edge; force training's materialised rows). */
int e3k_tp_bwd_xw(const e3k_tp_plan* plan, const float* x, const float* sh, const float* w, const float* g_out, const int32_t* dst,
                  const int32_t* src_ptr, const int32_t* src_perm, int64_t N, int64_t E, float* g_x, float* g_w, void* stream);
/* Force training on the table (GradientOutput: nn/output.py:31-53 with create_graph = self.training; the per-edge weights
 * then depend on pos through the radius, nn/message_passing.py:93).  With F = <g, TP(x[src], sh, w(T, coef))>, linear in each
 * of (g, x, sh, T, coef), every first and second derivative is one of the walks below (plans with e3k_tp_table2_supported:
 * channel-complete, one wave per group -- the l_max <= 2 models):
 * D [K + 1, W] is the SLOPE table dT/dr on the knots (e3k_radial_slope_fwd), interpolated with the same weights:
 * dw/dr[e] = sum_k coef[e,k] D[bin[e] - 1 + k]  (differentiating the weights instead would amplify the table's fp32 rounding by
 * 1 / knot spacing: measured 7e-6 .. 4e-5 relative slope error at any knot count, against 5e-8 this way).
 *   e3k_tp_bwd_e_table      g_sh [E, d_sh] = dF/dsh, g_r [E] = <dF/dw, dw/dr> (either may be NULL), g_w [E, W] = dF/dw written when
 *                           non-null -- the first backward of a force evaluation.  A work item (node, group) holds only its group's
 *                           share of an edge's sum: with e_partials (e3k_tp_edge_partials_floats(plan, E) floats of scratch) every
 *                           item STORES its share into its own slice and a second launch adds the slices in item order into g_sh /
 *                           g_r (overwritten: no zero fill, bit-reproducible -- round 6); with e_partials = NULL the shares are
 *                           float atomics onto g_sh / g_r, which the caller zero-fills (not reproducible run to run)
 *   e3k_tp_fwd_jvp_table    out = TP(x2, sh, w) + TP(x, sh2, w) + TP(x, sh, s2 * dw/dr)          (s2 [E]: the radius' partner)
 *   e3k_tp_bwd_x_dual_table g_x = dF/dx at (sh2, w) + dF/dx at (sh, s2 * dw/dr)
 *   e3k_tp_bwd_w_dual       g_w = dF/dw at (x2, sh) + dF/dw at (x, sh2)           (w is the open slot: no table involved)
 * With bin = coef = NULL the first three take w and dw/dr MATERIALISED: T = w [E, W], D = dw/dr [E, W] (e3k_rtable_interp_fwd of
 * either table): five or six kernels per layer read them, and one 7.7 KB row per edge is a quarter of four table rows. */
int e3k_tp_table2_supported(const e3k_tp_plan* plan);
/* ... and with the weights and their slope passed MATERIALISED (bin = coef = NULL): every channel-complete plan, the l_max 3 ones
 * (two waves per group) included */
int e3k_tp_second_order_streamed_supported(const e3k_tp_plan* plan);
int e3k_tp_bwd_e_table(const e3k_tp_plan* plan, const float* x, const float* sh, const float* T, const float* D,
                       const int32_t* bin, const float* coef, const float* g_out, const int32_t* src, const int32_t* dst_ptr,
                       const int32_t* dst_perm, int64_t N, int64_t E, float* g_sh, float* g_r, float* g_w, float* e_partials,
                       void* stream);
int64_t e3k_tp_edge_partials_floats(const e3k_tp_plan* plan, int64_t E);
int e3k_tp_fwd_jvp_table(const e3k_tp_plan* plan, const float* x, const float* x2, const float* sh, const float* sh2,
                         const float* T, const float* D, const int32_t* bin, const float* coef, const float* s2,
                         const int32_t* src, const int32_t* dst_ptr, const int32_t* dst_perm, int64_t N, int64_t E, float* out,
                         void* stream);
int e3k_tp_bwd_x_dual_table(const e3k_tp_plan* plan, const float* sh, const float* sh2, const float* T, const float* D,
                            const int32_t* bin, const float* coef, const float* s2, const float* g_out, const int32_t* dst,
                            const int32_t* src_ptr, const int32_t* src_perm, int64_t N, int64_t E, float* g_x, void* stream);
/* The first backward of a force evaluation w.r.t. everything an edge touches, in ONE walk of the source CSR (streamed rows w,
 * dw [E, W]): g_x (e3k_tp_bwd_x), g_sh [E, d_sh] and g_r [E] (as e3k_tp_bwd_e_table, e_partials included; either may be NULL, not
 * both) and, when g_w != NULL, the per-edge weight gradient (e3k_tp_bwd_w).  Channel-complete plans. */
int e3k_tp_bwd_xe(const e3k_tp_plan* plan, const float* x, const float* sh, const float* w, const float* dw, const float* g_out,
                  const int32_t* dst, const int32_t* src_ptr, const int32_t* src_perm, int64_t N, int64_t E, float* g_x, float* g_sh,
                  float* g_r, float* g_w, float* e_partials, void* stream);
/* e3k_tp_bwd_x_dual_table on streamed rows (w, dw [E, W]; bin = coef = NULL there) that ALSO writes the weight gradients sharing its
 * per-edge sums: g_w = e3k_tp_bwd_w_dual's, g_w_plain (may be NULL) = e3k_tp_bwd_w's -- one walk instead of three (the u-sweep of
 * force training: the adjoint of nn/output.py:42-50's first backward) */
int e3k_tp_bwd_xw_dual(const e3k_tp_plan* plan, const float* x, const float* x2, const float* sh, const float* sh2, const float* w,
                       const float* dw, const float* s2, const float* g_out, const int32_t* dst, const int32_t* src_ptr,
                       const int32_t* src_perm, int64_t N, int64_t E, float* g_x, float* g_w, float* g_w_plain, void* stream);
int e3k_tp_bwd_w_dual(const e3k_tp_plan* plan, const float* x, const float* x2, const float* sh, const float* sh2,
                      const float* g_out, const int32_t* src, const int32_t* dst_ptr, const int32_t* dst_perm, int64_t N, int64_t E,
                      float* g_w, void* stream);
int e3k_tp_bwd_x(const e3k_tp_plan* plan, const float* sh, const float* w, const float* g_out, const int32_t* dst,
                 const int32_t* src_ptr, const int32_t* src_perm, int64_t N, int64_t E, float* g_x, void* stream);

/* ------------------------------------------------------------------------------------------
 * Per-batch graph topology (csrc/e3k_graph.hip).
 * Replaces what the reference leaves to scatter(ef, edge_dst) (nn/message_passing.py:109) and Batch's segment
 * bookkeeping (data/batch.py:164-178): from edge_index int64 [2,E] (row 0 sources, row 1 destinations) builds
 *   src, dst [E] int32; dst_ptr/src_ptr [N+1] row pointers; dst_perm/src_perm [E] edge ids grouped by endpoint,
 *   ASCENDING inside a row (= a stable sort by endpoint, the reference's CPU summation order);
 * workspace: e3k_csr_workspace_ints(N, E) int32.  bad_flag [1]: set to 1 when an endpoint is outside [0, N)
 * (such edges are attached to node 0; the caller reads the flag when it chooses to).
 * ------------------------------------------------------------------------------------------ */
int64_t e3k_csr_workspace_ints(int64_t N, int64_t E);
int e3k_csr_build(const int64_t* edge_index, int64_t N, int64_t E, int32_t* src, int32_t* dst, int32_t* dst_ptr,
                  int32_t* dst_perm, int32_t* src_ptr, int32_t* src_perm, int32_t* workspace, int32_t* bad_flag,
                  void* stream);

/* Edge records: what a tensor-product kernel needs about the t-th edge of a CSR walk (perm [E]: the walk's edge ids, nbr [E]: the
 * neighbour node of an edge -- src for the destination CSR, dst for the source CSR) as ONE 64-byte block per edge, in walk order:
 *   rec[t] = { nbr[e], bin[e], coef[e, 0..3], sh[e, 0..8], e },  e = perm[t]        (bin / coef NULL: zeros; d_sh < 9: zero padded)
 * replacing the chain perm[t] -> e -> {nbr, bin, coef, sh}[e] of dependent scalar loads in front of every edge's row loads
 * (reference: what `x[edge_src]`, `edge_spherical[e]`, `weight[e]` index, nn/message_passing.py:96-109).  rec: 64-byte aligned. */
int e3k_edge_records(const int32_t* perm, const int32_t* nbr, const int32_t* bin, const float* coef, const float* sh, int32_t d_sh,
                     int64_t E, int32_t* rec, void* stream);

/* Rows grouped by a small categorical key (the keyed self-connection groups nodes by species: node_attrs =
 * Linear(one_hot(species)), layer_configs.py:104-118 feeding nn/message_passing.py:81-87,100): perm [R] int32 = row ids
 * sorted by key, stable; bounds [K,2] int32 = {start, count} per key; reps [K] int64 = first row of each key (0 for an
 * absent key).  K <= 256.  bad_flag [1]: set when a key is outside [0, K). */
int e3k_group_rows(const int64_t* key, int64_t R, int32_t K, int32_t* perm, int32_t* bounds, int64_t* reps,
                   int32_t* bad_flag, void* stream);

/* ------------------------------------------------------------------------------------------
 * Radial weights through a knot table (csrc/e3k_rtable.hip).
 * Replaces weight = fc(edge_radial) (nn/message_passing.py:74-79,93) evaluated per edge when edge_radial is
 * RadialBasisEncoding(edge_length) (nn/embedding.py:210-219): the MLP is evaluated on the K + 1 knots j * r_max / K
 * (T [K+1, W], by the caller, with the same kernels) and every edge interpolates the FOUR knots around it (cubic Lagrange
 * weights: error 3/128 h^4 max|d4f|; ~1e-7 relative at K = 512 for the shipped models, the fp32 rounding of the table).
 *   e3k_rtable_bins       r [E], h_inv = 1 / knot spacing (a power of two makes r / h and the offsets exact; the knots are then
 *                         exactly representable too) -> bin [E] (knot i = floor(r / h) clamped to [1, K - 2]; stencil rows
 *                         i - 1 .. i + 2),
 *                         coef [E, 4] (the four weights), and the edges grouped
 *                         by knot with a STABLE counting sort: bin_ptr [K + 2], bin_perm [E] (ascending edge id inside a knot),
 *                         bin_seg [K + 2] (first segment of <= 64 edges of every knot: the transpose's work list).
 *                         workspace: e3k_rtable_bins_workspace_ints(E, K) int32.  Three launches, no atomics, no sort.
 *   e3k_rtable_interp_fwd w[e,:] = sum_k coef[e,k] T[bin[e] - 1 + k,:]     (edges visited in knot order, bin_perm, so that
 *                         neighbouring waves share table rows)
 *   e3k_rtable_interp_bwd g_T[j,:] (+)= sum over the edges whose stencil holds row j of their weight for it (times scale[e] when
 *                         scale is given) times g_w[e,:]; every row written; no atomics: per-segment partial sums in the
 *                         workspace, then a fixed-order combination (bit-identical run to run).
 * W must be a multiple of 4.
 * ------------------------------------------------------------------------------------------ */
int64_t e3k_rtable_bins_workspace_ints(int64_t E, int32_t K);
int e3k_rtable_bins(const float* r, int64_t E, float h_inv, int32_t K, int32_t* bin, float* coef,
                    int32_t* bin_ptr, int32_t* bin_seg, int32_t* bin_perm, int32_t* workspace, void* stream);
/* KEYED tables: the edge embedding is a function of the radius and of a small categorical key per edge (config_diffusion.py:73-82:
 * the Bessel basis concatenated with a 4-way bond-type one-hot) -- n_keys tables of K + 1 rows stacked into one of n_keys (K + 1)
 * rows, bin[e] = key[e] (K + 1) + i.  bin_ptr / bin_seg [n_keys (K + 1) + 1]; workspace e3k_rtable_bins_workspace_ints(E,
 * n_keys (K + 1) - 1).  Every consumer takes the stacked table with K := n_keys (K + 1) - 1.
 * bad_flag (may be NULL): bit 3 is ORed in when a key lies outside [0, n_keys) (such an edge interpolates in block 0); the flag is
 * never cleared here, so it may be a persistent one shared with other checks (e3k_flag_fetch_clear hands it to the host). */
int e3k_rtable_bins_keyed(const float* r, const int64_t* key, int32_t n_keys, int64_t E, float h_inv, int32_t K, int32_t* bin,
                          float* coef, int32_t* bin_ptr, int32_t* bin_seg, int32_t* bin_perm, int32_t* workspace, int32_t* bad_flag,
                          void* stream);
int e3k_rtable_interp_fwd(const float* T, const int32_t* bin_perm, const int32_t* bin, const float* coef, int64_t E, int32_t K,
                          int32_t W, float* w, void* stream);
/* two tables of one shape through the same weights in one pass (force training: w from T and dw/dr from the slope table) */
int e3k_rtable_interp_fwd2(const float* T, const float* T2, const int32_t* bin_perm, const int32_t* bin, const float* coef, int64_t E,
                           int32_t K, int32_t W, float* w, float* w2, void* stream);
/* workspace: e3k_rtable_bwd_workspace_floats(E, K, W) floats (per-segment partial sums, combined in a fixed order) */
int64_t e3k_rtable_bwd_workspace_floats(int64_t E, int32_t K, int32_t W);
int e3k_rtable_interp_bwd(const float* g_w, const float* coef, const float* scale, const int32_t* bin_ptr,
                          const int32_t* bin_seg, const int32_t* bin_perm, int64_t E, int32_t K, int32_t W, float* workspace,
                          float* g_T, int32_t accumulate, void* stream);

/* The table packed for the tensor-product kernels: 12 bytes per (knot, weight) in ONE row -- the cubic through rows i-1 .. i+2 as
 * its Taylor polynomial about the middle of knot interval i, {d0: f32, d1: f32, d2 * 2^10: f16, d3 * 2^16: f16} -- instead of four
 * 4-byte values in four rows 4 W bytes apart: e3k_tp_fwd_ptable / e3k_tp_bwd_x_ptable read one dwordx2 and one dword per path slot
 * and edge (23 instead of 31 KB of table per edge through L1).  P [K + 1, 3 W] dwords: a row = its W (d0, d1) pairs, then its W
 * f16 pairs.  Same function as the four-row form up to the fp16
 * rounding of the two small coefficients (economised into the fp32 pair: <= 2^-11 (|d2| / 8 + |d3| / 32), ~6e-8 of the values on
 * 512 knots).
 * e3k_rtable_interp_packed materialises w[e, :] from P with the kernels' own arithmetic (bit-identical; tests). */
int e3k_rtable_pack(const float* T, int32_t K, int32_t W, void* P, void* stream);
/* ... n <= 16 tables of one row count (the layers of a radial stack) in ONE launch; the same bits as n e3k_rtable_pack calls */
int e3k_rtable_pack_multi(const float* const* T, int32_t K, const int32_t* W, void* const* P, int32_t n, void* stream);
int e3k_rtable_interp_packed(const void* P, const int32_t* bin_perm, const int32_t* bin, const float* coef, int64_t E, int32_t K,
                             int32_t W, float* w, void* stream);

/* A-posteriori bound of the table's interpolation error, evaluated ON THE DEVICE (the reference evaluates fc(edge_radial)
 * exactly on every edge, nn/message_passing.py:74-79,93: the table must know when it stops being a stand-in).  With
 * err_c = 3/128 max_i |4th difference of T[:, c] at i| (cubic Lagrange on the table's knots; packed != 0: plus the fp16 rounding of
 * the packed table's two small coefficients, 2^-11 (|2nd difference| / 16 + |3rd difference| / 192)):
 *     table-wide ratio  est_g = max_c err_c / max |T|
 *     per-column ratio  est_c = max_c err_c / max(max_i |T[i, c]|, floor_rel * max |T|)
 *     reported          est   = max(est_g, col_weight * est_c)        (col_weight = table-wide tolerance / per-column tolerance)
 * for up to 16 tables of `rows` rows in ONE launch (the layers of a radial stack).  states[t] float [4]: [0] running maximum of
 * est since the caller last zeroed it (atomic max: survives HIP-graph replays, which never re-enter the host code that would
 * read a per-launch value), [1] est of this launch, [2] internal ticket counter (zero it once at allocation), [3] est_c of this
 * launch.  scratch[t] float [16 * widths[t]].  A non-finite table entry gives est = +inf.  rows < 5: nothing to do. */
int e3k_rtable_guard(const float* const* tables, float* const* states, float* const* scratch, const int32_t* widths, int32_t n,
                     int32_t rows, float floor_rel, float col_weight, int32_t packed, void* stream);

/* ------------------------------------------------------------------------------------------
 * Node-side elementwise kernels.
 * ------------------------------------------------------------------------------------------ */
/* activation ids: 0 identity, 1 ssp, 2 silu, 3 tanhlu, 4 tanh, 5 abs */
int e3k_act_fwd(const float* x, int64_t n, int32_t act, float cst, float* y, void* stream);
int e3k_act_bwd(const float* x, const float* g_y, int64_t n, int32_t act, float cst, float* g_x, void* stream);
/* same derivative from the OUTPUT y = cst*act(x) (act = 1 ssp only: cst*sigmoid(x) = cst*(1 - 0.5*exp(-y/cst))),
 * so a fused linear+activation layer keeps only its output */
int e3k_act_bwd_from_output(const float* y, const float* g_y, int64_t n, int32_t act, float cst, float* g_x,
                            void* stream);
/* double backward (GradientOutput with create_graph=True, e3_layers/nn/output.py:42-50 — force training):
 * e3k_act_bwd computed g_x = g_y*cst*act'(x); given the cotangent g_hat of that g_x,
 *   g_gy = g_hat*cst*act'(x)   and   g_x = g_hat*g_y*cst*act''(x)   (either output may be NULL). */
int e3k_act_bwd2(const float* x, const float* g_y, const float* g_hat, int64_t n, int32_t act, float cst,
                 float* g_gy, float* g_x, void* stream);

/* layout change of a feature row: blocks (off, mul, dim=2l+1); to_cf=1: [mul][dim] -> [dim][mul]. */
typedef struct {
  int32_t off, mul, dim, _pad;
} e3k_block;
int e3k_relayout(const float* x, int64_t rows, int32_t row_dim, const e3k_block* blocks, int32_t n_blocks,
                 int32_t to_cf, float* y, void* stream);

/* Gate (e3nn.nn.Gate, nn/message_passing.py:195-205,249): input row in cf layout
 * [scalars | gates | gated blocks], output row in e3nn layout [act(scalars) | gated*act(gate)]. */
typedef struct {
  int32_t kind;    /* 0: scalar block with activation, 1: gated block */
  int32_t in_off;  /* offset of the block in the input row */
  int32_t gate_off;/* kind 1: offset of its gate scalars in the input row */
  int32_t out_off; /* offset in the output row */
  int32_t mul, dim;
  int32_t act;     /* activation of the scalars (kind 0) or of the gates (kind 1) */
  float cst;       /* its second-moment normalisation constant */
} e3k_gate_seg;
/* out_cf: layout of the OUTPUT row (y, g_y, g_y2, g_gy): 0 = e3nn blocks [mul][2l+1], 1 = channel-fastest [2l+1][mul]
 * (consecutive MessagePassing layers hand their features over in cf: no relayout between them).  The input row x is
 * always channel-fastest.  g_y2 (may be NULL): a second addend of the incoming gradient, summed inside the kernel. */
int e3k_gate_fwd(const float* x, int64_t rows, int32_t in_dim, int32_t out_dim, const e3k_gate_seg* segs,
                 int32_t n_segs, int32_t out_cf, float* y, void* stream);
int e3k_gate_bwd(const float* x, const float* g_y, const float* g_y2, int64_t rows, int32_t in_dim, int32_t out_dim,
                 const e3k_gate_seg* segs, int32_t n_segs, int32_t out_cf, float* g_x, void* stream);
/* double backward of the gate: g_hat [rows,in_dim] is the cotangent of e3k_gate_bwd's g_x;
 * g_gy [rows,out_dim] = (dy/dx) g_hat,  g_x [rows,in_dim] = d/dx <g_hat, gate_bwd(x, g_y)>  (either may be NULL). */
int e3k_gate_bwd2(const float* x, const float* g_y, const float* g_hat, int64_t rows, int32_t in_dim, int32_t out_dim,
                  const e3k_gate_seg* segs, int32_t n_segs, int32_t out_cf, float* g_gy, float* g_x, void* stream);

/* NormActivation (e3nn.nn.NormActivation as built at e3_layers/nn/message_passing.py:212-219; the 'norm'
 * nonlinearity_type of MessagePassing): per irrep channel n2 = max(sum_m x_m^2, epsilon^2), n = sqrt(n2),
 * y_m = x_m * act(n) / n (normalize = 1).  blocks cover the row (uncovered columns come out zero); input
 * channel-fastest [2l+1][mul], output e3nn layout [mul][2l+1] at the same offsets; act ids as e3k_act_fwd, raw
 * (no second-moment constant). */
int e3k_norm_act_fwd(const float* x, int64_t rows, int32_t row_dim, const e3k_block* blocks, int32_t n_blocks, int32_t act,
                     float epsilon, int32_t normalize, float* y, void* stream);
int e3k_norm_act_bwd(const float* x, const float* g_y, int64_t rows, int32_t row_dim, const e3k_block* blocks,
                     int32_t n_blocks, int32_t act, float epsilon, int32_t normalize, float* g_x, void* stream);
/* backward of e3k_norm_act_bwd (force training through a 'norm' nonlinearity: GradientOutput, create_graph=True): with the
 * cotangent h on g_x (layout of x), g_gy (layout of g_y) and g_x; either may be NULL. */
int e3k_norm_act_bwd2(const float* x, const float* g_y, const float* h, int64_t rows, int32_t row_dim, const e3k_block* blocks,
                      int32_t n_blocks, int32_t act, float epsilon, int32_t normalize, float* g_gy, float* g_x, void* stream);

/* per-irreps-block RMS normalisation (LayerNormalization, nn/pointwise.py:32-51), e3nn layout */
int e3k_layernorm_fwd(const float* x, int64_t rows, int32_t row_dim, const e3k_block* blocks, int32_t n_blocks,
                      const float* std, float* y, float* inv_norm, void* stream);
int e3k_layernorm_bwd(const float* x, const float* g_y, const float* inv_norm, int64_t rows, int32_t row_dim,
                      const e3k_block* blocks, int32_t n_blocks, const float* std, float* g_x, float* g_std,
                      void* stream);
/* double backward: h [rows,row_dim] is the cotangent of e3k_layernorm_bwd's g_x, h_std [n_blocks] that of its g_std (may
 * be NULL); g_gy, g_x [rows,row_dim] written on the block columns (the caller zero-fills uncovered columns), g_std
 * [n_blocks] ACCUMULATED (any of the three may be NULL). */
int e3k_layernorm_bwd2(const float* x, const float* g_y, const float* h, const float* h_std, const float* inv_norm,
                       int64_t rows, int32_t row_dim, const e3k_block* blocks, int32_t n_blocks, const float* std,
                       float* g_gy, float* g_x, float* g_std, void* stream);

/* segment sizes -> row pointers (what Pooling derives from the batch's per-graph node counts, nn/output.py:66-74):
 * ptr[0] = 0, ptr[s + 1] = counts[0] + .. + counts[s]; ptr has n_seg + 1 entries */
int e3k_counts_to_ptr(const int64_t* counts, int32_t n_seg, int32_t* ptr, void* stream);
/* one-hot rows of a type index (OneHotEncoding, nn/embedding.py:271-281: torch.nn.functional.one_hot(...).to(float)):
 * out [rows, num_types], out[r, t] = (idx[r] == t); an index outside [0, num_types) gives a zero row AND ORs bit 2 into
 * *bad_flag (may be NULL; never cleared here: the reference's one_hot raises on such an index) */
int e3k_onehot(const int64_t* idx, int64_t rows, int32_t num_types, float* out, int32_t* bad_flag, void* stream);
/* *host_out = *flag; *flag = 0 -- one launch, in stream order (host_out: pinned, device-visible host memory).  How a persistent
 * device error flag that kernels only ever OR into (captured index checks, the two above) reaches the host exactly once per
 * flagged batch. */
int e3k_flag_fetch_clear(int32_t* flag, int32_t* host_out, void* stream);
/* sorted-segment sum (Pooling, nn/output.py:66-74): out[s, :] = sum_{r in [ptr[s], ptr[s+1])} x[r, :] (* 1/count if mean) */
int e3k_segment_sum(const float* x, const int32_t* ptr, int64_t n_seg, int32_t dim, int32_t mean, float* out,
                    void* stream);

/* ------------------------------------------------------------------------------------------
 * Keyed self-connection weights (see e3k_gemm_grouped): rows of node_attrs that carry the same key share
 *     M[t, (j,u,w)] = sum_v a[t,v] * W_j[u,v,w]
 * with W_j the [U][V][Wout] weight block of instruction j of FullyConnectedTensorProduct(x, node_attrs)
 * (e3_layers/nn/message_passing.py:81-87, e3nn 'uvw' weight order) at element offset w_off of the flat weight,
 * and its U*Wout columns at m_off of M (columns packed in instruction order, row stride ld_m).
 * a [n_keys, V] (V <= 32; the keys are tiled 64 per workgroup).  backward: g_a [n_keys,V] ACCUMULATED (caller zeroes), g_W flat like W:
 * written (accumulate_w = 0) or added to (accumulate_w = 1); either may be NULL.  g_a needs a device `workspace` of
 * e3k_keyed_weights_bwd_workspace(...) floats (per-block partial sums, reduced by a second launch — same-address
 * atomics from every block serialise).  `instr` is a HOST array. */
typedef struct {
  int64_t w_off, m_off;
  int32_t u, w_out;
} e3k_kw_instr;
int e3k_keyed_weights_fwd(const float* a, const float* W, const e3k_kw_instr* instr, int32_t n_instr, int32_t n_keys,
                          int32_t V, int64_t ld_m, float* M, void* stream);
int64_t e3k_keyed_weights_bwd_workspace(const e3k_kw_instr* instr, int32_t n_instr, int32_t n_keys, int32_t V);
int e3k_keyed_weights_bwd(const float* a, const float* W, const float* g_M, const e3k_kw_instr* instr, int32_t n_instr,
                          int32_t n_keys, int32_t V, int64_t ld_m, float* g_a, float* g_W, int32_t accumulate_w,
                          float* workspace, void* stream);

/* The same for the self-connections of several layers that read the SAME attribute rows `a` (the convolutions of one network:
 * nn/message_passing.py:81-87 x layers), one launch per pass instead of one per layer.  e3k_kw_args: a layer's instruction
 * table in device memory, created once.  items[i].M: forward OUT [n_keys, ld_m_i]; backward IN (g_M of layer i).  g_W as in
 * e3k_keyed_weights_bwd, per item; g_a [n_keys, V] ACCUMULATED over all the layers (caller zeroes).  At most 8 layers. */
typedef struct e3k_kw_args e3k_kw_args;
int e3k_kw_args_create(const e3k_kw_instr* instr, int32_t n_instr, int32_t V, int64_t ld_m, e3k_kw_args** out);
void e3k_kw_args_destroy(e3k_kw_args* args);
typedef struct {
  const e3k_kw_args* args;
  const float* W;
  float* M;
  float* g_W;
  int32_t accumulate_w, _pad;
} e3k_kw_multi_item;
int e3k_keyed_weights_fwd_multi(const e3k_kw_multi_item* items, int32_t n, const float* a, int32_t n_keys, void* stream);
int64_t e3k_keyed_weights_bwd_multi_workspace(const e3k_kw_multi_item* items, int32_t n, int32_t n_keys);
int e3k_keyed_weights_bwd_multi(const e3k_kw_multi_item* items, int32_t n, const float* a, int32_t n_keys, float* g_a,
                                float* workspace, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused hidden chain of the radial MLP.
 * Replaces the hidden layers of e3nn.nn.FullyConnectedNet([n_radial, H, ..., H, weight_numel], act)
 * (e3_layers/nn/message_passing.py:74-79, call :93): per layer l < n_layers
 *     z_l = alphas[l] * (prev @ W_l),   h_l = cst * act(z_l),   prev = x for l = 0, else h_{l-1}
 * x [E,k0] (k0 <= 64), W_0 [k0,h], W_l [h,h] row-major, h in {32, 64}, n_layers <= 4; out = h_{n_layers-1} [E,h].
 * `weights`, `z`, `g_weights` are HOST arrays of n_layers device pointers.
 *   forward : z[l] [E,h] receive the pre-activations the backward needs (z or any z[l] may be NULL: inference).
 *   backward: g_out = gradient wrt out; g_weights[l] (same shapes as W_l) are ACCUMULATED with atomics
 *             (caller zeroes; NULL entries / NULL array are skipped); g_x [E,k0] written when not NULL. */
int e3k_mlp_hidden_fwd(const float* x, int64_t E, int32_t k0, int32_t h, int32_t n_layers, const float* const* weights,
                       const float* alphas, int32_t act, float cst, float* const* z, float* out, void* stream);
int e3k_mlp_hidden_bwd(const float* x, int64_t E, int32_t k0, int32_t h, int32_t n_layers, const float* const* weights,
                       const float* alphas, int32_t act, float cst, const float* const* z, const float* g_out,
                       float* const* g_weights, float* g_x, void* stream);

/* Several MLPs of ONE shape over the SAME input rows in one launch each way (the radial MLPs of all the layers of a
 * network read one edge embedding: nn/message_passing.py:74-79,93 per layer).  g_x is per net (the caller sums). */
typedef struct {
  const float* weights[4];
  float* z[4];            /* forward: pre-activations out (NULL entries: not kept); backward: in */
  float* out;             /* forward: [E, h] */
  const float* g_out;     /* backward: gradient w.r.t. out */
  float* g_weights[4];    /* backward: accumulated (NULL entries: skipped) */
  float* g_x;             /* backward: [E, k0] or NULL */
} e3k_mlp_net;
int e3k_mlp_hidden_fwd_multi(const e3k_mlp_net* nets, int32_t n_nets, const float* x, int64_t E, int32_t k0, int32_t h,
                             int32_t n_layers, const float* alphas, int32_t act, float cst, void* stream);
int e3k_mlp_hidden_bwd_multi(const e3k_mlp_net* nets, int32_t n_nets, const float* x, int64_t E, int32_t k0, int32_t h,
                             int32_t n_layers, const float* alphas, int32_t act, float cst, void* stream);

/* ------------------------------------------------------------------------------------------
 * Radius graph on the device.
 * Replaces computeEdgeIndex (e3_layers/data/compute_edge.py:38-113) for the criteria-free case: per graph all
 * ordered pairs (i, j), i != j, with fp32 sqrt(|pos_i - pos_j|^2) < r_max (strict), in the reference's order
 * (graphs concatenated, i slow, j fast); pre-existing edges (old_ptr [N+1] CSR by source with old_dst ascending
 * inside a row; NULL = none) are kept whatever their length.
 *   graph_start / graph_end [N]: first node / one-past-last node of the graph node i belongs to.
 *   pass 1 writes counts [N] (kept pairs per source); the caller scans them (exclusive, int64) into offsets [N];
 *   pass 2 writes edge_index int64 [2, E] (row 0 sources, row 1 destinations), E = total count. */
int e3k_radius_graph_count(const float* pos, const int32_t* graph_start, const int32_t* graph_end, int64_t N,
                           float r_max, const int32_t* old_ptr, const int32_t* old_dst, int32_t* counts, void* stream);
int e3k_radius_graph_fill(const float* pos, const int32_t* graph_start, const int32_t* graph_end, int64_t N,
                          float r_max, const int32_t* old_ptr, const int32_t* old_dst, const int64_t* offsets, int64_t E,
                          int64_t* edge_index, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training-step plumbing on the flat parameter vector.
 * Replaces clip_grad_norm_ + optim.step() + ema.update() (e3_layers/run/trainer.py:374-386) and the variant that
 * skips the optimizer step on a non-finite gradient (e3_layers/run/sde_utils.py:233-248): torch.optim.Adam
 * (amsgrad off) and torch_ema.ExponentialMovingAverage semantics, one pass over HBM.
 *   param, grad, exp_avg, exp_avg_sq, ema (NULL = no EMA): flat fp32 [n], 16-byte aligned.
 *   max_grad_norm <= 0: no clipping.  skip_nonfinite: leave param / moments untouched when the gradient norm is
 *   not finite (the EMA still updates, as in the reference).
 *   state: DEVICE float[16], zero-initialised once by the caller; holds the step count, bias corrections, clip
 *   coefficient, skip flag, gradient norm ([7]) and the EMA update count — so a captured HIP graph replays
 *   correctly. */
/* A squared-error loss term and its gradient in one launch (a trainer's coefficient x MSELoss term, e3_layers/run/loss.py:186-288
 * with torch.nn.MSELoss; weight != NULL: a weighted sum instead of the mean -- padded batches give their ghost entries weight 0):
 *   loss[0] = scale * sum_i w_i (pred_i - target_i)^2,   grad[i] = 2 scale w_i (pred_i - target_i),   w_i = weight[i / w_group] (one
 * weight per w_group consecutive entries: per node for its three force components), or 1 / n when weight == NULL.
 * One workgroup, fixed summation order. */
int e3k_sq_error(const float* pred, const float* target, const float* weight, int32_t w_group, int64_t n, float scale, float* loss,
                 float* grad, void* stream);
int e3k_adam_ema_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* ema, int64_t n,
                      float lr, float beta1, float beta2, float eps, float weight_decay, float ema_decay,
                      int32_t ema_use_num_updates, float max_grad_norm, int32_t skip_nonfinite, float* state,
                      void* stream);

/* ------------------------------------------------------------------------------------------
 * A convolution layer as ONE call (csrc/e3k_layer.hip): FactorizedConvolution + Gate
 * (e3_layers/nn/message_passing.py:91-124, 249) forward and backward -- the launch sequence of
 * backend/conv_block.py issued from native code on the caller's streams, with the library's own events
 * for the cross-stream edges.  The arithmetic is the kernels above, unchanged; what this removes is
 * the host: ~15 ctypes calls, ~12 stream switches and ~20 tensor allocations per layer and pass
 * (0.22 ms forward / 0.30 ms backward of Python per layer: at <= 128 molecules the whole training
 * step was bound by it, at 256 it equalled the GPU time).
 *
 * A layer is described once (template sets as for e3k_gemm_rebased / e3k_gemm_grouped_rebased: byte
 * offsets in the pointer fields; the arrays are copied).  Every buffer is the caller's (PyTorch's
 * allocator): inputs, outputs, saved activations and scratch arrive as device pointers.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const e3k_gemm_problem* p;   /* n problems, round-major: round r = p[round_start[r] .. round_start[r + 1]) */
  int32_t n;
  int32_t n_rounds;            /* problems of one round write distinct blocks; round r + 1 accumulates on top of round r */
  int32_t round_start[5];      /* (a Linear whose input block feeds two output blocks -- scalars and gates -- has a */
  int32_t _pad;                /*  two-round input gradient) */
} e3k_gemm_set;

typedef struct {
  const e3k_tp_plan* tp;
  e3k_gemm_set lin1_fwd, lin1_dgrad, lin1_dgrad_acc, lin1_wgrad;
  e3k_gemm_set post_fwd, post_dgrad, post_wgrad; /* post_fwd: conv (+)= scale * Linear(mid); accumulates iff a self-connection exists */
  e3k_gemm_set sc_fwd, sc_dgrad, sc_wgrad;       /* keyed (e3k_gemm_grouped) templates; sc_fwd.n == 0: no self-connection */
  e3k_gemm_set last_fwd, last_dgrad, last_wgrad; /* last layer of the radial MLP: [rows, h] -> [rows, W] */
  const e3k_gate_seg* gate;
  int32_t n_gate;
  int32_t n_in_blocks;
  const e3k_block* in_blocks;                    /* e3nn <-> cf relayout of the layer input */
  const e3k_kw_instr* kw;
  int32_t n_kw, V;
  int64_t ld_m;
  int32_t k0, h, n_hidden, act;                  /* radial MLP hidden chain (e3k_mlp_hidden_*) */
  float cst;
  float alphas[4];
  int32_t d_in, d_x1, d_mid, d_conv, d_out, W;
  int32_t post_in_covered, sc_in_covered, lin1_in_covered, sc_out_covered, post_out_covered, tp_bwd_x_overwrites;
} e3k_layer_desc;

typedef struct e3k_layer e3k_layer;
int e3k_layer_create(const e3k_layer_desc* desc, e3k_layer** out);
void e3k_layer_destroy(e3k_layer* layer);

/* radial branch of a layer: hidden chain + last layer on R rows (R = E edges, or knots + 1 table rows) and, with the
 * table, the interpolation to the E edges */
typedef struct {
  int64_t R, E;
  int32_t use_table, keep, knots;
  int32_t have_rows;         /* the MLP's output rows (T with the table, else w) were computed by e3k_radial_stack_fwd:
                                the layer only interpolates (table) / uses w as it is; its backward stops at the gradient
                                of those rows (g_T resp. g_w), which e3k_radial_stack_bwd takes from there */
  int32_t in_kernel;         /* table only, plans with e3k_tp_table_supported: the tensor-product kernels interpolate the
                                path weights from T themselves (e3k_tp_fwd_table / e3k_tp_bwd_x_table): no interpolation
                                pass, w unused (may be null); the backward needs T */
  int32_t packed;            /* P already holds the packed form of T (e3k_radial_stack_fwd packs the tables of all its layers
                                in one launch when their rads carry in_kernel and P): the layer does not pack again */
  const float* radial;       /* [R, k0] */
  const int32_t* bin;        /* table: knot per edge, */
  const int32_t* bin_ptr;    /*        edges grouped by knot (backward), */
  const int32_t* bin_perm;
  const float* bin_coef;     /*        the four interpolation weights per edge [E, 4], */
  const int32_t* bin_seg;    /*        first <= 64-edge segment of every knot [knots + 2] (e3k_rtable_bins) */
  const float* w_last;
  const float* w_hidden[4];
  float* h;                  /* [R, h] out */
  float* z[4];               /* [R, h] pre-activations (keep) */
  float* T;                  /* table: [R, W] out; without the table unused */
  float* w;                  /* [E, W] out */
  const int32_t* erec_dst;   /* with P: the batch's edge records over the destination CSR (forward) and over the source CSR */
  const int32_t* erec_src;   /* (input gradient), e3k_edge_records with this table's bin / coef */
  void* P;                   /* in_kernel only: the table packed for the tensor-product kernels [R, W, 3] dwords (e3k_rtable_pack),
                              * written behind T by the forward, read by tp_fwd / tp_bwd_x; NULL: they gather four rows of T */
} e3k_layer_radial;

typedef struct {
  int64_t N, E;
  int32_t in_cf, out_cf, keep, fork, has_w, n_keys;
  int32_t have_m, _pad;              /* have_m: the per-key self-connection weights `m` were computed by e3k_kw_stack_fwd */
  void *main, *side, *side2;
  const float *x, *node_attrs, *sh;
  const int32_t *src, *dst_ptr, *dst_perm;
  const int32_t *perm, *bounds;
  const int64_t* reps;
  const float *w_lin1, *w_post, *w_sc;
  e3k_layer_radial rad;              /* this layer's radial branch (skipped when has_w: rad.w already holds the weights) */
  const e3k_layer* next;             /* look-ahead: the next layer's radial branch, issued behind this layer's tensor product */
  const e3k_layer_radial* next_rad;
  float *x_cf, *a_rep, *m, *conv, *x1, *mid, *y;
} e3k_layer_fwd_args;
int e3k_layer_fwd(const e3k_layer* layer, const e3k_layer_fwd_args* a);

typedef struct {
  int64_t N, E;
  int32_t in_cf, out_cf, fork, n_keys, need_x, need_attrs, need_radial, acc_sc;
  int32_t have_m, fuse_xw;           /* have_m: `gm` (gradient of the per-key weights) is an OUTPUT handed to e3k_kw_stack_bwd;
                                        no weight / attribute gradient of the self-connection is formed here (2: gm arrives
                                        zero-filled, 1: it is zero-filled here).
                                        fuse_xw: with rad.P, the per-edge weight gradient g_w is formed by the input-gradient
                                        walk (e3k_tp_bwd_xw_ptable) instead of a pass of its own (e3k_tp_bwd_w) */
  void *main, *side, *side2, *side3;
  /* saved by the forward */
  const float *x_cf, *sh, *x1, *mid, *conv, *a_rep, *m;
  const int32_t *src, *dst, *dst_ptr, *dst_perm, *src_ptr, *src_perm;
  const int32_t *perm, *bounds;
  const int64_t* reps;
  const float *w_lin1, *w_post, *w_sc;
  e3k_layer_radial rad;              /* as in the forward (h, z, w, radial, bins, weights) */
  const float* gy;
  /* gradient buffers (NULL = not needed); accumulated (gradient sink) or written into zero-filled temporaries */
  float *gb_lin1, *gb_post, *gb_sc, *gb_last;
  float* gb_hidden[4];
  /* outputs */
  float *g_x;        /* [N, d_in] in the layer input's layout (need_x) */
  float *g_attrs;    /* [N, V] (need_attrs): zero-filled here, the per-key rows scattered to the keys' representatives */
  float *g_radial;   /* [R, k0] (need_radial) */
  /* scratch */
  float *g_conv, *g_mid, *g_x1, *g_xcf, *g_w, *g_T, *table_ws, *g_h, *gm, *ga, *kw_ws;
} e3k_layer_bwd_args;
int e3k_layer_bwd(const e3k_layer* layer, const e3k_layer_bwd_args* a);

/* The radial MLPs of several layers that read the same rows (one edge embedding / one knot basis), batched: hidden
 * chains in ONE launch (e3k_mlp_hidden_fwd_multi), last layers in ONE e3k_gemm_multi call -> rads[i].T (table) or
 * rads[i].w (per edge); backward: last-layer weight gradients in one call, input gradients in one, hidden chains in
 * one.  Per layer this was 2 launches forward and 4-5 backward on 4 097 knot rows -- latency, not work: 0.6 ms of a
 * 3.4 ms step at 64 molecules.  All layers must share k0 / h / depth / activation (the shipped models do). */
typedef struct {
  e3k_layer_radial rad;      /* radial rows, weights, h, z as in the forward */
  const float* g_rows;       /* [R, W] gradient of the MLP's output rows (from e3k_layer_bwd: g_T or g_w) */
  float* gb_last;            /* weight-gradient buffers (NULL = not needed), accumulated */
  float* gb_hidden[4];
  float* g_h;                /* scratch [R, h] */
  float* g_radial;           /* [R, k0] per layer (the caller sums) or NULL */
  /* force training on the table: the gradient of the SLOPE table D = H' W_last (e3k_radial_slope_fwd) joins the table's --
   * gb_last += H'^T g_slope here by the ordinary GEMM; g_hp = g_slope W_last^T is handed to the float64 reverse sweep of the
   * tangent chain (e3k_slope_tangent_bwd), which adds the hidden weights' share into gb_hidden.  All NULL: no slope table. */
  const float* g_slope;      /* [R, W] gradient of D */
  const float* hp;           /* [R, h] H' (fp32) as e3k_radial_slope_fwd wrote it */
  float* g_hp;               /* scratch [R, h] */
} e3k_radial_stack_item;
/* what the slope tables' chain needs besides the layers' weights: the knot radii, the basis and float64 scratch */
typedef struct {
  const float* knots;        /* [R] */
  const float* bessel_w;     /* [k0] */
  float r_max, r_min, p;
  int32_t one_over_r, cutoff_kind, _pad;
  double* acc;               /* backward: e3k_slope_tangent_bwd_scratch(n, n_hidden, k0, h, R) doubles */
  float* g_bessel;           /* backward: [k0] fp32, ADDED to (NULL: the frequencies need no gradient) */
} e3k_slope_ctx;
int e3k_radial_stack_fwd(const e3k_layer* const* layers, const e3k_layer_radial* rads, int32_t n, void* stream);
/* slope: NULL unless some item carries a slope gradient */
int e3k_radial_stack_bwd(const e3k_layer* const* layers, const e3k_radial_stack_item* items, int32_t n, const e3k_slope_ctx* slope,
                         void* stream);
/* The slope tables D_l [R, W] = d/dr fc_l(basis(r)) on the R = knots + 1 knots for the n layers of a radial stack
 * (csrc/e3k_slope.hip): H'_l = forward-mode derivative of the hidden chain, per knot in float64 (fp32 out, hp[l] [R, h], kept for
 * the backward) -> D_l = H'_l W_last_l (the layers' LAST_FWD GEMM sets). */
int e3k_radial_slope_fwd(const e3k_layer* const* layers, const e3k_layer_radial* rads, int32_t n, const e3k_slope_ctx* slope,
                         float* const* hp, float* const* D, void* stream);
/* pieces (also exported for tests); w_hidden[i * 4 + l]: layer l of net i */
int e3k_slope_tangent_fwd(const float* const* w_hidden, int32_t n_nets, int32_t n_hidden, const float* alphas, const float* knots,
                          int64_t R, const float* bessel_w, int32_t k0, int32_t H, float r_max, float r_min, float p,
                          int32_t one_over_r, int32_t cutoff_kind, int32_t act, float cst, float* const* hp, void* stream);
int64_t e3k_slope_tangent_bwd_scratch(int32_t n_nets, int32_t n_hidden, int32_t k0, int32_t H, int64_t R);
int e3k_slope_tangent_bwd(const float* const* w_hidden, int32_t n_nets, int32_t n_hidden, const float* alphas, const float* knots,
                          int64_t R, const float* bessel_w, int32_t k0, int32_t H, float r_max, float r_min, float p,
                          int32_t one_over_r, int32_t cutoff_kind, int32_t act, float cst, const float* const* g_hp, double* acc,
                          float* const* g_hidden, float* g_bessel, void* stream);

/* The per-key self-connection weights M_l = sum_v a[key, v] W_l[:, v, :] of several layers that read the same node attributes
 * (nn/message_passing.py:81-87, 100 x layers), batched: forward = one gather of the keys' representative rows + ONE launch for
 * all layers; backward (behind the first layer's backward, with the gm each layer's e3k_layer_bwd wrote) = ONE launch for the
 * weight gradients, one + a reduction for the attribute gradient summed over the layers, one scatter to the representatives.
 * Per layer that was 2 launches forward and 6 backward + an autograd add. */
typedef struct {
  const float* w_sc;
  float* m;                  /* forward: OUT [n_keys, ld_m]; backward: IN, the gradient gm of that block */
  float* gb_sc;              /* backward: weight-gradient buffer (NULL = not needed) */
  int32_t acc_sc, _pad;      /*           1 = accumulate into it (gradient sink), 0 = write */
} e3k_kw_stack_item;
int e3k_kw_stack_fwd(const e3k_layer* const* layers, const e3k_kw_stack_item* items, int32_t n, const float* node_attrs,
                     const int64_t* reps, int32_t n_keys, float* a_rep, void* stream);
int64_t e3k_kw_stack_bwd_workspace(const e3k_layer* const* layers, int32_t n, int32_t n_keys);
int e3k_kw_stack_bwd(const e3k_layer* const* layers, const e3k_kw_stack_item* items, int32_t n, const float* a_rep,
                     const int64_t* reps, const int32_t* bounds, int64_t N, int32_t n_keys, float* ga, float* g_attrs,
                     float* workspace, void* stream);

/* Per-kernel timing of a layer's edge kernels (bench.py's roofline block: HIP events on the stream that runs the kernel).
 * e3k_layer_profile(layer, capacity): capacity > 0 arms `capacity` event pairs per kind, 0 disarms.
 * e3k_layer_profile_read: waits for the recorded pairs of `kind` and returns their count (<= cap), filling the elapsed
 * milliseconds and the node / edge counts (table kinds: table rows / edges) of each launch. */
#define E3K_PROF_TP_FWD 0
#define E3K_PROF_TP_BWD_X 1
#define E3K_PROF_TP_BWD_W 2
#define E3K_PROF_RTABLE_FWD 3
#define E3K_PROF_RTABLE_BWD 4
#define E3K_PROF_RADIAL_LAST_FWD 5
int e3k_layer_profile(e3k_layer* layer, int32_t capacity);
/* the same, recording only the kinds whose bit is set in `kinds` (bit = E3K_PROF_* value): a timed region that wants the
 * dominant kernel's durations live without event pairs around every other launch */
int e3k_layer_profile_mask(e3k_layer* layer, int32_t capacity, uint32_t kinds);
int e3k_layer_profile_read(e3k_layer* layer, int32_t kind, float* ms, int64_t* n, int64_t* e, int32_t cap);

#ifdef __cplusplus
}
#endif
#endif /* E3K_H */
