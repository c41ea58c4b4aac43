/*
 * vsde_hip.h -- C ABI of libvsde_hip.so, the MI355X (gfx950) implementation of the
 * variational-SDE hot path.  Plain pointers and sizes only; every pointer is a DEVICE
 * pointer unless stated otherwise; `stream` is a hipStream_t passed as void*.
 *
 * Each entry point replaces one interface of the reference (Tom-Ryder/VIforSDEs):
 *
 *   vsde_head_forward        launch_fwd + sde_fwd_kernel   src/variational_sde/kernels/forward.py:378-563
 *                            (called from _SDEFunction.forward, kernels/autograd.py:79-87, and from
 *                             kernels.autograd.sample_diffusion_paths, autograd.py:244-268)
 *   vsde_head_backward       launch_bwd + sde_bwd_kernel   src/variational_sde/kernels/backward.py:627-784
 *                            (called from _SDEFunction.backward, kernels/autograd.py:191-201)
 *   vsde_elbo_path_terms     the B*T-sized part of compute_evidence_lower_bound
 *   vsde_elbo_path_terms_bwd   src/variational_sde/inference/evidence_lower_bound.py:42-50,77-83 and
 *                              DiffusionPathSample.log_jacobian, inference/types.py:23-24
 *
 * Weight tensors use torch.nn.GRU's native layout exactly as they are handed to
 * _SDEFunction.apply (kernels/autograd.py:46-56): W_ih_l0[3H][S+C+P] with input order
 * [state | context | theta] (models/head.py:75), W_hh_l0[3H][H], biases [3H], the stacked
 * layers >=1 as [L-1][3H][H] / [L-1][3H], out_weight[S+ntril][H], out_bias[S+ntril];
 * gate order r, z(u), n.  Gradients come back in the same layout and the same order as
 * launch_bwd's 13-tuple (backward.py:766-784).
 *
 * Error convention: every function returns 0 on success, a negative VSDE_E_* code for
 * argument errors (the Python host maps them to ValueError like the reference's own checks,
 * models/head.py:33-36) or a positive hipError_t.  vsde_last_error() returns a static,
 * thread-local, human-readable message for the last failure.
 *
 * Threading/streams: functions only enqueue work on `stream` (no device synchronisation, no
 * allocation); all scratch memory is the caller-provided workspace, sized by the matching
 * *_workspace_bytes query.  Inputs are never written; outputs are fully overwritten.
 */
#ifndef VSDE_HIP_H
#define VSDE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSDE_ABI_VERSION 1

#define VSDE_E_BADARG (-1)   /* shape/pointer/dtype argument invalid                    */
#define VSDE_E_LAYERS (-2)   /* num_layers outside [1, 4]  (kernels/constants.py:13)    */
#define VSDE_E_HIDDEN (-3)   /* hidden_dim not supported by the compiled kernels        */
#define VSDE_E_WORKSPACE (-4) /* workspace too small                                    */
#define VSDE_E_STATE (-5)    /* state_dim not supported by the compiled kernels         */

#define VSDE_MAX_LAYERS 4
#define VSDE_MAX_HIDDEN 1024 /* tuned kernels: hidden_dim <= 64; generic kernels: one thread per unit up to 1024 */
#define VSDE_MAX_STATE 32    /* tuned kernels: S + S(S+1)/2 <= 64 emission rows (S <= 9); generic kernels beyond */

#define VSDE_CTX_F32 0
#define VSDE_CTX_BF16 1

int vsde_abi_version(void);
/* 1: this library was built with -DVSDE_ABLATIONS (A/B switches read from the environment, losing kernel variants compiled in:
   tools and their tests only); 0: the shipped library. */
int vsde_build_ablations(void);
const char *vsde_last_error(void);

/* Problem dimensions shared by the head entry points (names as in SURVEY.md section 8). */
typedef struct vsde_head_dims {
    int B; /* Monte-Carlo sample paths in this launch          */
    int T; /* Euler-Maruyama steps (= context.shape[1])        */
    int S; /* state_dim                                        */
    int P; /* sde_param_dim                                    */
    int C; /* context_dim (= encoder hidden_dim)               */
    int H; /* GRU hidden_dim                                   */
    int L; /* GRU num_layers                                   */
} vsde_head_dims;

/* The ten weight tensors of the head (device pointers, fp32, contiguous). */
typedef struct vsde_head_weights {
    const float *W_ih_l0, *W_hh_l0, *b_ih_l0, *b_hh_l0;
    const float *W_ih_stack, *W_hh_stack, *b_ih_stack, *b_hh_stack; /* may be NULL when L == 1 */
    const float *out_weight, *out_bias;
} vsde_head_weights;

/* Strided view of the encoder context: element (b, t, c) lives at
 * base + (b*batch_stride + t*step_stride + c) * sizeof(elem).  This lets the caller pass the
 * reference's non-contiguous context[:, :-1] slice (inference/diffusion_path_sampler.py:61)
 * and the autocast bf16 context without the .contiguous().float() copies of forward.py:495. */
typedef struct vsde_context_view {
    const void *base;
    int dtype; /* VSDE_CTX_F32 or VSDE_CTX_BF16 */
    int64_t batch_stride;
    int64_t step_stride;
} vsde_context_view;

size_t vsde_head_forward_workspace_bytes(const vsde_head_dims *d);

/* Fused multi-layer-GRU + Gaussian emission + Euler-Maruyama time stepping.
 *   x0[B][S], theta[B][P], eps[B][T][S]
 *   paths[B][T+1][S], means[B][T][S], chol[B][T][S][S] (strict upper triangle written as 0)
 * Training mode (save != 0) additionally fills what the backward needs
 * (kernels/weights.py:11-23 SavedActivations), packed as
 *   chol_raw[B][T][ntril]  and  acts[B][T][L][5][H] with slots (h, r, z, n, n_hh). */
int vsde_head_forward(const vsde_head_dims *d, const float *x0, const vsde_context_view *ctx,
                      const float *theta, const float *eps, const vsde_head_weights *w,
                      double time_step, double diag_min, int save,
                      float *paths, float *means, float *chol, float *chol_raw, float *acts,
                      void *workspace, size_t workspace_bytes, void *stream);

/* The 13 gradient outputs of launch_bwd, same order (backward.py:766-784). */
typedef struct vsde_head_grads {
    float *x0;          /* [B][S]        */
    float *context;     /* [B][T][C]     */
    float *theta;       /* [B][P]        */
    float *W_ih_l0, *W_hh_l0, *b_ih_l0, *b_hh_l0;
    float *W_ih_stack, *W_hh_stack, *b_ih_stack, *b_hh_stack; /* unused when L == 1 */
    float *out_weight, *out_bias;
    /* placement of the context gradient (0, 0 = fp32 [B][T][C] contiguous): dtype 1 writes bf16 through the same pointer;
     * a batch stride (elements, >= T*C) lets it be the leading T rows of every slab of a [B][T+1][C] buffer, i.e. the
     * gradient of the encoder output itself, whose last token the head never reads (diffusion_path_sampler.py:66). */
    int context_dtype;
    int64_t context_batch_stride;
} vsde_head_grads;

size_t vsde_head_backward_workspace_bytes(const vsde_head_dims *d);

/* Reverse-time BPTT through the fused head.  g_paths[B][T+1][S], g_means[B][T][S],
 * g_chol[B][T][S][S] are the upstream gradients; paths/chol_raw/acts are the tensors the
 * forward saved.  Deterministic: weight gradients are tree-reduced, no atomics
 * (the reference uses global fp32 atomics, backward.py:108-139,575-590). */
int vsde_head_backward(const vsde_head_dims *d, const float *g_paths, const float *g_means,
                       const float *g_chol, const vsde_context_view *ctx, const float *theta,
                       const float *eps, const float *paths, const float *chol_raw,
                       const float *acts, const vsde_head_weights *w, double time_step,
                       double diag_min, const vsde_head_grads *grads, void *workspace,
                       size_t workspace_bytes, void *stream);

/* Per-sample path terms of the ELBO.  z, x: [B][T+1][S] (x = StateSpace.to_state(z));
 * means/drift: [B][T][S]; chol/diffusion: [B][T][S][S] lower-triangular factors;
 * positive_mask: S bytes on the HOST (state_space.positive_dims).
 * Outputs [B]: sde_lp, gen_lp, log_jac. */
int vsde_elbo_path_terms(int B, int T, int S, const float *z, const float *x, const float *means,
                         const float *chol, const float *drift, const float *diffusion,
                         const uint8_t *positive_mask_host, double time_step,
                         float *sde_lp, float *gen_lp, float *log_jac, void *stream);

/* Adjoint of vsde_elbo_path_terms for upstream per-sample gradients g_sde/g_gen/g_jac [B].
 * Outputs are fully overwritten: g_z, g_x [B][T+1][S]; g_means, g_drift [B][T][S];
 * g_chol, g_diffusion [B][T][S][S]. */
int vsde_elbo_path_terms_bwd(int B, int T, int S, const float *z, const float *x,
                             const float *means, const float *chol, const float *drift,
                             const float *diffusion, const uint8_t *positive_mask_host,
                             double time_step, const float *g_sde, const float *g_gen,
                             const float *g_jac, float *g_z, float *g_x, float *g_means,
                             float *g_chol, float *g_drift, float *g_diffusion, void *stream);

/* ---- Fused encoder operators (SiT blocks of the observation encoder) -------------------------
 * dtype: 0 = f32, 1 = bf16 for every `void *` tensor of the call; cos/sin tables, RMS weights,
 * lambda, mean/rstd are always f32.  Contiguous tensors; B = batch rows, N = tokens per row.
 * Replaces the unfused torch chains of primitives/sit.py:99-128 (LayerNorm -> (1+scale)x+shift,
 * gate*branch residual), primitives/attn.py:80-113 (QK RMS-norm, RoPE, value-residual mix, head
 * layout, sigmoid output gate) and primitives/mlp.py:21-24 (SwiGLU activation). */
/* mod_pitch: row pitch (elements) of the per-batch-row vectors scale / shift / gate AND of their gradient outputs; 0 = C
 * (contiguous [B][C]).  A pitch > C lets all of them be column ranges of one [B][pitch] buffer -- the output of the single
 * GEMM that produces every block's adaLN parameters, and its gradient -- with no copies in either direction. */
int vsde_ln_modulate_fwd(int dtype, const void *x, const void *scale, const void *shift, void *y, float *mean,
                         float *rstd, int64_t B, int N, int C, double eps, int64_t mod_pitch, void *stream);
/* The two backward passes below also produce per-(batch row, channel) sums over the tokens (dscale/dshift, dgate);
 * they need vsde_colsum_workspace_bytes(B, C) bytes of device scratch for the fp32 partials. */
size_t vsde_colsum_workspace_bytes(int64_t B, int C);
/* dres (optional, [B][N][C]) is added to dx: the gradient reaching x through the residual branch */
int vsde_ln_modulate_bwd(int dtype, const void *x, const void *scale, const void *dy, const float *mean,
                         const float *rstd, const void *dres, void *dx, void *dscale, void *dshift, int64_t B, int N, int C,
                         int64_t mod_pitch, void *workspace, size_t workspace_bytes, void *stream);
/* Gated residual fused with the LayerNorm-modulate that follows it (primitives/sit.py:73-79 + next norm):
 *   xnew = x + gate*y;  h = LN(xnew)*(1+scale) + shift.   Backward: dxnew (optional) is the gradient reaching xnew from its
 *   other consumers; dx = dxnew + LNbwd(dh) is the gradient of x, dy = gate*dx, dgate/dscale/dshift are token sums. */
int vsde_residual_ln_fwd(int dtype, const void *x, const void *y, const void *gate, const void *scale, const void *shift,
                         void *xnew, void *h, float *mean, float *rstd, int64_t B, int N, int C, double eps, int64_t mod_pitch,
                         void *stream);
int vsde_residual_ln_bwd(int dtype, const void *xnew, const void *y, const void *gate, const void *scale, const void *dh,
                         const void *dxnew, const float *mean, const float *rstd, void *dx, void *dy, void *dgate,
                         void *dscale, void *dshift, int64_t B, int N, int C, int64_t mod_pitch, void *workspace,
                         size_t workspace_bytes, void *stream);
int vsde_gated_residual_fwd(int dtype, const void *x, const void *y, const void *gate, void *out, int64_t B, int N, int C,
                            int64_t mod_pitch, void *stream);
int vsde_gated_residual_bwd(int dtype, const void *y, const void *gate, const void *dout, void *dy, void *dgate, int64_t B,
                            int N, int C, int64_t mod_pitch, void *workspace, size_t workspace_bytes, void *stream);
int vsde_swiglu_fwd(int dtype, const void *u, void *out, int64_t M, int H2, void *stream);
int vsde_swiglu_bwd(int dtype, const void *u, const void *dout, void *du, int64_t M, int H2, void *stream);
/* token_major selects the memory layout of the per-head tensors (attn, dattn, q, k, v, v0, dq, dk, dv, dv0):
 * 0 = [B][heads][N][d] (the reference's layout after attn.py:96), 1 = [B][N][heads][d] (what the memory-efficient
 * SDPA kernels read and write natively, so no transposing copies are needed around the attention call). */
int vsde_gate_merge_fwd(int dtype, const void *attn, const void *glog, void *out, int64_t B, int N, int heads, int d,
                        int token_major, int64_t glog_stride, void *stream);
int vsde_gate_merge_bwd(int dtype, const void *attn, const void *glog, const void *dout, void *dattn, void *dglog, int64_t B,
                        int N, int heads, int d, int token_major, int64_t glog_stride, void *stream);
/* qkv[B][N][3C] -> q, k, v; cosT/sinT [N][d/2]; wq/wk [d]; v0 optional, same layout as v.
 * row_stride (>= 3C elements) / glog_stride (>= d) are the row pitches of qkv+dqkv / glog+dglog, so both can be column
 * ranges of the output of ONE merged [qkv | gate] projection (and of its gradient buffer). */
int vsde_qk_norm_rope_fwd(int dtype, const void *qkv, const float *cosT, const float *sinT, const float *wq, const float *wk,
                          const void *v0, const float *lam, void *q, void *k, void *v, int64_t B, int N, int heads, int d,
                          double eps, int token_major, int64_t row_stride, void *stream);
int64_t vsde_qk_norm_rope_bwd_partials(int64_t B, int N, int heads, int d);
int vsde_qk_norm_rope_bwd(int dtype, const void *qkv, const float *cosT, const float *sinT, const float *wq, const float *wk,
                          const void *v0, const float *lam, const void *dq, const void *dk, const void *dv, void *dqkv,
                          void *dv0, float *dlam_partial, int64_t B, int N, int heads, int d, double eps, int token_major,
                          int64_t row_stride, int dv0_accumulate, const void *dv_extra, void *stream);
/* dv0_accumulate != 0: dv0 += (1-lam) dv (all blocks' value-residual gradients collect in one buffer); dv_extra (optional,
 * layout of dv) is added to dv first -- the block that produced v0 gets that buffer next to its own attention's dv. */

/* Attention core (bf16, head_dim 64 or 128).  head_dim 64 with N <= vsde_attention_max_tokens() runs the kernels that keep K
 * and V of one (batch, head) resident in LDS (the encoder's shape class at the OU / LV grids); longer sequences and head_dim
 * 128 (the 1001-token, 512 / 4-head stress configuration) run the kernels that stream 32-token tiles through LDS:  o = softmax(scale * q k^T) v  with q, k, v, o token-major [B][N][H][64];
 * lse [B][H][N] = natural-log sum-exp of the scaled scores (what a flash-attention backward consumes).
 * Replaces F.scaled_dot_product_attention at primitives/attn.py:104-106 for these shapes. */
int vsde_attention_max_tokens(void);
int vsde_attention_fwd_bf16(const void *q, const void *k, const void *v, void *o, float *lse, int64_t B, int N, int H,
                            int head_dim, double scale, void *stream);
/* Backward of the above (same limits): dq, dk, dv token-major bf16 from dout, q, k, v, o and the forward's lse;
 * delta [B][H][N] fp32 is scratch (row sums <dout, o>).  Deterministic (no atomics). */
int vsde_attention_bwd_bf16(const void *dout, const void *q, const void *k, const void *v, const void *o, const float *lse,
                            void *dq, void *dk, void *dv, float *delta, int64_t B, int N, int H, int head_dim, double scale,
                            void *stream);

/* ---- Training step with the projection-side elementwise work folded into the attention kernels ------------------------
 * (primitives/attn.py:80-113 between the [q | k | v | gate] projection and the output projection; head_dim 64 and
 * N <= vsde_attention_max_tokens(): vsde_attention_fused_supported != 0.)  The forward is vsde_linear_qknorm_bf16 (with its
 * rinv / vdiff outputs) + vsde_attention_fwd_gated_bf16; the backward is vsde_gate_bwd_delta + vsde_attention_bwd_fused_bf16,
 * which leaves the gradient dy [M][ldy] of the projection output ready for its input- and weight-gradient GEMMs.  Replaces
 * the separate qk_norm_rope fwd / bwd and gate_merge fwd passes of the unfused training chain.
 *   vsde_attention_fwd_gated_bf16   o[b,n,h,:] = softmax(..) v * s[b n][0..64), s = rnd(sigmoid(gate logit)) as written by
 *                                   vsde_linear_qknorm_bf16(gate_sigmoid = 1)          (the merged [B,N,(h d)] rows)
 *   vsde_gate_bwd_delta             dattn = dout * s, dgate[m][0..64) = (1 - s) sum_h dout * og (the gradient of the LOGITS),
 *                                   delta[b,h,n] = <dout, og>
 *   vsde_attention_bwd_fused_bf16   dattn, the forward's q, k, v, lse and that delta -> dy columns [dq_raw | dk_raw | dv_raw]:
 *       the RoPE / RMS-norm backward from the saved rotated rows and rinv [M][2H] (weights wq / wk all non-zero), the value mix
 *       from vdiff = v_raw - v0 (NULL: no mixing): dv_raw = lam dv, dv0 (+)= (1 - lam) dv, dlam_partial[B H ceil(N/32)] partial
 *       sums of <dv, vdiff> (vsde_attention_bwd_fused_partials entries); dv_extra (NULL or [M][64H]) is added to dv first. */
int vsde_attention_fused_supported(int N, int head_dim);
int vsde_attention_fwd_gated_bf16(const void *q, const void *k, const void *v, const void *gate, int64_t ldg, void *o, float *lse,
                                  int64_t B, int N, int H, double scale, void *stream);
int vsde_gate_bwd_delta(const void *dout, const void *og, const void *gate, int64_t ldg, void *dattn, void *dgate, int64_t ldd,
                        float *delta, int64_t B, int N, int H, void *stream);
int64_t vsde_attention_bwd_fused_partials(int64_t B, int N, int H);
int vsde_attention_bwd_fused_bf16(const void *dattn, const void *q, const void *k, const void *v, const float *lse, const float *delta,
                                  const float *rinv, const float *cosT, const float *sinT, const float *wq, const float *wk,
                                  const void *vdiff, const float *lam, void *dv0, int dv0_accumulate, const void *dv_extra, void *dy,
                                  int64_t ldy, float *dlam_partial, int64_t B, int N, int H, double scale, void *stream);

/* Weight and bias gradient of y = x W^T + b for bf16 activations:  dW[N][K] = dy^T x,  db[N] = colsum(dy)
 * (db may be NULL).  dy [M][N], x [M][K] bf16 contiguous, N % 8 == K % 8 == 0; results fp32, deterministic.
 * Replaces torch's hipBLASLt "wgrad" GEMM + bf16 column-sum kernel of every encoder nn.Linear
 * (reference: primitives/attn.py:46-47,54, primitives/mlp.py:41-44 via torch autograd). */
size_t vsde_linear_wgrad_workspace_bytes(int64_t M, int N, int K);
int vsde_linear_wgrad_bf16(const void *dy, const void *x, int64_t M, int N, int K, float *dW, float *db, void *workspace,
                           size_t workspace_bytes, void *stream);
/* Same, with the rows of the product scattered: row n of dY^T X (and of the column sums) is stored at dW[row_map[n]] /
 * db[row_map[n]] (int32 [N] on the device; negative = dropped).  Lets a weight packed in a permuted / padded row order (the
 * 16-row-interleaved SwiGLU input projection) hand its gradient back in the parameter's own row order without a gather. */
int vsde_linear_wgrad_bf16_rows(const void *dy, const void *x, int64_t M, int N, int K, float *dW, float *db,
                                const int32_t *row_map, void *workspace, size_t workspace_bytes, void *stream);
/* Several weight gradients in (at most) two launches per tile width plus their reductions instead of one pair per problem:
 * item i is exactly one vsde_linear_wgrad_bf16_rows call (with group_plan = 0: same plan, same fixed-order sums, bit-identical).  At a few
 * thousand rows every such problem is a ~20 us kernel that cannot fill the chip alone (the OU example: 25 per optimizer step);
 * the trainer collects the encoder's weight gradients of one backward pass and issues them together. */
typedef struct VsdeWgradItem {
    const void *dy, *x;      /* bf16 [M][N], [M][K], contiguous, 16-byte aligned */
    int64_t M;
    int32_t N, K;            /* multiples of 8 */
    float *dW, *db;          /* fp32 [rows][K], [rows]; db may be NULL */
    const int32_t *row_map;  /* device int32 [N] or NULL (identity) */
} VsdeWgradItem;
size_t vsde_linear_wgrad_group_workspace_bytes(int n, const void *items /* VsdeWgradItem[n], host */);
/* group_plan != 0: split counts chosen for the group as a whole (fewer, longer workgroups per problem: less partial-tile traffic);
 * the fixed-order sums then differ in order from a single launch's -- deterministic, but not bit-identical to it. */
int vsde_linear_wgrad_group_bf16(int n, const void *items, int group_plan, void *workspace, size_t workspace_bytes, void *stream);

/* ---- Dense contractions of the encoder on bf16 MFMA ----------------------------------------------------------------
 * y[M][N] = x[M][K] w[N][K]^T + bias[N]   (x, w, bias, y bf16; fp32 accumulation).  Replaces the hipBLASLt GEMMs behind
 * nn.Linear in primitives/attn.py:46-47,54,113 (qkv / gate / out projections), primitives/mlp.py:41-54 (SwiGLU pair) and
 * primitives/sit.py:186 (output projection), and -- called with the transposed weight -- their input gradients.
 * ldx / ldy: row pitches in elements (multiples of 8; 16-byte aligned bases).  epilogue:
 *   VSDE_EPI_PLAIN        y = acc + bias
 *   VSDE_EPI_SWIGLU       w packs the SwiGLU input projection with its two halves interleaved in blocks of 16 rows
 *                         ([a_0..15 | b_0..15 | a_16..31 | ...]); u = acc + bias goes to y (may be NULL when the caller
 *                         does not need it), s_out[M][N/2] = silu(a) * b  (mlp.py:21-24)
 *   VSDE_EPI_SWIGLU_BWD   acc is ds (gradient of s, N = width of s); u_in[M][2N] is the saved interleaved u and
 *                         y[M][2N] receives du = (da | db) in the same interleaved layout
 * vsde_linear_bf16_supported returns 0 when the shape is outside the compiled kernels (K in {128, 256} with N % 64 == 0;
 * for the plain epilogue also any K % 64 == 0 with N % 128 == 0); callers then keep their library GEMM. */
#define VSDE_EPI_PLAIN 0
#define VSDE_EPI_SWIGLU 1
#define VSDE_EPI_SWIGLU_BWD 2
int vsde_linear_bf16_supported(int64_t M, int N, int K, int epilogue);
int vsde_linear_bf16(const void *x, int64_t ldx, const void *w, const void *bias, void *y, int64_t ldy, int64_t M, int N, int K,
                     int epilogue, void *s_out, int64_t lds, const void *u_in, int64_t ldu, void *stream);
/* Attention projection (primitives/attn.py:80-103 in one kernel): y = x W^T + b with W = [q | k | v | gate] rows
 * (N = 3 heads*64 + gate_width, K in {128, 256}), and in the epilogue RMS-norm (weights wq / wk [64], eps) + RoPE (cos / sin tables
 * [tokens][32], row m is token m % tokens, rotary pairs (i, i + 32)) on every q / k head, v = lam v + (1 - lam) v0 when residual
 * values are given, everything written straight in the attention layout: q, k, v [M][heads*64] (token-major [B,N,h,64]),
 * gate logits [M][ldg].  Same rounding points as vsde_linear_bf16 followed by vsde_qk_norm_rope_fwd.  For the training step
 * (both may be NULL): rinv [M][2 heads] fp32 receives the inverse RMS of every q / k head row and vdiff [M][heads*64] bf16
 * v_raw - v0 -- all the backward needs, the raw projection is never written; gate_sigmoid != 0: the gate block is written as
 * rnd(sigmoid(logit)), the factor vsde_attention_fwd_gated_bf16 multiplies by (one sigmoid per token and channel, not per head). */
int vsde_linear_qknorm_bf16(const void *x, int64_t ldx, const void *w, const void *bias, int64_t M, int K, int heads, int gate_width,
                            int tokens, const float *cosT, const float *sinT, const float *wq, const float *wk, const void *v0,
                            const float *lam, double eps, void *q, void *k, void *v, void *gate, int64_t ldg, int gate_sigmoid,
                            float *rinv, void *vdiff, void *stream);
/* No-grad attention output projection with gate_merge folded into the operand load (primitives/attn.py:107-110):
 * y = (attn * sigmoid(gate[:, k % 64])) W^T + b for attn [M][K] (token-major heads of 64), gate logits [M][ldgate].
 * Rows kernel shapes only (K in {128, 256}, N % 64 == 0). */
int vsde_linear_gated_bf16(const void *attn, int64_t ldx, const void *gate, int64_t ldgate, const void *w, const void *bias, void *y,
                           int64_t ldy, int64_t M, int N, int K, void *stream);

/* Training step: input gradient of the attention output projection with the backward of the sigmoid output gate in its
 * epilogue (primitives/attn.py:107-113; replaces vsde_linear_bf16 on the transposed weight followed by vsde_gate_bwd_delta):
 * d = dy w_t^T is the gradient of the merged rows [M][heads*64] (w_t [heads*64][K] = the projection weight transposed; K in
 * {128, 256}); with og (the merged gated rows) and the gate factors s [M][lds] it leaves as
 *   dattn = d * s,   delta[b,h,n] = <d, og>,   dgate[m][0..64) = (1 - s) sum_h d og   (gradient of the gate LOGITS, row pitch ldd)
 * -- d itself is never written.  Rows are (batch, token) pairs, m = b * tokens + n. */
int vsde_linear_gate_bwd_bf16(const void *dy, int64_t ldy, const void *w_t, const void *og, const void *s, int64_t lds, void *dattn,
                              void *dgate, int64_t ldd, float *delta, int64_t M, int K, int heads, int tokens, void *stream);

/* ---- The SwiGLU feed-forward of a SiT block as one kernel per direction (csrc/vsde_mlp.hip, round 5) -------------------------
 * Replaces primitives/mlp.py:50-54 under autocast:  y = W_out (silu(a) * b) + b_out,  [a | b] = W_in x + b_in,  x [M][C] bf16,
 * hidden size H (a multiple of 64, zero-padded), C in {128, 256}.  The weights are handed over as tile IMAGES (T = H / 16 tiles;
 * sizes per tile from vsde_mlp_image_bytes; layouts in csrc/vsde_mlp.hip, built by primitives/fused.py::MlpImages):
 *   w1_img [T][32 rows][C + 8] bf16 (+ padding to whole KB): row 8 g + 4 h + i of tile t = (g < 2 ? a : b) unit 16 t + 8 h + 4 (g & 1) + i
 *   w2_img [T][2][C][8] bf16: W_out[n][16 t + 8 h + 0..7];   b1_img [T][64] fp32: b_in in w1_img's row order (32 used)
 * s_out (training only, else NULL): silu(a) * b [M][lds] bf16 in natural unit order, what the weight gradient of W_out needs; the
 * pre-activations are never written (the backward recomputes them). */
/* Block form (no-grad sampling): the gated residuals and modulated LayerNorms on either side of the MLP are the kernel's prologue
 * and epilogue (reference primitives/sit.py:112-128 under autocast, bf16 roundings where the unfused chain has them):
 *   x1 = x + ga * yin;  tok = x1 + gm * mlp(LN(x1) (1 + sc) + sh);  hnext = LN(tok) (1 + sn) + hs   (sn / hs / hnext NULL: last block)
 * x, yin, tok, hnext [M][C] bf16 contiguous; ga .. hs per-batch-row vectors [B][mp] bf16, batch row of row m = m / tokens. */
int vsde_mlp_block_fwd_bf16(const void *x, const void *yin, const void *ga, const void *sc, const void *sh, const void *gm,
                            const void *sn, const void *hs, int64_t mp, int tokens, double eps, double eps_next, const void *w1_img,
                            const void *w2_img, const float *b1_img, const void *b2, void *tok, void *hnext, int64_t M, int C, int H,
                            void *stream);
/* Block form with the attention branch's out projection as the prologue's first product (reference primitives/attn.py:107-110 +
 * sit.py:112-128):  yin = (attn * sigmoid(glog[:, k % 64])) W_o^T + b_o, then as vsde_mlp_block_fwd_bf16.  attn [M][C] bf16 (merged
 * heads, token-major), glog [M][ldg] bf16 (64 gate logits per row), wo_img = W_o as C / 16 tiles in w2_img's format, bo [C] bf16 or
 * NULL.  tok doubles as the scratch that holds x1 between the prologue and the epilogue: it must not alias x. */
int vsde_mlp_attn_block_fwd_bf16(const void *x, const void *attn, const void *glog, int64_t ldg, const void *wo_img, const void *bo,
                                 const void *ga, const void *sc, const void *sh, const void *gm, const void *sn, const void *hs,
                                 int64_t mp, int tokens, double eps, double eps_next, const void *w1_img, const void *w2_img,
                                 const float *b1_img, const void *b2, void *tok, void *hnext, int64_t M, int C, int H, void *stream);
/* y [M][256] = x [M][K] W^T (+ bias) for a DEEP reduction (K % 64 == 0, K >= 256) at the encoder's width 256 (csrc/vsde_mlp.hip,
 * round 5): the SwiGLU output projection (primitives/mlp.py:54) and the input-gradient GEMMs of mlp.py:50 / attn.py:80-82 -- the
 * three products the library ran until then.  w_img = W as K / 16 k-step images [2][256][8] bf16 (W[n][16 t + 8 h + 0..7]: the
 * layout of w2_img above; built by primitives/fused.py::DeepImage), bias [256] bf16 or NULL. */
int vsde_linear_deep256_bf16(const void *x, int64_t ldx, const void *w_img, const void *bias, void *y, int64_t ldy, int64_t M, int K,
                             void *stream);
/* Backward of the SwiGLU MLP in one pass (training step): du [M][2 H] = swiglu'(u) * (dy W_out) and dx [M][C] = du W_in, with the saved
 * pre-activations u and du in the 16-row interleaved layout of primitives/fused.py::swiglu_packs(interleave=True) (64 columns per 32
 * hidden units: [a16 | b16 | a16 | b16]).  img: H / 32 pair-tile images of vsde_mlp_bwd_image_bytes(C) bytes each (layout in
 * csrc/vsde_mlp.hip, built by primitives/fused.py::MlpBwdImages).  Replaces vsde_linear_bf16(EPI_SWIGLU_BWD) + the dx GEMM over du. */
int64_t vsde_mlp_bwd_image_bytes(int C);
int vsde_mlp_bwd_bf16(const void *dy, int64_t lddy, const void *u, int64_t ldu, const void *img, void *du, int64_t lddu, void *dx,
                      int64_t lddx, int64_t M, int C, int H, void *stream);
/* debugging aid (VSDE_MLP_DEBUG=16): device buffer that receives workgroup 0's per-phase cycle stamps */
int vsde_mlp_debug_trace(void *buf);
/* The same for the weight-gradient kernel (eight-wave TN = 256 form): [8 waves][4 phase cycle sums + step count] (tools/wgrad_trace.py). */
int vsde_wgrad_debug_trace(void *buf);
/* ... and for the persistent LDS-resident attention forward: [12 waves][4 phase cycle sums + pair count] (tools/attn_trace.py). */
int vsde_attn_debug_trace(void *buf);
int vsde_mlp_image_bytes(int C, int64_t *w1_tile, int64_t *w2_tile, int64_t *b1_tile);
int vsde_mlp_fwd_bf16(const void *x, int64_t ldx, const void *w1_img, const void *w2_img, const float *b1_img, const void *b2, void *y,
                      int64_t ldy, void *s_out, int64_t lds, int64_t M, int C, int H, void *stream);

/* ---- The optimizer step as two launches over all parameters -----------------------------------------------------------
 * Replaces, per training step (inference/trainer.py:197-204, inference/exponential_moving_average.py:27-32):
 * scaler.unscale_ (non-finite check + g *= 1/scale), clip_grad_norm_ (norms + g *= clip coefficient), the multi-tensor AdamW
 * (torch.optim.AdamW: decoupled weight decay, bias corrections) and the EMA lerp -- with the same arithmetic.
 * chunks: device array of n_chunks records of vsde_optim_chunk_bytes() = 64 bytes:
 *   { float *p, *m, *v, *ema; int32 param, n; int64 goff; int32 group, 0; int64 0 }   (<= vsde_optim_chunk_elems() elements each)
 * grads: device array of the step's gradient base pointers (one per parameter); scale: loss scale (device scalar) or NULL;
 * partials [n_chunks] scratch; tstate [2] = (t_cur, t_next) float step counts, t_next is the persistent one;
 * groups [n_groups][5] doubles = lr, beta1, beta2, eps, weight_decay; max_norm <= 0: no clipping; ema_weight = 1 - decay (< 0: no EMA);
 * out [2] = global gradient norm (unscaled, before clipping), found_inf (0 / 1).  With a loss scale, a non-finite gradient
 * skips the update (parameters, moments and step count unchanged, as GradScaler.step does; the EMA lerp still runs, as the
 * reference's ema.update() does).  Deterministic. */
int vsde_optim_chunk_bytes(void);
int vsde_optim_chunk_elems(void);
int vsde_optim_step(const void *chunks, int n_chunks, const void *grads, const float *scale, float *partials, float *tstate,
                    const double *groups, double max_norm, double ema_weight, float *out, void *stream);

/* ---- Refresh of the cached bf16 GEMM operands after an optimizer step -------------------------------------------------
 * The reference re-casts each nn.Linear weight to bf16 in every forward under autocast (primitives/attn.py:46-54,
 * primitives/mlp.py:41-54); this build keeps the bf16 operands (concatenated / padded / interleaved packs and their
 * transposes) as persistent buffers and re-fills ALL of them with one launch after the optimizer step.
 * tiles: device array of n_tiles records of vsde_pack_tile_bytes() = 64 bytes each:
 *   { const float *src; uint16_t *dst; uint16_t *dst_t; int64_t src_pitch, dst_pitch, pitch_t; int32_t rows, cols; int64_t reserved; }
 * = up to 16 rows x cols of one fp32 parameter row block -> bf16 at dst (row pitch dst_pitch) and, when dst_t != NULL, the
 * same values transposed: dst_t[col * pitch_t + row]. */
int vsde_pack_tile_bytes(void);
int vsde_pack_refresh(const void *tiles, int n_tiles, void *stream);

/* ---- Batched Euler-Maruyama simulator of the MODEL SDE (parameter pre-training stage) --------------------------------
 * Replaces the T-step Python loop of core/euler_maruyama.py:11-45 (called from trainer.py:246-259 with 4096 paths) for
 * SDEs whose drift / diffusion are built in:
 *   VSDE_SDE_OU               f = kappa (mu - x), G = sigma                        examples/ornstein_uhlenbeck.py:18-30
 *   VSDE_SDE_LOTKA_VOLTERRA   analytic 2x2 Cholesky diffusion, three 1e-6 clamps   examples/lotka_volterra.py:18-46
 *   VSDE_SDE_LINEAR_DIAGONAL  f = -a x, G = diag(softplus(b) + 1e-3), theta=(a,b)  (BASELINE.json config 5)
 * x0[B][S], theta[B][P], noise[B][T][S] -> traj[B][T+1][S] with traj[:,0] = x0 and the positive dims (positive_mask: S
 * bytes on the HOST) clamped at 1e-6 after every step.  _bwd: g_traj[B][T+1][S] -> g_x0[B][S], g_theta[B][P] (reverse-mode
 * derivative of exactly that recursion; a clamped entry passes no gradient).  User-defined SDEs keep the torch loop. */
#define VSDE_SDE_OU 1
#define VSDE_SDE_LOTKA_VOLTERRA 2
#define VSDE_SDE_LINEAR_DIAGONAL 3
int vsde_euler_maruyama_fwd(int kind, int B, int T, int S, int P, const float *x0, const float *theta, const float *noise,
                            double time_step, const uint8_t *positive_mask_host, float *traj, void *stream);
int vsde_euler_maruyama_bwd(int kind, int B, int T, int S, int P, const float *theta, const float *noise, const float *traj,
                            const float *g_traj, double time_step, const uint8_t *positive_mask_host, float *g_x0,
                            float *g_theta, void *stream);

/* The [B]-sized tail of the ELBO (inference/evidence_lower_bound.py:52-83): Gaussian observation log-density of the states at
 * the K observed grid points x_obs[B][K][S] (core/observations.py:57-74; obs_matrix [O][S] or NULL = identity), iid prior
 * (prior_type 0 Normal, 1 LogNormal; core/priors.py:46-60), mean-field posterior log q(theta)
 * (models/sde_parameter_posterior.py:44-66; theta_positive_mask_host: P bytes), combined with the three path terms into
 *   out6 = [mean_b(obs + sde - gen + jac + prior - post), mean obs, mean sde, mean gen, mean prior, mean post].
 * _bwd: upstream gradient of out6 -> gradients of x_obs, theta, the posterior's mean / log_std [P] and the three path terms
 * [B].  One single-workgroup kernel each; dims S, O, P <= 16. */
int vsde_elbo_tail_fwd(int B, int K, int S, int O, int P, const float *x_obs, const float *obs_values, const float *obs_matrix,
                       double variance, const float *theta, int prior_type, double prior_mean, double prior_std,
                       const float *post_mean, const float *post_log_std, const uint8_t *theta_positive_mask_host,
                       const float *sde_lp, const float *gen_lp, const float *log_jac, float *out6, void *stream);
int vsde_elbo_tail_bwd(int B, int K, int S, int O, int P, const float *x_obs, const float *obs_values, const float *obs_matrix,
                       double variance, const float *theta, int prior_type, double prior_mean, double prior_std,
                       const float *post_mean, const float *post_log_std, const uint8_t *theta_positive_mask_host,
                       const float *g_out6, float *g_x_obs, float *g_theta, float *g_post_mean, float *g_post_log_std,
                       float *g_sde, float *g_gen, float *g_jac, void *stream);

/* Drift and diffusion factor of a built-in SDE on every grid point of a batch of paths -- what the ELBO evaluates through the
 * user's Python callables on the flattened [(B T), S] states (inference/evidence_lower_bound.py:37-40) -- and the
 * vector-Jacobian product its backward needs.  x[B][T+1][S] (rows 0..T-1 are read), theta[B][P] -> drift[B][T][S],
 * diffusion[B][T][S][S].  _bwd: g_drift, g_diffusion -> g_x[B][T+1][S] (row T zero), g_theta[B][P] (sum over t in a fixed
 * order).  torch.clamp(min=1e-6) semantics: the gradient passes where the clamped quantity is >= the bound. */
int vsde_sde_coefficients_fwd(int kind, int B, int T, int S, int P, const float *x, const float *theta, float *drift,
                              float *diffusion, void *stream);
int vsde_sde_coefficients_bwd(int kind, int B, int T, int S, int P, const float *x, const float *theta, const float *g_drift,
                              const float *g_diffusion, float *g_x, float *g_theta, void *stream);

/* Measurement aid (no reference counterpart): when enabled, the launchers bracket their kernels with hipEvents on the
 * launch stream.  which: 0 = serial time-stepping forward kernel (training variant), 1 = serial backward kernel,
 * 2 = everything vsde_head_forward enqueues (training variant), 3 = everything vsde_head_backward enqueues,
 * 4 = the forward's context-projection GEMM, 5 = the backward's grad_context GEMM, 6 = the grouped weight-gradient
 * reduction.  vsde_profile_elapsed_ms waits for the end event of the LAST such launch. */
int vsde_profile_enable(int on);
/* Test hook: route L <= 2 through the LDS-resident one-wave-per-path kernels that serve L = 3, 4. */
int vsde_debug_force_v1(int on);
/* Which forward time-stepping kernel serves hidden_dim 64 / L <= 2 / state_dim <= 2 (reference kernels/forward.py:137-375):
 * mode 1 = the multi-path MFMA kernel (16 paths per workgroup, csrc/vsde_head_mp.hip) whenever applicable, 0 = never (the
 * four-waves-per-path kernel), < 0 = default (environment VSDE_HEAD_MP, else by batch size). */
int vsde_debug_head_mp(int mode);
/* 1 once a GRU head weight has left the f16 range of the multi-path MFMA kernels (|W| > 2.2e4; the weight preparation of every such
 * launch checks): the launch that found it returns NON-FINITE paths (never silently wrong ones), and every later launch of the process
 * takes the fp32 four-waves-per-path kernels.  The flag lives in host-mapped memory and is written in stream order: a caller that
 * REPLAYS a captured graph (whose kernel choice was fixed at capture) polls it after a replay and re-captures / steps eagerly.
 * mode > 0 clears the flag (tests).  No reference counterpart (the reference's Triton kernels are fp32 throughout). */
int vsde_head_mfma_range_exceeded(int clear);
int vsde_profile_elapsed_ms(int which, float *ms);

#ifdef __cplusplus
}
#endif
#endif /* VSDE_HIP_H */
