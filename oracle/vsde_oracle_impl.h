/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the variational-SDE hot path.
 *
 * This file is a plain-C restatement of the arithmetic of the reference
 * (Tom-Ryder/VIforSDEs, snapshot under /root/reference) for
 *   - the fused GRU DiffusionTransitionHead time-stepping forward
 *       (src/variational_sde/kernels/forward.py:137-375),
 *   - its hand-derived reverse-time backward
 *       (src/variational_sde/kernels/backward.py:208-624),
 *   - the per-sample ELBO terms
 *       (src/variational_sde/inference/evidence_lower_bound.py:29-83,
 *        src/variational_sde/inference/state_space.py:20-38,
 *        src/variational_sde/core/observations.py:52-74).
 *
 * It is included twice by vsde_oracle.c, once with REAL=float (suffix _f32)
 * and once with REAL=double (suffix _f64).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may call it; the shipped package never does.
 *
 * Parity pin: checked against golden vectors produced by importing the
 * reference's own eager per-step definition (models/head.py:68-97) and its
 * autograd, see tests/golden/make_golden.py and tests/test_oracle_golden.py.
 *
 * Weight layout is torch.nn.GRU's native one (gate-major rows r|z|n):
 *   W_ih0[3H][I] with I = S + C + P and input order [state | context | theta]
 *   (models/head.py:75), W_hh0[3H][H], stacks [L-1][3H][H], out_W[S+ntril][H].
 * Saved activations: acts[B][T][L][5][H] with slots (h, r, u, n, c_n) where
 *   c_n = b_hn + W_hn.h_prev is the reference's "n_hh" (forward.py:233,255).
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUFFIX)

static inline REAL FN(sigm)(REAL x) { return (REAL)1 / ((REAL)1 + EXP(-x)); }

/* ------------------------------------------------------------------ forward */
/* forward.py:137-375 (Appendix A.1 of SURVEY.md).  ctx rows are C contiguous
 * values; batch b, step t lives at ctx + b*ctx_bstride + t*C so the caller can
 * pass the non-contiguous context[:, :-1] view (diffusion_path_sampler.py:61). */
void FN(vsde_oracle_fwd)(
    int B, int T, int S, int P, int C, int H, int L,
    const REAL *x0, const REAL *ctx, long ctx_bstride, const REAL *theta, const REAL *eps,
    const REAL *W_ih0, const REAL *W_hh0, const REAL *b_ih0, const REAL *b_hh0,
    const REAL *W_ih_st, const REAL *W_hh_st, const REAL *b_ih_st, const REAL *b_hh_st,
    const REAL *out_W, const REAL *out_b, double dt_d, double diag_min_d,
    REAL *paths, REAL *means, REAL *chol, /* outputs */
    REAL *chol_raw, REAL *acts /* optional (NULL = eval mode, forward.py SAVE_ACTIVATIONS) */)
{
    const int I = S + C + P, G3 = 3 * H, ntril = S * (S + 1) / 2, NO = S + ntril;
    const REAL dt = (REAL)dt_d, sqdt = (REAL)sqrt(dt_d), diag_min = (REAL)diag_min_d;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        REAL *h = (REAL *)calloc((size_t)L * H, sizeof(REAL));      /* forward.py:143,408-414 */
        REAL *gth = (REAL *)malloc(sizeof(REAL) * G3);
        REAL *a = (REAL *)malloc(sizeof(REAL) * G3), *c = (REAL *)malloc(sizeof(REAL) * G3);
        REAL *z = (REAL *)malloc(sizeof(REAL) * S), *o = (REAL *)malloc(sizeof(REAL) * NO);
        REAL *hin = (REAL *)malloc(sizeof(REAL) * H), *le = (REAL *)malloc(sizeof(REAL) * S);
        /* hoisted theta projection, forward.py:157-175 */
        for (int row = 0; row < G3; ++row) {
            REAL acc = 0;
            for (int p = 0; p < P; ++p) acc += theta[(long)b * P + p] * W_ih0[(long)row * I + S + C + p];
            gth[row] = acc;
        }
        for (int i = 0; i < S; ++i) { z[i] = x0[(long)b * S + i]; paths[((long)b * (T + 1)) * S + i] = z[i]; }
        for (int t = 0; t < T; ++t) {
            const REAL *cx = ctx + (long)b * ctx_bstride + (long)t * C;
            /* layer 0 pre-activations, forward.py:195-233 */
            for (int row = 0; row < G3; ++row) {
                REAL xs = 0, cs = 0, hs = 0;
                const REAL *w = W_ih0 + (long)row * I;
                for (int i = 0; i < S; ++i) xs += z[i] * w[i];
                for (int k = 0; k < C; ++k) cs += cx[k] * w[S + k];
                for (int j = 0; j < H; ++j) hs += h[j] * W_hh0[(long)row * H + j];
                a[row] = ((b_ih0[row] + gth[row]) + xs) + cs;
                c[row] = b_hh0[row] + hs;
            }
            for (int l = 0; l < L; ++l) {
                REAL *hl = h + (long)l * H;
                if (l > 0) { /* gru_cell_standard, forward.py:33-88 */
                    const REAL *Wi = W_ih_st + (long)(l - 1) * G3 * H, *Wh = W_hh_st + (long)(l - 1) * G3 * H;
                    const REAL *bi = b_ih_st + (long)(l - 1) * G3, *bh = b_hh_st + (long)(l - 1) * G3;
                    for (int row = 0; row < G3; ++row) {
                        REAL is = 0, hs = 0;
                        for (int j = 0; j < H; ++j) { is += hin[j] * Wi[(long)row * H + j]; hs += hl[j] * Wh[(long)row * H + j]; }
                        a[row] = bi[row] + is;
                        c[row] = bh[row] + hs;
                    }
                }
                for (int j = 0; j < H; ++j) { /* forward.py:235-238 */
                    REAL r = FN(sigm)(a[j] + c[j]);
                    REAL u = FN(sigm)(a[H + j] + c[H + j]);
                    REAL n = TANH(a[2 * H + j] + r * c[2 * H + j]);
                    REAL hn = ((REAL)1 - u) * n + u * hl[j];
                    if (acts) {
                        REAL *A = acts + ((((long)b * T + t) * L + l) * 5) * H;
                        A[0 * H + j] = hn; A[1 * H + j] = r; A[2 * H + j] = u; A[3 * H + j] = n; A[4 * H + j] = c[2 * H + j];
                    }
                    hin[j] = hn;
                }
                for (int j = 0; j < H; ++j) hl[j] = hin[j];
            }
            /* emission, forward.py:314-362 */
            for (int k = 0; k < NO; ++k) {
                REAL acc = 0;
                for (int j = 0; j < H; ++j) acc += hin[j] * out_W[(long)k * H + j];
                o[k] = out_b[k] + acc;
            }
            REAL *Lt = chol + (((long)b * T + t) * S) * S;
            for (int i = 0; i < S * S; ++i) Lt[i] = 0; /* forward.py:404-406 */
            const REAL *e = eps + ((long)b * T + t) * S;
            int k = 0;
            for (int i = 0; i < S; ++i) {
                REAL row_sum = 0;
                for (int j = 0; j <= i; ++j, ++k) {
                    REAL raw = o[S + k];
                    REAL lij = (i == j) ? (raw > diag_min ? raw : diag_min) : raw; /* forward.py:346-351 */
                    Lt[i * S + j] = lij;
                    if (chol_raw) chol_raw[((long)b * T + t) * ntril + k] = raw;
                    row_sum += lij * e[j];
                }
                le[i] = row_sum;
            }
            for (int i = 0; i < S; ++i) {
                means[((long)b * T + t) * S + i] = o[i];
                z[i] = z[i] + o[i] * dt + le[i] * sqdt; /* forward.py:365 */
                paths[((long)b * (T + 1) + t + 1) * S + i] = z[i];
            }
        }
        free(h); free(gth); free(a); free(c); free(z); free(o); free(hin); free(le);
    }
}

/* ----------------------------------------------------------------- backward */
/* backward.py:208-624 (Appendix A.2).  All weight gradients come back in the
 * nn.GRU native layout, i.e. after the re-transpose of backward.py:766-784. */
void FN(vsde_oracle_bwd)(
    int B, int T, int S, int P, int C, int H, int L,
    const REAL *g_paths, const REAL *g_means, const REAL *g_chol,
    const REAL *ctx, long ctx_bstride, const REAL *theta, const REAL *eps,
    const REAL *paths, const REAL *chol_raw, const REAL *acts,
    const REAL *W_ih0, const REAL *W_hh0, const REAL *W_ih_st, const REAL *W_hh_st, const REAL *out_W,
    double dt_d, double diag_min_d,
    REAL *g_x0, REAL *g_ctx /* [B][T][C] contiguous */, REAL *g_theta,
    REAL *gW_ih0, REAL *gW_hh0, REAL *gb_ih0, REAL *gb_hh0,
    REAL *gW_ih_st, REAL *gW_hh_st, REAL *gb_ih_st, REAL *gb_hh_st,
    REAL *g_out_W, REAL *g_out_b)
{
    const int I = S + C + P, G3 = 3 * H, ntril = S * (S + 1) / 2, NO = S + ntril;
    const REAL dt = (REAL)dt_d, sqdt = (REAL)sqrt(dt_d), diag_min = (REAL)diag_min_d;
    memset(gW_ih0, 0, sizeof(REAL) * G3 * I); memset(gW_hh0, 0, sizeof(REAL) * G3 * H);
    memset(gb_ih0, 0, sizeof(REAL) * G3); memset(gb_hh0, 0, sizeof(REAL) * G3);
    if (L > 1) {
        memset(gW_ih_st, 0, sizeof(REAL) * (L - 1) * G3 * H); memset(gW_hh_st, 0, sizeof(REAL) * (L - 1) * G3 * H);
        memset(gb_ih_st, 0, sizeof(REAL) * (L - 1) * G3); memset(gb_hh_st, 0, sizeof(REAL) * (L - 1) * G3);
    }
    memset(g_out_W, 0, sizeof(REAL) * NO * H); memset(g_out_b, 0, sizeof(REAL) * NO);
    /* The reference accumulates weight gradients with order-nondeterministic atomics
     * (backward.py:108-139).  Here every OpenMP thread owns a private copy of all
     * weight-gradient buffers and the copies are summed in thread order at the end,
     * so the result is deterministic for a fixed thread count. */
    const long nWi0 = (long)G3 * I, nWh0 = (long)G3 * H, nSt = (long)(L > 1 ? (L - 1) : 0) * G3 * H;
    const long nBs = (long)(L > 1 ? (L - 1) : 0) * G3, nOW = (long)NO * H;
    const long oWi0 = 0, oWh0 = oWi0 + nWi0, oBi0 = oWh0 + nWh0, oBh0 = oBi0 + G3, oWis = oBh0 + G3,
               oWhs = oWis + nSt, oBis = oWhs + nSt, oBhs = oBis + nBs, oOW = oBhs + nBs, oOb = oOW + nOW,
               nTot = oOb + NO;
    REAL *outs[10] = {gW_ih0, gW_hh0, gb_ih0, gb_hh0, gW_ih_st, gW_hh_st, gb_ih_st, gb_hh_st, g_out_W, g_out_b};
    const long offs[11] = {oWi0, oWh0, oBi0, oBh0, oWis, oWhs, oBis, oBhs, oOW, oOb, nTot};
#pragma omp parallel
    {
    REAL *priv = (REAL *)calloc((size_t)nTot, sizeof(REAL));
    REAL *gW_ih0 = priv + oWi0, *gW_hh0 = priv + oWh0, *gb_ih0 = priv + oBi0, *gb_hh0 = priv + oBh0;
    REAL *gW_ih_st = priv + oWis, *gW_hh_st = priv + oWhs, *gb_ih_st = priv + oBis, *gb_hh_st = priv + oBhs;
    REAL *g_out_W = priv + oOW, *g_out_b = priv + oOb;
    REAL *dh = (REAL *)malloc(sizeof(REAL) * L * H);
    REAL *dx = (REAL *)malloc(sizeof(REAL) * S), *dO = (REAL *)malloc(sizeof(REAL) * NO);
    REAL *dcur = (REAL *)malloc(sizeof(REAL) * H), *dinp = (REAL *)malloc(sizeof(REAL) * H);
    REAL *pi = (REAL *)malloc(sizeof(REAL) * G3), *ph = (REAL *)malloc(sizeof(REAL) * G3);
    REAL *zero = (REAL *)calloc(H, sizeof(REAL));
#pragma omp for schedule(static)
    for (int b = 0; b < B; ++b) {
        for (int i = 0; i < S; ++i) dx[i] = 0;
        for (int i = 0; i < L * H; ++i) dh[i] = 0;
        for (int p = 0; p < P; ++p) g_theta[(long)b * P + p] = 0;
        for (int t = T - 1; t >= 0; --t) {
            const REAL *e = eps + ((long)b * T + t) * S;
            const REAL *A = acts + (((long)b * T + t) * L) * 5 * H;
            const REAL *Aprev = (t > 0) ? acts + (((long)b * T + t - 1) * L) * 5 * H : NULL;
            for (int i = 0; i < S; ++i) {
                dx[i] += g_paths[((long)b * (T + 1) + t + 1) * S + i];           /* backward.py:278 */
                dO[i] = dx[i] * dt + g_means[((long)b * T + t) * S + i];         /* backward.py:279 */
            }
            int k = 0;
            for (int i = 0; i < S; ++i)
                for (int j = 0; j <= i; ++j, ++k) {
                    REAL dL = dx[i] * e[j] * sqdt + g_chol[(((long)b * T + t) * S + i) * S + j]; /* :324 */
                    if (i == j) {
                        REAL raw = chol_raw[((long)b * T + t) * ntril + k];
                        if (!(raw >= diag_min || dL < 0)) dL = 0;                  /* :331-334, bounds.py:20 */
                    }
                    dO[S + k] = dL;
                }
            const REAL *htop = A + ((long)(L - 1) * 5) * H;
            for (int j = 0; j < H; ++j) dcur[j] = 0;
            for (int q = 0; q < NO; ++q) {                                          /* :296-349 */
                g_out_b[q] += dO[q];
                for (int j = 0; j < H; ++j) { g_out_W[(long)q * H + j] += dO[q] * htop[j]; dcur[j] += dO[q] * out_W[(long)q * H + j]; }
            }
            for (int l = L - 1; l >= 0; --l) {
                const REAL *Al = A + ((long)l * 5) * H;
                const REAL *r = Al + H, *u = Al + 2 * H, *n = Al + 3 * H, *cn = Al + 4 * H;
                const REAL *hprev = Aprev ? Aprev + ((long)l * 5) * H : zero;     /* :382-393,470-477 */
                REAL *dhl = dh + (long)l * H;
                for (int j = 0; j < H; ++j) {
                    REAL d = dcur[j] + dhl[j];                                      /* :360-367, :452-455 */
                    REAL dn = ((REAL)1 - u[j]) * d, du = (hprev[j] - n[j]) * d;     /* :59-61 */
                    REAL dn_pre = dn * ((REAL)1 - n[j] * n[j]);
                    REAL du_pre = du * (u[j] * ((REAL)1 - u[j]));
                    REAL dcn = dn_pre * r[j];
                    REAL dr_pre = (dn_pre * cn[j]) * (r[j] * ((REAL)1 - r[j]));     /* :63-67 */
                    pi[j] = dr_pre; pi[H + j] = du_pre; pi[2 * H + j] = dn_pre;
                    ph[j] = dr_pre; ph[H + j] = du_pre; ph[2 * H + j] = dcn;
                    dhl[j] = u[j] * d;                                              /* carry u.dh */
                }
                const REAL *Wh = (l == 0) ? W_hh0 : W_hh_st + (long)(l - 1) * G3 * H;
                REAL *gWh = (l == 0) ? gW_hh0 : gW_hh_st + (long)(l - 1) * G3 * H;
                REAL *gbi = (l == 0) ? gb_ih0 : gb_ih_st + (long)(l - 1) * G3;
                REAL *gbh = (l == 0) ? gb_hh0 : gb_hh_st + (long)(l - 1) * G3;
                for (int row = 0; row < G3; ++row) {
                    gbi[row] += pi[row]; gbh[row] += ph[row];                        /* :141-151, :592-618 */
                    for (int j = 0; j < H; ++j) {
                        dhl[j] += Wh[(long)row * H + j] * ph[row];                   /* :96-105, :566-573 */
                        gWh[(long)row * H + j] += ph[row] * hprev[j];                /* :124-139, :575-590 */
                    }
                }
                if (l > 0) {
                    const REAL *Wi = W_ih_st + (long)(l - 1) * G3 * H;
                    REAL *gWi = gW_ih_st + (long)(l - 1) * G3 * H;
                    const REAL *hinp = A + ((long)(l - 1) * 5) * H;
                    for (int j = 0; j < H; ++j) dinp[j] = 0;
                    for (int row = 0; row < G3; ++row)
                        for (int j = 0; j < H; ++j) {
                            dinp[j] += Wi[(long)row * H + j] * pi[row];              /* :83-94 */
                            gWi[(long)row * H + j] += pi[row] * hinp[j];             /* :107-122 */
                        }
                    for (int j = 0; j < H; ++j) dcur[j] = dinp[j];
                } else {
                    const REAL *zt = paths + ((long)b * (T + 1) + t) * S;
                    const REAL *cx = ctx + (long)b * ctx_bstride + (long)t * C;
                    REAL *gc = g_ctx + ((long)b * T + t) * C;
                    for (int q = 0; q < C; ++q) gc[q] = 0;
                    for (int row = 0; row < G3; ++row) {
                        const REAL *w = W_ih0 + (long)row * I;
                        REAL *gw = gW_ih0 + (long)row * I;
                        for (int i = 0; i < S; ++i) { dx[i] += w[i] * pi[row]; gw[i] += pi[row] * zt[i]; }             /* :494-509 */
                        for (int q = 0; q < C; ++q) { gc[q] += w[S + q] * pi[row]; gw[S + q] += pi[row] * cx[q]; }      /* :550-564 */
                        for (int p = 0; p < P; ++p) {                                                                  /* :511-548 */
                            g_theta[(long)b * P + p] += w[S + C + p] * pi[row];
                            gw[S + C + p] += pi[row] * theta[(long)b * P + p];
                        }
                    }
                }
            }
        }
        for (int i = 0; i < S; ++i) g_x0[(long)b * S + i] = dx[i] + g_paths[((long)b * (T + 1)) * S + i]; /* :620-624 */
    }
    free(dh); free(dx); free(dO); free(dcur); free(dinp); free(pi); free(ph); free(zero);
#pragma omp for ordered schedule(static, 1)
    for (int th = 0; th < omp_get_num_threads(); ++th) {
#pragma omp ordered
        for (int q = 0; q < 10; ++q)
            for (long i = 0; i < offs[q + 1] - offs[q]; ++i) outs[q][i] += priv[offs[q] + i];
    }
    free(priv);
    }
}

/* --------------------------------------------------------------- ELBO terms */
/* log N(y; m, A A^T) as torch.distributions.MultivariateNormal(scale_tril=A).log_prob
 * evaluates it (evidence_lower_bound.py:77-83): forward substitution, then
 * -1/2 |w|^2 - sum log A_ii - S/2 log 2pi.  w is returned for the backward. */
static REAL FN(tri_logpdf)(int S, const REAL *y, const REAL *m, const REAL *A, REAL scale, REAL *w)
{
    REAL quad = 0, logdet = 0;
    for (int i = 0; i < S; ++i) {
        REAL acc = y[i] - m[i];
        for (int j = 0; j < i; ++j) acc -= (A[i * S + j] * scale) * w[j];
        w[i] = acc / (A[i * S + i] * scale);
        quad += w[i] * w[i];
        logdet += LOG(A[i * S + i] * scale);
    }
    return (REAL)(-0.5) * ((REAL)S * (REAL)1.8378770664093453 + quad) - logdet;
}

/* Per-sample path terms of the ELBO (evidence_lower_bound.py:29-50, types.py:23-24,
 * state_space.py:35-38):
 *   sde_lp[b] = sum_t log N(x_{t+1}; x_t + f dt, (G sqrt dt)(G sqrt dt)^T)
 *   gen_lp[b] = sum_t log N(z_{t+1}; z_t + mu dt, (L sqrt dt)(L sqrt dt)^T)
 *   jac[b]    = sum_{t>=1} sum_{d in pos} log sigmoid(z_{t,d})
 * x is passed in (x = StateSpace.to_state(z), state_space.py:20-25). */
void FN(vsde_oracle_elbo_path_terms)(
    int B, int T, int S, const REAL *z, const REAL *x, const REAL *means, const REAL *chol,
    const REAL *drift, const REAL *diffusion, const unsigned char *pos_mask, double dt_d,
    REAL *sde_lp, REAL *gen_lp, REAL *jac)
{
    const REAL dt = (REAL)dt_d, sqdt = (REAL)POW(dt_d, 0.5);
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        REAL w[64], m[64];
        REAL s_acc = 0, g_acc = 0, j_acc = 0;
        for (int t = 0; t < T; ++t) {
            const REAL *zt = z + ((long)b * (T + 1) + t) * S, *zn = zt + S;
            const REAL *xt = x + ((long)b * (T + 1) + t) * S, *xn = xt + S;
            const REAL *f = drift + ((long)b * T + t) * S, *mu = means + ((long)b * T + t) * S;
            const REAL *G = diffusion + (((long)b * T + t) * S) * S, *Lc = chol + (((long)b * T + t) * S) * S;
            for (int i = 0; i < S; ++i) m[i] = xt[i] + f[i] * dt;
            s_acc += FN(tri_logpdf)(S, xn, m, G, sqdt, w);
            for (int i = 0; i < S; ++i) m[i] = zt[i] + mu[i] * dt;
            g_acc += FN(tri_logpdf)(S, zn, m, Lc, sqdt, w);
            for (int i = 0; i < S; ++i)
                if (pos_mask[i]) { /* logsigmoid(v) = min(v,0) - log1p(exp(-|v|)) */
                    REAL v = zn[i];
                    j_acc += (v < 0 ? v : 0) - LOG1P(EXP(-(v < 0 ? -v : v)));
                }
        }
        sde_lp[b] = s_acc; gen_lp[b] = g_acc; jac[b] = j_acc;
    }
}

/* Analytic adjoint of the three path terms.  For lp = log N(y; m, (A s)(A s)^T) with
 * w = (A s)^-1 (y-m), v = (A s)^-T w:   d/dy = -v, d/dm = +v,
 * d/dA = s * tril(v w^T) - diag(1/A_ii).  Upstream per-sample gradients g_sde, g_gen,
 * g_jac come from autograd of the [B]-vector combination (evidence_lower_bound.py:63-66). */
static void FN(tri_logpdf_bwd)(int S, const REAL *y, const REAL *m, const REAL *A, REAL scale, REAL g,
                               REAL *gy, REAL *gm, REAL *gA, REAL *w, REAL *v)
{
    (void)FN(tri_logpdf)(S, y, m, A, scale, w);
    for (int i = S - 1; i >= 0; --i) {
        REAL acc = w[i];
        for (int j = i + 1; j < S; ++j) acc -= (A[j * S + i] * scale) * v[j];
        v[i] = acc / (A[i * S + i] * scale);
    }
    for (int i = 0; i < S; ++i) {
        gy[i] += -g * v[i];
        gm[i] = g * v[i];
        for (int j = 0; j < S; ++j) gA[i * S + j] = 0;
        for (int j = 0; j <= i; ++j) gA[i * S + j] = g * scale * v[i] * w[j];
        gA[i * S + i] -= g / A[i * S + i];
    }
}

void FN(vsde_oracle_elbo_path_terms_bwd)(
    int B, int T, int S, const REAL *z, const REAL *x, const REAL *means, const REAL *chol,
    const REAL *drift, const REAL *diffusion, const unsigned char *pos_mask, double dt_d,
    const REAL *g_sde, const REAL *g_gen, const REAL *g_jac,
    REAL *g_z, REAL *g_x, REAL *g_means, REAL *g_chol, REAL *g_drift, REAL *g_diffusion)
{
    const REAL dt = (REAL)dt_d, sqdt = (REAL)POW(dt_d, 0.5);
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        REAL w[64], v[64], m[64], gm[64];
        for (int i = 0; i < (T + 1) * S; ++i) { g_z[(long)b * (T + 1) * S + i] = 0; g_x[(long)b * (T + 1) * S + i] = 0; }
        for (int t = 0; t < T; ++t) {
            const long o1 = ((long)b * (T + 1) + t) * S, o2 = ((long)b * T + t) * S;
            for (int i = 0; i < S; ++i) m[i] = x[o1 + i] + drift[o2 + i] * dt;
            FN(tri_logpdf_bwd)(S, x + o1 + S, m, diffusion + o2 * S, sqdt, g_sde[b], g_x + o1 + S, gm, g_diffusion + o2 * S, w, v);
            for (int i = 0; i < S; ++i) { g_x[o1 + i] += gm[i]; g_drift[o2 + i] = gm[i] * dt; }
            for (int i = 0; i < S; ++i) m[i] = z[o1 + i] + means[o2 + i] * dt;
            FN(tri_logpdf_bwd)(S, z + o1 + S, m, chol + o2 * S, sqdt, g_gen[b], g_z + o1 + S, gm, g_chol + o2 * S, w, v);
            for (int i = 0; i < S; ++i) { g_z[o1 + i] += gm[i]; g_means[o2 + i] = gm[i] * dt; }
            for (int i = 0; i < S; ++i)
                if (pos_mask[i]) { REAL zv = z[o1 + S + i]; g_z[o1 + S + i] += g_jac[b] * ((REAL)1 / ((REAL)1 + EXP(zv))); }
        }
    }
}

/* Gaussian observation term, core/observations.py:52-74 with the gather of
 * evidence_lower_bound.py:52-56: obs_lp[b] = sum_k sum_o [-(y-Hx)^2/(2 var) - 1/2 log(2 pi var)].
 * obs_matrix may be NULL (identity, obs_dim == S). */
void FN(vsde_oracle_obs_log_prob)(
    int B, int T, int S, int n_obs, int obs_dim, const REAL *x, const long *obs_idx,
    const REAL *obs_values, const REAL *obs_matrix, double variance, REAL *obs_lp)
{
    const REAL var = (REAL)variance, lognorm = (REAL)(0.5 * log(2.0 * 3.14159265358979323846 * variance));
    for (int b = 0; b < B; ++b) {
        REAL acc = 0;
        for (int k = 0; k < n_obs; ++k) {
            const REAL *xs = x + ((long)b * (T + 1) + obs_idx[k]) * S;
            for (int o = 0; o < obs_dim; ++o) {
                REAL pred;
                if (obs_matrix) { pred = 0; for (int d = 0; d < S; ++d) pred += obs_matrix[o * S + d] * xs[d]; }
                else pred = xs[o];
                REAL diff = obs_values[k * obs_dim + o] - pred;
                acc += (REAL)(-0.5) * (diff * diff) / var - lognorm;
            }
        }
        obs_lp[b] = acc;
    }
}

/* StateSpace.to_state / to_latent, state_space.py:20-33. */
void FN(vsde_oracle_to_state)(long n, int S, const REAL *z, const unsigned char *pos_mask, REAL *x)
{
    for (long r = 0; r < n; ++r)
        for (int i = 0; i < S; ++i) {
            REAL v = z[r * S + i];
            /* F.softplus with beta=1, threshold=20 */
            x[r * S + i] = pos_mask[i] ? (v > (REAL)20 ? v : LOG1P(EXP(v))) : v;
        }
}

void FN(vsde_oracle_to_latent)(long n, int S, const REAL *x, const unsigned char *pos_mask, REAL *z)
{
    for (long r = 0; r < n; ++r)
        for (int i = 0; i < S; ++i) {
            REAL v = x[r * S + i];
            if (pos_mask[i]) { if (v < (REAL)1e-6) v = (REAL)1e-6; v = v + LOG(-EXPM1(-v)); }
            z[r * S + i] = v;
        }
}

/* log p(theta) for the iid Normal / LogNormal prior (core/priors.py:46-60) and
 * log q(theta) of the mean-field posterior (models/sde_parameter_posterior.py:48-59). */
void FN(vsde_oracle_theta_log_probs)(
    int B, int P, const REAL *theta, const REAL *q_mean, const REAL *q_log_std,
    const unsigned char *positive_mask, int prior_is_lognormal, double prior_mean, double prior_std,
    REAL *prior_lp, REAL *post_lp)
{
    const REAL half_log_2pi = (REAL)0.91893853320467274178;
    for (int b = 0; b < B; ++b) {
        REAL pa = 0, qa = 0;
        for (int p = 0; p < P; ++p) {
            REAL th = theta[(long)b * P + p];
            REAL pv = prior_is_lognormal ? LOG(th) : th;
            REAL pz = (pv - (REAL)prior_mean) / (REAL)prior_std;
            pa += (REAL)(-0.5) * pz * pz - LOG((REAL)prior_std) - half_log_2pi - (prior_is_lognormal ? LOG(th) : (REAL)0);
            REAL sd = EXP(q_log_std[p]);
            REAL qv = positive_mask[p] ? LOG(th) : th;
            REAL qz = (qv - q_mean[p]) / sd;
            qa += (REAL)(-0.5) * qz * qz - q_log_std[p] - half_log_2pi - (positive_mask[p] ? LOG(th) : (REAL)0);
        }
        prior_lp[b] = pa; post_lp[b] = qa;
    }
}

/* ------------------------------------------------- Euler-Maruyama simulator of the MODEL SDE (pre-training stage)
 * core/euler_maruyama.py:11-45:  x <- x + f(x, theta) dt + (G(x, theta) eps_t) sqrt(dt);  x[positive_dims] <- max(x, 1e-6).
 * The reference takes f and G as user callables; restated here for the SDEs of its two example scripts and the
 * benchmark's synthetic config:
 *   kind 1  Ornstein-Uhlenbeck   examples/ornstein_uhlenbeck.py:18-30   f = kappa (mu - x),  G = sigma
 *   kind 2  Lotka-Volterra       examples/lotka_volterra.py:18-46       f = (t1 u - t2 u v, t2 u v - t3 v), G = analytic 2x2
 *                                Cholesky factor with three clamp(min=1e-6)
 *   kind 3  linear, diagonal     (BASELINE config 5)                     f = -a x,  G = diag(softplus(b) + 1e-3), theta = (a, b)
 * traj[B][T+1][S].  The backward is the reverse-mode derivative of exactly this recursion (what torch autograd computes for
 * trainer.py:208-259): clamp(min) passes the gradient where its input >= the floor; a clamped trajectory entry is
 * recognised by its value (== 1e-6 exactly). */
static void FN(em_step)(int kind, int S, const REAL *x, const REAL *th, const REAL *e, REAL dt, REAL sqdt, REAL *y)
{
    if (kind == 1) {
        y[0] = x[0] + th[0] * (th[1] - x[0]) * dt + th[2] * e[0] * sqdt;
    } else if (kind == 2) {
        const REAL u = x[0], v = x[1], t1 = th[0], t2 = th[1], t3 = th[2], floor = (REAL)1e-6;
        const REAL uv = t2 * u * v;
        REAL q00 = t1 * u + uv; if (q00 < floor) q00 = floor;
        const REAL l00 = (REAL)sqrt((double)q00);
        const REAL c = l00 < floor ? floor : l00;
        const REAL l10 = -uv / c;
        REAL q11 = t3 * v + uv - l10 * l10; if (q11 < floor) q11 = floor;
        const REAL l11 = (REAL)sqrt((double)q11);
        y[0] = u + (t1 * u - uv) * dt + (l00 * e[0]) * sqdt;
        y[1] = v + (uv - t3 * v) * dt + (l10 * e[0] + l11 * e[1]) * sqdt;
    } else {
        for (int i = 0; i < S; ++i) {
            const REAL b = th[S + i];
            const REAL sp = (b > (REAL)20 ? b : LOG1P(EXP(b))) + (REAL)1e-3;
            y[i] = x[i] + (-th[i] * x[i]) * dt + (sp * e[i]) * sqdt;
        }
    }
}

void FN(vsde_oracle_em_fwd)(int kind, int B, int T, int S, int P, const REAL *x0, const REAL *theta, const REAL *noise,
                            double dt_d, const unsigned char *pos_mask, REAL *traj)
{
    const REAL dt = (REAL)dt_d, sqdt = (REAL)POW(dt_d, 0.5);
    for (int b = 0; b < B; ++b) {
        REAL *tr = traj + (long)b * (T + 1) * S;
        for (int i = 0; i < S; ++i) tr[i] = x0[(long)b * S + i];
        for (int t = 0; t < T; ++t) {
            REAL *y = tr + (long)(t + 1) * S;
            FN(em_step)(kind, S, tr + (long)t * S, theta + (long)b * P, noise + ((long)b * T + t) * S, dt, sqdt, y);
            for (int i = 0; i < S; ++i) if (pos_mask[i] && y[i] < (REAL)1e-6) y[i] = (REAL)1e-6;
        }
    }
}

void FN(vsde_oracle_em_bwd)(int kind, int B, int T, int S, int P, const REAL *theta, const REAL *noise, const REAL *traj,
                            const REAL *g_traj, double dt_d, const unsigned char *pos_mask, REAL *g_x0, REAL *g_theta)
{
    const REAL dt = (REAL)dt_d, sqdt = (REAL)POW(dt_d, 0.5), floor = (REAL)1e-6;
    REAL a[64], ax[64];
    for (int b = 0; b < B; ++b) {
        const REAL *th = theta + (long)b * P;
        REAL *gth = g_theta + (long)b * P;
        for (int p = 0; p < P; ++p) gth[p] = 0;
        for (int i = 0; i < S; ++i) a[i] = 0;
        for (int t = T - 1; t >= 0; --t) {
            const REAL *x = traj + ((long)b * (T + 1) + t) * S, *xn = x + S, *e = noise + ((long)b * T + t) * S;
            for (int i = 0; i < S; ++i) {
                a[i] += g_traj[((long)b * (T + 1) + t + 1) * S + i];
                if (pos_mask[i] && xn[i] == floor) a[i] = 0;           /* clamped: no gradient through this entry */
            }
            if (kind == 1) {
                gth[0] += a[0] * (th[1] - x[0]) * dt; gth[1] += a[0] * th[0] * dt; gth[2] += a[0] * e[0] * sqdt;
                ax[0] = a[0] * ((REAL)1 - th[0] * dt);
            } else if (kind == 2) {
                const REAL u = x[0], v = x[1], t1 = th[0], t2 = th[1], t3 = th[2];
                const REAL uv = t2 * u * v;
                const REAL q00r = t1 * u + uv, q00 = q00r < floor ? floor : q00r, l00 = (REAL)sqrt((double)q00);
                const REAL c = l00 < floor ? floor : l00, l10 = -uv / c;
                const REAL q11r = t3 * v + uv - l10 * l10, q11 = q11r < floor ? floor : q11r, l11 = (REAL)sqrt((double)q11);
                const REAL d_f0 = a[0] * dt, d_f1 = a[1] * dt;
                REAL d_l00 = a[0] * e[0] * sqdt, d_l10 = a[1] * e[0] * sqdt;
                const REAL d_l11 = a[1] * e[1] * sqdt;
                REAL d_u = a[0], d_v = a[1], d_uv = 0, d_t1 = 0, d_t2 = 0, d_t3 = 0;
                const REAL d_q11 = q11r >= floor ? d_l11 / ((REAL)2 * l11) : (REAL)0;
                d_t3 += d_q11 * v; d_v += d_q11 * t3; d_uv += d_q11; d_l10 += (REAL)(-2) * l10 * d_q11;
                d_uv += -d_l10 / c;
                const REAL d_c = d_l10 * uv / (c * c);
                if (l00 >= floor) d_l00 += d_c;
                const REAL d_q00 = q00r >= floor ? d_l00 / ((REAL)2 * l00) : (REAL)0;
                d_t1 += d_q00 * u; d_u += d_q00 * t1; d_uv += d_q00;
                d_t1 += d_f0 * u; d_u += d_f0 * t1; d_uv -= d_f0;
                d_uv += d_f1; d_t3 -= d_f1 * v; d_v -= d_f1 * t3;
                d_t2 += d_uv * u * v; d_u += d_uv * t2 * v; d_v += d_uv * t2 * u;
                gth[0] += d_t1; gth[1] += d_t2; gth[2] += d_t3;
                ax[0] = d_u; ax[1] = d_v;
            } else {
                for (int i = 0; i < S; ++i) {
                    const REAL bb = th[S + i];
                    const REAL sg = (REAL)1 / ((REAL)1 + EXP(-bb));    /* d softplus(b) / db (threshold branch: 1) */
                    gth[i] += a[i] * (-x[i]) * dt;
                    gth[S + i] += a[i] * (bb > (REAL)20 ? (REAL)1 : sg) * e[i] * sqdt;
                    ax[i] = a[i] * ((REAL)1 - th[i] * dt);
                }
            }
            for (int i = 0; i < S; ++i) a[i] = ax[i];
        }
        for (int i = 0; i < S; ++i) g_x0[(long)b * S + i] = a[i] + g_traj[((long)b * (T + 1)) * S + i];
    }
}

#undef FN
#undef CAT
#undef CAT_
