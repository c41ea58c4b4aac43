// Monte-Carlo ELBO accumulation over sample paths (gfx950): SDE transition log-density,
// variational path entropy (generative log-density) and softplus log-Jacobian, forward and
// analytic adjoint.  Replaces the two MultivariateNormal(scale_tril=...).log_prob calls over
// B*T tiny matrices plus ~10 elementwise torch kernels of the reference
// (src/variational_sde/inference/evidence_lower_bound.py:42-50,77-83; inference/types.py:23-24;
// inference/state_space.py:35-38).  HBM-bound streaming kernels: every tensor is read once.
#include "vsde_common.h"

namespace vsde {

constexpr float kLog2Pi = 1.8378770664093453f;

struct ElboParams {
    int B, T;
    const float *z, *x, *means, *chol, *drift, *diffusion;
    uint32_t pos_mask;
    float dt, sqdt;
    // forward outputs
    float *sde_lp, *gen_lp, *jac;
    // backward
    const float *g_sde, *g_gen, *g_jac;
    float *g_z, *g_x, *g_means, *g_chol, *g_drift, *g_diffusion;
};

// w = (A*s)^-1 (y - (m0 + d*dt)); returns log N(y; m0 + d dt, (A s)(A s)^T)
template <int S>
__device__ __forceinline__ float tri_logpdf(const float *__restrict__ y, const float *__restrict__ m0,
                                            const float *__restrict__ d, const float *__restrict__ A, float dt, float s,
                                            float (&w)[S]) {
    float quad = 0.f, logdet = 0.f;
#pragma unroll
    for (int i = 0; i < S; ++i) {
        float acc = y[i] - (m0[i] + d[i] * dt);
#pragma unroll
        for (int j = 0; j < i; ++j) acc -= (A[i * S + j] * s) * w[j];
        float dii = A[i * S + i] * s;
        w[i] = acc / dii;
        quad += w[i] * w[i];
        logdet += __logf(dii);
    }
    return -0.5f * ((float)S * kLog2Pi + quad) - logdet;
}

__device__ __forceinline__ float log_sigmoid(float v) { return fminf(v, 0.f) - log1pf(__expf(-fabsf(v))); }

template <int S>
__global__ void __launch_bounds__(256) elbo_path_terms_kernel(ElboParams p) {
    const int b = blockIdx.x;
    float s_acc = 0.f, g_acc = 0.f, j_acc = 0.f;
    for (int t = threadIdx.x; t < p.T; t += blockDim.x) {
        const int64_t o1 = ((int64_t)b * (p.T + 1) + t) * S, o2 = ((int64_t)b * p.T + t) * S;
        float w[S];
        s_acc += tri_logpdf<S>(p.x + o1 + S, p.x + o1, p.drift + o2, p.diffusion + o2 * S, p.dt, p.sqdt, w);
        g_acc += tri_logpdf<S>(p.z + o1 + S, p.z + o1, p.means + o2, p.chol + o2 * S, p.dt, p.sqdt, w);
#pragma unroll
        for (int i = 0; i < S; ++i)
            if ((p.pos_mask >> i) & 1u) j_acc += log_sigmoid(p.z[o1 + S + i]);
    }
    __shared__ float red[3][4];
    s_acc = wave_sum(s_acc); g_acc = wave_sum(g_acc); j_acc = wave_sum(j_acc);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { red[0][wave] = s_acc; red[1][wave] = g_acc; red[2][wave] = j_acc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = blockDim.x >> 6;
        float a = 0.f, c = 0.f, d = 0.f;
        for (int i = 0; i < nw; ++i) { a += red[0][i]; c += red[1][i]; d += red[2][i]; }
        p.sde_lp[b] = a; p.gen_lp[b] = c; p.jac[b] = d;
    }
}

// v = (A s)^-T w
template <int S>
__device__ __forceinline__ void tri_solve_t(const float *__restrict__ A, float s, const float (&w)[S], float (&v)[S]) {
#pragma unroll
    for (int i = S - 1; i >= 0; --i) {
        float acc = w[i];
#pragma unroll
        for (int j = i + 1; j < S; ++j) acc -= (A[j * S + i] * s) * v[j];
        v[i] = acc / (A[i * S + i] * s);
    }
}

// One thread per (b, tau), tau in [0, T]: the "current" role of step tau (local gradients and the
// +v term of g_z/g_x[tau]) and the "next" role of step tau-1 (the -v term), so nothing races.
template <int S>
__global__ void __launch_bounds__(256) elbo_path_terms_bwd_kernel(ElboParams p) {
    const int b = blockIdx.y, tau = blockIdx.x * blockDim.x + threadIdx.x;
    if (tau > p.T) return;
    const float gs = p.g_sde[b], gg = p.g_gen[b], gj = p.g_jac[b];
    float gz[S], gx[S];
#pragma unroll
    for (int i = 0; i < S; ++i) { gz[i] = 0.f; gx[i] = 0.f; }
    float w[S], v[S];
    if (tau < p.T) {
        const int64_t o1 = ((int64_t)b * (p.T + 1) + tau) * S, o2 = ((int64_t)b * p.T + tau) * S;
        const float *G = p.diffusion + o2 * S, *Lc = p.chol + o2 * S;
        (void)tri_logpdf<S>(p.x + o1 + S, p.x + o1, p.drift + o2, G, p.dt, p.sqdt, w);
        tri_solve_t<S>(G, p.sqdt, w, v);
#pragma unroll
        for (int i = 0; i < S; ++i) {
            gx[i] += gs * v[i];
            p.g_drift[o2 + i] = gs * v[i] * p.dt;
#pragma unroll
            for (int j = 0; j < S; ++j) {
                float val = j <= i ? gs * p.sqdt * v[i] * w[j] : 0.f;
                if (i == j) val -= gs / G[i * S + i];
                p.g_diffusion[o2 * S + i * S + j] = val;
            }
        }
        (void)tri_logpdf<S>(p.z + o1 + S, p.z + o1, p.means + o2, Lc, p.dt, p.sqdt, w);
        tri_solve_t<S>(Lc, p.sqdt, w, v);
#pragma unroll
        for (int i = 0; i < S; ++i) {
            gz[i] += gg * v[i];
            p.g_means[o2 + i] = gg * v[i] * p.dt;
#pragma unroll
            for (int j = 0; j < S; ++j) {
                float val = j <= i ? gg * p.sqdt * v[i] * w[j] : 0.f;
                if (i == j) val -= gg / Lc[i * S + i];
                p.g_chol[o2 * S + i * S + j] = val;
            }
        }
    }
    if (tau > 0) {
        const int64_t o1 = ((int64_t)b * (p.T + 1) + tau - 1) * S, o2 = ((int64_t)b * p.T + tau - 1) * S;
        (void)tri_logpdf<S>(p.x + o1 + S, p.x + o1, p.drift + o2, p.diffusion + o2 * S, p.dt, p.sqdt, w);
        tri_solve_t<S>(p.diffusion + o2 * S, p.sqdt, w, v);
#pragma unroll
        for (int i = 0; i < S; ++i) gx[i] -= gs * v[i];
        (void)tri_logpdf<S>(p.z + o1 + S, p.z + o1, p.means + o2, p.chol + o2 * S, p.dt, p.sqdt, w);
        tri_solve_t<S>(p.chol + o2 * S, p.sqdt, w, v);
#pragma unroll
        for (int i = 0; i < S; ++i) {
            gz[i] -= gg * v[i];
            if ((p.pos_mask >> i) & 1u) gz[i] += gj * fast_rcp(1.0f + __expf(p.z[o1 + S + i]));  // d logsigmoid = sigmoid(-z)
        }
    }
    const int64_t o = ((int64_t)b * (p.T + 1) + tau) * S;
#pragma unroll
    for (int i = 0; i < S; ++i) { p.g_z[o + i] = gz[i]; p.g_x[o + i] = gx[i]; }
}

template <int S>
static int launch_elbo(const ElboParams &p, bool bwd, hipStream_t s) {
    if (!bwd) hipLaunchKernelGGL((elbo_path_terms_kernel<S>), dim3(p.B), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((elbo_path_terms_bwd_kernel<S>), dim3((p.T + 1 + 255) / 256, p.B), dim3(256), 0, s, p);
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}

static int dispatch_elbo(int S, const ElboParams &p, bool bwd, hipStream_t s) {
    switch (S) {
        case 1: return launch_elbo<1>(p, bwd, s);
        case 2: return launch_elbo<2>(p, bwd, s);
        case 3: return launch_elbo<3>(p, bwd, s);
        case 4: return launch_elbo<4>(p, bwd, s);
        case 5: return launch_elbo<5>(p, bwd, s);
        case 6: return launch_elbo<6>(p, bwd, s);
        case 7: return launch_elbo<7>(p, bwd, s);
        case 8: return launch_elbo<8>(p, bwd, s);
        case 9: return launch_elbo<9>(p, bwd, s);
        case 10: return launch_elbo<10>(p, bwd, s);
        case 11: return launch_elbo<11>(p, bwd, s);
        case 12: return launch_elbo<12>(p, bwd, s);
        case 13: return launch_elbo<13>(p, bwd, s);
        case 14: return launch_elbo<14>(p, bwd, s);
        case 15: return launch_elbo<15>(p, bwd, s);
        case 16: return launch_elbo<16>(p, bwd, s);
        default:
            set_error("state_dim %d not supported by the ELBO kernels (1..16)", S);
            return VSDE_E_STATE;
    }
}

static uint32_t mask_bits(const uint8_t *m, int S) {
    uint32_t r = 0;
    if (m) for (int i = 0; i < S; ++i) if (m[i]) r |= 1u << i;
    return r;
}

}  // namespace vsde

using namespace vsde;

extern "C" int vsde_elbo_path_terms(int B, int T, int S, const float *z, const float *x, const float *means,
                                    const float *chol, const float *drift, const float *diffusion,
                                    const uint8_t *positive_mask_host, double time_step, float *sde_lp,
                                    float *gen_lp, float *log_jac, void *stream) {
    VSDE_CHECK_ARG(B > 0 && T > 0 && S > 0, VSDE_E_BADARG, "bad dims B=%d T=%d S=%d", B, T, S);
    VSDE_CHECK_ARG(z && x && means && chol && drift && diffusion && sde_lp && gen_lp && log_jac, VSDE_E_BADARG, "NULL argument");
    ElboParams p = {};
    p.B = B; p.T = T; p.z = z; p.x = x; p.means = means; p.chol = chol; p.drift = drift; p.diffusion = diffusion;
    p.pos_mask = mask_bits(positive_mask_host, S); p.dt = (float)time_step; p.sqdt = (float)sqrt(time_step);
    p.sde_lp = sde_lp; p.gen_lp = gen_lp; p.jac = log_jac;
    return dispatch_elbo(S, p, false, (hipStream_t)stream);
}

extern "C" int vsde_elbo_path_terms_bwd(int B, int T, int S, const float *z, const float *x, const float *means,
                                        const float *chol, const float *drift, const float *diffusion,
                                        const uint8_t *positive_mask_host, double time_step, const float *g_sde,
                                        const float *g_gen, const float *g_jac, float *g_z, float *g_x,
                                        float *g_means, float *g_chol, float *g_drift, float *g_diffusion,
                                        void *stream) {
    VSDE_CHECK_ARG(B > 0 && T > 0 && S > 0, VSDE_E_BADARG, "bad dims B=%d T=%d S=%d", B, T, S);
    VSDE_CHECK_ARG(z && x && means && chol && drift && diffusion && g_sde && g_gen && g_jac && g_z && g_x && g_means && g_chol &&
                       g_drift && g_diffusion, VSDE_E_BADARG, "NULL argument");
    ElboParams p = {};
    p.B = B; p.T = T; p.z = z; p.x = x; p.means = means; p.chol = chol; p.drift = drift; p.diffusion = diffusion;
    p.pos_mask = mask_bits(positive_mask_host, S); p.dt = (float)time_step; p.sqdt = (float)sqrt(time_step);
    p.g_sde = g_sde; p.g_gen = g_gen; p.g_jac = g_jac;
    p.g_z = g_z; p.g_x = g_x; p.g_means = g_means; p.g_chol = g_chol; p.g_drift = g_drift; p.g_diffusion = g_diffusion;
    return dispatch_elbo(S, p, true, (hipStream_t)stream);
}
