// Batched Euler-Maruyama simulator of the MODEL SDE and its reverse-mode gradient (gfx950).
// Replaces the T-step Python loop of the reference's core/euler_maruyama.py:11-45 (two user callbacks, an einsum, an
// in-place clamp and a slice-assign per step: ~10 tiny kernels x T) for the SDEs whose drift / diffusion are built in:
//   kind 1  Ornstein-Uhlenbeck  (examples/ornstein_uhlenbeck.py:18-30)   f = kappa (mu - x),   G = sigma
//   kind 2  Lotka-Volterra      (examples/lotka_volterra.py:18-46)       analytic 2x2 Cholesky factor, three clamp(min=1e-6)
//   kind 3  linear / diagonal   (BASELINE config 5)                       f = -a x, G = diag(softplus(b) + 1e-3)
// One thread per sample path, the time loop inside the kernel, x_t in registers.  HBM-bound streaming (noise read once,
// trajectory written once; the backward reads noise, trajectory and the upstream gradient once): a path's records are
// contiguous over time, so the 64 paths of a workgroup move CH steps at a time through LDS as coalesced row segments.
// The backward is what torch autograd computes for trainer.py:208-259 (pre-training differentiates the simulation with
// respect to theta): reverse recursion over the saved trajectory, a clamped entry (== 1e-6) passes no gradient.
#include "vsde_common.h"

namespace vsde {

constexpr int kEmPaths = 64;   // paths per workgroup (one wavefront)
constexpr int kEmChunk = 32;   // time steps staged per LDS round
constexpr float kEmFloor = 1e-6f;
// clamp(min = 1e-6) that propagates NaN like torch.clamp / torch.maximum do (fmaxf(NaN, floor) would return the floor and hide a
// diverged path from the non-finite-loss guards of the pre-training loop and the ELBO)
__device__ __forceinline__ float floor_nan(float y) { return y < kEmFloor ? kEmFloor : y; }

struct EmParams {
    int B, T, S, P;
    const float *x0, *theta, *noise, *traj_in, *g_traj;
    float *traj, *g_x0, *g_theta;
    uint32_t pos_mask;
    float dt, sqdt;
};

template <int KIND> struct EmDims { static constexpr int S = KIND == 2 ? 2 : 1; static constexpr int P = KIND == 3 ? 2 : 3; };

__device__ __forceinline__ float softplus_f(float b) { return b > 20.f ? b : log1pf(__expf(b)); }

// y = x + f dt + (G eps) sqrt(dt)
template <int KIND>
__device__ __forceinline__ void em_step(const float *x, const float *th, const float *e, float dt, float sqdt, float *y) {
    if constexpr (KIND == 1) {
        y[0] = x[0] + th[0] * (th[1] - x[0]) * dt + th[2] * e[0] * sqdt;
    } else if constexpr (KIND == 2) {
        const float u = x[0], v = x[1], uv = th[1] * u * v;
        const float l00 = sqrtf(floor_nan(th[0] * u + uv));
        const float l10 = -uv / floor_nan(l00);
        const float l11 = sqrtf(floor_nan(th[2] * v + uv - l10 * l10));
        y[0] = u + (th[0] * u - uv) * dt + (l00 * e[0]) * sqdt;
        y[1] = v + (uv - th[2] * v) * dt + (l10 * e[0] + l11 * e[1]) * sqdt;
    } else {
        y[0] = x[0] + (-th[0] * x[0]) * dt + ((softplus_f(th[1]) + 1e-3f) * e[0]) * sqdt;
    }
}

// reverse-mode derivative of em_step: a = dL/dy (already masked by the clamp) -> ax = dL/dx, gth += dL/dtheta
template <int KIND>
__device__ __forceinline__ void em_step_bwd(const float *x, const float *th, const float *e, const float *a, float dt,
                                            float sqdt, float *ax, float *gth) {
    if constexpr (KIND == 1) {
        gth[0] += a[0] * (th[1] - x[0]) * dt; gth[1] += a[0] * th[0] * dt; gth[2] += a[0] * e[0] * sqdt;
        ax[0] = a[0] * (1.f - th[0] * dt);
    } else if constexpr (KIND == 2) {
        const float u = x[0], v = x[1], t1 = th[0], t2 = th[1], t3 = th[2], uv = t2 * u * v;
        const float q00r = t1 * u + uv, l00 = sqrtf(floor_nan(q00r));
        const float c = floor_nan(l00), l10 = -uv / c;
        const float q11r = t3 * v + uv - l10 * l10, l11 = sqrtf(floor_nan(q11r));
        const float d_f0 = a[0] * dt, d_f1 = a[1] * dt, d_l11 = a[1] * e[1] * sqdt;
        float d_l00 = a[0] * e[0] * sqdt, d_l10 = a[1] * e[0] * sqdt;
        float d_u = a[0], d_v = a[1], d_uv = 0.f, d_t1 = 0.f, d_t3 = 0.f;
        const float d_q11 = q11r >= kEmFloor ? d_l11 / (2.f * l11) : 0.f;
        d_t3 += d_q11 * v; d_v += d_q11 * t3; d_uv += d_q11; d_l10 += -2.f * l10 * d_q11;
        d_uv += -d_l10 / c;
        if (l00 >= kEmFloor) d_l00 += d_l10 * uv / (c * c);
        const float d_q00 = q00r >= kEmFloor ? d_l00 / (2.f * l00) : 0.f;
        d_t1 += d_q00 * u; d_u += d_q00 * t1; d_uv += d_q00;
        d_t1 += d_f0 * u; d_u += d_f0 * t1; d_uv -= d_f0;
        d_uv += d_f1; d_t3 -= d_f1 * v; d_v -= d_f1 * t3;
        gth[0] += d_t1; gth[1] += d_uv * u * v; gth[2] += d_t3;
        ax[0] = d_u + d_uv * t2 * v; ax[1] = d_v + d_uv * t2 * u;
    } else {
        const float b = th[1];
        gth[0] += a[0] * (-x[0]) * dt;
        gth[1] += a[0] * (b > 20.f ? 1.f : fast_rcp(1.f + __expf(-b))) * e[0] * sqdt;
        ax[0] = a[0] * (1.f - th[0] * dt);
    }
}

// Move `count` floats of each of the workgroup's 64 path rows between global memory (row r at g + r*gstride) and LDS
// (row r at s + r*sstride): each wave instruction covers one contiguous run of a single row.
template <bool LOAD>
__device__ __forceinline__ void em_rows(float *s, int sstride, float *g, int64_t gstride, int count, int rows, int lane) {
    for (int r = 0; r < rows; ++r)
        for (int k = lane; k < count; k += kEmPaths) {
            if constexpr (LOAD) s[r * sstride + k] = g[(int64_t)r * gstride + k];
            else g[(int64_t)r * gstride + k] = s[r * sstride + k];
        }
}

// kinds 1, 2: thread = path
template <int KIND>
__global__ void __launch_bounds__(kEmPaths) em_fwd_kernel(EmParams p) {
    constexpr int S = EmDims<KIND>::S, P = EmDims<KIND>::P, RS = kEmChunk * S + 1;
    __shared__ float noise_s[kEmPaths * RS], traj_s[kEmPaths * RS];
    const int lane = threadIdx.x, b0 = blockIdx.x * kEmPaths, b = b0 + lane;
    const int rows = min(kEmPaths, p.B - b0);
    const bool valid = b < p.B;
    float x[S], th[P];
#pragma unroll
    for (int i = 0; i < S; ++i) x[i] = valid ? p.x0[(int64_t)b * S + i] : 1.f;
#pragma unroll
    for (int k = 0; k < P; ++k) th[k] = valid ? p.theta[(int64_t)b * P + k] : 1.f;
    if (valid)
#pragma unroll
        for (int i = 0; i < S; ++i) p.traj[(int64_t)b * (p.T + 1) * S + i] = x[i];
    for (int t0 = 0; t0 < p.T; t0 += kEmChunk) {
        const int n = min(kEmChunk, p.T - t0);
        em_rows<true>(noise_s, RS, const_cast<float *>(p.noise) + ((int64_t)b0 * p.T + t0) * S, (int64_t)p.T * S, n * S, rows, lane);
        __syncthreads();
        for (int k = 0; k < n; ++k) {
            float y[S];
            em_step<KIND>(x, th, noise_s + lane * RS + k * S, p.dt, p.sqdt, y);
#pragma unroll
            for (int i = 0; i < S; ++i) {
                x[i] = ((p.pos_mask >> i) & 1u) ? floor_nan(y[i]) : y[i];
                traj_s[lane * RS + k * S + i] = x[i];
            }
        }
        __syncthreads();
        em_rows<false>(traj_s, RS, p.traj + ((int64_t)b0 * (p.T + 1) + t0 + 1) * S, (int64_t)(p.T + 1) * S, n * S, rows, lane);
        __syncthreads();
    }
}

template <int KIND>
__global__ void __launch_bounds__(kEmPaths) em_bwd_kernel(EmParams p) {
    constexpr int S = EmDims<KIND>::S, P = EmDims<KIND>::P, RS = (kEmChunk + 1) * S + 1;
    __shared__ float noise_s[kEmPaths * RS], traj_s[kEmPaths * RS], g_s[kEmPaths * RS];
    const int lane = threadIdx.x, b0 = blockIdx.x * kEmPaths, b = b0 + lane;
    const int rows = min(kEmPaths, p.B - b0);
    const bool valid = b < p.B;
    float a[S], th[P], gth[P];
#pragma unroll
    for (int i = 0; i < S; ++i) a[i] = 0.f;
#pragma unroll
    for (int k = 0; k < P; ++k) { th[k] = valid ? p.theta[(int64_t)b * P + k] : 1.f; gth[k] = 0.f; }
    const int nchunks = (p.T + kEmChunk - 1) / kEmChunk;
    for (int c = nchunks - 1; c >= 0; --c) {
        const int t0 = c * kEmChunk, n = min(kEmChunk, p.T - t0);
        em_rows<true>(noise_s, RS, const_cast<float *>(p.noise) + ((int64_t)b0 * p.T + t0) * S, (int64_t)p.T * S, n * S, rows, lane);
        em_rows<true>(traj_s, RS, const_cast<float *>(p.traj_in) + ((int64_t)b0 * (p.T + 1) + t0) * S, (int64_t)(p.T + 1) * S,
                      (n + 1) * S, rows, lane);                                             // x_{t0} .. x_{t0+n}
        em_rows<true>(g_s, RS, const_cast<float *>(p.g_traj) + ((int64_t)b0 * (p.T + 1) + t0 + 1) * S, (int64_t)(p.T + 1) * S,
                      n * S, rows, lane);                                                   // upstream of x_{t0+1} .. x_{t0+n}
        __syncthreads();
        for (int k = n - 1; k >= 0; --k) {
            const float *xs = traj_s + lane * RS + k * S;
#pragma unroll
            for (int i = 0; i < S; ++i) {
                a[i] += g_s[lane * RS + k * S + i];
                if (((p.pos_mask >> i) & 1u) && xs[S + i] == kEmFloor) a[i] = 0.f;          // clamped entry: no gradient
            }
            float ax[S];
            em_step_bwd<KIND>(xs, th, noise_s + lane * RS + k * S, a, p.dt, p.sqdt, ax, gth);
#pragma unroll
            for (int i = 0; i < S; ++i) a[i] = ax[i];
        }
        __syncthreads();
    }
    if (valid) {
#pragma unroll
        for (int i = 0; i < S; ++i) p.g_x0[(int64_t)b * S + i] = a[i] + p.g_traj[(int64_t)b * (p.T + 1) * S + i];
#pragma unroll
        for (int k = 0; k < P; ++k) p.g_theta[(int64_t)b * P + k] = gth[k];
    }
}

// kind 3: every state dimension is an independent scalar SDE -> thread = (path, dim); neighbouring threads are neighbouring
// addresses at every step, so the accesses coalesce without staging
__global__ void __launch_bounds__(256) em_diag_fwd_kernel(EmParams p) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (int64_t)p.B * p.S) return;
    const int b = (int)(q / p.S), i = (int)(q % p.S);
    const float th[2] = {p.theta[(int64_t)b * p.P + i], p.theta[(int64_t)b * p.P + p.S + i]};
    const bool pos = (p.pos_mask >> i) & 1u;
    float x = p.x0[q];
    float *tr = p.traj + (int64_t)b * (p.T + 1) * p.S + i;
    const float *nz = p.noise + (int64_t)b * p.T * p.S + i;
    tr[0] = x;
    for (int t = 0; t < p.T; ++t) {
        float y;
        em_step<3>(&x, th, nz + (int64_t)t * p.S, p.dt, p.sqdt, &y);
        x = pos ? floor_nan(y) : y;
        tr[(int64_t)(t + 1) * p.S] = x;
    }
}

__global__ void __launch_bounds__(256) em_diag_bwd_kernel(EmParams p) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (int64_t)p.B * p.S) return;
    const int b = (int)(q / p.S), i = (int)(q % p.S);
    const float th[2] = {p.theta[(int64_t)b * p.P + i], p.theta[(int64_t)b * p.P + p.S + i]};
    const bool pos = (p.pos_mask >> i) & 1u;
    const float *tr = p.traj_in + (int64_t)b * (p.T + 1) * p.S + i, *gt = p.g_traj + (int64_t)b * (p.T + 1) * p.S + i;
    const float *nz = p.noise + (int64_t)b * p.T * p.S + i;
    float a = 0.f, gth[2] = {0.f, 0.f};
    for (int t = p.T - 1; t >= 0; --t) {
        a += gt[(int64_t)(t + 1) * p.S];
        if (pos && tr[(int64_t)(t + 1) * p.S] == kEmFloor) a = 0.f;
        float ax;
        em_step_bwd<3>(tr + (int64_t)t * p.S, th, nz + (int64_t)t * p.S, &a, p.dt, p.sqdt, &ax, gth);
        a = ax;
    }
    p.g_x0[q] = a + gt[0];
    p.g_theta[(int64_t)b * p.P + i] = gth[0];
    p.g_theta[(int64_t)b * p.P + p.S + i] = gth[1];
}

// ---------------------------------------------------------------------------------------------------------------------
// Drift f(x_t, theta) and diffusion factor G(x_t, theta) of the built-in SDEs on every grid point of a batch of paths, and
// the vector-Jacobian product the ELBO backward needs.  Replaces the ~35 (forward) + ~70 (autograd backward) tiny torch
// kernels the Python drift / diffusion callables expand to on the flattened [(B T), S] states
// (inference/evidence_lower_bound.py:37-40 of the reference evaluates them exactly there).
//   x [B][T+1][S] (rows 0..T-1 used), theta [B][P]  ->  drift [B][T][S], diffusion [B][T][S][S]
//   backward: g_drift, g_diff -> g_x [B][T+1][S] (row T = 0), g_theta [B][P] (sum over t, fixed order)
// torch.clamp(min=1e-6) passes the gradient where the input is >= the bound.
struct CoefParams {
    int B, T, S, P;
    const float *x, *theta, *g_drift, *g_diff;
    float *drift, *diff, *g_x, *g_theta;
};

template <int KIND> __device__ __forceinline__ void coef_fwd(const float *x, const float *th, float *f, float *G) {
    if constexpr (KIND == 1) {
        f[0] = th[0] * (th[1] - x[0]); G[0] = th[2];
    } else {
        const float u = x[0], v = x[1], uv = th[1] * u * v;
        const float l00 = sqrtf(floor_nan(th[0] * u + uv));
        const float l10 = -uv / floor_nan(l00);
        const float l11 = sqrtf(floor_nan(th[2] * v + uv - l10 * l10));
        f[0] = th[0] * u - uv; f[1] = uv - th[2] * v;
        G[0] = l00; G[1] = 0.f; G[2] = l10; G[3] = l11;
    }
}

template <int KIND>
__device__ __forceinline__ void coef_bwd(const float *x, const float *th, const float *gf, const float *gG, float *gx, float *gth) {
    if constexpr (KIND == 1) {
        gth[0] += gf[0] * (th[1] - x[0]); gth[1] += gf[0] * th[0]; gth[2] += gG[0];
        gx[0] = -gf[0] * th[0];
    } else {
        const float u = x[0], v = x[1], t1 = th[0], t2 = th[1], t3 = th[2], uv = t2 * u * v;
        const float q00r = t1 * u + uv, l00 = sqrtf(floor_nan(q00r));
        const float c = floor_nan(l00), l10 = -uv / c;
        const float q11r = t3 * v + uv - l10 * l10, l11 = sqrtf(floor_nan(q11r));
        float d_l00 = gG[0], d_l10 = gG[2];
        float d_u = 0.f, d_v = 0.f, d_uv = 0.f, d_t1 = 0.f, d_t3 = 0.f;
        const float d_q11 = q11r >= kEmFloor ? gG[3] / (2.f * l11) : 0.f;
        d_t3 += d_q11 * v; d_v += d_q11 * t3; d_uv += d_q11; d_l10 += -2.f * l10 * d_q11;
        d_uv += -d_l10 / c;
        if (l00 >= kEmFloor) d_l00 += d_l10 * uv / (c * c);
        const float d_q00 = q00r >= kEmFloor ? d_l00 / (2.f * l00) : 0.f;
        d_t1 += d_q00 * u; d_u += d_q00 * t1; d_uv += d_q00;
        d_t1 += gf[0] * u; d_u += gf[0] * t1; d_uv -= gf[0];
        d_uv += gf[1]; d_t3 -= gf[1] * v; d_v -= gf[1] * t3;
        gth[0] += d_t1; gth[1] += d_uv * u * v; gth[2] += d_t3;
        gx[0] = d_u + d_uv * t2 * v; gx[1] = d_v + d_uv * t2 * u;
    }
}

// thread = grid point (b, t)
template <int KIND>
__global__ void __launch_bounds__(256) coef_fwd_kernel(CoefParams p) {
    constexpr int S = EmDims<KIND>::S, P = EmDims<KIND>::P;
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= (int64_t)p.B * p.T) return;
    const int b = (int)(q / p.T), t = (int)(q % p.T);
    float x[S], th[P], f[S], G[S * S];
#pragma unroll
    for (int i = 0; i < S; ++i) x[i] = p.x[((int64_t)b * (p.T + 1) + t) * S + i];
#pragma unroll
    for (int i = 0; i < P; ++i) th[i] = p.theta[(int64_t)b * P + i];
    coef_fwd<KIND>(x, th, f, G);
#pragma unroll
    for (int i = 0; i < S; ++i) p.drift[q * S + i] = f[i];
#pragma unroll
    for (int i = 0; i < S * S; ++i) p.diff[q * S * S + i] = G[i];
}

// workgroup = path: threads walk the time steps, the theta gradient is reduced through LDS in a fixed tree
template <int KIND>
__global__ void __launch_bounds__(256) coef_bwd_kernel(CoefParams p) {
    constexpr int S = EmDims<KIND>::S, P = EmDims<KIND>::P;
    __shared__ float red[256][P + 1];
    const int b = blockIdx.x, tid = threadIdx.x;
    float th[P], gth[P];
#pragma unroll
    for (int i = 0; i < P; ++i) { th[i] = p.theta[(int64_t)b * P + i]; gth[i] = 0.f; }
    for (int t = tid; t <= p.T; t += 256) {
        float gx[S];
        if (t < p.T) {
            const int64_t q = (int64_t)b * p.T + t;
            float x[S], gf[S], gG[S * S];
#pragma unroll
            for (int i = 0; i < S; ++i) { x[i] = p.x[((int64_t)b * (p.T + 1) + t) * S + i]; gf[i] = p.g_drift[q * S + i]; }
#pragma unroll
            for (int i = 0; i < S * S; ++i) gG[i] = p.g_diff[q * S * S + i];
            coef_bwd<KIND>(x, th, gf, gG, gx, gth);
        } else {
#pragma unroll
            for (int i = 0; i < S; ++i) gx[i] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < S; ++i) p.g_x[((int64_t)b * (p.T + 1) + t) * S + i] = gx[i];
    }
#pragma unroll
    for (int i = 0; i < P; ++i) red[tid][i] = gth[i];
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w)
#pragma unroll
            for (int i = 0; i < P; ++i) red[tid][i] += red[tid + w][i];
        __syncthreads();
    }
    if (tid < P) p.g_theta[(int64_t)b * P + tid] = red[0][tid];
}

// kind 3 (f_i = -a_i x_i, G = diag(softplus(b_i) + 1e-3)): thread = (b, t, i) forward; workgroup = path backward with the threads
// laid out as (time slot, i)
__global__ void __launch_bounds__(256) coef_diag_fwd_kernel(CoefParams p) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= (int64_t)p.B * p.T * p.S) return;
    const int i = (int)(q % p.S);
    const int64_t bt = q / p.S;
    const int b = (int)(bt / p.T), t = (int)(bt % p.T);
    const float a = p.theta[(int64_t)b * p.P + i], g = softplus_f(p.theta[(int64_t)b * p.P + p.S + i]) + 1e-3f;
    p.drift[q] = -a * p.x[((int64_t)b * (p.T + 1) + t) * p.S + i];
    float *row = p.diff + q * p.S;
    for (int j = 0; j < p.S; ++j) row[j] = j == i ? g : 0.f;
}

__global__ void __launch_bounds__(256) coef_diag_bwd_kernel(CoefParams p) {
    __shared__ float red[256][2];
    const int b = blockIdx.x, tid = threadIdx.x, S = p.S;
    const int slots = 256 / S, slot = tid / S, i = tid % S;
    float ga = 0.f, gb = 0.f;
    if (slot < slots) {
        const float a = p.theta[(int64_t)b * p.P + i];
        for (int t = slot; t <= p.T; t += slots) {
            float gx = 0.f;
            if (t < p.T) {
                const int64_t q = ((int64_t)b * p.T + t) * S + i;
                const float gf = p.g_drift[q];
                gx = -gf * a;
                ga -= gf * p.x[((int64_t)b * (p.T + 1) + t) * S + i];
                gb += p.g_diff[q * S + i];
            }
            p.g_x[((int64_t)b * (p.T + 1) + t) * S + i] = gx;
        }
    }
    red[tid][0] = ga; red[tid][1] = gb;
    __syncthreads();
    if (tid < S) {
        float sa = 0.f, sb = 0.f;
        for (int k = 0; k < slots; ++k) { sa += red[k * S + tid][0]; sb += red[k * S + tid][1]; }
        const float bb = p.theta[(int64_t)b * p.P + S + tid];
        p.g_theta[(int64_t)b * p.P + tid] = sa;
        p.g_theta[(int64_t)b * p.P + S + tid] = sb * (bb > 20.f ? 1.f : 1.f / (1.f + expf(-bb)));
    }
}

static int em_check(int kind, int B, int T, int S, int P) {
    VSDE_CHECK_ARG(B > 0 && T > 0, VSDE_E_BADARG, "bad Euler-Maruyama dims B=%d T=%d", B, T);
    VSDE_CHECK_ARG(kind >= 1 && kind <= 3, VSDE_E_BADARG, "unknown built-in SDE kind %d", kind);
    VSDE_CHECK_ARG(kind != 1 || (S == 1 && P == 3), VSDE_E_BADARG, "Ornstein-Uhlenbeck needs state_dim 1, sde_param_dim 3");
    VSDE_CHECK_ARG(kind != 2 || (S == 2 && P == 3), VSDE_E_BADARG, "Lotka-Volterra needs state_dim 2, sde_param_dim 3");
    VSDE_CHECK_ARG(kind != 3 || (S >= 1 && S <= 32 && P == 2 * S), VSDE_E_BADARG, "linear-diagonal SDE needs sde_param_dim = 2 state_dim <= 64");
    return 0;
}

static uint32_t em_mask(const uint8_t *m, int S) {
    uint32_t r = 0;
    for (int i = 0; i < S && m; ++i) r |= (m[i] ? 1u : 0u) << i;
    return r;
}

}  // namespace vsde

using namespace vsde;

extern "C" int vsde_euler_maruyama_fwd(int kind, int B, int T, int S, int P, const float *x0, const float *theta,
                                       const float *noise, double time_step, const uint8_t *positive_mask_host, float *traj,
                                       void *stream) {
    int rc = em_check(kind, B, T, S, P);
    if (rc) return rc;
    VSDE_CHECK_ARG(x0 && theta && noise && traj && time_step > 0, VSDE_E_BADARG, "NULL argument / bad time_step");
    EmParams p = {};
    p.B = B; p.T = T; p.S = S; p.P = P; p.x0 = x0; p.theta = theta; p.noise = noise; p.traj = traj;
    p.pos_mask = em_mask(positive_mask_host, S); p.dt = (float)time_step; p.sqdt = (float)sqrt(time_step);
    hipStream_t s = (hipStream_t)stream;
    const int grid = (B + kEmPaths - 1) / kEmPaths;
    if (kind == 1) hipLaunchKernelGGL(em_fwd_kernel<1>, dim3(grid), dim3(kEmPaths), 0, s, p);
    else if (kind == 2) hipLaunchKernelGGL(em_fwd_kernel<2>, dim3(grid), dim3(kEmPaths), 0, s, p);
    else hipLaunchKernelGGL(em_diag_fwd_kernel, dim3((unsigned)(((int64_t)B * S + 255) / 256)), dim3(256), 0, s, p);
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int vsde_euler_maruyama_bwd(int kind, int B, int T, int S, int P, const float *theta, const float *noise,
                                       const float *traj, const float *g_traj, double time_step,
                                       const uint8_t *positive_mask_host, float *g_x0, float *g_theta, void *stream) {
    int rc = em_check(kind, B, T, S, P);
    if (rc) return rc;
    VSDE_CHECK_ARG(theta && noise && traj && g_traj && g_x0 && g_theta && time_step > 0, VSDE_E_BADARG, "NULL argument / bad time_step");
    EmParams p = {};
    p.B = B; p.T = T; p.S = S; p.P = P; p.theta = theta; p.noise = noise; p.traj_in = traj; p.g_traj = g_traj;
    p.g_x0 = g_x0; p.g_theta = g_theta;
    p.pos_mask = em_mask(positive_mask_host, S); p.dt = (float)time_step; p.sqdt = (float)sqrt(time_step);
    hipStream_t s = (hipStream_t)stream;
    const int grid = (B + kEmPaths - 1) / kEmPaths;
    if (kind == 1) hipLaunchKernelGGL(em_bwd_kernel<1>, dim3(grid), dim3(kEmPaths), 0, s, p);
    else if (kind == 2) hipLaunchKernelGGL(em_bwd_kernel<2>, dim3(grid), dim3(kEmPaths), 0, s, p);
    else hipLaunchKernelGGL(em_diag_bwd_kernel, dim3((unsigned)(((int64_t)B * S + 255) / 256)), dim3(256), 0, s, p);
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int vsde_sde_coefficients_fwd(int kind, int B, int T, int S, int P, const float *x, const float *theta, float *drift,
                                         float *diffusion, void *stream) {
    int rc = em_check(kind, B, T, S, P);
    if (rc) return rc;
    VSDE_CHECK_ARG(x && theta && drift && diffusion, VSDE_E_BADARG, "NULL argument");
    CoefParams p = {};
    p.B = B; p.T = T; p.S = S; p.P = P; p.x = x; p.theta = theta; p.drift = drift; p.diff = diffusion;
    hipStream_t s = (hipStream_t)stream;
    const unsigned grid = (unsigned)(((int64_t)B * T + 255) / 256);
    if (kind == 1) hipLaunchKernelGGL(coef_fwd_kernel<1>, dim3(grid), dim3(256), 0, s, p);
    else if (kind == 2) hipLaunchKernelGGL(coef_fwd_kernel<2>, dim3(grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(coef_diag_fwd_kernel, dim3((unsigned)(((int64_t)B * T * S + 255) / 256)), dim3(256), 0, s, p);
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int vsde_sde_coefficients_bwd(int kind, int B, int T, int S, int P, const float *x, const float *theta,
                                         const float *g_drift, const float *g_diffusion, float *g_x, float *g_theta, void *stream) {
    int rc = em_check(kind, B, T, S, P);
    if (rc) return rc;
    VSDE_CHECK_ARG(x && theta && g_drift && g_diffusion && g_x && g_theta, VSDE_E_BADARG, "NULL argument");
    CoefParams p = {};
    p.B = B; p.T = T; p.S = S; p.P = P; p.x = x; p.theta = theta; p.g_drift = g_drift; p.g_diff = g_diffusion;
    p.g_x = g_x; p.g_theta = g_theta;
    hipStream_t s = (hipStream_t)stream;
    if (kind == 1) hipLaunchKernelGGL(coef_bwd_kernel<1>, dim3(B), dim3(256), 0, s, p);
    else if (kind == 2) hipLaunchKernelGGL(coef_bwd_kernel<2>, dim3(B), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(coef_diag_bwd_kernel, dim3(B), dim3(256), 0, s, p);
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}
