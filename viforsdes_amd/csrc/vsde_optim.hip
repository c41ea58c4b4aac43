// The optimizer step of the ELBO trainer as TWO launches over all parameters (gfx950).
//
// Reference (inference/trainer.py:197-204, inference/exponential_moving_average.py:27-32), per step:
//     scaler.unscale_(optimizer)                         one multi-tensor pass over the gradients (non-finite check, g *= 1/scale)
//     clip_grad_norm_(parameters, max_norm)              a second (norms) and a third (g *= clip coefficient) pass
//     scaler.step(optimizer)   [AdamW]                   multi-tensor AdamW: 220 kernels of ~40 us for the ~200 parameter tensors
//     ema.update()                                       one more pass over parameters + shadow
// = ~390 us of multi-tensor kernels at the LV model (8.4 M parameters) for 0.33 GB of traffic.  Here:
//   optim_stats_kernel       partial sums of (g / scale)^2 per 4096-element chunk of every gradient (one read of the gradients);
//   optim_update_kernel      every workgroup adds the partials in index order (the same value in every workgroup: deterministic),
//                            derives found_inf = !isfinite(sum), the global norm and the clip coefficient, and -- unless found_inf --
//                            applies unscale x clip, the decoupled-weight-decay AdamW update (torch's fused kernel, same operation
//                            order) and the EMA lerp to its chunk: parameters, both moments and the shadow are read and written once.
// The step count is ONE device scalar pair (t_cur, t_next): the stats kernel publishes t_cur = t_next before anybody reads it, the
// update kernel's first workgroup writes t_next = t_cur + (found_inf ? 0 : 1); graph-capture safe (no host-side counter).
// The work is a table of chunks built once by the host (viforsdes_amd/inference/fused_optimizer.py); the gradient tensors are new
// allocations every step, so their base pointers come through a small per-parameter pointer array.
#include "vsde_common.h"

namespace vsde {

struct OptChunk {        // 64 bytes; all pointers device pointers to this chunk's first element
    float *p, *m, *v, *ema;
    int32_t param;       // index into the gradient pointer array
    int32_t n;           // elements in this chunk (1 .. OPT_CHUNK)
    int64_t goff;        // element offset of the chunk inside its gradient tensor
    int32_t group;       // hyper-parameter row
    int32_t reserved0;
    int64_t reserved1;
};
static_assert(sizeof(OptChunk) == 64, "the host builds the table as int64 [n_chunks][8]");

constexpr int OPT_CHUNK = 4096, OPT_THREADS = 256;

struct OptParams {
    const OptChunk *chunks; int n_chunks;
    const float *const *grads;      // [n_params] base pointers of this step's gradient tensors
    const float *scale;             // loss scale (device scalar) or nullptr (= 1)
    float *partials;                // [n_chunks]
    float *tstate;                  // [2]: t_cur, t_next (float step counts, as torch keeps them)
    const double *groups;           // [n_groups][5]: lr, beta1, beta2, eps, weight_decay (double, as torch's kernel takes them)
    float max_norm;                 // gradient clipping threshold (<= 0: no clipping)
    float ema_weight;               // 1 - decay; < 0: no EMA
    float *out;                     // [2]: global gradient norm (unscaled, before clipping), found_inf (0 / 1)
};

__device__ __forceinline__ float block_sum_256(float v, float *red) {   // fixed order: lanes by butterfly, waves 0..3 in order
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float t = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return t;
}

__global__ void __launch_bounds__(OPT_THREADS) optim_stats_kernel(OptParams q) {
    __shared__ float red[4];
    const OptChunk c = q.chunks[blockIdx.x];
    const float *g = q.grads[c.param] + c.goff;
    const float inv = q.scale ? (float)(1.0 / (double)q.scale[0]) : 1.0f;   // GradScaler: scale.double().reciprocal().float()
    float ss = 0.f;
    for (int i = threadIdx.x * 4; i < c.n; i += OPT_THREADS * 4) {
        if (i + 4 <= c.n && ((((uintptr_t)(g + i)) & 15) == 0)) {
            const float4 x = *(const float4 *)(g + i);
            const float a = x.x * inv, b = x.y * inv, d = x.z * inv, e = x.w * inv;
            ss += (a * a + b * b) + (d * d + e * e);
        } else {
            for (int j = i; j < c.n && j < i + 4; ++j) { const float a = g[j] * inv; ss += a * a; }
        }
    }
    ss = block_sum_256(ss, red);
    if (threadIdx.x == 0) {
        q.partials[blockIdx.x] = ss;
        if (blockIdx.x == 0) q.tstate[0] = q.tstate[1];   // this step's count: nobody reads tstate during this kernel
    }
}

__global__ void __launch_bounds__(OPT_THREADS) optim_update_kernel(OptParams q) {
    __shared__ float red[4];
    // ---- the global sum of squares, identically in every workgroup
    float t = 0.f;
    for (int i = threadIdx.x; i < q.n_chunks; i += OPT_THREADS) t += q.partials[i];
    const float total = block_sum_256(t, red);
    const bool found_inf = !(fabsf(total) <= 3.402823466e38f);   // inf or nan
    const float norm = sqrtf(total);
    float clip = 1.0f;
    // clip_grad_norm_: clamp(max_norm / (norm + 1e-6), max = 1); torch.clamp propagates NaN, so a NaN norm without loss scaling
    // poisons every gradient (and parameter) like the reference sequence does -- loud, not a silent unclipped step
    if (q.max_norm > 0.f) { clip = q.max_norm / (norm + 1e-6f); clip = (clip < 1.0f || clip != clip) ? clip : 1.0f; }
    const float t_cur = q.tstate[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        q.out[0] = norm; q.out[1] = found_inf ? 1.0f : 0.0f;
        q.tstate[1] = t_cur + ((found_inf && q.scale != nullptr) ? 0.0f : 1.0f);
    }
    const OptChunk c = q.chunks[blockIdx.x];
    const bool ema = q.ema_weight >= 0.f;
    if (found_inf && q.scale != nullptr) {   // GradScaler skips the optimizer step (no loss scaling: no check, as the reference) ...
        if (ema)                             // ... but ema.update() still runs on the unchanged parameters
            for (int i = threadIdx.x; i < c.n; i += OPT_THREADS) { const float sh = c.ema[i]; c.ema[i] = sh + q.ema_weight * (c.p[i] - sh); }
        return;
    }
    const double *hp = q.groups + 5 * c.group;
    const double lr = hp[0], beta1 = hp[1], beta2 = hp[2], eps = hp[3], wd = hp[4];
    const float step = t_cur + 1.0f;
    // torch's fused AdamW (fused_adam_utils.cuh adam_math, ADAMW mode, fp32 parameters): the same operations in the same order
    // with the same promotions (hyper-parameters are doubles there, the tensors' opmath is float)
    const float bc1 = (float)(1.0 - pow(beta1, (double)step));
    const float bc2_sqrt = sqrtf((float)(1.0 - pow(beta2, (double)step)));
    const float step_size = (float)(lr / (double)bc1);
    const float inv = q.scale ? (float)(1.0 / (double)q.scale[0]) : 1.0f;
    const float *g = q.grads[c.param] + c.goff;
    auto adam = [&](float grad, float &p, float &m, float &v, float &sh) {
        grad *= inv;                 // unscale_ ...
        grad *= clip;                // ... clip_grad_norm_ (always multiplies, by 1 when the norm is small)
        p = (float)((double)p - lr * wd * (double)p);
        m = (float)((double)m + (1.0 - beta1) * (double)(grad - m));          // lerp(exp_avg, grad, 1 - beta1), weight < 0.5
        v = (float)(beta2 * (double)v + (1.0 - beta2) * (double)grad * (double)grad);
        const float denom = (float)((double)(sqrtf(v) / bc2_sqrt) + eps);
        p -= step_size * m / denom;
        if (ema) sh = sh + q.ema_weight * (p - sh);                            // torch.lerp, scalar weight < 0.5
    };
    // four elements per thread and trip (16-byte accesses, all of a trip's loads in flight together); parameter / moment / shadow
    // chunks start on 16-byte boundaries, a gradient may be an arbitrary view: scalar trips then
    const bool vec = ((((uintptr_t)g) | ((uintptr_t)c.p) | ((uintptr_t)c.m) | ((uintptr_t)c.v) | (ema ? (uintptr_t)c.ema : 0)) & 15) == 0;
    const int nvec = vec ? (c.n & ~3) : 0;
    for (int i = threadIdx.x * 4; i < nvec; i += OPT_THREADS * 4) {
        const float4 g4 = *(const float4 *)(g + i);
        float4 p4 = *(const float4 *)(c.p + i), m4 = *(const float4 *)(c.m + i), v4 = *(const float4 *)(c.v + i);
        float4 s4 = ema ? *(const float4 *)(c.ema + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        adam(g4.x, p4.x, m4.x, v4.x, s4.x); adam(g4.y, p4.y, m4.y, v4.y, s4.y);
        adam(g4.z, p4.z, m4.z, v4.z, s4.z); adam(g4.w, p4.w, m4.w, v4.w, s4.w);
        *(float4 *)(c.p + i) = p4; *(float4 *)(c.m + i) = m4; *(float4 *)(c.v + i) = v4;
        if (ema) *(float4 *)(c.ema + i) = s4;
    }
    for (int i = nvec + threadIdx.x; i < c.n; i += OPT_THREADS) {
        float p = c.p[i], m = c.m[i], v = c.v[i], sh = ema ? c.ema[i] : 0.f;
        adam(g[i], p, m, v, sh);
        c.p[i] = p; c.m[i] = m; c.v[i] = v;
        if (ema) c.ema[i] = sh;
    }
}

}  // namespace vsde

using namespace vsde;

extern "C" int vsde_optim_chunk_bytes(void) { return (int)sizeof(OptChunk); }
extern "C" int vsde_optim_chunk_elems(void) { return OPT_CHUNK; }

extern "C" int vsde_optim_step(const void *chunks, int n_chunks, const void *grads, const float *scale, float *partials, float *tstate,
                               const double *groups, double max_norm, double ema_weight, float *out, void *stream) {
    VSDE_CHECK_ARG(chunks && grads && partials && tstate && groups && out && n_chunks > 0, VSDE_E_BADARG, "bad optimizer-step arguments");
    OptParams q;
    q.chunks = (const OptChunk *)chunks; q.n_chunks = n_chunks; q.grads = (const float *const *)grads; q.scale = scale;
    q.partials = partials; q.tstate = tstate; q.groups = groups; q.max_norm = (float)max_norm; q.ema_weight = (float)ema_weight; q.out = out;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(optim_stats_kernel, dim3((unsigned)n_chunks), dim3(OPT_THREADS), 0, s, q);
    hipLaunchKernelGGL(optim_update_kernel, dim3((unsigned)n_chunks), dim3(OPT_THREADS), 0, s, q);
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}
