// Fused elementwise / normalisation operators of the SiT observation encoder (gfx950).
//
// The reference runs the encoder as stock torch ops (primitives/sit.py:99-128, attn.py:80-111,
// mlp.py:50-54, cond.py:19-23); under bf16 autocast one block costs ~100 small kernels (casts,
// LayerNorm, RMS, complex RoPE, cat, mul/add chains).  Here every chain between two GEMMs / the
// attention call is one HBM pass, forward and backward:
//   ln_modulate        y = LayerNorm(x) * (1 + scale_b) + shift_b          (sit.py:99-101,124-126)
//   qk_norm_rope       q,k = RoPE(RMS(q,k)); v = lam v + (1-lam) v0; -> [B,h,N,d]  (attn.py:80-103)
//   gate_merge         out[b,n,(h d)] = attn[b,h,n,d] * sigmoid(gate[b,n,d])        (attn.py:108-113)
//   gated_residual     out = x + gate_b * y                                 (sit.py:114-115,127-128)
//   swiglu             out = silu(a) * b,  [a | b] = u                      (mlp.py:21-24)
// Memory-bound streaming kernels: 8 (bf16) / 4 (f32) contiguous elements per lane, math in fp32.
// T = float (mixed_precision off; parity tests) or bf16 (autocast).  Per-batch-row conditioning
// vectors are [B, C]; reductions over the tokens of a batch row are deterministic (no atomics).
#include <stdlib.h>

#include "vsde_common.h"

namespace vsde {

struct bf16_t { uint16_t v; };

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return __uint_as_float(((uint32_t)x.v) << 16); }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) {  // round to nearest even
    uint32_t u = __float_as_uint(x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return bf16_t{(uint16_t)((u >> 16) | 0x40)};
    u += 0x7fffu + ((u >> 16) & 1u);
    return bf16_t{(uint16_t)(u >> 16)};
}
template <typename T> __device__ __forceinline__ float rnd(float x) { return to_f32(from_f32<T>(x)); }

__device__ __forceinline__ float sigm(float x) { return fast_rcp(1.0f + __expf(-x)); }

// V contiguous elements of T <-> V floats with one (8/16-byte) memory instruction
template <typename T, int V> struct Pack;
template <int V> struct Pack<float, V> {
    static __device__ __forceinline__ void load(const float *p, float (&o)[V]) {
        if (V == 4) { float4 t = *(const float4 *)p; o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w; }
        else if (V == 2) { float2 t = *(const float2 *)p; o[0] = t.x; o[1] = t.y; }
        else {
#pragma unroll
            for (int i = 0; i < V; ++i) o[i] = p[i];
        }
    }
    static __device__ __forceinline__ void store(float *p, const float (&o)[V]) {
        if (V == 4) *(float4 *)p = make_float4(o[0], o[1], o[2], o[3]);
        else if (V == 2) *(float2 *)p = make_float2(o[0], o[1]);
        else {
#pragma unroll
            for (int i = 0; i < V; ++i) p[i] = o[i];
        }
    }
};
template <int V> struct Pack<bf16_t, V> {
    static __device__ __forceinline__ void load(const bf16_t *p, float (&o)[V]) {
        if (V == 8) { uint4 t = *(const uint4 *)p; const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) { o[2 * i] = __uint_as_float(w[i] << 16); o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); } }
        else if (V == 4) { uint2 t = *(const uint2 *)p; const uint32_t w[2] = {t.x, t.y};
#pragma unroll
            for (int i = 0; i < 2; ++i) { o[2 * i] = __uint_as_float(w[i] << 16); o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); } }
        else if (V == 2) { uint32_t w = *(const uint32_t *)p; o[0] = __uint_as_float(w << 16); o[1] = __uint_as_float(w & 0xffff0000u); }
        else {
#pragma unroll
            for (int i = 0; i < V; ++i) o[i] = to_f32(p[i]);
        }
    }
    static __device__ __forceinline__ void store(bf16_t *p, const float (&o)[V]) {
        if (V == 8 || V == 4 || V == 2) {
            uint32_t w[V / 2 > 0 ? V / 2 : 1];
#pragma unroll
            for (int i = 0; i < V / 2; ++i) w[i] = (uint32_t)from_f32<bf16_t>(o[2 * i]).v | ((uint32_t)from_f32<bf16_t>(o[2 * i + 1]).v << 16);
            if (V == 8) *(uint4 *)p = make_uint4(w[0], w[1], w[2 % (V / 2)], w[3 % (V / 2)]);
            else if (V == 4) *(uint2 *)p = make_uint2(w[0], w[1 % (V / 2)]);
            else *(uint32_t *)p = w[0];
        } else {
#pragma unroll
            for (int i = 0; i < V; ++i) p[i] = from_f32<bf16_t>(o[i]);
        }
    }
};
template <typename T> struct VecOf { static constexpr int v = 4; };
template <> struct VecOf<bf16_t> { static constexpr int v = 8; };

// ------------------------------------------------------------------------------ ln_modulate
// sum over aligned groups of `width` (power of two) consecutive lanes
__device__ __forceinline__ float seg_sum(float v, int width) {
    for (int off = width >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// LPR lanes (64 or 32) per token, 256/LPR tokens per block; a lane owns V contiguous channels of each
// LPR*V-wide slab (<= 4 slabs).  C = 256 in bf16 runs as 32 lanes x 16 bytes: two tokens per wavefront.
template <typename T, int V, int LPR, int NSLAB>
__global__ void __launch_bounds__(256) ln_mod_fwd_kernel(const T *__restrict__ x, const T *__restrict__ scale,
                                                         const T *__restrict__ shift, T *__restrict__ y,
                                                         float *__restrict__ mean, float *__restrict__ rstd, int64_t M,
                                                         int N, int C, float eps, const T *__restrict__ res_y,
                                                         const T *__restrict__ res_gate, T *__restrict__ xnew, int64_t mp) {
    // res_y != nullptr: the input of the norm is the gated residual x + gate * res_y, which is also written to xnew
    // (same rounding points as gated_residual followed by ln_modulate).
    // Grid-stride over the tokens with the next token's x / res_y rows requested before the current one is normalised: a capped
    // grid of resident workgroups keeps more bytes in flight per CU than one tiny workgroup per 8 tokens (round 3).
    const int lane = threadIdx.x & (LPR - 1);
    constexpr int nslab = NSLAB;  // C == NSLAB * LPR * V
    const int64_t stride = (int64_t)gridDim.x * (256 / LPR);
    int64_t m = (int64_t)blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
    float v[NSLAB][V], ry[NSLAB][V], nv[NSLAB][V], nry[NSLAB][V];
    auto fetch = [&](int64_t mm, float (&vx)[NSLAB][V], float (&vy)[NSLAB][V]) {
#pragma unroll
        for (int sl = 0; sl < NSLAB; ++sl)
            if (sl < nslab) {
                Pack<T, V>::load(x + mm * C + (sl * LPR + lane) * V, vx[sl]);
                if (res_y) Pack<T, V>::load(res_y + mm * C + (sl * LPR + lane) * V, vy[sl]);
            }
    };
    if (m < M) fetch(m, v, ry);
    for (; m < M; m += stride) {
        const int64_t mn = m + stride;
        if (mn < M) fetch(mn, nv, nry);
        const int64_t b = m / N;
        float s = 0.f;
#pragma unroll
        for (int sl = 0; sl < NSLAB; ++sl)
            if (sl < nslab) {
                if (res_y) {
                    float rg[V];
                    Pack<T, V>::load(res_gate + b * mp + (sl * LPR + lane) * V, rg);
#pragma unroll
                    for (int e = 0; e < V; ++e) v[sl][e] = rnd<T>(v[sl][e] + rnd<T>(rg[e] * ry[sl][e]));
                    Pack<T, V>::store(xnew + m * C + (sl * LPR + lane) * V, v[sl]);
                }
#pragma unroll
                for (int e = 0; e < V; ++e) s += v[sl][e];
            }
        const float mu = seg_sum(s, LPR) / C;
        float q = 0.f;
#pragma unroll
        for (int sl = 0; sl < NSLAB; ++sl)
            if (sl < nslab)
#pragma unroll
                for (int e = 0; e < V; ++e) { const float d = v[sl][e] - mu; q += d * d; }
        const float rs = rsqrtf(seg_sum(q, LPR) / C + eps);
        if (lane == 0) { mean[m] = mu; rstd[m] = rs; }
#pragma unroll
        for (int sl = 0; sl < NSLAB; ++sl)
            if (sl < nslab) {
                const int c = (sl * LPR + lane) * V;
                float sc[V], sh[V], o[V];
                Pack<T, V>::load(scale + b * mp + c, sc); Pack<T, V>::load(shift + b * mp + c, sh);
#pragma unroll
                for (int e = 0; e < V; ++e) o[e] = (v[sl][e] - mu) * rs * (1.0f + sc[e]) + sh[e];
                Pack<T, V>::store(y + m * C + c, o);
            }
#pragma unroll
        for (int sl = 0; sl < NSLAB; ++sl)
#pragma unroll
            for (int e = 0; e < V; ++e) { v[sl][e] = nv[sl][e]; ry[sl][e] = nry[sl][e]; }
    }
}

// Backward passes that also need a per-(batch row, channel) sum over the N tokens.  Grid (nchunk, B): a block
// walks one chunk of the tokens of one batch row, 256/LPR tokens at a time, writes the token gradients and keeps
// the channel sums in registers; the token slots are combined through LDS and each block stores one fp32 partial
// row part[k][b][chunk][C].  colsum_finish_kernel adds the nchunk partials in a fixed order (deterministic).
constexpr int kColsumChunksMax = 64;
// token chunks per batch row: ~768 workgroups in total, at least four per row (per-workgroup prologue/epilogue is the overhead that grows
// with the chunk count: 4 chunks beat 8 by 5 % and 32 by 2x at B = 512), at least 8 tokens per chunk
static int colsum_chunks(int64_t B, int N) {
    static int force = -1;   // VSDE_COLSUM_CHUNKS: token chunks per batch row (A/B runs)
    if (force < 0) force = (int)vsde_knob("VSDE_COLSUM_CHUNKS", 0);
    if (force > 0) return force > kColsumChunksMax ? kColsumChunksMax : (force > N ? (N > 0 ? N : 1) : force);
    int64_t c = (768 + B - 1) / B;   // ~768 workgroups (B = 128, 101 tokens: 12 | 8 | 6 | 4 | 2 chunks = 3.73 | 3.59 | 3.57 | 3.58 | 3.62 ms per OU step)
    if (c < 4) c = 4;
    if (c > kColsumChunksMax) c = kColsumChunksMax;
    if (c > N / 8) c = N / 8 > 0 ? N / 8 : 1;
    return (int)c;
}

template <int NACC>
__device__ __forceinline__ void colsum_block_reduce(float *red, const float (&acc)[NACC], int slot, int slots, int lane_off,
                                                    int width, float *__restrict__ part, int C) {
    // red: [slots][width] floats; acc holds NACC consecutive channels starting at lane_off
    __syncthreads();
#pragma unroll
    for (int e = 0; e < NACC; ++e) red[slot * width + lane_off + e] = acc[e];
    __syncthreads();
    for (int c = threadIdx.x; c < width && c < C; c += blockDim.x) {
        float t = 0.f;
        for (int sidx = 0; sidx < slots; ++sidx) t += red[sidx * width + c];
        part[c] = t;
    }
}

// ln_modulate backward: dx (+ dres) per token, and partial sums of dy*xhat (part0) and dy (part1)
template <typename T, int V, int LPR, int NSLAB>
__global__ void __launch_bounds__(256) ln_mod_bwd_kernel(const T *__restrict__ x, const T *__restrict__ scale,
                                                         const T *__restrict__ dy, const float *__restrict__ mean,
                                                         const float *__restrict__ rstd, const T *__restrict__ dres,
                                                         T *__restrict__ dx, float *__restrict__ part, int N, int C,
                                                         const T *__restrict__ res_y, const T *__restrict__ res_gate,
                                                         T *__restrict__ res_dy, int64_t mp) {
    // res_y != nullptr: x is the output of a gated residual x0 + gate * res_y; dx is then also the gradient of x0, and the
    // kernel additionally writes res_dy = gate * dx and the partial sums of dx * res_y (part2: gradient of the gate)
    extern __shared__ float red[];
    constexpr int SLOTS = 256 / LPR;
    const int lane = threadIdx.x & (LPR - 1), slot = threadIdx.x / LPR;
    const int b = blockIdx.y, nchunk = gridDim.x, B = gridDim.y;
    const int CL = (N + nchunk - 1) / nchunk;
    const int n0 = blockIdx.x * CL, n1 = min(N, n0 + CL);
    constexpr int nslab = NSLAB;  // C == NSLAB * LPR * V
    float sc1[NSLAB][V], a1[NSLAB][V], a2[NSLAB][V], a3[NSLAB][V], gt[NSLAB][V];
#pragma unroll
    for (int sl = 0; sl < NSLAB; ++sl) {
        if (sl < nslab) {
            Pack<T, V>::load(scale + (int64_t)b * mp + (sl * LPR + lane) * V, sc1[sl]);
#pragma unroll
            for (int e = 0; e < V; ++e) sc1[sl][e] += 1.0f;
            if (res_y) Pack<T, V>::load(res_gate + (int64_t)b * mp + (sl * LPR + lane) * V, gt[sl]);
        }
#pragma unroll
        for (int e = 0; e < V; ++e) { a1[sl][e] = 0.f; a2[sl][e] = 0.f; a3[sl][e] = 0.f; }
    }
    for (int n = n0 + slot; n < n1; n += SLOTS) {
        const int64_t m = (int64_t)b * N + n;
        const float mu = mean[m], rs = rstd[m];
        float g[NSLAB][V], xh[NSLAB][V];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int sl = 0; sl < NSLAB; ++sl)
            if (sl < nslab) {
                const int c = (sl * LPR + lane) * V;
                Pack<T, V>::load(x + m * C + c, xh[sl]); Pack<T, V>::load(dy + m * C + c, g[sl]);
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    xh[sl][e] = (xh[sl][e] - mu) * rs;
                    a1[sl][e] += g[sl][e] * xh[sl][e]; a2[sl][e] += g[sl][e];
                    g[sl][e] *= sc1[sl][e];
                    s1 += g[sl][e]; s2 += g[sl][e] * xh[sl][e];
                }
            }
        s1 = seg_sum(s1, LPR) / C; s2 = seg_sum(s2, LPR) / C;
#pragma unroll
        for (int sl = 0; sl < NSLAB; ++sl)
            if (sl < nslab) {
                const int c = (sl * LPR + lane) * V;
                float o[V];
#pragma unroll
                for (int e = 0; e < V; ++e) o[e] = rs * (g[sl][e] - s1 - xh[sl][e] * s2);
                if (dres) {  // x also feeds the residual branch: fold that gradient in here instead of a separate add
                    float r[V];
                    Pack<T, V>::load(dres + m * C + c, r);
#pragma unroll
                    for (int e = 0; e < V; ++e) o[e] += r[e];
                }
                Pack<T, V>::store(dx + m * C + c, o);
                if (res_y) {
                    float ry[V], od[V];
                    Pack<T, V>::load(res_y + m * C + c, ry);
#pragma unroll
                    for (int e = 0; e < V; ++e) { const float t = rnd<T>(o[e]); a3[sl][e] += t * ry[e]; od[e] = gt[sl][e] * t; }
                    Pack<T, V>::store(res_dy + m * C + c, od);
                }
            }
    }
    float *p0 = part + (((int64_t)0 * B + b) * nchunk + blockIdx.x) * C;
    float *p1 = part + (((int64_t)1 * B + b) * nchunk + blockIdx.x) * C;
#pragma unroll
    for (int sl = 0; sl < NSLAB; ++sl)
        if (sl < nslab) {
            const int off = sl * LPR * V;
            colsum_block_reduce<V>(red, a1[sl], slot, SLOTS, lane * V, LPR * V, p0 + off, C - off);
            colsum_block_reduce<V>(red, a2[sl], slot, SLOTS, lane * V, LPR * V, p1 + off, C - off);
            if (res_y) {
                float *p2 = part + (((int64_t)2 * B + b) * nchunk + blockIdx.x) * C;
                colsum_block_reduce<V>(red, a3[sl], slot, SLOTS, lane * V, LPR * V, p2 + off, C - off);
            }
        }
}

// gated_residual backward: dy = gate * dout per token, and partial sums of dout * y (part0)
template <typename T, int V>
__global__ void __launch_bounds__(256) gated_residual_bwd_kernel(const T *__restrict__ y, const T *__restrict__ gate,
                                                                 const T *__restrict__ dout, T *__restrict__ dy,
                                                                 float *__restrict__ part, int N, int C, int64_t mp) {
    extern __shared__ float red[];
    const int lpt = C / V;                 // lanes per token
    const int slots = 256 / lpt;           // tokens in flight per block (threads beyond slots*lpt idle)
    const int slot = threadIdx.x / lpt, lane = threadIdx.x - slot * lpt;
    const int b = blockIdx.y, nchunk = gridDim.x;
    const int CL = (N + nchunk - 1) / nchunk;
    const int n0 = blockIdx.x * CL, n1 = min(N, n0 + CL);
    const bool live = slot < slots;
    float gv[V], acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) { gv[e] = 0.f; acc[e] = 0.f; }
    if (live) Pack<T, V>::load(gate + (int64_t)b * mp + lane * V, gv);
    if (live)
        for (int n = n0 + slot; n < n1; n += slots) {
            const int64_t o = ((int64_t)b * N + n) * C + lane * V;
            float g[V], yv[V], r[V];
            Pack<T, V>::load(dout + o, g); Pack<T, V>::load(y + o, yv);
#pragma unroll
            for (int e = 0; e < V; ++e) { r[e] = gv[e] * g[e]; acc[e] += g[e] * yv[e]; }
            Pack<T, V>::store(dy + o, r);
        }
    float *p0 = part + ((int64_t)b * nchunk + blockIdx.x) * C;
    __syncthreads();
    if (live) {
#pragma unroll
        for (int e = 0; e < V; ++e) red[slot * C + lane * V + e] = acc[e];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float t = 0.f;
        for (int sidx = 0; sidx < slots; ++sidx) t += red[sidx * C + c];
        p0[c] = t;
    }
}

// r_k[b][c] = sum_chunk part[k][b][chunk][c]  (k < nout), converted to T
template <typename T>
__global__ void __launch_bounds__(256) colsum_finish_kernel(const float *__restrict__ part, T *__restrict__ r0, T *__restrict__ r1,
                                                            T *__restrict__ r2, int64_t BC, int C, int nchunk, int64_t mp) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BC) return;
    const int64_t b = i / C;
    const int c = (int)(i - b * C);
    T *const outs[3] = {r0, r1, r2};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (!outs[k]) continue;
        const float *p = part + ((int64_t)k * (BC / C) + b) * nchunk * C + c;
        float t = 0.f;
        for (int ch0 = 0; ch0 < nchunk; ch0 += 8) {   // eight loads in flight, added in the fixed order (a serial chain of loads took ~0.5 us each)
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = ch0 + j < nchunk ? p[(int64_t)(ch0 + j) * C] : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) if (ch0 + j < nchunk) t += v[j];
        }
        outs[k][b * mp + c] = from_f32<T>(t);
    }
}

// --------------------------------------------------------------------------- gated_residual
template <typename T, int V>
__global__ void __launch_bounds__(256) gated_residual_kernel(const T *__restrict__ x, const T *__restrict__ y,
                                                             const T *__restrict__ gate, T *__restrict__ out, int64_t total,
                                                             int N, int C, int mode, int64_t mp) {
    // mode 0: out = x + gate*y ; mode 1: out = gate * y   (backward: dy = gate * dout, x unused)
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * V;
    for (int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * V; i0 < total; i0 += stride) {
        const int64_t m = i0 / C;
        const int c = (int)(i0 - m * C);
        const int64_t b = m / N;
        float gv[V], yv[V], xv[V], o[V];
        Pack<T, V>::load(gate + b * mp + c, gv); Pack<T, V>::load(y + i0, yv);
        if (mode == 0) Pack<T, V>::load(x + i0, xv);
#pragma unroll
        for (int e = 0; e < V; ++e) { const float g = gv[e] * yv[e]; o[e] = mode == 0 ? xv[e] + rnd<T>(g) : g; }
        Pack<T, V>::store(out + i0, o);
    }
}

// ----------------------------------------------------------------------------------- swiglu
template <typename T, int V>
__global__ void __launch_bounds__(256) swiglu_fwd_kernel(const T *__restrict__ u, T *__restrict__ out, int64_t M, int H2) {
    const int hv = H2 / V;
    const int64_t total = M * hv, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t m = i / hv;
        const int j = (int)(i - m * hv) * V;
        float a[V], bv[V], o[V];
        Pack<T, V>::load(u + m * 2 * H2 + j, a); Pack<T, V>::load(u + m * 2 * H2 + H2 + j, bv);
#pragma unroll
        for (int e = 0; e < V; ++e) o[e] = rnd<T>(a[e] * sigm(a[e])) * bv[e];
        Pack<T, V>::store(out + m * H2 + j, o);
    }
}

template <typename T, int V>
__global__ void __launch_bounds__(256) swiglu_bwd_kernel(const T *__restrict__ u, const T *__restrict__ dout, T *__restrict__ du,
                                                         int64_t M, int H2) {
    const int hv = H2 / V;
    const int64_t total = M * hv, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t m = i / hv;
        const int j = (int)(i - m * hv) * V;
        float a[V], bv[V], g[V], da[V], db[V];
        Pack<T, V>::load(u + m * 2 * H2 + j, a); Pack<T, V>::load(u + m * 2 * H2 + H2 + j, bv); Pack<T, V>::load(dout + m * H2 + j, g);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float sg = sigm(a[e]);
            da[e] = g[e] * bv[e] * sg * (1.0f + a[e] * (1.0f - sg)); db[e] = g[e] * a[e] * sg;
        }
        Pack<T, V>::store(du + m * 2 * H2 + j, da); Pack<T, V>::store(du + m * 2 * H2 + H2 + j, db);
    }
}

// ------------------------------------------------------------------------------- gate_merge
template <typename T, int V>
__global__ void __launch_bounds__(256) gate_merge_fwd_kernel(const T *__restrict__ attn, const T *__restrict__ glog,
                                                             T *__restrict__ out, int64_t M, int N, int heads, int d, int token_major,
                                                             int64_t gstride) {
    const int C = heads * d, cv = C / V;
    const int64_t total = M * cv, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t m = i / cv;
        const int c = (int)(i - m * cv) * V, hh = c / d, k = c - hh * d;
        const int64_t b = m / N, n = m - b * N;
        float gl[V], av[V], o[V];
        Pack<T, V>::load(glog + m * gstride + k, gl);
        Pack<T, V>::load(attn + (token_major ? m * C + c : ((b * heads + hh) * N + n) * d + k), av);
#pragma unroll
        for (int e = 0; e < V; ++e) o[e] = av[e] * rnd<T>(sigm(gl[e]));
        Pack<T, V>::store(out + m * C + c, o);
    }
}

template <typename T, int V>
__global__ void __launch_bounds__(256) gate_merge_bwd_kernel(const T *__restrict__ attn, const T *__restrict__ glog,
                                                             const T *__restrict__ dout, T *__restrict__ dattn,
                                                             T *__restrict__ dglog, int64_t M, int N, int heads, int d, int token_major,
                                                             int64_t gstride) {
    const int C = heads * d, dv = d / V;
    const int64_t total = M * dv, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t m = i / dv;
        const int k = (int)(i - m * dv) * V;
        const int64_t b = m / N, n = m - b * N;
        float gl[V], sg[V], acc[V];
        Pack<T, V>::load(glog + m * gstride + k, gl);
#pragma unroll
        for (int e = 0; e < V; ++e) { sg[e] = sigm(gl[e]); acc[e] = 0.f; }
        for (int hh = 0; hh < heads; ++hh) {
            const int64_t ai = token_major ? m * C + hh * d + k : ((b * heads + hh) * N + n) * d + k;
            float g[V], av[V], o[V];
            Pack<T, V>::load(dout + m * C + hh * d + k, g); Pack<T, V>::load(attn + ai, av);
#pragma unroll
            for (int e = 0; e < V; ++e) { o[e] = g[e] * sg[e]; acc[e] += g[e] * av[e]; }
            Pack<T, V>::store(dattn + ai, o);
        }
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] *= sg[e] * (1.0f - sg[e]);
        Pack<T, V>::store(dglog + m * gstride + k, acc);
    }
}

// ----------------------------------------------------------------------------- qk_norm_rope
// one thread per rotary pair (i, i + d/2) of one head of one token; the d/2 threads of a head are
// consecutive lanes (d/2 is a power of two <= 64), so the RMS reduction is a segmented butterfly.
// PV consecutive rotary pairs per thread (PV = 4: 8-byte bf16 / 16-byte f32 accesses); the head's
// half/PV threads are consecutive lanes, so the RMS reduction is a segmented butterfly of that width.
template <typename T, int PV>
__global__ void __launch_bounds__(256) qk_norm_rope_fwd_kernel(const T *__restrict__ qkv, const float *__restrict__ cosT,
                                                               const float *__restrict__ sinT, const float *__restrict__ wq,
                                                               const float *__restrict__ wk, const T *__restrict__ v0,
                                                               const float *__restrict__ lam, T *__restrict__ q, T *__restrict__ k,
                                                               T *__restrict__ v, int64_t M, int N, int heads, int d, float eps, int token_major,
                                                               int64_t rstride) {
    const int half = d >> 1, C = heads * d, tph = half / PV, TPT = heads * tph;  // threads per head / per token
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = gid < M * TPT;
    const int64_t m = ok ? gid / TPT : 0;
    const int pp = ok ? (int)(gid - m * TPT) : 0, hh = pp / tph, i = (pp - hh * tph) * PV;
    const int64_t b = m / N, n = m - b * N;
    const T *row = qkv + m * rstride + hh * d;
    float x[6][PV];  // q lo, q hi, k lo, k hi, v lo, v hi
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int e = 0; e < PV; ++e) x[t][e] = 0.f;
    if (ok) {
#pragma unroll
        for (int t = 0; t < 3; ++t) { Pack<T, PV>::load(row + t * C + i, x[2 * t]); Pack<T, PV>::load(row + t * C + i + half, x[2 * t + 1]); }
    }
    float sq = 0.f, sk = 0.f;
#pragma unroll
    for (int e = 0; e < PV; ++e) { sq += x[0][e] * x[0][e] + x[1][e] * x[1][e]; sk += x[2][e] * x[2][e] + x[3][e] * x[3][e]; }
    const float rq = rsqrtf(seg_sum(sq, tph) / d + eps);
    const float rk = rsqrtf(seg_sum(sk, tph) / d + eps);
    if (!ok) return;
    float cs[PV], sn[PV], wql[PV], wqh[PV], wkl[PV], wkh[PV];
    Pack<float, PV>::load(cosT + n * half + i, cs); Pack<float, PV>::load(sinT + n * half + i, sn);
    Pack<float, PV>::load(wq + i, wql); Pack<float, PV>::load(wq + i + half, wqh);
    Pack<float, PV>::load(wk + i, wkl); Pack<float, PV>::load(wk + i + half, wkh);
    const int64_t o = token_major ? (m * heads + hh) * d + i : ((b * heads + hh) * N + n) * d + i;
    float o0[PV], o1[PV];
#pragma unroll
    for (int e = 0; e < PV; ++e) {
        const float a0 = rnd<T>(x[0][e] * rq * wql[e]), a1 = rnd<T>(x[1][e] * rq * wqh[e]);
        o0[e] = a0 * cs[e] - a1 * sn[e]; o1[e] = a0 * sn[e] + a1 * cs[e];
    }
    Pack<T, PV>::store(q + o, o0); Pack<T, PV>::store(q + o + half, o1);
#pragma unroll
    for (int e = 0; e < PV; ++e) {
        const float a0 = rnd<T>(x[2][e] * rk * wkl[e]), a1 = rnd<T>(x[3][e] * rk * wkh[e]);
        o0[e] = a0 * cs[e] - a1 * sn[e]; o1[e] = a0 * sn[e] + a1 * cs[e];
    }
    Pack<T, PV>::store(k + o, o0); Pack<T, PV>::store(k + o + half, o1);
    if (v0) {
        const float l = lam[0];
        float p0[PV], p1[PV];
        Pack<T, PV>::load(v0 + o, p0); Pack<T, PV>::load(v0 + o + half, p1);
#pragma unroll
        for (int e = 0; e < PV; ++e) { x[4][e] = l * x[4][e] + (1.0f - l) * p0[e]; x[5][e] = l * x[5][e] + (1.0f - l) * p1[e]; }
    }
    Pack<T, PV>::store(v + o, x[4]); Pack<T, PV>::store(v + o + half, x[5]);
}

template <typename T, int PV>
__global__ void __launch_bounds__(256) qk_norm_rope_bwd_kernel(const T *__restrict__ qkv, const float *__restrict__ cosT,
                                                               const float *__restrict__ sinT, const float *__restrict__ wq,
                                                               const float *__restrict__ wk, const T *__restrict__ v0,
                                                               const float *__restrict__ lam, const T *__restrict__ dq,
                                                               const T *__restrict__ dk, const T *__restrict__ dv,
                                                               T *__restrict__ dqkv, T *__restrict__ dv0,
                                                               float *__restrict__ dlam_partial, int64_t M, int N, int heads, int d,
                                                               float eps, int token_major, int64_t rstride, int dv0_accumulate,
                                                               const T *__restrict__ dv_extra) {
    // dv0_accumulate: dv0 += (1 - lam) dv instead of dv0 = ... (the value-residual gradient of all blocks collects in ONE buffer);
    // dv_extra: added to dv first (block 0 receives that buffer next to the gradient of its own attention)
    __shared__ float red[4];
    const int half = d >> 1, C = heads * d, tph = half / PV, TPT = heads * tph;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = gid < M * TPT;
    const int64_t m = ok ? gid / TPT : 0;
    const int pp = ok ? (int)(gid - m * TPT) : 0, hh = pp / tph, i = (pp - hh * tph) * PV;
    const int64_t b = m / N, n = m - b * N;
    const T *row = qkv + m * rstride + hh * d;
    T *drow = dqkv + m * rstride + hh * d;
    const int64_t o = token_major ? (m * heads + hh) * d + i : ((b * heads + hh) * N + n) * d + i;
    float x[4][PV], gy[4][PV], w[4][PV];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < PV; ++e) { x[t][e] = 0.f; gy[t][e] = 0.f; w[t][e] = 0.f; }
    float dlam = 0.f;
    if (ok) {
        float cs[PV], sn[PV], g0[PV], g1[PV];
        Pack<float, PV>::load(cosT + n * half + i, cs); Pack<float, PV>::load(sinT + n * half + i, sn);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            Pack<T, PV>::load(row + t * C + i, x[2 * t]); Pack<T, PV>::load(row + t * C + i + half, x[2 * t + 1]);
            Pack<T, PV>::load((t == 0 ? dq : dk) + o, g0); Pack<T, PV>::load((t == 0 ? dq : dk) + o + half, g1);
            Pack<float, PV>::load((t == 0 ? wq : wk) + i, w[2 * t]); Pack<float, PV>::load((t == 0 ? wq : wk) + i + half, w[2 * t + 1]);
#pragma unroll
            for (int e = 0; e < PV; ++e) { gy[2 * t][e] = g0[e] * cs[e] + g1[e] * sn[e]; gy[2 * t + 1][e] = -g0[e] * sn[e] + g1[e] * cs[e]; }  // inverse rotation
        }
        Pack<T, PV>::load(dv + o, g0); Pack<T, PV>::load(dv + o + half, g1);
        if (dv_extra) {
            float e0[PV], e1[PV];
            Pack<T, PV>::load(dv_extra + o, e0); Pack<T, PV>::load(dv_extra + o + half, e1);
#pragma unroll
            for (int e = 0; e < PV; ++e) { g0[e] = rnd<T>(g0[e] + e0[e]); g1[e] = rnd<T>(g1[e] + e1[e]); }
        }
        if (v0) {
            const float l = lam[0];
            float a0[PV], a1[PV], p0[PV], p1[PV], z0[PV], z1[PV];
            Pack<T, PV>::load(row + 2 * C + i, a0); Pack<T, PV>::load(row + 2 * C + i + half, a1);
            Pack<T, PV>::load(v0 + o, p0); Pack<T, PV>::load(v0 + o + half, p1);
#pragma unroll
            for (int e = 0; e < PV; ++e) {
                dlam += g0[e] * (a0[e] - p0[e]) + g1[e] * (a1[e] - p1[e]);
                z0[e] = (1.0f - l) * g0[e]; z1[e] = (1.0f - l) * g1[e]; g0[e] *= l; g1[e] *= l;
            }
            if (dv0_accumulate) {
                float y0[PV], y1[PV];
                Pack<T, PV>::load(dv0 + o, y0); Pack<T, PV>::load(dv0 + o + half, y1);
#pragma unroll
                for (int e = 0; e < PV; ++e) { z0[e] = y0[e] + rnd<T>(z0[e]); z1[e] = y1[e] + rnd<T>(z1[e]); }
            }
            Pack<T, PV>::store(dv0 + o, z0); Pack<T, PV>::store(dv0 + o + half, z1);
        }
        Pack<T, PV>::store(drow + 2 * C + i, g0); Pack<T, PV>::store(drow + 2 * C + i + half, g1);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {  // RMS backward: dx = r w dy - x r^3 mean(x w dy)
        float ss = 0.f, dt = 0.f;
#pragma unroll
        for (int e = 0; e < PV; ++e) {
            ss += x[2 * t][e] * x[2 * t][e] + x[2 * t + 1][e] * x[2 * t + 1][e];
            dt += x[2 * t][e] * w[2 * t][e] * gy[2 * t][e] + x[2 * t + 1][e] * w[2 * t + 1][e] * gy[2 * t + 1][e];
        }
        const float r = rsqrtf(seg_sum(ss, tph) / d + eps);
        const float dot = seg_sum(dt, tph) / d;
        if (ok) {
            float o0[PV], o1[PV];
#pragma unroll
            for (int e = 0; e < PV; ++e) {
                o0[e] = r * w[2 * t][e] * gy[2 * t][e] - x[2 * t][e] * r * r * r * dot;
                o1[e] = r * w[2 * t + 1][e] * gy[2 * t + 1][e] - x[2 * t + 1][e] * r * r * r * dot;
            }
            Pack<T, PV>::store(drow + t * C + i, o0); Pack<T, PV>::store(drow + t * C + i + half, o1);
        }
    }
    if (dlam_partial) {
        dlam = wave_sum(dlam);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dlam;
        __syncthreads();
        if (threadIdx.x == 0) dlam_partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    }
}

static inline int ew_grid(int64_t total, int per_block) {
    int64_t g = (total + per_block - 1) / per_block;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

static int64_t ln_fwd_grid_cap() {   // resident workgroups of the grid-stride forward (VSDE_LN_GRID overrides, A/B runs)
    static int64_t v = 0;
    if (!v) { v = vsde_knob("VSDE_LN_GRID", 4096); if (v <= 0) v = 4096; }
    return v;
}

struct LnResidual {  // optional gated residual fused in front of the norm (forward) / behind its backward
    const void *y = nullptr;     // residual branch [B][N][C]
    const void *gate = nullptr;  // [B][C]
    void *out = nullptr;         // forward: x + gate*y; backward: gate * dx
    int64_t mp = 0;              // row pitch (elements) of scale / shift / gate and of their gradients; 0 = C (contiguous)
};

template <typename T>
static int ln_mod_dispatch(int which, const void *x, const void *scale, const void *shift_or_dy, const void *dres, void *out,
                           float *mean, float *rstd, float *part, int64_t B, int N, int C, float eps, hipStream_t s,
                           LnResidual res = LnResidual()) {
    VSDE_CHECK_ARG(C % 64 == 0 && C <= 1024, VSDE_E_BADARG, "ln_modulate needs C %% 64 == 0 and C <= 1024, got %d", C);
    constexpr int VF = VecOf<T>::v;  // 8 bf16 / 4 f32 = 16 bytes per lane
    const int64_t M = B * N;
    dim3 block(256);
#define LNM_N(V, LPR, NS)                                                                                                     \
    do {                                                                                                                      \
        if (which == 0) {                                                                                                     \
            int64_t nblk = (M + 256 / LPR - 1) / (256 / LPR);                                                                 \
            const int64_t cap = ln_fwd_grid_cap();                                                                            \
            dim3 grid((unsigned)(nblk < cap ? nblk : cap));                                                                   \
            hipLaunchKernelGGL((ln_mod_fwd_kernel<T, V, LPR, NS>), grid, block, 0, s, (const T *)x, (const T *)scale,         \
                               (const T *)shift_or_dy, (T *)out, mean, rstd, M, N, C, eps, (const T *)res.y,                  \
                               (const T *)res.gate, (T *)res.out, res.mp ? res.mp : (int64_t)C);                              \
        } else {                                                                                                              \
            hipLaunchKernelGGL((ln_mod_bwd_kernel<T, V, LPR, NS>), dim3(colsum_chunks(B, N), (unsigned)B), block,                   \
                               256 * V * sizeof(float), s, (const T *)x, (const T *)scale, (const T *)shift_or_dy,            \
                               (const float *)mean, (const float *)rstd, (const T *)dres, (T *)out, part, N, C,               \
                               (const T *)res.y, (const T *)res.gate, (T *)res.out, res.mp ? res.mp : (int64_t)C);            \
        }                                                                                                                     \
    } while (0)
#define LNM(V, LPR)                                                                                                           \
    do {                                                                                                                      \
        switch (C / ((LPR) * (V))) {                                                                                          \
            case 1: LNM_N(V, LPR, 1); break;                                                                                  \
            case 2: LNM_N(V, LPR, 2); break;                                                                                  \
            case 3: LNM_N(V, LPR, 3); break;                                                                                  \
            default: LNM_N(V, LPR, 4); break;                                                                                 \
        }                                                                                                                     \
    } while (0)
    if (C % (64 * VF) == 0 && C / (64 * VF) <= 4) LNM(VF, 64);
    else if (C % (32 * VF) == 0 && C / (32 * VF) <= 4) LNM(VF, 32);
    else if (C % (64 * (VF / 2)) == 0 && C / (64 * (VF / 2)) <= 4) LNM(VF / 2, 64);
    else if (C % (64 * (VF / 4)) == 0 && C / (64 * (VF / 4)) <= 4) LNM(VF / 4, 64);
    else {
        VSDE_CHECK_ARG(C / 64 <= 4, VSDE_E_BADARG, "ln_modulate: unsupported channel count %d", C);
        LNM(1, 64);
    }
#undef LNM_N
#undef LNM
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace vsde

using namespace vsde;

#define VSDE_DTYPE_SWITCH(dtype, CALL)                                                                       \
    do {                                                                                                     \
        if ((dtype) == 0) { using T = float; CALL; }                                                         \
        else if ((dtype) == 1) { using T = bf16_t; CALL; }                                                   \
        else { set_error("dtype must be 0 (f32) or 1 (bf16), got %d", (dtype)); return VSDE_E_BADARG; }      \
    } while (0)

// per-batch-row vectors (scale, shift, gate and their gradients) may be column ranges of one wider [B][pitch] buffer
#define MOD_PITCH_OK(mp, C) ((mp) == 0 || ((mp) >= (C) && (mp) % 8 == 0))

extern "C" int vsde_ln_modulate_fwd(int dtype, const void *x, const void *scale, const void *shift, void *y, float *mean,
                                    float *rstd, int64_t B, int N, int C, double eps, int64_t mod_pitch, void *stream) {
    VSDE_CHECK_ARG(x && scale && shift && y && mean && rstd && B > 0 && N > 0 && MOD_PITCH_OK(mod_pitch, C), VSDE_E_BADARG,
                   "bad ln_modulate arguments");
    LnResidual res; res.mp = mod_pitch;
    VSDE_DTYPE_SWITCH(dtype, return ln_mod_dispatch<T>(0, x, scale, shift, nullptr, y, mean, rstd, nullptr, B, N, C, (float)eps,
                                                       (hipStream_t)stream, res));
}

extern "C" size_t vsde_colsum_workspace_bytes(int64_t B, int C) {
    return (size_t)3 * (size_t)B * kColsumChunksMax * (size_t)C * sizeof(float);
}

extern "C" int vsde_ln_modulate_bwd(int dtype, const void *x, const void *scale, const void *dy, const float *mean,
                                    const float *rstd, const void *dres, void *dx, void *dscale, void *dshift, int64_t B,
                                    int N, int C, int64_t mod_pitch, void *workspace, size_t workspace_bytes, void *stream) {
    VSDE_CHECK_ARG(x && scale && dy && mean && rstd && dx && dscale && dshift && B > 0 && N > 0 && MOD_PITCH_OK(mod_pitch, C),
                   VSDE_E_BADARG, "bad ln_modulate_bwd arguments");
    VSDE_CHECK_ARG(workspace && workspace_bytes >= vsde_colsum_workspace_bytes(B, C), VSDE_E_WORKSPACE,
                   "ln_modulate_bwd workspace too small");
    hipStream_t s = (hipStream_t)stream;
    float *part = (float *)workspace;
    VSDE_DTYPE_SWITCH(dtype, {
        LnResidual res; res.mp = mod_pitch;
        int rc = ln_mod_dispatch<T>(1, x, scale, dy, dres, dx, (float *)mean, (float *)rstd, part, B, N, C, 0.f, s, res);
        if (rc) return rc;
        const int64_t BC = B * C;
        hipLaunchKernelGGL((colsum_finish_kernel<T>), dim3((unsigned)((BC + 255) / 256)), dim3(256), 0, s, (const float *)part,
                           (T *)dscale, (T *)dshift, (T *)nullptr, BC, C, colsum_chunks(B, N), mod_pitch ? mod_pitch : (int64_t)C);
    });
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int vsde_residual_ln_fwd(int dtype, const void *x, const void *y, const void *gate, const void *scale, const void *shift,
                                    void *xnew, void *h, float *mean, float *rstd, int64_t B, int N, int C, double eps,
                                    int64_t mod_pitch, void *stream) {
    VSDE_CHECK_ARG(x && y && gate && scale && shift && xnew && h && mean && rstd && B > 0 && N > 0 && MOD_PITCH_OK(mod_pitch, C),
                   VSDE_E_BADARG, "bad residual_ln arguments");
    LnResidual res; res.y = y; res.gate = gate; res.out = xnew; res.mp = mod_pitch;
    VSDE_DTYPE_SWITCH(dtype, return ln_mod_dispatch<T>(0, x, scale, shift, nullptr, h, mean, rstd, nullptr, B, N, C, (float)eps,
                                                       (hipStream_t)stream, res));
}

extern "C" int vsde_residual_ln_bwd(int dtype, const void *xnew, const void *y, const void *gate, const void *scale, const void *dh,
                                    const void *dxnew, const float *mean, const float *rstd, void *dx, void *dy, void *dgate,
                                    void *dscale, void *dshift, int64_t B, int N, int C, int64_t mod_pitch, void *workspace,
                                    size_t workspace_bytes, void *stream) {
    VSDE_CHECK_ARG(xnew && y && gate && scale && dh && mean && rstd && dx && dy && dgate && dscale && dshift && B > 0 && N > 0 &&
                       MOD_PITCH_OK(mod_pitch, C), VSDE_E_BADARG, "bad residual_ln_bwd arguments");
    VSDE_CHECK_ARG(workspace && workspace_bytes >= vsde_colsum_workspace_bytes(B, C), VSDE_E_WORKSPACE,
                   "residual_ln_bwd workspace too small");
    hipStream_t s = (hipStream_t)stream;
    float *part = (float *)workspace;
    LnResidual res; res.y = y; res.gate = gate; res.out = dy; res.mp = mod_pitch;
    VSDE_DTYPE_SWITCH(dtype, {
        int rc = ln_mod_dispatch<T>(1, xnew, scale, dh, dxnew, dx, (float *)mean, (float *)rstd, part, B, N, C, 0.f, s, res);
        if (rc) return rc;
        const int64_t BC = B * C;
        hipLaunchKernelGGL((colsum_finish_kernel<T>), dim3((unsigned)((BC + 255) / 256)), dim3(256), 0, s, (const float *)part,
                           (T *)dscale, (T *)dshift, (T *)dgate, BC, C, colsum_chunks(B, N), mod_pitch ? mod_pitch : (int64_t)C);
    });
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int vsde_gated_residual_fwd(int dtype, const void *x, const void *y, const void *gate, void *out, int64_t B, int N,
                                       int C, int64_t mod_pitch, void *stream) {
    VSDE_CHECK_ARG(x && y && gate && out && C % 4 == 0 && MOD_PITCH_OK(mod_pitch, C), VSDE_E_BADARG, "bad gated_residual arguments");
    const int64_t total = B * N * C, mp = mod_pitch ? mod_pitch : (int64_t)C;
    VSDE_DTYPE_SWITCH(dtype, {
        constexpr int VF = VecOf<T>::v;
        if (C % VF == 0) hipLaunchKernelGGL((gated_residual_kernel<T, VF>), dim3(ew_grid(total, 256 * VF)), dim3(256), 0, (hipStream_t)stream,
                                            (const T *)x, (const T *)y, (const T *)gate, (T *)out, total, N, C, 0, mp);
        else hipLaunchKernelGGL((gated_residual_kernel<T, 4>), dim3(ew_grid(total, 1024)), dim3(256), 0, (hipStream_t)stream,
                                (const T *)x, (const T *)y, (const T *)gate, (T *)out, total, N, C, 0, mp);
    });
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int vsde_gated_residual_bwd(int dtype, const void *y, const void *gate, const void *dout, void *dy, void *dgate,
                                       int64_t B, int N, int C, int64_t mod_pitch, void *workspace, size_t workspace_bytes,
                                       void *stream) {
    VSDE_CHECK_ARG(y && gate && dout && dy && dgate && C % 4 == 0 && C <= 2048 && MOD_PITCH_OK(mod_pitch, C), VSDE_E_BADARG,
                   "bad gated_residual_bwd arguments");
    const int64_t mp = mod_pitch ? mod_pitch : (int64_t)C;
    VSDE_CHECK_ARG(workspace && workspace_bytes >= vsde_colsum_workspace_bytes(B, C) / 2, VSDE_E_WORKSPACE,
                   "gated_residual_bwd workspace too small");
    hipStream_t s = (hipStream_t)stream;
    float *part = (float *)workspace;
    const int64_t BC = B * C;
    VSDE_DTYPE_SWITCH(dtype, {
        constexpr int VF = VecOf<T>::v;
        dim3 grid(colsum_chunks(B, N), (unsigned)B);
        if (C % VF == 0 && C / VF <= 256)
            hipLaunchKernelGGL((gated_residual_bwd_kernel<T, VF>), grid, dim3(256), 256 * VF * sizeof(float), s, (const T *)y,
                               (const T *)gate, (const T *)dout, (T *)dy, part, N, C, mp);
        else
            hipLaunchKernelGGL((gated_residual_bwd_kernel<T, 4>), grid, dim3(256), 256 * 4 * sizeof(float), s, (const T *)y,
                               (const T *)gate, (const T *)dout, (T *)dy, part, N, C, mp);
        hipLaunchKernelGGL((colsum_finish_kernel<T>), dim3((unsigned)((BC + 255) / 256)), dim3(256), 0, s, (const float *)part,
                           (T *)dgate, (T *)nullptr, (T *)nullptr, BC, C, colsum_chunks(B, N), mp);
    });
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int vsde_swiglu_fwd(int dtype, const void *u, void *out, int64_t M, int H2, void *stream) {
    VSDE_CHECK_ARG(u && out && M > 0 && H2 > 0, VSDE_E_BADARG, "bad swiglu arguments");
    VSDE_DTYPE_SWITCH(dtype, {
        constexpr int VF = VecOf<T>::v;  // 16-byte accesses when the (padded) hidden width allows
        if (H2 % VF == 0) hipLaunchKernelGGL((swiglu_fwd_kernel<T, VF>), dim3(ew_grid(M * H2 / VF, 256)), dim3(256), 0, (hipStream_t)stream,
                                             (const T *)u, (T *)out, M, H2);
        else if (H2 % 2 == 0) hipLaunchKernelGGL((swiglu_fwd_kernel<T, 2>), dim3(ew_grid(M * H2 / 2, 256)), dim3(256), 0, (hipStream_t)stream,
                                                 (const T *)u, (T *)out, M, H2);
        else hipLaunchKernelGGL((swiglu_fwd_kernel<T, 1>), dim3(ew_grid(M * H2, 256)), dim3(256), 0, (hipStream_t)stream,
                                (const T *)u, (T *)out, M, H2);
    });
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int vsde_swiglu_bwd(int dtype, const void *u, const void *dout, void *du, int64_t M, int H2, void *stream) {
    VSDE_CHECK_ARG(u && dout && du && M > 0 && H2 > 0, VSDE_E_BADARG, "bad swiglu_bwd arguments");
    VSDE_DTYPE_SWITCH(dtype, {
        constexpr int VF = VecOf<T>::v;
        if (H2 % VF == 0) hipLaunchKernelGGL((swiglu_bwd_kernel<T, VF>), dim3(ew_grid(M * H2 / VF, 256)), dim3(256), 0, (hipStream_t)stream,
                                             (const T *)u, (const T *)dout, (T *)du, M, H2);
        else if (H2 % 2 == 0) hipLaunchKernelGGL((swiglu_bwd_kernel<T, 2>), dim3(ew_grid(M * H2 / 2, 256)), dim3(256), 0, (hipStream_t)stream,
                                                 (const T *)u, (const T *)dout, (T *)du, M, H2);
        else hipLaunchKernelGGL((swiglu_bwd_kernel<T, 1>), dim3(ew_grid(M * H2, 256)), dim3(256), 0, (hipStream_t)stream,
                                (const T *)u, (const T *)dout, (T *)du, M, H2);
    });
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int vsde_gate_merge_fwd(int dtype, const void *attn, const void *glog, void *out, int64_t B, int N, int heads, int d,
                                   int token_major, int64_t glog_stride, void *stream) {
    VSDE_CHECK_ARG(attn && glog && out && glog_stride >= d, VSDE_E_BADARG, "bad gate_merge arguments");
    VSDE_DTYPE_SWITCH(dtype, {
        constexpr int VF = VecOf<T>::v;
        if (d % VF == 0) hipLaunchKernelGGL((gate_merge_fwd_kernel<T, VF>), dim3(ew_grid(B * N * heads * d / VF, 256)), dim3(256), 0,
                                            (hipStream_t)stream, (const T *)attn, (const T *)glog, (T *)out, B * N, N, heads, d, token_major, glog_stride);
        else hipLaunchKernelGGL((gate_merge_fwd_kernel<T, 1>), dim3(ew_grid(B * N * heads * d, 256)), dim3(256), 0,
                                (hipStream_t)stream, (const T *)attn, (const T *)glog, (T *)out, B * N, N, heads, d, token_major, glog_stride);
    });
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int vsde_gate_merge_bwd(int dtype, const void *attn, const void *glog, const void *dout, void *dattn, void *dglog,
                                   int64_t B, int N, int heads, int d, int token_major, int64_t glog_stride,
                                   void *stream) {
    VSDE_CHECK_ARG(attn && glog && dout && dattn && dglog && glog_stride >= d, VSDE_E_BADARG, "bad gate_merge_bwd arguments");
    VSDE_DTYPE_SWITCH(dtype, {
        constexpr int VF = VecOf<T>::v;
        if (d % VF == 0) hipLaunchKernelGGL((gate_merge_bwd_kernel<T, VF>), dim3(ew_grid(B * N * d / VF, 256)), dim3(256), 0,
                                            (hipStream_t)stream, (const T *)attn, (const T *)glog, (const T *)dout, (T *)dattn,
                                            (T *)dglog, B * N, N, heads, d, token_major, glog_stride);
        else hipLaunchKernelGGL((gate_merge_bwd_kernel<T, 1>), dim3(ew_grid(B * N * d, 256)), dim3(256), 0,
                                (hipStream_t)stream, (const T *)attn, (const T *)glog, (const T *)dout, (T *)dattn,
                                (T *)dglog, B * N, N, heads, d, token_major, glog_stride);
    });
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}

static int qk_pv(int d) { return ((d / 2) % 4 == 0) ? 4 : 1; }  // rotary pairs per thread

static int qk_check(int heads, int d) {
    const int half = d / 2;
    VSDE_CHECK_ARG(d % 2 == 0 && half >= 1 && half <= 64 && (half & (half - 1)) == 0, VSDE_E_BADARG,
                   "qk_norm_rope needs head_dim/2 to be a power of two <= 64, got head_dim %d", d);
    VSDE_CHECK_ARG(heads > 0 && (256 % half) == 0, VSDE_E_BADARG, "bad head count");
    return 0;
}

extern "C" int vsde_qk_norm_rope_fwd(int dtype, const void *qkv, const float *cosT, const float *sinT, const float *wq,
                                     const float *wk, const void *v0, const float *lam, void *q, void *k, void *v, int64_t B,
                                     int N, int heads, int d, double eps, int token_major, int64_t row_stride,
                                     void *stream) {
    VSDE_CHECK_ARG(qkv && cosT && sinT && wq && wk && q && k && v && (!v0 || lam) && row_stride >= 3 * (int64_t)heads * d,
                   VSDE_E_BADARG, "bad qk_norm_rope arguments");
    int rc = qk_check(heads, d);
    if (rc) return rc;
    const int pv = qk_pv(d);
    const int64_t threads = B * N * heads * (d / 2 / pv);
    VSDE_DTYPE_SWITCH(dtype, {
        if (pv == 4) hipLaunchKernelGGL((qk_norm_rope_fwd_kernel<T, 4>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                                        (hipStream_t)stream, (const T *)qkv, cosT, sinT, wq, wk, (const T *)v0, lam, (T *)q, (T *)k,
                                        (T *)v, B * N, N, heads, d, (float)eps, token_major, row_stride);
        else hipLaunchKernelGGL((qk_norm_rope_fwd_kernel<T, 1>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                                (hipStream_t)stream, (const T *)qkv, cosT, sinT, wq, wk, (const T *)v0, lam, (T *)q, (T *)k, (T *)v,
                                B * N, N, heads, d, (float)eps, token_major, row_stride);
    });
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int64_t vsde_qk_norm_rope_bwd_partials(int64_t B, int N, int heads, int d) {
    return (B * N * heads * (d / 2 / qk_pv(d)) + 255) / 256;
}

extern "C" int vsde_qk_norm_rope_bwd(int dtype, const void *qkv, const float *cosT, const float *sinT, const float *wq,
                                     const float *wk, const void *v0, const float *lam, const void *dq, const void *dk,
                                     const void *dv, void *dqkv, void *dv0, float *dlam_partial, int64_t B, int N, int heads, int d,
                                     double eps, int token_major, int64_t row_stride, int dv0_accumulate, const void *dv_extra,
                                     void *stream) {
    VSDE_CHECK_ARG(qkv && cosT && sinT && wq && wk && dq && dk && dv && dqkv && (!v0 || (lam && dv0 && dlam_partial)) &&
                       row_stride >= 3 * (int64_t)heads * d, VSDE_E_BADARG,
                   "bad qk_norm_rope_bwd arguments");
    int rc = qk_check(heads, d);
    if (rc) return rc;
    const int pv = qk_pv(d);
    const int64_t threads = B * N * heads * (d / 2 / pv);
    VSDE_DTYPE_SWITCH(dtype, {
        if (pv == 4) hipLaunchKernelGGL((qk_norm_rope_bwd_kernel<T, 4>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                                        (hipStream_t)stream, (const T *)qkv, cosT, sinT, wq, wk, (const T *)v0, lam, (const T *)dq,
                                        (const T *)dk, (const T *)dv, (T *)dqkv, (T *)dv0, v0 ? dlam_partial : nullptr, B * N, N, heads, d,
                                        (float)eps, token_major, row_stride, dv0_accumulate, (const T *)dv_extra);
        else hipLaunchKernelGGL((qk_norm_rope_bwd_kernel<T, 1>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                                (hipStream_t)stream, (const T *)qkv, cosT, sinT, wq, wk, (const T *)v0, lam, (const T *)dq,
                                (const T *)dk, (const T *)dv, (T *)dqkv, (T *)dv0, v0 ? dlam_partial : nullptr, B * N, N, heads, d,
                                (float)eps, token_major, row_stride, dv0_accumulate, (const T *)dv_extra);
    });
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}
