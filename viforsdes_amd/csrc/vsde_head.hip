// Fused GRU DiffusionTransitionHead: batched Euler-Maruyama path sampler, forward and
// reverse-time backward, for gfx950 (MI355X).
//
// Replaces sde_fwd_kernel / sde_bwd_kernel of the reference
// (src/variational_sde/kernels/forward.py:91-375, backward.py:156-624).  Design:
//   * one wavefront (64 lanes) per sample path, lane j owns hidden unit j of every layer;
//   * the recurrent matrices (W_hh_l0, W_ih_l1, W_hh_l1: 3 x 48 KB fp32) are staged once per
//     workgroup in LDS in a [k/4][gate][j][k%4] layout so that a lane reads its weights with
//     conflict-free ds_read_b128 and the hidden state is broadcast with ds_read_b128;
//   * the context projection ctx_t . W_c (the reference's C-long scalar loop per step,
//     kernels/helpers.py:42-72) is NOT recurrent: it is hoisted into one fp32-MFMA GEMM over all
//     B*T path-steps (vsde_gemm.hip) and the time loop only streams 3H coalesced floats/step;
//   * after h^l_t is known the wave computes, in one pass over h^l_t, both the input
//     pre-activations of layer l+1 (or the emission rows) and the recurrent pre-activations
//     c^l_{t+1} = b_hh + W_hh h^l_t of the NEXT step, so each step has L broadcast rounds;
//   * the backward only carries d_x and d_h^l through time and emits the gate pre-activation
//     gradients; every weight gradient is a deterministic MFMA reduction afterwards instead of
//     ~87k global atomics per path-step (backward.py:108-139,575-590).
// All arithmetic is fp32 (README.md:95 of the reference; kernels/autograd.py:80-83).
#include "vsde_common.h"

namespace vsde {

#ifdef VSDE_TRACE
// debug build only: cycle stamps of one wave for one time step (tools/trace_fwd.py)
__device__ long long g_trace[64];
#define VSDE_TP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0 && c == 1 && tt == 5) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); g_trace[(k)] = clock64(); } } while (0)
#define VSDE_TPB(k) do { if (blockIdx.x == 0 && threadIdx.x == 0 && c == 1 && (tt == 5 || ((k) == 31 && tt == 4))) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); g_trace[(k)] = clock64(); } } while (0)
#else
#define VSDE_TP(k) do { } while (0)
#define VSDE_TPB(k) do { } while (0)
#endif

// ----------------------------------------------------------------------------- packing
// Matrix order q: 0 = W_hh_l0, 2l-1 = W_ih_l(l), 2l = W_hh_l(l)  (l >= 1).
struct PackParams {
    int S, P, C, H, L, NO;
    const float *W_ih0, *W_hh0, *W_ih_st, *W_hh_st, *out_W;
    float4 *packF;  // [2L-1][16][3][64]  forward layout  (k-chunk, gate, lane j)      or nullptr
    float4 *packO;  // [16][NO]            emission rows   (k-chunk, row)                or nullptr
    float *Wc;      // [3H][C]             context block of W_ih_l0, contiguous          or nullptr
    float4 *packB;  // [2L-1][3][16][64]  backward layout (gate, j-chunk, lane i)       or nullptr
    float *WcT;     // [C][3H]                                                          or nullptr
};

__device__ __forceinline__ const float *mat_ptr(const PackParams &p, int q) {
    if (q == 0) return p.W_hh0;
    int l = (q + 1) >> 1;
    return ((q & 1) ? p.W_ih_st : p.W_hh_st) + (int64_t)(l - 1) * 3 * p.H * p.H;
}

__global__ void pack_weights_kernel(PackParams p) {
    const int H = p.H, I = p.S + p.C + p.P, nmat = 2 * p.L - 1;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, tid0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p.packF) {
        for (int64_t e = tid0; e < (int64_t)nmat * kMatF4; e += stride) {
            int j = e % kHP, g = (e / kHP) % 3, c = (e / (kHP * 3)) % kChunks, q = e / kMatF4;
            const float *W = mat_ptr(p, q);
            float v[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                int k = 4 * c + x;
                v[x] = (j < H && k < H) ? W[(int64_t)(g * H + j) * H + k] : 0.f;
            }
            p.packF[e] = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
    if (p.packO) {
        for (int64_t e = tid0; e < (int64_t)kChunks * p.NO; e += stride) {
            int r = e % p.NO, c = e / p.NO;
            float v[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) { int k = 4 * c + x; v[x] = k < H ? p.out_W[(int64_t)r * H + k] : 0.f; }
            p.packO[e] = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
    if (p.Wc) {
        for (int64_t e = tid0; e < (int64_t)3 * H * p.C; e += stride) {
            int c = e % p.C, n = e / p.C;
            p.Wc[e] = p.W_ih0[(int64_t)n * I + p.S + c];
        }
    }
    if (p.packB) {
        for (int64_t e = tid0; e < (int64_t)nmat * kMatF4; e += stride) {
            int i = e % kHP, jc = (e / kHP) % kChunks, g = (e / (kHP * kChunks)) % 3, q = e / kMatF4;
            const float *W = mat_ptr(p, q);
            float v[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                int j = 4 * jc + x;
                v[x] = (j < H && i < H) ? W[(int64_t)(g * H + j) * H + i] : 0.f;
            }
            p.packB[e] = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
    if (p.WcT) {
        for (int64_t e = tid0; e < (int64_t)3 * H * p.C; e += stride) {
            int n = e % (3 * H), c = e / (3 * H);
            p.WcT[e] = p.W_ih0[(int64_t)n * I + p.S + c];
        }
    }
}

// ----------------------------------------------------------------------------- forward
struct FwdParams {
    int B, T, S, P, C, H, NO, ntril, wpb;
    const float *x0, *theta, *eps;
    const float *G;  // [B*T][3H] = ctx . W_c^T + b_ih_l0
    const float *W_ih0, *b_hh0, *b_ih_st, *b_hh_st, *out_b;
    const float *W_hh0, *W_ih_st, *W_hh_st, *out_W;  // native tensors (v2 kernels load them straight into VGPRs)
    const float4 *packF, *packO;
    float dt, sqdt, diag_min;
    float *paths, *means, *chol, *chol_raw, *acts;
};

constexpr int kScratchPerWave = 3 * 64;  // hbuf, obuf, ebuf
constexpr int kMaxSReg = 8;              // state rows of W_ih_l0 kept in registers
constexpr int kMaxSRegV2 = 4;            // same for the register-heavy v2 kernels

__host__ __device__ inline int fwd_lds_matrices(int L) { int n = 2 * L - 1; return n < 3 ? n : 3; }

// acc[g] += sum_k h[k] * W[k][g][lane] for one packed matrix; hv = 4 broadcast values of chunk c
__device__ __forceinline__ void fma3(const float4 *__restrict__ W, int c, int lane, const float4 &hv, float (&acc)[3]) {
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        float4 w = W[(c * 3 + g) * kHP + lane];
        acc[g] = fmaf(hv.x, w.x, acc[g]); acc[g] = fmaf(hv.y, w.y, acc[g]);
        acc[g] = fmaf(hv.z, w.z, acc[g]); acc[g] = fmaf(hv.w, w.w, acc[g]);
    }
}

template <int L, bool SAVE>
__global__ void __launch_bounds__(512) head_fwd_kernel(FwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float4 smem4[];
    constexpr int NMAT = 2 * L - 1;
    constexpr int NLDS = NMAT < 3 ? NMAT : 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * p.wpb + wave;
    const int H = p.H, S = p.S, T = p.T, NO = p.NO;

    float4 *Wl = smem4;                       // NLDS packed matrices
    float4 *POl = Wl + NLDS * kMatF4;         // emission rows [16][NO]
    float *scratch = (float *)(POl + kChunks * NO) + wave * kScratchPerWave;
    float *hbuf = scratch, *obuf = scratch + 64, *ebuf = scratch + 128;

    for (int i = threadIdx.x; i < NLDS * kMatF4; i += blockDim.x) Wl[i] = p.packF[i];
    for (int i = threadIdx.x; i < kChunks * NO; i += blockDim.x) POl[i] = p.packO[i];
    __syncthreads();
    if (b >= p.B) return;

    const int I = S + p.C + p.P, G3 = 3 * H;
    const bool act = lane < H;
    const int nchunk = (H + 3) >> 2;
    // per-lane constants
    float bhh[L][3], bih[L][3], gth[3], wx[kMaxSReg][3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        bhh[0][g] = act ? p.b_hh0[g * H + lane] : 0.f;
        bih[0][g] = 0.f;  // folded into G by the projection GEMM
#pragma unroll
        for (int l = 1; l < L; ++l) {
            bhh[l][g] = act ? p.b_hh_st[(l - 1) * G3 + g * H + lane] : 0.f;
            bih[l][g] = act ? p.b_ih_st[(l - 1) * G3 + g * H + lane] : 0.f;
        }
        float acc = 0.f;  // hoisted theta projection (reference: forward.py:157-175)
        if (act)
            for (int q = 0; q < p.P; ++q) acc = fmaf(p.theta[(int64_t)b * p.P + q], p.W_ih0[(int64_t)(g * H + lane) * I + S + p.C + q], acc);
        gth[g] = acc;
#pragma unroll
        for (int i = 0; i < kMaxSReg; ++i) wx[i][g] = (act && i < S) ? p.W_ih0[(int64_t)(g * H + lane) * I + i] : 0.f;
    }
    const float outb = lane < NO ? p.out_b[lane] : 0.f;
    // lane k >= S owns tril entry q = k - S -> (row, col)
    int trow = 0, tcol = 0;
    if (lane >= S && lane < NO) {
        int q = lane - S, r = 0;
        while ((r + 1) * (r + 2) / 2 <= q) ++r;
        trow = r; tcol = q - r * (r + 1) / 2;
    }
    const bool is_diag = lane >= S && lane < NO && trow == tcol;

    float h[L], cc[L][3];
#pragma unroll
    for (int l = 0; l < L; ++l) {
        h[l] = 0.f;
#pragma unroll
        for (int g = 0; g < 3; ++g) cc[l][g] = bhh[l][g];  // c = b_hh + W_hh . 0
    }
    float xreg = lane < S ? p.x0[(int64_t)b * S + lane] : 0.f;  // lane i holds z_t[i]
    if (lane < S) p.paths[(int64_t)b * (T + 1) * S + lane] = xreg;

    const float *Gb = p.G + (int64_t)b * T * G3;
    const float *eb = p.eps + (int64_t)b * T * S;
    float gcur[3], gnext[3], ecur, enext;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        gcur[g] = act ? Gb[g * H + lane] : 0.f;
        gnext[g] = (act && T > 1) ? Gb[G3 + g * H + lane] : 0.f;
    }
    ecur = lane < S ? eb[lane] : 0.f;
    enext = (lane < S && T > 1) ? eb[S + lane] : 0.f;

    for (int t = 0; t < T; ++t) {
        float gpre[3] = {0.f, 0.f, 0.f}, epre = 0.f;  // prefetch step t+2
        if (t + 2 < T) {
#pragma unroll
            for (int g = 0; g < 3; ++g) gpre[g] = act ? Gb[(int64_t)(t + 2) * G3 + g * H + lane] : 0.f;
            epre = lane < S ? eb[(int64_t)(t + 2) * S + lane] : 0.f;
        }
        // ---- layer-0 input pre-activations: a = G[b,t] + theta part + z_t . W_x  (forward.py:195-219)
        float a[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) a[g] = gcur[g] + gth[g];
#pragma unroll
        for (int i = 0; i < kMaxSReg; ++i) {
            if (i < S) {
                float xi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xreg), i));
#pragma unroll
                for (int g = 0; g < 3; ++g) a[g] = fmaf(xi, wx[i][g], a[g]);
            }
        }
        for (int i = kMaxSReg; i < S; ++i) {
            float xi = __shfl(xreg, i, 64);
            if (act)
#pragma unroll
                for (int g = 0; g < 3; ++g) a[g] = fmaf(xi, p.W_ih0[(int64_t)(g * H + lane) * I + i], a[g]);
        }
        float o = outb;
#pragma unroll
        for (int l = 0; l < L; ++l) {
            // ---- GRU cell (forward.py:235-238)
            float r = fast_sigmoid(a[0] + cc[l][0]);
            float u = fast_sigmoid(a[1] + cc[l][1]);
            float n = fast_tanh(a[2] + r * cc[l][2]);
            float hn = (1.0f - u) * n + u * h[l];
            if (SAVE && act) {
                float *A = p.acts + ((((int64_t)b * T + t) * L + l) * 5) * H + lane;
                A[0] = hn; A[H] = r; A[2 * H] = u; A[3 * H] = n; A[4 * H] = cc[l][2];
            }
            h[l] = hn;
            hbuf[lane] = hn;
            wave_lds_fence();
            // ---- one pass over h^l_t: recurrent pre-activations of step t+1 and the inputs of layer l+1
            float cn[3] = {bhh[l][0], bhh[l][1], bhh[l][2]};
            if (l < L - 1) {
                float an[3] = {bih[l + 1][0], bih[l + 1][1], bih[l + 1][2]};
                const int qh = 2 * l, qi = 2 * l + 1;
                const float4 *Wh = (qh < NLDS) ? (const float4 *)(Wl + qh * kMatF4) : (p.packF + (int64_t)qh * kMatF4);
                const float4 *Wi = (qi < NLDS) ? (const float4 *)(Wl + qi * kMatF4) : (p.packF + (int64_t)qi * kMatF4);
#pragma unroll 4
                for (int c = 0; c < nchunk; ++c) {
                    float4 hv = *(const float4 *)&hbuf[4 * c];
                    if (qh < NLDS) fma3(Wl + qh * kMatF4, c, lane, hv, cn); else fma3(Wh, c, lane, hv, cn);
                    if (qi < NLDS) fma3(Wl + qi * kMatF4, c, lane, hv, an); else fma3(Wi, c, lane, hv, an);
                }
#pragma unroll
                for (int g = 0; g < 3; ++g) a[g] = an[g];
            } else {
                const int qh = 2 * l;
                const float4 *Wh = p.packF + (int64_t)qh * kMatF4;
                const int orow = lane < NO ? lane : NO - 1;
#pragma unroll 4
                for (int c = 0; c < nchunk; ++c) {
                    float4 hv = *(const float4 *)&hbuf[4 * c];
                    if (qh < NLDS) fma3(Wl + qh * kMatF4, c, lane, hv, cn); else fma3(Wh, c, lane, hv, cn);
                    float4 w = POl[c * NO + orow];
                    o = fmaf(hv.x, w.x, o); o = fmaf(hv.y, w.y, o); o = fmaf(hv.z, w.z, o); o = fmaf(hv.w, w.w, o);
                }
            }
#pragma unroll
            for (int g = 0; g < 3; ++g) cc[l][g] = cn[g];
            wave_lds_fence();  // hbuf is rewritten by the next layer
        }
        // ---- emission (forward.py:314-375): lane k < S holds mu_k, lane S+q holds tril entry q
        float lval = (is_diag && o < p.diag_min) ? p.diag_min : o;  // NaN propagates (torch.max semantics)
        obuf[lane] = lane < S ? o : lval;
        if (lane < S) ebuf[lane] = ecur;
        wave_lds_fence();
        if (lane < S) {
            float acc = 0.f;
            const int base = S + lane * (lane + 1) / 2;
            for (int j = 0; j <= lane; ++j) acc = fmaf(obuf[base + j], ebuf[j], acc);
            xreg = xreg + o * p.dt + acc * p.sqdt;
            p.means[((int64_t)b * T + t) * S + lane] = o;
            p.paths[((int64_t)b * (T + 1) + t + 1) * S + lane] = xreg;
        }
        if (SAVE && lane >= S && lane < NO) p.chol_raw[((int64_t)b * T + t) * p.ntril + (lane - S)] = o;
        for (int e = lane; e < S * S; e += 64) {  // full S x S matrix, strict upper triangle = 0
            int rr = e / S, cl = e - rr * S;
            float v = cl <= rr ? obuf[S + rr * (rr + 1) / 2 + cl] : 0.f;
            p.chol[((int64_t)b * T + t) * S * S + e] = v;
        }
        wave_lds_fence();
#pragma unroll
        for (int g = 0; g < 3; ++g) { gcur[g] = gnext[g]; gnext[g] = gpre[g]; }
        ecur = enext; enext = epre;
    }
}

// ---------------------------------------------------------------------------- backward
struct BwdParams {
    int B, T, S, P, C, H, NO, ntril, wpb;
    const float *g_paths, *g_means, *g_chol, *theta, *eps, *chol_raw, *acts;
    const float *W_ih0, *out_W;
    const float *W_hh0, *W_ih_st, *W_hh_st;  // native tensors for the register-resident v2 kernel
    const float4 *packB;
    float dt, sqdt, diag_min;
    float *D4;   // [B*T][L][4H]  (dr_pre, du_pre, dn_pre, dc_n)
    float *DO;   // [B*T][NO]     gradient wrt the emission rows
    float *g_x0, *g_theta;
};

// acc += sum_j W[g][j][lane] * d[g][j] for one packed (backward layout) matrix and gate g
__device__ __forceinline__ float bdot(const float4 *__restrict__ W, int g, int jc, int lane, const float4 &dv, float acc) {
    float4 w = W[(g * kChunks + jc) * kHP + lane];
    acc = fmaf(dv.x, w.x, acc); acc = fmaf(dv.y, w.y, acc); acc = fmaf(dv.z, w.z, acc); acc = fmaf(dv.w, w.w, acc);
    return acc;
}

constexpr int kBwdScratchPerWave = 4 * 64 + 3 * 64;  // dbuf[4][64], dobuf, xbuf, ebuf

template <int L>
__global__ void __launch_bounds__(512) head_bwd_kernel(BwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float4 smem4[];
    constexpr int NMAT = 2 * L - 1;
    constexpr int NLDS = NMAT < 3 ? NMAT : 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * p.wpb + wave;
    const int H = p.H, S = p.S, T = p.T, NO = p.NO;

    float4 *Wl = smem4;
    float *OWl = (float *)(Wl + NLDS * kMatF4);  // out_W native [NO][64] (zero padded columns)
    float *scratch = OWl + NO * kHP + wave * kBwdScratchPerWave;
    float *dbuf = scratch, *dobuf = scratch + 256, *xbuf = scratch + 320, *ebuf = scratch + 384;

    for (int i = threadIdx.x; i < NLDS * kMatF4; i += blockDim.x) Wl[i] = p.packB[i];
    for (int i = threadIdx.x; i < NO * kHP; i += blockDim.x) {
        int j = i & 63, k = i >> 6;
        OWl[i] = j < H ? p.out_W[(int64_t)k * H + j] : 0.f;
    }
    __syncthreads();
    if (b >= p.B) return;

    const int I = S + p.C + p.P;
    const bool act = lane < H;
    const int nchunk = (H + 3) >> 2;
    float wx[kMaxSReg][3];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int i = 0; i < kMaxSReg; ++i) wx[i][g] = (act && i < S) ? p.W_ih0[(int64_t)(g * H + lane) * I + i] : 0.f;
    int trow = 0, tcol = 0;
    if (lane >= S && lane < NO) {
        int q = lane - S, r = 0;
        while ((r + 1) * (r + 2) / 2 <= q) ++r;
        trow = r; tcol = q - r * (r + 1) / 2;
    }
    const bool is_tril = lane >= S && lane < NO;
    const bool is_diag = is_tril && trow == tcol;

    float dh[L], spi[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int l = 0; l < L; ++l) dh[l] = 0.f;
    float dx = 0.f;  // lane i < S carries d z_t[i]

    const int64_t bt0 = (int64_t)b * T;
    for (int t = T - 1; t >= 0; --t) {
        const int64_t bt = bt0 + t;
        // ---- upstream gradients for this step (backward.py:257-349)
        float gl = 0.f, raw = 0.f;
        if (lane < S) {
            dx += p.g_paths[((int64_t)b * (T + 1) + t + 1) * S + lane];
            xbuf[lane] = dx;
            ebuf[lane] = p.eps[bt * S + lane];
        }
        if (is_tril) {
            gl = p.g_chol[(bt * S + trow) * S + tcol];
            raw = p.chol_raw[bt * p.ntril + (lane - S)];
        }
        wave_lds_fence();
        float dO = 0.f;
        if (lane < S) dO = dx * p.dt + p.g_means[bt * S + lane];
        else if (is_tril) {
            float dL = xbuf[trow] * ebuf[tcol] * p.sqdt + gl;
            if (is_diag && !(raw >= p.diag_min || dL < 0.f)) dL = 0.f;  // bounds.py:20 / backward.py:331-334
            dO = dL;
        }
        dobuf[lane] = dO;
        if (lane < NO) p.DO[bt * NO + lane] = dO;
        wave_lds_fence();
        float dcur = 0.f;  // d h_top[j] = sum_k dO_k out_W[k][j]
        for (int k = 0; k < NO; ++k) dcur = fmaf(dobuf[k], OWl[k * kHP + lane], dcur);

#pragma unroll
        for (int l = L - 1; l >= 0; --l) {
            const float *A = p.acts + ((bt * L + l) * 5) * H + lane;
            float r = 0.f, u = 0.f, n = 0.f, cn = 0.f, hprev = 0.f;
            if (act) {
                r = A[H]; u = A[2 * H]; n = A[3 * H]; cn = A[4 * H];
                if (t > 0) hprev = A[-(int64_t)L * 5 * H];
            }
            // ---- GRU cell adjoint (backward.py:59-67, 452-486)
            float d = dcur + dh[l];
            float dn = (1.0f - u) * d, du = (hprev - n) * d;
            float dn_pre = dn * (1.0f - n * n);
            float du_pre = du * (u * (1.0f - u));
            float dcn = dn_pre * r;
            float dr_pre = (dn_pre * cn) * (r * (1.0f - r));
            float carry = u * d;
            if (act) {
                float *D = p.D4 + (bt * L + l) * 4 * H + lane;
                D[0] = dr_pre; D[H] = du_pre; D[2 * H] = dn_pre; D[3 * H] = dcn;
            }
            dbuf[lane] = dr_pre; dbuf[64 + lane] = du_pre; dbuf[128 + lane] = dn_pre; dbuf[192 + lane] = dcn;
            wave_lds_fence();
            const int qh = 2 * l, qi = 2 * l - 1;
            const float4 *Wh = (qh < NLDS) ? (const float4 *)(Wl + qh * kMatF4) : (p.packB + (int64_t)qh * kMatF4);
            float acc_h = carry, acc_i = 0.f;
            if (l > 0) {
                const float4 *Wi = (qi < NLDS) ? (const float4 *)(Wl + qi * kMatF4) : (p.packB + (int64_t)qi * kMatF4);
#pragma unroll 4
                for (int jc = 0; jc < nchunk; ++jc) {
                    float4 vr = *(const float4 *)&dbuf[4 * jc], vu = *(const float4 *)&dbuf[64 + 4 * jc];
                    float4 vn = *(const float4 *)&dbuf[128 + 4 * jc], vc = *(const float4 *)&dbuf[192 + 4 * jc];
                    if (qh < NLDS) {
                        acc_h = bdot(Wl + qh * kMatF4, 0, jc, lane, vr, acc_h); acc_h = bdot(Wl + qh * kMatF4, 1, jc, lane, vu, acc_h);
                        acc_h = bdot(Wl + qh * kMatF4, 2, jc, lane, vc, acc_h);
                    } else {
                        acc_h = bdot(Wh, 0, jc, lane, vr, acc_h); acc_h = bdot(Wh, 1, jc, lane, vu, acc_h); acc_h = bdot(Wh, 2, jc, lane, vc, acc_h);
                    }
                    if (qi < NLDS) {
                        acc_i = bdot(Wl + qi * kMatF4, 0, jc, lane, vr, acc_i); acc_i = bdot(Wl + qi * kMatF4, 1, jc, lane, vu, acc_i);
                        acc_i = bdot(Wl + qi * kMatF4, 2, jc, lane, vn, acc_i);
                    } else {
                        acc_i = bdot(Wi, 0, jc, lane, vr, acc_i); acc_i = bdot(Wi, 1, jc, lane, vu, acc_i); acc_i = bdot(Wi, 2, jc, lane, vn, acc_i);
                    }
                }
                dcur = acc_i;
            } else {
#pragma unroll 4
                for (int jc = 0; jc < nchunk; ++jc) {
                    float4 vr = *(const float4 *)&dbuf[4 * jc], vu = *(const float4 *)&dbuf[64 + 4 * jc];
                    float4 vc = *(const float4 *)&dbuf[192 + 4 * jc];
                    acc_h = bdot(Wl, 0, jc, lane, vr, acc_h); acc_h = bdot(Wl, 1, jc, lane, vu, acc_h); acc_h = bdot(Wl, 2, jc, lane, vc, acc_h);
                }
                // d z_t += W_ih_l0[:, state rows]^T . d_pre  (backward.py:494-509): wave reductions
                spi[0] += dr_pre; spi[1] += du_pre; spi[2] += dn_pre;
#pragma unroll
                for (int i = 0; i < kMaxSReg; ++i) {
                    if (i < S) {
                        float v = wx[i][0] * dr_pre + wx[i][1] * du_pre + wx[i][2] * dn_pre;
                        v = wave_sum(v);
                        if (lane == i) dx += v;
                    }
                }
                for (int i = kMaxSReg; i < S; ++i) {
                    float v = 0.f;
                    if (act) v = p.W_ih0[(int64_t)lane * I + i] * dr_pre + p.W_ih0[(int64_t)(H + lane) * I + i] * du_pre +
                                 p.W_ih0[(int64_t)(2 * H + lane) * I + i] * dn_pre;
                    v = wave_sum(v);
                    if (lane == i) dx += v;
                }
            }
            dh[l] = acc_h;
            wave_lds_fence();
        }
    }
    if (lane < S) p.g_x0[(int64_t)b * S + lane] = dx + p.g_paths[(int64_t)b * (T + 1) * S + lane];  // backward.py:620-624
    // d theta_b = W_ih_l0[:, theta rows]^T . sum_t d_pre_0   (backward.py:511-548)
    for (int q = 0; q < p.P; ++q) {
        float v = 0.f;
        if (act) v = p.W_ih0[(int64_t)lane * I + S + p.C + q] * spi[0] + p.W_ih0[(int64_t)(H + lane) * I + S + p.C + q] * spi[1] +
                     p.W_ih0[(int64_t)(2 * H + lane) * I + S + p.C + q] * spi[2];
        v = wave_sum(v);
        if (lane == 0) p.g_theta[(int64_t)b * p.P + q] = v;
    }
}

// =========================================================================================
// v2 kernels (L <= 2): four wavefronts per sample path, recurrent weights resident in VGPRs.
//
// Thread (wave w, lane) owns hidden unit j = 16 w + (lane >> 2) together with the three other
// lanes of its quad; lane kq = lane & 3 of the quad multiplies the k-slice [16 kq, 16 kq + 16) of
// every dot product and the quad sums with two DPP quad_perm adds.  Each lane therefore keeps
// 3 x 16 floats per recurrent matrix (W_hh_l0, W_ih_l1, W_hh_l1) + 16 floats of the emission
// rows = 160 VGPRs of weights and the time loop issues NO weight loads at all; the hidden state
// crosses waves through a 256-byte LDS buffer and one workgroup barrier per GRU layer.
// 512 paths -> 2048 waves = 2 per SIMD on all 256 CUs (the v1 kernel put one wave on half of
// the SIMDs and was bound by the latency of 224 dependent ds_read_b128 per step).
// =========================================================================================
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ __forceinline__ float dpp_quad(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float quad_sum(float v) {
    v += dpp_quad<0xB1>(v);  // quad_perm [1,0,3,2]
    v += dpp_quad<0x4E>(v);  // quad_perm [2,3,0,1]
    return v;
}
__device__ __forceinline__ float quad_bcast(float v, int q) {
    float a = dpp_quad<0x00>(v), b = dpp_quad<0x55>(v), c = dpp_quad<0xAA>(v), d = dpp_quad<0xFF>(v);
    return q == 0 ? a : (q == 1 ? b : (q == 2 ? c : d));
}

// LDS carve of the v2 forward kernel (floats); CH = time steps staged per chunk.
constexpr int kWoStride = 68;
struct FwdV2Lds {
    int hb, obuf, wx, wo, gth, gbuf, ebuf, s_paths, s_means, s_chol, s_raw, s_acts, dummy, total;
};
__host__ __device__ inline FwdV2Lds fwd_v2_lds(int H, int S, int L, int CH, bool save) {
    const int ntril = S * (S + 1) / 2;
    FwdV2Lds o; int off = 0;
    auto take = [&](int n) { int r = off; off += (n + 3) & ~3; return r; };
    o.hb = take(L * 64); o.obuf = take(64); o.wx = take(S * 3 * 64);
    o.wo = take((S + ntril + 1) * kWoStride);  // emission rows (+ one all-zero row), row stride 68 floats: conflict-free ds_read_b128
    o.gth = take(3 * 64);                      // hoisted theta projection [gate][unit]
    o.gbuf = take(2 * CH * 3 * H); o.ebuf = take(2 * CH * S);
    o.s_paths = take(CH * S); o.s_means = take(CH * S); o.s_chol = take(CH * S * S);
    o.s_raw = take(save ? CH * ntril : 0);
    o.s_acts = take(save ? CH * L * 5 * 64 : 0);   // staged activation records, unit stride padded to 64: compile-time offsets
    o.dummy = take(save ? L * 5 * 64 : 0);         // where the lanes of units >= hidden_dim park their (unused) stores
    o.total = off;
    return o;
}

// LDS byte address of a pointer into the dynamic shared segment (generic -> local = the low 32 bits)
__device__ __forceinline__ uint32_t lds_addr(const void *p) { return (uint32_t)(uintptr_t)p; }
// ds_write_b32 with an immediate byte offset, issued in program order among its kind: hipcc neither reorders, merges
// (ds_write2) nor counts these, which is what the counted s_waitcnt in front of the per-layer barrier relies on
template <int OFF> __device__ __forceinline__ void lds_store(uint32_t addr, float v) {
    asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(addr), "v"(v), "i"(OFF) : "memory");
}

// All global traffic of the v2 kernels happens at chunk boundaries as wide coalesced bursts
// (the record layouts [b][t][...] are contiguous over a run of time steps); the per-step loop
// only touches LDS, so no s_waitcnt vmcnt ever sits on the recurrence's critical path.
// MODE: 1, 2 = state_dim known at compile time (z_t kept as wave-uniform registers, no LDS on the
// emission -> Euler update -> next-step-input path); 3 = generic state_dim with NO <= 16; 4 = generic.
//
// Round 3 (what the cycle stamps of one wave asked for -- a step is a serial chain, ~40 % FMA issue, the rest latency):
//  * gate pre-activations live in the exp2 domain: W_r, W_z, their biases, the state / theta / context terms are scaled by
//    -log2(e) and the n-gate terms by 2 log2(e) when they are loaded, so sigmoid / tanh are exp2 -> +1 -> rcp with no multiply;
//  * biases ride in the accumulators (lane kq == 0 of a quad starts from the bias, the others from 0): no adds after the quad sums;
//    the hoisted theta projection is added (and the context record scaled) once per chunk when the record goes to LDS;
//  * one accumulator per dot product (a wave cannot issue faster than one VALU per ~4.6 cycles, dependent or not);
//  * the exchanged h^l is stored FIRST, the five saved-activation stores behind it, and the barrier waits with a COUNTED
//    s_waitcnt lgkmcnt(5): the recurrence never waits for the staging stores (the stores are inline asm so that their
//    number and order are exactly what the count assumes);
//  * eps of the step is fetched a step ahead;
//  * workgroups start with first chunks of different lengths: their flush bursts (41 KB of activations per chunk) are spread
//    over time instead of all 512 workgroups hitting HBM in the same microsecond while the memory system idles in between.
template <int L, bool SAVE, int CH, int MODE>
__global__ void __launch_bounds__(256, 2) head_fwd_v2_kernel(FwdParams p) {
    constexpr bool SMALL = MODE <= 3;
    constexpr int SS = MODE <= 2 ? MODE : 0;   // compile-time state_dim (0 = runtime)
    constexpr int SSN = SS > 0 ? SS : 1;
    constexpr float kSr = -1.4426950408889634f, kSn = 2.8853900817779268f, kInvSn = 1.0f / kSn;
    static_assert(L >= 1 && L <= 2, "v2 keeps at most three 64x192 matrices in registers");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, u = lane >> 2, kq = lane & 3;
    const int j = 16 * wave + u, k0 = 16 * kq;
    const int b = blockIdx.x;
    const int H = p.H, S = SS > 0 ? SS : p.S, T = p.T, NO = p.NO;
    const int I = S + p.C + p.P, G3 = 3 * H;
    // SMALL (NO <= 16): every wave computes all emission rows (row = quad index u) redundantly, so the
    // Euler update needs no third workgroup barrier; otherwise row r = j is owned by one quad of the block.
    const int orow = SMALL ? u : j;
    const bool unit_ok = j < H, row_ok = orow < NO;
    const FwdV2Lds lay = fwd_v2_lds(H, S, L, CH, SAVE);
    float *hb = smem + lay.hb, *obuf = smem + lay.obuf, *wxl = smem + lay.wx, *wol = smem + lay.wo, *gthl = smem + lay.gth;
    float *gbuf = smem + lay.gbuf, *ebuf = smem + lay.ebuf;
    float *s_paths = smem + lay.s_paths, *s_means = smem + lay.s_means, *s_chol = smem + lay.s_chol;
    float *s_raw = smem + lay.s_raw, *s_acts = smem + lay.s_acts;
    for (int e = tid; e < lay.total; e += 256) smem[e] = 0.f;  // finite data everywhere (see head_bwd_v2_kernel)
    __syncthreads();

    // ---- register-resident weights (gate rows pre-scaled into the exp2 domain) -------------
    const float gsc[3] = {kSr, kSr, kSn};
    float wh[L][3][16], wi[L][3][16];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const bool ok = unit_ok && (k0 + i) < H;
            wh[0][g][i] = ok ? gsc[g] * p.W_hh0[(int64_t)(g * H + j) * H + k0 + i] : 0.f;
            wi[0][g][i] = 0.f;
            if (L > 1) {
                wh[L - 1][g][i] = ok ? gsc[g] * p.W_hh_st[(int64_t)(g * H + j) * H + k0 + i] : 0.f;
                wi[L - 1][g][i] = ok ? gsc[g] * p.W_ih_st[(int64_t)(g * H + j) * H + k0 + i] : 0.f;
            }
        }
    for (int e = tid; e < NO * 64; e += 256) {  // emission rows -> LDS (kept out of the VGPR budget); row NO stays zero
        int kk = e & 63, r = e >> 6;
        wol[r * kWoStride + kk] = kk < H ? p.out_W[(int64_t)r * H + kk] : 0.f;
    }
    const float *wop = wol + (row_ok ? orow : NO) * kWoStride + k0;
    // wide variant (NO > 16, S <= 9): element e = lane, 64 + lane of the dense Cholesky block reads obuf[cidx] (-1: strict upper
    // triangle = 0, -2: beyond S * S) -- the division by the run-time S happens here, not in wave 1's every time step
    int cidx[2] = {-2, -2};
    if (!SMALL) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = 64 * q + lane;
            if (e < S * S) { const int rr = e / S, cl = e - rr * S; cidx[q] = cl <= rr ? S + rr * (rr + 1) / 2 + cl : -1; }
        }
    }
    // biases enter through the accumulators: only lane kq == 0 of a quad carries them into the quad sum
    float bhh[L][3], bih[L][3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const bool lead = unit_ok && kq == 0;
        bhh[0][g] = lead ? gsc[g] * p.b_hh0[g * H + j] : 0.f;
        bih[0][g] = 0.f;
        if (L > 1) {
            bhh[L - 1][g] = lead ? gsc[g] * p.b_hh_st[g * H + j] : 0.f;
            bih[L - 1][g] = lead ? gsc[g] * p.b_ih_st[g * H + j] : 0.f;
        }
        float acc = 0.f;  // hoisted theta projection (forward.py:157-175) -> LDS, added to the context record at staging time
        if (unit_ok)
            for (int q = 0; q < p.P; ++q) acc = fmaf(p.theta[(int64_t)b * p.P + q], p.W_ih0[(int64_t)(g * H + j) * I + S + p.C + q], acc);
        if (lead) gthl[g * H + j] = gsc[g] * acc;
    }
    for (int e = tid; e < S * 3 * 64; e += 256) {  // state rows of W_ih_l0: wxl[i][g][unit]
        int un = e & 63, g = (e >> 6) % 3, i = e / 192;
        // wide variant: the four lanes of a quad read four different i (= kq mod 4) of the same unit: rotating the unit index by
        // 16 (i & 3) puts them on different banks
        const int slot = SMALL ? un : ((un + 16 * (i & 3)) & 63);
        wxl[(e - un) + slot] = un < H ? gsc[g] * p.W_ih0[(int64_t)(g * H + un) * I + i] : 0.f;
    }
    float wxr[SSN][3], xu[SSN];  // SS > 0: state rows of W_ih_l0 and z_t as wave-uniform values
#pragma unroll
    for (int i = 0; i < SSN; ++i) {
        xu[i] = (SS > 0) ? p.x0[(int64_t)b * S + i] : 0.f;
#pragma unroll
        for (int g = 0; g < 3; ++g) wxr[i][g] = (SS > 0 && unit_ok) ? gsc[g] * p.W_ih0[(int64_t)(g * H + j) * I + i] : 0.f;
    }
    const float outb = (row_ok && kq == 0) ? p.out_b[orow] : 0.f;
    int trow = 0, tcol = 0;
    if (orow >= S && orow < NO) {
        int q = orow - S, r = 0;
        while ((r + 1) * (r + 2) / 2 <= q) ++r;
        trow = r; tcol = q - r * (r + 1) / 2;
    }
    const bool is_tril = orow >= S && orow < NO;
    const bool is_diag = is_tril && trow == tcol;

    float h[L], cc[L][3];
#pragma unroll
    for (int l = 0; l < L; ++l) {
        h[l] = 0.f;
#pragma unroll
        for (int g = 0; g < 3; ++g) cc[l][g] = quad_sum(bhh[l][g]);   // W_hh h_{-1} + b_hh with h_{-1} = 0
    }
    // every wave keeps z_t[i] on its lane i -- wide variant: on the four lanes of its quad i (u = i)
    const int xi_own = SMALL ? lane : u;
    const bool x_own = SMALL ? lane < S : (u < S);
    float xreg = x_own ? p.x0[(int64_t)b * S + xi_own] : 0.f;
    if (wave == 0 && x_own && (SMALL || kq == 0)) p.paths[(int64_t)b * (T + 1) * S + xi_own] = xreg;
    __syncthreads();   // gthl complete

    // ---- chunked staging of the projected context record G[b, t, 3H] and eps ---------------
    const float *Gb = p.G + (int64_t)b * T * G3;
    const float *eb = p.eps + (int64_t)b * T * S;
    constexpr int PF = (CH * 3 * 64 + 255) / 256;  // floats of G per thread and chunk (H <= 64)
    float pf[PF], pe = 0.f;
    auto issue_loads = [&](int t0, int n) {
        const int nG = n * G3, nE = n * S;
#pragma unroll
        for (int r = 0; r < PF; ++r) { int e = tid + 256 * r; pf[r] = (e < nG) ? Gb[(int64_t)t0 * G3 + e] : 0.f; }
        pe = (tid < nE) ? eb[(int64_t)t0 * S + tid] : 0.f;
    };
    auto commit_loads = [&](int buf) {   // record -> exp2 domain, + theta term
#pragma unroll
        for (int r = 0; r < PF; ++r) {
            int e = tid + 256 * r;
            if (e < CH * G3) {
                int c = tid + 64 * (r % 3);            // e mod 192 when hidden_dim is 64
                if (G3 == 192) { if (c >= 192) c -= 192; } else c = e % G3;
                gbuf[buf * CH * G3 + e] = fmaf(pf[r], c < 2 * H ? kSr : kSn, gthl[c]);
            }
        }
        if (tid < CH * S) ebuf[buf * CH * S + tid] = pe;
    };
    for (int e = tid; e < CH * S * S; e += 256) s_chol[e] = 0.f;
    // first chunk of 1..CH steps: the workgroups' chunk boundaries (bursts of staged outputs) are spread over time; the two
    // workgroups that share a CU (b and b + 256 in dispatch order) sit half a chunk apart
    int t0 = 0, nsteps = min(CH - (b * 5 + (b >> 8) * (CH / 2)) % CH, T), cur = 0;
    issue_loads(0, nsteps);
    commit_loads(0);
    __syncthreads();

    const uint32_t hb_a = lds_addr(hb + j);
    const uint32_t act_a0 = lds_addr(SAVE ? (unit_ok ? s_acts + j : smem + lay.dummy) : smem);
    const uint32_t act_step = unit_ok ? (uint32_t)(L * 5 * 64 * sizeof(float)) : 0u;
    for (int c = 0; t0 < T; ++c) {   // c: chunk counter (trace builds)
        const float *gch = gbuf + cur * CH * G3, *ech = ebuf + cur * CH * S;
        float gcur[3], ev[SSN];
        gcur[0] = unit_ok ? gch[j] : 0.f; gcur[1] = unit_ok ? gch[H + j] : 0.f; gcur[2] = unit_ok ? gch[2 * H + j] : 0.f;
#pragma unroll
        for (int i = 0; i < SSN; ++i) ev[i] = SS > 0 ? ech[i] : 0.f;
        uint32_t act_a = act_a0;
        for (int tt = 0; tt < nsteps; ++tt) {
            VSDE_TP(0);
            float a[3], evc[SSN];
#pragma unroll
            for (int g = 0; g < 3; ++g) a[g] = gcur[g];
#pragma unroll
            for (int i = 0; i < SSN; ++i) evc[i] = ev[i];
            {   // prefetch the next step's projected record and eps from LDS (off the critical path)
                const int tn = min(tt + 1, nsteps - 1);
                const float *gr = gch + tn * G3 + j;
                gcur[0] = unit_ok ? gr[0] : 0.f; gcur[1] = unit_ok ? gr[H] : 0.f; gcur[2] = unit_ok ? gr[2 * H] : 0.f;
#pragma unroll
                for (int i = 0; i < SSN; ++i) ev[i] = SS > 0 ? ech[tn * SS + i] : 0.f;
            }
            if (SS > 0) {
#pragma unroll
                for (int i = 0; i < SSN; ++i)
#pragma unroll
                    for (int g = 0; g < 3; ++g) a[g] = fmaf(xu[i], wxr[i][g], a[g]);
            } else {
                if (SMALL) {
                    for (int i = 0; i < S; ++i) {
                        const float xi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xreg), i));
#pragma unroll
                        for (int g = 0; g < 3; ++g) a[g] = fmaf(xi, wxl[(i * 3 + g) * 64 + j], a[g]);
                    }
                } else {
                    // S <= 12 here (NO <= 64 gives S <= 9).  Lane kq of a quad multiplies the components i = kq, kq + 4, kq + 8 (z_t[i]
                    // from quad i by one cross-lane read each, all LDS reads in flight together), the quad adds up: a loop over
                    // a run-time S waited for an LDS round trip per component and ran in all four lanes of the quad.
                    float xs[3], wv[3][3], part[3] = {0.f, 0.f, 0.f};
#pragma unroll
                    for (int ii = 0; ii < 3; ++ii) {
                        const int i = kq + 4 * ii, ic = i < S ? i : 0;
                        xs[ii] = __shfl(xreg, 4 * ic, 64);
#pragma unroll
                        for (int g = 0; g < 3; ++g) wv[ii][g] = wxl[(ic * 3 + g) * 64 + ((j + 16 * kq) & 63)];   // ic & 3 == kq (or 0: unused)
                    }
#pragma unroll
                    for (int ii = 0; ii < 3; ++ii)
#pragma unroll
                        for (int g = 0; g < 3; ++g) part[g] = (kq + 4 * ii) < S ? fmaf(xs[ii], wv[ii][g], part[g]) : part[g];
#pragma unroll
                    for (int g = 0; g < 3; ++g) a[g] += quad_sum(part[g]);
                }
            }
            VSDE_TP(1);
            float o = 0.f;
#pragma unroll
            for (int l = 0; l < L; ++l) {
                // exp2-domain gates: r = 1 / (1 + 2^(a_r + c_r)), n = 1 - 2 / (1 + 2^(a_n + r c_n))
                const float r = fast_rcp(1.0f + fast_exp2(a[0] + cc[l][0]));
                const float uu = fast_rcp(1.0f + fast_exp2(a[1] + cc[l][1]));
                const float n = fmaf(-2.0f, fast_rcp(1.0f + fast_exp2(fmaf(r, cc[l][2], a[2]))), 1.0f);
                const float hn = fmaf(uu, h[l] - n, n);   // (1 - u) n + u h
                h[l] = hn;
                VSDE_TP(2 + 4 * l);
                // h^l first, the saved activations (h, r, z, n, n_hh of unit j: lane kq == 0) behind it; the barrier waits for
                // everything but the five staging stores
                asm volatile("" ::: "memory");
                if (kq == 0) {
                    if (l == 0) lds_store<0>(hb_a, hn); else lds_store<256>(hb_a, hn);
                    if (SAVE) {
                        const float nhh = cc[l][2] * kInvSn;
                        if (l == 0) {
                            lds_store<0>(act_a, hn); lds_store<256>(act_a, r); lds_store<512>(act_a, uu); lds_store<768>(act_a, n);
                            lds_store<1024>(act_a, nhh);
                        } else {
                            lds_store<1280>(act_a, hn); lds_store<1536>(act_a, r); lds_store<1792>(act_a, uu); lds_store<2048>(act_a, n);
                            lds_store<2304>(act_a, nhh);
                        }
                    }
                }
                if (SAVE) asm volatile("s_waitcnt lgkmcnt(5)\n\ts_barrier" ::: "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                VSDE_TP(3 + 4 * l);
                // Plain v_fma_f32 chains, one accumulator per dot product, started from the (lane-masked) bias.  The 16 staged h
                // values (and the emission-row weights) are consumed 8 at a time: with three matrices in registers there is no
                // room for more.
                float eh[3] = {bhh[l][0], bhh[l][1], bhh[l][2]};             // W_hh^l h^l + b_hh^l
                float ei[3];                                                   // W_ih^{l+1} h^l + b_ih^{l+1}, or [2] = emission row
                if (l < L - 1) { constexpr int ln = (L > 1) ? 1 : 0; ei[0] = bih[ln][0]; ei[1] = bih[ln][1]; ei[2] = bih[ln][2]; }
                else { ei[0] = 0.f; ei[1] = 0.f; ei[2] = outb; }
                VSDE_TP(4 + 4 * l);
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const float4 ha = *(const float4 *)&hb[l * 64 + k0 + 8 * hf], hc = *(const float4 *)&hb[l * 64 + k0 + 8 * hf + 4];
                    const float hs[8] = {ha.x, ha.y, ha.z, ha.w, hc.x, hc.y, hc.z, hc.w};
                    if (l < L - 1) {
                        constexpr int ln = (L > 1) ? 1 : 0;
#pragma unroll
                        for (int i8 = 0; i8 < 8; ++i8) {
                            const int i = 8 * hf + i8;
#pragma unroll
                            for (int g = 0; g < 3; ++g) { ei[g] = fmaf(hs[i8], wi[ln][g][i], ei[g]); eh[g] = fmaf(hs[i8], wh[l][g][i], eh[g]); }
                        }
                    } else {
                        const float4 wa = *(const float4 *)(wop + 8 * hf), wc = *(const float4 *)(wop + 8 * hf + 4);
                        const float wo[8] = {wa.x, wa.y, wa.z, wa.w, wc.x, wc.y, wc.z, wc.w};
#pragma unroll
                        for (int i8 = 0; i8 < 8; ++i8) {
                            const int i = 8 * hf + i8;
                            ei[2] = fmaf(hs[i8], wo[i8], ei[2]);
#pragma unroll
                            for (int g = 0; g < 3; ++g) eh[g] = fmaf(hs[i8], wh[l][g][i], eh[g]);
                        }
                    }
                    if (hf == 0) __builtin_amdgcn_sched_barrier(0);  // second half's LDS reads stay behind the first half's FMAs
                        }
                if (l < L - 1) {
#pragma unroll
                    for (int g = 0; g < 3; ++g) a[g] = quad_sum(ei[g]);
                } else {
                    o = quad_sum(ei[2]);
                }
#pragma unroll
                for (int g = 0; g < 3; ++g) cc[l][g] = quad_sum(eh[g]);
                VSDE_TP(5 + 4 * l);
            }
            act_a += act_step;
            // ---- emission (forward.py:314-375)
            VSDE_TP(10);
            if (SMALL) {
                // o of row r sits on lanes 4r..4r+3 of EVERY wave: no LDS exchange, no barrier
                const float oc = (is_diag && o < p.diag_min) ? p.diag_min : o;  // NaN propagates (torch.max semantics)
                if (wave == 0 && kq == 0 && is_tril) {
                    s_chol[tt * S * S + trow * S + tcol] = oc;  // strict upper triangle stays 0 (zeroed once)
                    if (SAVE) s_raw[tt * p.ntril + (orow - S)] = o;
                }
                if (SS > 0) {
#pragma unroll
                    for (int i = 0; i < SSN; ++i) {
                        const float mu = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(oc), 4 * i));
                        float acc = 0.f;
#pragma unroll
                        for (int q = 0; q <= i; ++q)
                            acc = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(oc), 4 * (SS + i * (i + 1) / 2 + q))), evc[q], acc);
                        xu[i] = xu[i] + mu * p.dt + acc * p.sqdt;
                        if (tid == 0) { s_means[tt * SS + i] = mu; s_paths[tt * SS + i] = xu[i]; }
                    }
                } else
                for (int i = 0; i < S; ++i) {
                    const float mu = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(oc), 4 * i));
                    float acc = 0.f;
                    const int base = S + i * (i + 1) / 2;
                    for (int q = 0; q <= i; ++q)
                        acc = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(oc), 4 * (base + q))), ech[tt * S + q], acc);
                    if (lane == i) {
                        xreg = xreg + mu * p.dt + acc * p.sqdt;
                        if (wave == 0) { s_means[tt * S + i] = mu; s_paths[tt * S + i] = xreg; }
                    }
                }
                VSDE_TP(11);
            } else {
                if (kq == 0 && row_ok) {
                    obuf[j] = (is_diag && o < p.diag_min) ? p.diag_min : o;
                    if (SAVE && j >= S) s_raw[tt * p.ntril + (j - S)] = o;
                }
                __syncthreads();
                VSDE_TP(11);
                if (u < S) {
                    // row u of L_t eps_t by quad u: lane kq takes the columns kq, kq + 4, kq + 8 (<= u), the quad adds up; z_t[u] lives
                    // on the quad's four lanes.  (A loop `q <= lane` on lane = row waited for an LDS round trip per column.)
                    const int base = S + u * (u + 1) / 2;
                    float lv[3], evv[3], acc = 0.f;
#pragma unroll
                    for (int ii = 0; ii < 3; ++ii) {
                        const int q = kq + 4 * ii, qc = q <= u ? q : 0;
                        lv[ii] = obuf[base + qc]; evv[ii] = ech[tt * S + qc];
                    }
#pragma unroll
                    for (int ii = 0; ii < 3; ++ii) acc = (kq + 4 * ii) <= u ? fmaf(lv[ii], evv[ii], acc) : acc;
                    acc = quad_sum(acc);
                    const float mu = obuf[u];
                    xreg = xreg + mu * p.dt + acc * p.sqdt;
                    if (wave == 0 && kq == 0) { s_means[tt * S + u] = mu; s_paths[tt * S + u] = xreg; }
                }
                if (wave == 1) {   // L_t as a dense [S][S] block; the source index of each element is fixed (cidx, set up once)
                    if (cidx[0] >= -1) s_chol[tt * S * S + lane] = cidx[0] >= 0 ? obuf[cidx[0]] : 0.f;
                    if (cidx[1] >= -1) s_chol[tt * S * S + 64 + lane] = cidx[1] >= 0 ? obuf[cidx[1]] : 0.f;
                }
            }
            VSDE_TP(12);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the last steps' staging stores
        __syncthreads();                 // chunk complete: staging buffers final
        const int tnext = t0 + nsteps, nnext = min(CH, T - tnext);
        issue_loads(tnext, nnext > 0 ? nnext : 0);   // next chunk's records; their latency hides behind the flush below
        // ---- flush the staged outputs (contiguous runs in the [b][t][...] layouts)
        {
            const int64_t bt = (int64_t)b * T + t0;
#pragma unroll 1
            for (int e = tid; e < nsteps * S; e += 256) {
                p.means[bt * S + e] = s_means[e];
                p.paths[(bt + b + 1) * S + e] = s_paths[e];
            }
#pragma unroll 1
            for (int e = tid; e < nsteps * S * S; e += 256) p.chol[bt * S * S + e] = s_chol[e];
            if (SAVE) {
#pragma unroll 1
                for (int e = tid; e < nsteps * p.ntril; e += 256) p.chol_raw[bt * p.ntril + e] = s_raw[e];
                float *dst = p.acts + bt * L * 5 * H;
                if (H == 64) {
                    const int nA = nsteps * L * 5 * 64;
#pragma unroll 1
                    for (int e = tid; e < nA / 4; e += 256) ((float4 *)dst)[e] = ((const float4 *)s_acts)[e];
                } else {   // staged with a unit stride of 64
                    const int nA = nsteps * L * 5 * H;
#pragma unroll 1
                    for (int e = tid; e < nA; e += 256) dst[e] = s_acts[(e / H) * 64 + e % H];
                }
            }
        }
        commit_loads(cur ^ 1);
        __syncthreads();                 // staging reusable, gbuf[next] visible
        t0 = tnext; nsteps = nnext; cur ^= 1;
    }
}

// ---------------------------------------------------------------------------- backward v2
// Same decomposition as head_fwd_v2_kernel for the reverse-time sweep (L <= 2, NO <= 16):
// quad (wave w, u) owns unit i = 16 w + u as an OUTPUT of the transposed products
//   d h^l_{t-1}[i] += sum_{g,j} W_hh^l[gH+j][i] ph_g[j],   d in^l[i] = sum_{g,j} W_ih^l[gH+j][i] pi_g[j]
// and lane kq multiplies the j-slice [16 kq, 16 kq + 16) with weights held in VGPRs.  The gate
// gradients cross waves through the staged D4 record itself (it is both the exchange buffer and
// the kernel's output), one workgroup barrier per layer.  All global traffic (saved activations,
// upstream gradients in; D4/DO out) is staged per chunk of CH steps as contiguous bursts.
struct BwdV2Lds {
    int acts, d4, dO, gp, gm, gl, raw, eps, owl, dxp, wxl, dOT, total;
};
__host__ __device__ inline BwdV2Lds bwd_v2_lds(int H, int S, int L, int CH, bool two_act_buffers = false, bool wide = false) {
    const int ntril = S * (S + 1) / 2, NO = S + ntril;
    BwdV2Lds o; int off = 0;
    auto take = [&](int n) { int r = off; off += (n + 3) & ~3; return r; };
    o.acts = take((two_act_buffers ? 2 : 1) * (CH + 1) * L * 5 * H); o.d4 = take(CH * L * 4 * H); o.dO = take(CH * NO);
    o.gp = take(CH * S); o.gm = take(CH * S); o.gl = take(CH * S * S); o.raw = take(CH * ntril); o.eps = take(CH * S);
    // wide: the emission weights transposed per lane, [unit][kq][16 rows r = kq + 4 q] with a pitch of 20 floats (conflict-free
    // ds_read_b128), and this step's dO in the same order [kq][16]
    o.owl = take(wide ? 256 * 20 : NO * 64); o.dOT = take(wide ? 64 : 0);
    o.dxp = take(wide ? 16 * 16 : 4 * 16);   // wide: [state component][wave x 16-lane row] partial sums
    o.wxl = take(wide ? S * 3 * 64 : 0);  // wide variant: state rows of W_ih_l0 live in LDS
    o.total = off;
    return o;
}

// SS > 0: compile-time state dimension (loops over S / NO unroll).  DMA: the saved-activation records of the next
// (earlier) chunk are copied global -> LDS by global_load_lds_dwordx4 into a second buffer while the current chunk's time
// steps run (no registers, no wait until the chunk boundary); needs 16-byte aligned records (H % 4 == 0).
// WIDE: up to 64 emission rows / 16 state dims (e.g. S = 8: NO = 44).  The rows are spread over the four waves
// (row = 16*wave + quad) instead of replicated in every wave, so the emission gradients cross waves through the staged dO
// record (one more barrier per step) and the state rows of W_ih_l0 come from LDS instead of registers.
template <int L, int CH, int SS, bool DMA, bool WIDE = false>
__global__ void __launch_bounds__(256, 2) head_bwd_v2_kernel(BwdParams p) {
    static_assert(L >= 1 && L <= 2, "v2 keeps at most three 64x192 matrices in registers");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, u = lane >> 2, kq = lane & 3;
    const int i_unit = 16 * wave + u, k0 = 16 * kq;
    const int b = blockIdx.x;
    const int H = p.H, T = p.T;
    const int S = SS > 0 ? SS : p.S, ntril = SS > 0 ? SS * (SS + 1) / 2 : p.ntril, NO = SS > 0 ? SS + SS * (SS + 1) / 2 : p.NO;
    const int I = S + p.C + p.P;
    const bool unit_ok = i_unit < H;
    const int orow = WIDE ? 16 * wave + u : u;  // !WIDE: every wave owns all emission rows (NO <= 16)
    const bool row_ok = orow < NO;
    const BwdV2Lds lay = bwd_v2_lds(H, S, L, CH, DMA, WIDE);
    float *s_acts = smem + lay.acts, *s_d4 = smem + lay.d4, *s_dO = smem + lay.dO;
    float *const acts_buf0 = smem + lay.acts;
    float *s_gp = smem + lay.gp, *s_gm = smem + lay.gm, *s_gl = smem + lay.gl, *s_raw = smem + lay.raw, *s_eps = smem + lay.eps;
    float *owl = smem + lay.owl, *dxp = smem + lay.dxp, *wxl = smem + lay.wxl, *dOT = smem + lay.dOT;
    const int REC = L * 5 * H, DREC = L * 4 * H;
    // For H < 64 a lane's 16-wide j-slice can reach past the H valid entries of a staged record; its
    // weights are zero there, so the data only has to be finite: start from an all-zero LDS image.
    for (int e = tid; e < lay.total; e += 256) smem[e] = 0.f;
    __syncthreads();

    // ---- register-resident (transposed-use) weights: w[g][jj] = W[(g H + 16 kq + jj)][i]
    float whh[L][3][16], wih[3][16];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            const bool ok = unit_ok && (k0 + jj) < H;
            whh[0][g][jj] = ok ? p.W_hh0[(int64_t)(g * H + k0 + jj) * H + i_unit] : 0.f;
            if (L > 1) {
                whh[L - 1][g][jj] = ok ? p.W_hh_st[(int64_t)(g * H + k0 + jj) * H + i_unit] : 0.f;
                wih[g][jj] = ok ? p.W_ih_st[(int64_t)(g * H + k0 + jj) * H + i_unit] : 0.f;
            } else {
                wih[g][jj] = 0.f;
            }
        }
    if (WIDE) {
        for (int e = tid; e < 256 * 16; e += 256) {   // owl[(unit * 4 + kq) * 20 + q] = W_out[kq + 4 q][unit], zero beyond NO / H
            const int q = e & 15, kqq = (e >> 4) & 3, un = e >> 6, r = kqq + 4 * q;
            owl[(un * 4 + kqq) * 20 + q] = (r < NO && un < H) ? p.out_W[(int64_t)r * H + un] : 0.f;
        }
    } else {
        for (int e = tid; e < NO * 64; e += 256) {
            int kk = e & 63, r = e >> 6;
            owl[e] = kk < H ? p.out_W[(int64_t)r * H + kk] : 0.f;
        }
    }
    float wxr[kMaxSRegV2][3];  // state rows of W_ih_l0 for this lane's unit
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int i = 0; i < kMaxSRegV2; ++i) wxr[i][g] = (!WIDE && unit_ok && i < S) ? p.W_ih0[(int64_t)(g * H + i_unit) * I + i] : 0.f;
    if (WIDE)
        for (int e = tid; e < S * 3 * 64; e += 256) {  // wxl[i][g][unit rotated by 16 (i & 3)]: lane kq reads i = kq mod 4, see below
            const int un = e & 63, g = (e >> 6) % 3, i = e / 192;
            wxl[(e - un) + ((un + 16 * (i & 3)) & 63)] = un < H ? p.W_ih0[(int64_t)(g * H + un) * I + i] : 0.f;
        }
    int trow = 0, tcol = 0;
    if (orow >= S && orow < NO) {
        int q = orow - S, r = 0;
        while ((r + 1) * (r + 2) / 2 <= q) ++r;
        trow = r; tcol = q - r * (r + 1) / 2;
    }
    const bool is_tril = orow >= S && orow < NO;
    const bool is_diag = is_tril && trow == tcol;

    float dh[L], spi[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int l = 0; l < L; ++l) dh[l] = 0.f;
    float dxreg = 0.f;  // lane i < S of every wave carries d z_t[i]

    const int64_t bt0 = (int64_t)b * T;
    // Chunk c covers the steps [max(0, c CH - phase), min(T, (c + 1) CH - phase)): the workgroups' chunk boundaries (bursts of
    // D4 stores and activation loads) are spread over time; the two workgroups that share a CU sit half a chunk apart.
    const int phase = (b * 5 + (b >> 8) * (CH / 2)) % CH;
    const int nchunks = (T + phase + CH - 1) / CH;
    auto chunk_t0 = [&](int c) { return max(0, c * CH - phase); };
    auto chunk_n = [&](int c) { return min(T, (c + 1) * CH - phase) - max(0, c * CH - phase); };
    const int ABUF = (CH + 1) * REC;  // floats of one activation buffer
    // the per-chunk loads below address global memory with tid_c / lane_c, opaque copies renewed once per chunk: their per-thread
    // base addresses are loop-invariant, and hoisted out of the chunk loop (all 256 VGPRs are in use) they are spilled and every
    // reload is followed by a full vmcnt(0) drain
    int tid_c = tid, lane_c = lane;
    // activation records t0-1 .. t0+n-1 (record -1 of the path is all zero)
    auto load_acts_sync = [&](int t0, int n, float *dst) {
        const int nrec = (n + 1) * REC;
        const float *src = p.acts + (bt0 + t0 - 1) * REC;
#pragma unroll 1
        for (int e = tid; e < nrec; e += 256) dst[e] = (t0 == 0 && e < REC) ? 0.f : src[e];
    };
    auto load_acts_dma = [&](int t0, int n, float *dst) {
        const int skip = t0 == 0 ? REC : 0;  // record -1 does not exist: zero it by hand, copy the rest
        if (skip) for (int e = tid; e < REC; e += 256) dst[e] = 0.f;
        const char *src = (const char *)(p.acts + (bt0 + t0 - 1) * REC + skip);
        const int nbytes = ((n + 1) * REC - skip) * 4;
#pragma unroll 1
        for (int off = wave * 1024; off < nbytes; off += 4096) {  // one instruction: 64 lanes x 16 B -> 1 KB of LDS
            if (off + lane_c * 16 < nbytes)
                __builtin_amdgcn_global_load_lds((const void *)(src + off + lane_c * 16),
                                                 (__attribute__((address_space(3))) void *)((char *)(dst + skip) + off), 16, 0, 0);
        }
    };
    // upstream gradients / eps / raw factors of a chunk: a few hundred floats, one or two per thread, prefetched in registers
    const int small_per_step = 3 * S + S * S + ntril;
    constexpr int NSV = WIDE ? 8 : 3;  // CH * (3S + S^2 + ntril) floats: <= 16 * 38 (S <= 4), <= 2048 in the wide variant
    float sv[NSV];
#pragma unroll
    for (int q = 0; q < NSV; ++q) sv[q] = 0.f;
    auto small_issue = [&](int t0, int n) {
#pragma unroll
        for (int q = 0; q < NSV; ++q) {
            const int e = tid_c + 256 * q;
            float v = 0.f;
            if (e < n * small_per_step) {
                int r = e;
                if (r < n * S) v = p.g_paths[((int64_t)b * (T + 1) + t0 + 1) * S + r];
                else if ((r -= n * S) < n * S) v = p.g_means[(bt0 + t0) * S + r];
                else if ((r -= n * S) < n * S) v = p.eps[(bt0 + t0) * S + r];
                else if ((r -= n * S) < n * S * S) v = p.g_chol[(bt0 + t0) * S * S + r];
                else { r -= n * S * S; v = p.chol_raw[(bt0 + t0) * ntril + r]; }
            }
            sv[q] = v;
        }
    };
    auto small_commit = [&](int n) {
#pragma unroll
        for (int q = 0; q < NSV; ++q) {
            const int e = tid + 256 * q;
            if (e < n * small_per_step) {
                int r = e;
                if (r < n * S) s_gp[r] = sv[q];
                else if ((r -= n * S) < n * S) s_gm[r] = sv[q];
                else if ((r -= n * S) < n * S) s_eps[r] = sv[q];
                else if ((r -= n * S) < n * S * S) s_gl[r] = sv[q];
                else { r -= n * S * S; s_raw[r] = sv[q]; }
            }
        }
    };
    {
        const int t0 = chunk_t0(nchunks - 1), n = chunk_n(nchunks - 1);
        s_acts = acts_buf0 + (DMA ? ((nchunks - 1) & 1) * ABUF : 0);
        if (DMA) load_acts_dma(t0, n, s_acts); else load_acts_sync(t0, n, s_acts);
        small_issue(t0, n);
        small_commit(n);
        if (DMA) __builtin_amdgcn_s_waitcnt(0);
    }
    __syncthreads();

    for (int c = nchunks - 1; c >= 0; --c) {
        const int t0 = chunk_t0(c), nsteps = chunk_n(c);
        const int tprev = c > 0 ? chunk_t0(c - 1) : 0, nprev = c > 0 ? chunk_n(c - 1) : 0;
        asm volatile("" : "+v"(tid_c), "+v"(lane_c));
        if (c > 0) {  // the next (earlier) chunk: its loads fly during this chunk's time steps
            if (DMA) load_acts_dma(tprev, nprev, acts_buf0 + ((c - 1) & 1) * ABUF);
            small_issue(tprev, nprev);
        }
        for (int tt = nsteps - 1; tt >= 0; --tt) {
            VSDE_TPB(tt == 5 ? 20 : 31);
            // ---- saved activations of this lane's unit (independent of the recurrence: issue first)
            float sr[L], su[L], sn[L], scn[L], shp[L];
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const float *A = s_acts + ((tt + 1) * L + l) * 5 * H + i_unit;
                sr[l] = unit_ok ? A[H] : 0.f; su[l] = unit_ok ? A[2 * H] : 0.f; sn[l] = unit_ok ? A[3 * H] : 0.f;
                scn[l] = unit_ok ? A[4 * H] : 0.f; shp[l] = unit_ok ? A[-REC] : 0.f;
            }
            // ---- upstream gradients for this step (backward.py:257-349); every wave redundantly
            if (lane < S) dxreg += s_gp[tt * S + lane];
            float mydx = 0.f;
            if (WIDE || SS == 0) {   // run-time state dimension: one cross-lane read instead of S x (v_readlane, compare, select)
                const int sel = orow < S ? orow : trow;
                mydx = __shfl(dxreg, (sel >= 0 && sel < S) ? sel : 0, 64);
                if (!(sel >= 0 && sel < S)) mydx = 0.f;
            } else {
                for (int i = 0; i < S; ++i) {
                    const float dxi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dxreg), i));
                    if ((orow < S ? orow : trow) == i) mydx = dxi;
                }
            }
            float dO = 0.f;
            if (orow < S) dO = mydx * p.dt + s_gm[tt * S + orow];
            else if (is_tril) {
                float dL = mydx * s_eps[tt * S + tcol] * p.sqdt + s_gl[(tt * S + trow) * S + tcol];
                const float raw = s_raw[tt * ntril + (orow - S)];
                if (is_diag && !(raw >= p.diag_min || dL < 0.f)) dL = 0.f;  // bounds.py:20 / backward.py:331-334
                dO = dL;
            }
            if ((WIDE || wave == 0) && kq == 0 && row_ok) s_dO[tt * NO + orow] = dO;
            if (WIDE && kq == 0 && row_ok) dOT[(orow & 3) * 16 + (orow >> 2)] = dO;   // the order the lanes read it back in (rows >= NO stay 0)
            VSDE_TPB(21);
            float dcur = 0.f;  // d h_top[i] = sum_r dO_r out_W[r][i]
            if (WIDE) {
                __syncthreads();  // the rows live in different waves: exchange through the staged record
                // a quarter of the rows per lane (r = kq + 4 q), weights and dO in that order: float4 reads, zero padding instead of
                // a run-time trip count (a loop over the rows waited for two LDS round trips per row)
                const float *wT = owl + (i_unit * 4 + kq) * 20, *dT = dOT + kq * 16;
                float dc2 = 0.f;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    if (hf == 1 && NO <= 32) break;   // rows 32.. exist from S = 7 on (workgroup-uniform)
                    const float4 w0 = *(const float4 *)(wT + 8 * hf), w1 = *(const float4 *)(wT + 8 * hf + 4);
                    const float4 d0 = *(const float4 *)(dT + 8 * hf), d1 = *(const float4 *)(dT + 8 * hf + 4);
                    dcur = fmaf(d0.x, w0.x, dcur); dc2 = fmaf(d0.y, w0.y, dc2); dcur = fmaf(d0.z, w0.z, dcur); dc2 = fmaf(d0.w, w0.w, dc2);
                    dcur = fmaf(d1.x, w1.x, dcur); dc2 = fmaf(d1.y, w1.y, dc2); dcur = fmaf(d1.z, w1.z, dcur); dc2 = fmaf(d1.w, w1.w, dc2);
                }
                dcur = quad_sum(dcur + dc2);
            } else {
                for (int r = 0; r < NO; ++r)
                    dcur = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(dO), 4 * r)), owl[r * 64 + i_unit], dcur);
            }
            VSDE_TPB(22);

#pragma unroll
            for (int l = L - 1; l >= 0; --l) {
                // ---- GRU cell adjoint (backward.py:59-67, 452-486)
                const float d = dcur + dh[l];
                const float dn = (1.0f - su[l]) * d, du = (shp[l] - sn[l]) * d;
                const float dn_pre = dn * (1.0f - sn[l] * sn[l]);
                const float du_pre = du * (su[l] * (1.0f - su[l]));
                const float dcn = dn_pre * sr[l];
                const float dr_pre = (dn_pre * scn[l]) * (sr[l] * (1.0f - sr[l]));
                const float carry = su[l] * d;
                float *D = s_d4 + (tt * L + l) * 4 * H;
                if (unit_ok) D[kq * H + i_unit] = kq == 0 ? dr_pre : (kq == 1 ? du_pre : (kq == 2 ? dn_pre : dcn));
                if (l == 0) {
                    spi[0] += dr_pre; spi[1] += du_pre; spi[2] += dn_pre;
                    // d z_t += W_ih_l0[:, state rows]^T . d_pre (backward.py:494-509): per-wave partial sums
                    if (WIDE) {
                        // lane kq of a quad takes the state components i = kq, kq + 4, kq + 8 (S <= 9 here; the four lanes of a quad hold the
                        // same gate gradients), two DPP row rotations add the four units of a 16-lane row per kq, and the 16 row sums per
                        // component (4 waves x 4 rows) are added by the lane that consumes them, after the barrier below.  (A loop over the
                        // run-time S in every lane, with v_readlane + scalar adds per component, sat on the critical path.)
                        float wv[3][3], v[3];
#pragma unroll
                        for (int ii = 0; ii < 3; ++ii) {
                            const int i = kq + 4 * ii, ic = i < S ? i : 0;
                            const float *wx = wxl + ic * 192 + ((i_unit + 16 * kq) & 63);
                            wv[ii][0] = wx[0]; wv[ii][1] = wx[64]; wv[ii][2] = wx[128];
                        }
#pragma unroll
                        for (int ii = 0; ii < 3; ++ii) {
                            v[ii] = wv[ii][0] * dr_pre + wv[ii][1] * du_pre + wv[ii][2] * dn_pre;
                            v[ii] = dpp_add<0x124>(v[ii]);   // row_ror:4
                            v[ii] = dpp_add<0x128>(v[ii]);   // row_ror:8: every lane holds the sum over its row's four units for its kq
                            if ((lane & 15) < 4 && kq + 4 * ii < S) dxp[(kq + 4 * ii) * 16 + wave * 4 + (lane >> 4)] = v[ii];
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < kMaxSRegV2; ++i) {
                            if (i < S) {
                                float v = wxr[i][0] * dr_pre + wxr[i][1] * du_pre + wxr[i][2] * dn_pre;
                                v = wave_sum_of_quads(v);  // the four lanes of a quad hold identical values
                                if (lane == 0) dxp[wave * 16 + i] = v;
                            }
                        }
                    }
                }
                VSDE_TPB(23 + 3 * (L - 1 - l));
                __syncthreads();
                VSDE_TPB(24 + 3 * (L - 1 - l));
                // ---- transposed products over this lane's j-slice, 8 j's at a time: with three 64x192 matrices in
                //      registers there is no room to hold all 64 staged values of the slice at once (scratch spills)
                const float *Dk = D + k0;
                // plain v_fma_f32 chains (v_pk_fma_f32 has no rate advantage on gfx950 and needs operand moves);
                // l > 0: (acc_h, acc_i) per gate; l == 0: even / odd j per gate
                float ax[3] = {0.f, 0.f, 0.f}, ay[3] = {0.f, 0.f, 0.f};
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const float4 r0 = *(const float4 *)(Dk + 8 * hf), r1 = *(const float4 *)(Dk + 8 * hf + 4);
                    const float4 u0 = *(const float4 *)(Dk + H + 8 * hf), u1 = *(const float4 *)(Dk + H + 8 * hf + 4);
                    const float4 c0 = *(const float4 *)(Dk + 3 * H + 8 * hf), c1 = *(const float4 *)(Dk + 3 * H + 8 * hf + 4);
                    const float vr[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
                    const float vu[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
                    const float vc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
                    if (l > 0) {
                        const float4 n0 = *(const float4 *)(Dk + 2 * H + 8 * hf), n1 = *(const float4 *)(Dk + 2 * H + 8 * hf + 4);
                        const float vn[8] = {n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w};
#pragma unroll
                        for (int j8 = 0; j8 < 8; ++j8) {
                            const int jj = 8 * hf + j8;
                            ax[0] = fmaf(vr[j8], whh[l][0][jj], ax[0]); ay[0] = fmaf(vr[j8], wih[0][jj], ay[0]);
                            ax[1] = fmaf(vu[j8], whh[l][1][jj], ax[1]); ay[1] = fmaf(vu[j8], wih[1][jj], ay[1]);
                            ax[2] = fmaf(vc[j8], whh[l][2][jj], ax[2]); ay[2] = fmaf(vn[j8], wih[2][jj], ay[2]);
                        }
                    } else {
#pragma unroll
                        for (int j8 = 0; j8 < 8; j8 += 2) {
                            const int jj = 8 * hf + j8;
                            ax[0] = fmaf(vr[j8], whh[0][0][jj], ax[0]); ay[0] = fmaf(vr[j8 + 1], whh[0][0][jj + 1], ay[0]);
                            ax[1] = fmaf(vu[j8], whh[0][1][jj], ax[1]); ay[1] = fmaf(vu[j8 + 1], whh[0][1][jj + 1], ay[1]);
                            ax[2] = fmaf(vc[j8], whh[0][2][jj], ax[2]); ay[2] = fmaf(vc[j8 + 1], whh[0][2][jj + 1], ay[2]);
                        }
                    }
                    if (hf == 0) __builtin_amdgcn_sched_barrier(0);  // keep the second half's loads behind the first half's FMAs
                }
                const float sx = (ax[0] + ax[1]) + ax[2], sy = (ay[0] + ay[1]) + ay[2];
                if (l > 0) {
                    dh[l] = carry + quad_sum(sx);
                    dcur = quad_sum(sy);
                } else {
                    dh[0] = carry + quad_sum(sx + sy);
                    if (WIDE) {
                        if (lane < S) {
                            const float4 q0 = *(const float4 *)(dxp + lane * 16), q1 = *(const float4 *)(dxp + lane * 16 + 4);
                            const float4 q2 = *(const float4 *)(dxp + lane * 16 + 8), q3 = *(const float4 *)(dxp + lane * 16 + 12);
                            dxreg += (((q0.x + q0.y) + (q0.z + q0.w)) + ((q1.x + q1.y) + (q1.z + q1.w))) +
                                     (((q2.x + q2.y) + (q2.z + q2.w)) + ((q3.x + q3.y) + (q3.z + q3.w)));
                        }
                    } else if (lane < S) dxreg += dxp[lane] + dxp[16 + lane] + dxp[32 + lane] + dxp[48 + lane];
                }
                VSDE_TPB(25 + 3 * (L - 1 - l));
            }
        }
        __syncthreads();  // chunk done: D4 / DO staging final, input staging free
        // ---- flush D4 / DO (contiguous runs), then stage the next (earlier) chunk
        {
            const int nD = nsteps * DREC;
            float *dst = p.D4 + (bt0 + t0) * DREC;
            if ((H & 3) == 0) {
#pragma unroll 1
                for (int e = tid; e < nD / 4; e += 256) ((float4 *)dst)[e] = ((const float4 *)s_d4)[e];
            } else {
#pragma unroll 1
                for (int e = tid; e < nD; e += 256) dst[e] = s_d4[e];
            }
#pragma unroll 1
            for (int e = tid; e < nsteps * NO; e += 256) p.DO[(bt0 + t0) * NO + e] = s_dO[e];
        }
        if (c > 0) {
            small_commit(nprev);
            if (DMA) { s_acts = acts_buf0 + ((c - 1) & 1) * ABUF; __builtin_amdgcn_s_waitcnt(0); }
            else load_acts_sync(tprev, nprev, s_acts);
        }
        __syncthreads();
    }
    if (wave == 0 && lane < S) p.g_x0[(int64_t)b * S + lane] = dxreg + p.g_paths[(int64_t)b * (T + 1) * S + lane];  // :620-624
    // d theta_b = W_ih_l0[:, theta rows]^T . sum_t d_pre_0   (backward.py:511-548)
    for (int q = 0; q < p.P; ++q) {
        float v = 0.f;
        if (unit_ok && kq == 0)
            v = p.W_ih0[(int64_t)i_unit * I + S + p.C + q] * spi[0] + p.W_ih0[(int64_t)(H + i_unit) * I + S + p.C + q] * spi[1] +
                p.W_ih0[(int64_t)(2 * H + i_unit) * I + S + p.C + q] * spi[2];
        v = wave_sum(v);
        __syncthreads();
        if (lane == 0) dxp[wave] = v;
        __syncthreads();
        if (tid == 0) p.g_theta[(int64_t)b * p.P + q] = dxp[0] + dxp[1] + dxp[2] + dxp[3];
    }
}

// =========================================================================================
// Generic ("wide") kernels: any hidden_dim <= 1024, any state_dim <= 32, L <= 4.  One workgroup
// per sample path, thread j = hidden unit j, weights streamed from global memory (L2-resident),
// hidden state and gate gradients exchanged through LDS.  Correctness-first path so that every
// HeadConfig the reference accepts (models/head.py:33-36 only bounds num_layers) runs; the tuned
// kernels above cover hidden_dim <= 64.
// =========================================================================================
template <bool SAVE>
__global__ void head_fwd_wide_kernel(FwdParams p, int L) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, nthr = blockDim.x, b = blockIdx.x;
    const int H = p.H, S = p.S, T = p.T, NO = p.NO, I = S + p.C + p.P, G3 = 3 * H;
    float *hst = smem;              // [L][H] hidden state of every layer
    float *hnew = hst + L * H;      // [H]    output of the layer just evaluated
    float *xbuf = hnew + H;         // [S]
    float *obuf = xbuf + S;         // [NO]
    for (int e = tid; e < L * H; e += nthr) hst[e] = 0.f;
    if (tid < S) { xbuf[tid] = p.x0[(int64_t)b * S + tid]; p.paths[(int64_t)b * (T + 1) * S + tid] = xbuf[tid]; }
    const int j = tid;
    const bool act = j < H;
    float gth[3] = {0.f, 0.f, 0.f};
    if (act)
        for (int g = 0; g < 3; ++g)
            for (int q = 0; q < p.P; ++q) gth[g] = fmaf(p.theta[(int64_t)b * p.P + q], p.W_ih0[(int64_t)(g * H + j) * I + S + p.C + q], gth[g]);
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const int64_t bt = (int64_t)b * T + t;
        for (int l = 0; l < L; ++l) {
            float a[3] = {0.f, 0.f, 0.f}, c[3] = {0.f, 0.f, 0.f};
            if (act) {
                const float *Wh = l == 0 ? p.W_hh0 : p.W_hh_st + (int64_t)(l - 1) * G3 * H;
                const float *bh = l == 0 ? p.b_hh0 : p.b_hh_st + (int64_t)(l - 1) * G3;
                for (int g = 0; g < 3; ++g) {
                    const float *wr = Wh + (int64_t)(g * H + j) * H;
                    float acc = bh[g * H + j];
                    for (int k = 0; k < H; ++k) acc = fmaf(hst[l * H + k], wr[k], acc);
                    c[g] = acc;
                    if (l == 0) {
                        float ai = p.G[bt * G3 + g * H + j] + gth[g];
                        for (int i = 0; i < S; ++i) ai = fmaf(xbuf[i], p.W_ih0[(int64_t)(g * H + j) * I + i], ai);
                        a[g] = ai;
                    } else {
                        const float *wi = p.W_ih_st + (int64_t)(l - 1) * G3 * H + (int64_t)(g * H + j) * H;
                        float ai = p.b_ih_st[(int64_t)(l - 1) * G3 + g * H + j];
                        for (int k = 0; k < H; ++k) ai = fmaf(hnew[k], wi[k], ai);
                        a[g] = ai;
                    }
                }
            }
            __syncthreads();  // every thread finished reading hnew / hst[l]
            if (act) {
                const float r = fast_sigmoid(a[0] + c[0]), u = fast_sigmoid(a[1] + c[1]);
                const float n = fast_tanh(a[2] + r * c[2]);
                const float hn = (1.0f - u) * n + u * hst[l * H + j];
                if (SAVE) {
                    float *A = p.acts + ((bt * L + l) * 5) * H + j;
                    A[0] = hn; A[H] = r; A[2 * H] = u; A[3 * H] = n; A[4 * H] = c[2];
                }
                hst[l * H + j] = hn; hnew[j] = hn;
            }
            __syncthreads();
        }
        for (int r = tid; r < NO; r += nthr) {  // emission rows (forward.py:314-362)
            float acc = p.out_b[r];
            for (int k = 0; k < H; ++k) acc = fmaf(hnew[k], p.out_W[(int64_t)r * H + k], acc);
            if (SAVE && r >= S) p.chol_raw[bt * p.ntril + (r - S)] = acc;
            if (r >= S) {
                int q = r - S, rr = 0;
                while ((rr + 1) * (rr + 2) / 2 <= q) ++rr;
                if (q - rr * (rr + 1) / 2 == rr && acc < p.diag_min) acc = p.diag_min;
            }
            obuf[r] = acc;
        }
        __syncthreads();
        float xn = 0.f;
        if (tid < S) {
            float acc = 0.f;
            const int base = S + tid * (tid + 1) / 2;
            for (int q = 0; q <= tid; ++q) acc = fmaf(obuf[base + q], p.eps[bt * S + q], acc);
            xn = xbuf[tid] + obuf[tid] * p.dt + acc * p.sqdt;
            p.means[bt * S + tid] = obuf[tid];
            p.paths[((int64_t)b * (T + 1) + t + 1) * S + tid] = xn;
        }
        for (int e = tid; e < S * S; e += nthr) {
            int rr = e / S, cl = e - rr * S;
            p.chol[bt * S * S + e] = cl <= rr ? obuf[S + rr * (rr + 1) / 2 + cl] : 0.f;
        }
        __syncthreads();
        if (tid < S) xbuf[tid] = xn;
        __syncthreads();
    }
}

__global__ void head_bwd_wide_kernel(BwdParams p, int L) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, nthr = blockDim.x, b = blockIdx.x;
    const int H = p.H, S = p.S, T = p.T, NO = p.NO, I = S + p.C + p.P, G3 = 3 * H;
    float *dbuf = smem;             // [4][H] gate gradients of the layer being processed
    float *dobuf = dbuf + 4 * H;    // [NO]
    float *xbuf = dobuf + NO;       // [S]   d z_t
    float *red = xbuf + S;          // [S][H] per-unit contributions to d z_t
    const int j = tid;
    const bool act = j < H;
    float dh[VSDE_MAX_LAYERS] = {0.f, 0.f, 0.f, 0.f}, spi[3] = {0.f, 0.f, 0.f};
    if (tid < S) xbuf[tid] = 0.f;
    __syncthreads();
    for (int t = T - 1; t >= 0; --t) {
        const int64_t bt = (int64_t)b * T + t;
        if (tid < S) xbuf[tid] += p.g_paths[((int64_t)b * (T + 1) + t + 1) * S + tid];
        __syncthreads();
        for (int r = tid; r < NO; r += nthr) {
            float dO;
            if (r < S) dO = xbuf[r] * p.dt + p.g_means[bt * S + r];
            else {
                int q = r - S, rr = 0;
                while ((rr + 1) * (rr + 2) / 2 <= q) ++rr;
                const int cl = q - rr * (rr + 1) / 2;
                dO = xbuf[rr] * p.eps[bt * S + cl] * p.sqdt + p.g_chol[(bt * S + rr) * S + cl];
                if (rr == cl && !(p.chol_raw[bt * p.ntril + q] >= p.diag_min || dO < 0.f)) dO = 0.f;
            }
            dobuf[r] = dO;
            p.DO[bt * NO + r] = dO;
        }
        __syncthreads();
        float dcur = 0.f;
        if (act)
            for (int r = 0; r < NO; ++r) dcur = fmaf(dobuf[r], p.out_W[(int64_t)r * H + j], dcur);
        for (int l = L - 1; l >= 0; --l) {
            float carry = 0.f, pi0 = 0.f, pi1 = 0.f, pi2 = 0.f;
            if (act) {
                const float *A = p.acts + ((bt * L + l) * 5) * H + j;
                const float r = A[H], u = A[2 * H], n = A[3 * H], cn = A[4 * H];
                const float hprev = t > 0 ? A[-(int64_t)L * 5 * H] : 0.f;
                const float d = dcur + dh[l];
                const float dn = (1.0f - u) * d, du = (hprev - n) * d;
                const float dn_pre = dn * (1.0f - n * n), du_pre = du * (u * (1.0f - u));
                const float dcn = dn_pre * r, dr_pre = (dn_pre * cn) * (r * (1.0f - r));
                carry = u * d;
                float *D = p.D4 + (bt * L + l) * 4 * H + j;
                D[0] = dr_pre; D[H] = du_pre; D[2 * H] = dn_pre; D[3 * H] = dcn;
                dbuf[j] = dr_pre; dbuf[H + j] = du_pre; dbuf[2 * H + j] = dn_pre; dbuf[3 * H + j] = dcn;
                pi0 = dr_pre; pi1 = du_pre; pi2 = dn_pre;
            }
            __syncthreads();
            if (act) {
                const float *Wh = l == 0 ? p.W_hh0 : p.W_hh_st + (int64_t)(l - 1) * G3 * H;
                float acc_h = carry, acc_i = 0.f;
                for (int k = 0; k < H; ++k) {
                    acc_h = fmaf(Wh[(int64_t)k * H + j], dbuf[k], acc_h);
                    acc_h = fmaf(Wh[(int64_t)(H + k) * H + j], dbuf[H + k], acc_h);
                    acc_h = fmaf(Wh[(int64_t)(2 * H + k) * H + j], dbuf[3 * H + k], acc_h);
                }
                if (l > 0) {
                    const float *Wi = p.W_ih_st + (int64_t)(l - 1) * G3 * H;
                    for (int k = 0; k < H; ++k) {
                        acc_i = fmaf(Wi[(int64_t)k * H + j], dbuf[k], acc_i);
                        acc_i = fmaf(Wi[(int64_t)(H + k) * H + j], dbuf[H + k], acc_i);
                        acc_i = fmaf(Wi[(int64_t)(2 * H + k) * H + j], dbuf[2 * H + k], acc_i);
                    }
                    dcur = acc_i;
                } else {
                    spi[0] += pi0; spi[1] += pi1; spi[2] += pi2;
                    for (int i = 0; i < S; ++i)
                        red[i * H + j] = p.W_ih0[(int64_t)j * I + i] * pi0 + p.W_ih0[(int64_t)(H + j) * I + i] * pi1 +
                                         p.W_ih0[(int64_t)(2 * H + j) * I + i] * pi2;
                }
                dh[l] = acc_h;
            }
            __syncthreads();
            if (l == 0 && tid < S) {  // fixed-order sum over the units
                float acc = 0.f;
                for (int k = 0; k < H; ++k) acc += red[tid * H + k];
                xbuf[tid] += acc;
            }
        }
        __syncthreads();
    }
    if (tid < S) p.g_x0[(int64_t)b * S + tid] = xbuf[tid] + p.g_paths[(int64_t)b * (T + 1) * S + tid];
    for (int q = 0; q < p.P; ++q) {  // d theta (backward.py:511-548)
        __syncthreads();
        if (act) red[j] = p.W_ih0[(int64_t)j * I + S + p.C + q] * spi[0] + p.W_ih0[(int64_t)(H + j) * I + S + p.C + q] * spi[1] +
                          p.W_ih0[(int64_t)(2 * H + j) * I + S + p.C + q] * spi[2];
        __syncthreads();
        if (tid == 0) {
            float acc = 0.f;
            for (int k = 0; k < H; ++k) acc += red[k];
            p.g_theta[(int64_t)b * p.P + q] = acc;
        }
    }
}

// ------------------------------------------------------------------------ host launchers
// Optional per-kernel timing with HIP events on the launch stream (bench.py roofline block).
static bool g_force_v1 = false;  // test hook: run L<=2 through the LDS-resident v1 kernels
static bool g_prof_on = false;
// which: 0 = serial forward kernel (training variant), 1 = serial backward kernel, 2 = whole vsde_head_forward (training),
// 3 = whole vsde_head_backward, 4 = forward context-projection GEMM, 5 = grad_context GEMM, 6 = grouped weight-gradient reduction
constexpr int kProfSlots = 7;
static hipEvent_t g_prof_ev[kProfSlots][2];
static bool g_prof_init = false;
static bool g_prof_valid[kProfSlots] = {};
static void prof_mark(int which, int edge, hipStream_t s) {
    if (!g_prof_on) return;
    if (!g_prof_init) {
        for (int i = 0; i < kProfSlots; ++i) for (int j = 0; j < 2; ++j) (void)hipEventCreate(&g_prof_ev[i][j]);
        g_prof_init = true;
    }
    (void)hipEventRecord(g_prof_ev[which][edge], s);
    if (edge == 1) g_prof_valid[which] = true;
}

static int check_dims(const vsde_head_dims *d) {
    VSDE_CHECK_ARG(d != nullptr, VSDE_E_BADARG, "dims is NULL");
    VSDE_CHECK_ARG(d->B > 0 && d->T > 0 && d->S > 0 && d->P >= 0 && d->C > 0 && d->H > 0, VSDE_E_BADARG,
                   "bad dims B=%d T=%d S=%d P=%d C=%d H=%d", d->B, d->T, d->S, d->P, d->C, d->H);
    VSDE_CHECK_ARG(d->L >= 1 && d->L <= VSDE_MAX_LAYERS, VSDE_E_LAYERS, "num_layers must be in [1, %d], got %d",
                   VSDE_MAX_LAYERS, d->L);
    VSDE_CHECK_ARG(d->H <= VSDE_MAX_HIDDEN, VSDE_E_HIDDEN, "hidden_dim %d > %d is not supported by the gfx950 kernels", d->H,
                   VSDE_MAX_HIDDEN);
    VSDE_CHECK_ARG(d->S <= VSDE_MAX_STATE, VSDE_E_STATE, "state_dim %d > %d is not supported by the gfx950 kernels", d->S,
                   VSDE_MAX_STATE);
    VSDE_CHECK_ARG((size_t)(d->L * d->H + 2 * d->H + 4 * d->H + (d->S + 2) * (d->S + d->H + 4)) * sizeof(float) <= 150 * 1024,
                   VSDE_E_HIDDEN, "hidden_dim %d x state_dim %d exceeds the LDS budget of the generic kernels", d->H, d->S);
    return 0;
}

static inline size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }
// VSDE_PROJ_GENERIC=1: run the context projection / grad_context on the generic fp32-MFMA kernel (A/B against vsde_proj.hip)
static bool proj_fast_path() {
    static int generic = -1;
    if (generic < 0) generic = (int)vsde_knob("VSDE_PROJ_GENERIC", 0);
    return generic == 0;
}

struct FwdLayout { size_t packF, packO, Wc, planes, G, frags, total; };
static FwdLayout fwd_layout(const vsde_head_dims *d) {
    const int NO = d->S + d->S * (d->S + 1) / 2;
    FwdLayout o; size_t off = 0;
    o.packF = off; off += align256((size_t)(2 * d->L - 1) * kMatF4 * sizeof(float4));
    o.packO = off; off += align256((size_t)kChunks * NO * sizeof(float4));
    o.Wc = off; off += align256((size_t)3 * d->H * d->C * sizeof(float));
    o.planes = off; off += align256(proj_planes_bytes(3 * d->H, d->C));   // bf16 planes of W_c (vsde_proj.hip)
    o.G = off; off += align256((size_t)d->B * d->T * 3 * d->H * sizeof(float));
    o.frags = off; off += mp_applicable(d->H, d->L, d->S) ? align256(mp_frag_bytes(d->L, d->S)) : 0;   // vsde_head_mp.hip
    o.total = off;
    return o;
}

// Which forward time-stepping kernel takes hidden_dim 64 / L <= 2 / state_dim <= 2: the multi-path MFMA kernel (vsde_head_mp.hip) or
// the four-waves-per-path v2 kernel.  VSDE_HEAD_MP = 0 / 1 forces one of them (A/B runs); vsde_debug_head_mp() does the same
// from tests.  Default: see mp_auto().
static int g_mp_mode = -1;   // -1 = environment / auto, 0 = never, 1 = whenever applicable; 4 / 8 / 16 = applicable + that many paths per group
static int mp_env() {
    static int mode = -2;
    if (mode == -2) mode = (int)vsde_knob("VSDE_HEAD_MP", -1);
    return mode;
}
static bool mp_auto(const vsde_head_dims *d, int save) {
    // Measured at the LV head dims (profiles/r04_head_mp.txt; us, training / no-grad launch, v2 vs multi-path): 128 paths 508 / 426 vs
    // 461 / 441, 256 paths 535 / 440 vs 462 / 441, 512 paths 676 / 549 vs 492 / 445, 4096 paths 4709 / 3967 vs 1104 / 591.  The
    // no-grad launch of the v2 kernel (one path per CU up to 256 paths) keeps its edge there; the training launch does not since the
    // layer-1 role's records leave through LDS.
    // Round 5 (spread forms, groups of 2 paths up to 512; tools/head_mp_check.py time): 128 paths 507 / 423 (v2) vs 306 / 301, 256 paths
    // 530 / 434 vs 319 / 306, 512 paths 688 / 539 vs 380 / 329: the multi-path kernel from 32 paths on, both launches.
    (void)save;
    return d->B >= 32;
}
static bool use_mp(const vsde_head_dims *d, int save) {
    if (!mp_applicable(d->H, d->L, d->S) || mp_weights_overflowed()) return false;
    const int mode = g_mp_mode >= 0 ? g_mp_mode : mp_env();
    if (mode == 0) return false;
    if (mode > 0) return true;
    return mp_auto(d, save);
}
// The reverse-time sweep on the matrix cores pays from ~700 paths on (profiles/r04_head_mp.txt: 512 paths 993 us against 800 us for
// the v2 kernel, 1024 paths 1017 against 1571 us, 4096 paths 2235 against 6253 us).  What holds it back at 512 paths is one CU's LDS
// bandwidth: K = 192 products want six B fragments each and four published gradient vectors per layer
// (profiles/r04_head_bwd_ablation.txt).
static bool use_mp_bwd(const vsde_head_dims *d) {
    if (!mp_bwd_applicable(d->H, d->L, d->S) || mp_weights_overflowed()) return false;
    const int mode = g_mp_mode >= 0 ? g_mp_mode : mp_env();
    if (mode == 0) return false;
    if (mode > 0) return true;
    // Round 5 (spread sweep head_bwd_mps_kernel, tools/head_mp_check.py time): 128 paths 721 (v2) vs 490 us, 512 paths 812 vs 501, 1024 paths
    // 1567 vs 817: the matrix-core sweep from 32 paths on
    return d->B >= 32;
}

static int pick_wpb(int B, size_t lds_fixed, size_t lds_per_wave) {
    int wpb = (B + 255) / 256;  // fill all 256 CUs first
    if (wpb < 1) wpb = 1;
    if (wpb > 8) wpb = 8;
    while (wpb > 1 && lds_fixed + (size_t)wpb * lds_per_wave > 160 * 1024) --wpb;
    return wpb;
}

static RowView ctx_rowview(const vsde_context_view *c, int T, int C) {
    RowView v;
    v.base = c->base; v.batch_stride = c->batch_stride; v.row_stride = c->step_stride; v.rows_per_batch = T;
    v.shift = 0; v.col_split = C; v.col_skip = 0; v.dtype = c->dtype == VSDE_CTX_BF16 ? 1 : 0;
    return v;
}

template <int L>
static int launch_fwd_L(const FwdParams &p, bool save, int grid, int block, size_t lds, hipStream_t s) {
    if (save) {
        VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)head_fwd_kernel<L, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        prof_mark(0, 0, s);
        hipLaunchKernelGGL((head_fwd_kernel<L, true>), dim3(grid), dim3(block), lds, s, p);
        prof_mark(0, 1, s);
    } else {
        VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)head_fwd_kernel<L, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((head_fwd_kernel<L, false>), dim3(grid), dim3(block), lds, s, p);
    }
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}

template <int L>
static int launch_bwd_L(const BwdParams &p, int grid, int block, size_t lds, hipStream_t s) {
    VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)head_bwd_kernel<L>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    prof_mark(1, 0, s);
    hipLaunchKernelGGL((head_bwd_kernel<L>), dim3(grid), dim3(block), lds, s, p);
    prof_mark(1, 1, s);
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace vsde

using namespace vsde;

#ifdef VSDE_TRACE
extern "C" int vsde_debug_read_trace(long long *host64) {
    VSDE_CHECK_HIP(hipMemcpyFromSymbol(host64, HIP_SYMBOL(vsde::g_trace), sizeof(long long) * 64));
    return 0;
}
#endif

extern "C" int vsde_debug_force_v1(int on) {
    g_force_v1 = on != 0;
    return 0;
}

extern "C" int vsde_debug_head_mp(int mode) {
    g_mp_mode = mode < 0 ? -1 : mode;
    return 0;
}

extern "C" int vsde_head_mfma_range_exceeded(int clear) {
    if (clear > 0) { mp_clear_overflow(); return 0; }
    return mp_weights_overflowed() ? 1 : 0;
}

extern "C" int vsde_profile_enable(int on) {
    g_prof_on = on != 0;
    for (int i = 0; i < kProfSlots; ++i) g_prof_valid[i] = false;
    return 0;
}

extern "C" int vsde_profile_elapsed_ms(int which, float *ms) {
    VSDE_CHECK_ARG(which >= 0 && which < kProfSlots && ms, VSDE_E_BADARG, "bad profile query");
    VSDE_CHECK_ARG(g_prof_valid[which], VSDE_E_BADARG, "no timed launch of kernel %d recorded", which);
    VSDE_CHECK_HIP(hipEventSynchronize(g_prof_ev[which][1]));
    VSDE_CHECK_HIP(hipEventElapsedTime(ms, g_prof_ev[which][0], g_prof_ev[which][1]));
    return 0;
}

extern "C" size_t vsde_head_forward_workspace_bytes(const vsde_head_dims *d) {
    if (check_dims(d) != 0) return 0;
    return fwd_layout(d).total;
}

extern "C" int vsde_head_forward(const vsde_head_dims *d, const float *x0, const vsde_context_view *ctx,
                                 const float *theta, const float *eps, const vsde_head_weights *w,
                                 double time_step, double diag_min, int save, float *paths, float *means,
                                 float *chol, float *chol_raw, float *acts, void *workspace,
                                 size_t workspace_bytes, void *stream) {
    int rc = check_dims(d);
    if (rc) return rc;
    VSDE_CHECK_ARG(x0 && ctx && ctx->base && eps && w && paths && means && chol && workspace, VSDE_E_BADARG, "NULL argument");
    VSDE_CHECK_ARG(d->P == 0 || theta, VSDE_E_BADARG, "theta is NULL");
    VSDE_CHECK_ARG(!save || (chol_raw && acts), VSDE_E_BADARG, "training mode needs chol_raw and acts buffers");
    VSDE_CHECK_ARG(d->L == 1 || (w->W_ih_stack && w->W_hh_stack && w->b_ih_stack && w->b_hh_stack), VSDE_E_BADARG,
                   "stacked layer weights are NULL");
    VSDE_CHECK_ARG(time_step > 0, VSDE_E_BADARG, "time_step must be positive");
    const FwdLayout lay = fwd_layout(d);
    VSDE_CHECK_ARG(workspace_bytes >= lay.total, VSDE_E_WORKSPACE, "forward workspace too small: %zu < %zu", workspace_bytes, lay.total);
    hipStream_t s = (hipStream_t)stream;
    char *ws = (char *)workspace;
    const int NO = d->S + d->S * (d->S + 1) / 2;

    PackParams pk = {};
    pk.S = d->S; pk.P = d->P; pk.C = d->C; pk.H = d->H; pk.L = d->L; pk.NO = NO;
    pk.W_ih0 = w->W_ih_l0; pk.W_hh0 = w->W_hh_l0; pk.W_ih_st = w->W_ih_stack; pk.W_hh_st = w->W_hh_stack; pk.out_W = w->out_weight;
    const bool wide = d->H > kHP || NO > kWave;
    if (!wide) { pk.packF = (float4 *)(ws + lay.packF); pk.packO = (float4 *)(ws + lay.packO); }
    pk.Wc = (float *)(ws + lay.Wc);
    if (save) prof_mark(2, 0, s);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(64), dim3(256), 0, s, pk);
    VSDE_CHECK_HIP(hipGetLastError());

    float *G = (float *)(ws + lay.G);
    if (save) prof_mark(4, 0, s);
    rc = proj_fast_path() ? launch_proj_fwd_bf16(ctx_rowview(ctx, d->T, d->C), (int64_t)d->B * d->T, d->C, pk.Wc, d->C, 3 * d->H, w->b_ih_l0, G,
                                                 3 * d->H, ws + lay.planes, proj_planes_bytes(3 * d->H, d->C), s) : 0;
    if (rc < 0) return rc;
    if (rc == 0) rc = launch_gemm_nt(ctx_rowview(ctx, d->T, d->C), d->B * d->T, d->C, pk.Wc, d->C, 3 * d->H, w->b_ih_l0, G, 3 * d->H, s);
    else rc = 0;
    if (save) prof_mark(4, 1, s);
    if (rc) return rc;

    FwdParams p = {};
    p.B = d->B; p.T = d->T; p.S = d->S; p.P = d->P; p.C = d->C; p.H = d->H; p.NO = NO; p.ntril = NO - d->S;
    p.x0 = x0; p.theta = theta; p.eps = eps; p.G = G;
    p.W_ih0 = w->W_ih_l0; p.b_hh0 = w->b_hh_l0; p.b_ih_st = w->b_ih_stack; p.b_hh_st = w->b_hh_stack; p.out_b = w->out_bias;
    p.W_hh0 = w->W_hh_l0; p.W_ih_st = w->W_ih_stack; p.W_hh_st = w->W_hh_stack; p.out_W = w->out_weight;
    p.packF = pk.packF; p.packO = pk.packO;
    p.dt = (float)time_step; p.sqdt = (float)sqrt(time_step); p.diag_min = (float)diag_min;
    p.paths = paths; p.means = means; p.chol = chol; p.chol_raw = chol_raw; p.acts = acts;
    if (d->H > kHP || NO > kWave) {  // generic kernels: any hidden_dim / state_dim
        const int block = ((d->H > NO ? d->H : NO) + 63) / 64 * 64 > 1024 ? 1024 : ((d->H > NO ? d->H : NO) + 63) / 64 * 64;
        const size_t ldsw = (size_t)(d->L * d->H + d->H + d->S + NO + 8) * sizeof(float);
        if (save) hipLaunchKernelGGL((head_fwd_wide_kernel<true>), dim3(d->B), dim3(block), ldsw, s, p, d->L);
        else hipLaunchKernelGGL((head_fwd_wide_kernel<false>), dim3(d->B), dim3(block), ldsw, s, p, d->L);
        VSDE_CHECK_HIP(hipGetLastError());
        if (save) prof_mark(2, 1, s);
        return 0;
    }
    if (!g_force_v1 && use_mp(d, save)) {   // 16 paths per workgroup on the matrix cores
        MpLaunch a = {};
        a.B = d->B; a.T = d->T; a.S = d->S; a.P = d->P; a.C = d->C; a.L = d->L; a.save = save;
        a.np = g_mp_mode >= 0 ? g_mp_mode : mp_env();   // 4 / 8 / 16 force the group size, anything else: by batch size
        a.x0 = x0; a.theta = theta; a.eps = eps; a.G = G;
        a.W_ih0 = w->W_ih_l0; a.W_hh0 = w->W_hh_l0; a.W_ih_st = w->W_ih_stack; a.W_hh_st = w->W_hh_stack; a.out_W = w->out_weight;
        a.b_hh0 = w->b_hh_l0; a.b_ih_st = w->b_ih_stack; a.b_hh_st = w->b_hh_stack; a.out_b = w->out_bias;
        a.frags = ws + lay.frags;
        a.dt = p.dt; a.sqdt = p.sqdt; a.diag_min = p.diag_min;
        a.paths = paths; a.means = means; a.chol = chol; a.chol_raw = chol_raw; a.acts = acts;
        rc = launch_head_fwd_mp(a, s, prof_mark);
        if (save && rc == 0) prof_mark(2, 1, s);
        return rc;
    }
    if (d->L <= 2 && !g_force_v1) {  // register-resident 4-waves-per-path kernels
        p.wpb = 1;
        int ch = 16;  // two workgroups per CU need <= 80 KB of LDS each
        if ((size_t)fwd_v2_lds(d->H, d->S, d->L, 16, save != 0).total * sizeof(float) > 80 * 1024) ch = 8;
        const size_t lds2 = (size_t)fwd_v2_lds(d->H, d->S, d->L, ch, save != 0).total * sizeof(float);
        VSDE_CHECK_ARG(lds2 <= 160 * 1024, VSDE_E_STATE, "LDS budget exceeded (%zu B)", lds2);
#define VSDE_LAUNCH_FWD_V2_(LL, SV, CC, SM)                                                                         \
    do {                                                                                                            \
        VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)head_fwd_v2_kernel<LL, SV, CC, SM>,                        \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));                  \
        prof_mark(0, 0, s);                                                                                         \
        hipLaunchKernelGGL((head_fwd_v2_kernel<LL, SV, CC, SM>), dim3(d->B), dim3(256), lds2, s, p);                \
        prof_mark(0, 1, s);                                                                                         \
    } while (0)
#define VSDE_LAUNCH_FWD_V2(LL, SV, CC)                                                                              \
    do {                                                                                                            \
        if (d->S == 1 && CC == 16) VSDE_LAUNCH_FWD_V2_(LL, SV, 16, 1);                                              \
        else if (d->S == 2 && CC == 16) VSDE_LAUNCH_FWD_V2_(LL, SV, 16, 2);                                         \
        else if (NO <= 16) VSDE_LAUNCH_FWD_V2_(LL, SV, CC, 3);                                                      \
        else VSDE_LAUNCH_FWD_V2_(LL, SV, CC, 4);                                                                    \
    } while (0)
        if (d->L == 1) {
            if (save) { if (ch == 16) VSDE_LAUNCH_FWD_V2(1, true, 16); else VSDE_LAUNCH_FWD_V2(1, true, 8); }
            else { if (ch == 16) VSDE_LAUNCH_FWD_V2(1, false, 16); else VSDE_LAUNCH_FWD_V2(1, false, 8); }
        } else {
            if (save) { if (ch == 16) VSDE_LAUNCH_FWD_V2(2, true, 16); else VSDE_LAUNCH_FWD_V2(2, true, 8); }
            else { if (ch == 16) VSDE_LAUNCH_FWD_V2(2, false, 16); else VSDE_LAUNCH_FWD_V2(2, false, 8); }
        }
#undef VSDE_LAUNCH_FWD_V2
#undef VSDE_LAUNCH_FWD_V2_
        VSDE_CHECK_HIP(hipGetLastError());
        if (save) prof_mark(2, 1, s);
        return 0;
    }
    const size_t lds_fixed = (size_t)fwd_lds_matrices(d->L) * kMatF4 * sizeof(float4) + (size_t)kChunks * NO * sizeof(float4);
    const size_t lds_wave = kScratchPerWave * sizeof(float);
    p.wpb = pick_wpb(d->B, lds_fixed, lds_wave);
    const size_t lds = lds_fixed + p.wpb * lds_wave;
    VSDE_CHECK_ARG(lds <= 160 * 1024, VSDE_E_STATE, "LDS budget exceeded (%zu B)", lds);
    const int grid = (d->B + p.wpb - 1) / p.wpb, block = 64 * p.wpb;
    switch (d->L) {
        case 1: rc = launch_fwd_L<1>(p, save != 0, grid, block, lds, s); break;
        case 2: rc = launch_fwd_L<2>(p, save != 0, grid, block, lds, s); break;
        case 3: rc = launch_fwd_L<3>(p, save != 0, grid, block, lds, s); break;
        default: rc = launch_fwd_L<4>(p, save != 0, grid, block, lds, s); break;
    }
    if (save && rc == 0) prof_mark(2, 1, s);
    return rc;
}

namespace vsde {
struct BwdLayout { size_t packB, WcT, D4, DO, frags, tn, total; };

static int build_tn(const vsde_head_dims *d, const vsde_context_view *ctx, const float *theta, const float *paths,
                    const float *acts, const float *D4, const float *DO, const vsde_head_grads *g, TnProblem *pr) {
    const int H = d->H, L = d->L, S = d->S, C = d->C, P = d->P, T = d->T, I = S + C + P, NO = S + S * (S + 1) / 2;
    int n = 0;
    auto d4view = [&](int l, bool hh) {
        RowView v; v.base = D4 + (int64_t)l * 4 * H; v.batch_stride = (int64_t)T * L * 4 * H; v.row_stride = (int64_t)L * 4 * H;
        v.rows_per_batch = T; v.shift = 0; v.dtype = 0;
        if (hh) { v.col_split = 2 * H; v.col_skip = H; } else { v.col_split = 3 * H; v.col_skip = 0; }
        return v;
    };
    auto actview = [&](int l, int shift) {  // hidden state h^l_{t+shift}
        RowView v; v.base = acts + (int64_t)l * 5 * H; v.batch_stride = (int64_t)T * L * 5 * H; v.row_stride = (int64_t)L * 5 * H;
        v.rows_per_batch = T; v.shift = shift; v.col_split = H; v.col_skip = 0; v.dtype = 0;
        return v;
    };
    // layer 0, W_ih_l0 = [state | context | theta] column blocks  (backward.py:494-564)
    {
        TnProblem &q = pr[n++]; q.X = d4view(0, false); q.NX = 3 * H;
        q.Y = ctx_rowview(ctx, T, C); q.NY = C; q.out = g->W_ih_l0; q.ldo = I; q.col_off = S; q.bias_out = g->b_ih_l0;
    }
    {
        TnProblem &q = pr[n++]; q.X = d4view(0, false); q.NX = 3 * H;
        RowView y; y.base = paths; y.batch_stride = (int64_t)(T + 1) * S; y.row_stride = S; y.rows_per_batch = T; y.shift = 0;
        y.col_split = S; y.col_skip = 0; y.dtype = 0;
        q.Y = y; q.NY = S; q.out = g->W_ih_l0; q.ldo = I; q.col_off = 0; q.bias_out = nullptr;
    }
    if (P > 0) {
        TnProblem &q = pr[n++]; q.X = d4view(0, false); q.NX = 3 * H;
        RowView y; y.base = theta; y.batch_stride = P; y.row_stride = 0; y.rows_per_batch = T; y.shift = 0;
        y.col_split = P; y.col_skip = 0; y.dtype = 0;
        q.Y = y; q.NY = P; q.out = g->W_ih_l0; q.ldo = I; q.col_off = S + C; q.bias_out = nullptr;
    }
    {
        TnProblem &q = pr[n++]; q.X = d4view(0, true); q.NX = 3 * H;
        q.Y = actview(0, -1); q.NY = H; q.out = g->W_hh_l0; q.ldo = H; q.col_off = 0; q.bias_out = g->b_hh_l0;
    }
    for (int l = 1; l < L; ++l) {
        {
            TnProblem &q = pr[n++]; q.X = d4view(l, false); q.NX = 3 * H;
            q.Y = actview(l - 1, 0); q.NY = H; q.out = g->W_ih_stack + (int64_t)(l - 1) * 3 * H * H; q.ldo = H; q.col_off = 0;
            q.bias_out = g->b_ih_stack + (int64_t)(l - 1) * 3 * H;
        }
        {
            TnProblem &q = pr[n++]; q.X = d4view(l, true); q.NX = 3 * H;
            q.Y = actview(l, -1); q.NY = H; q.out = g->W_hh_stack + (int64_t)(l - 1) * 3 * H * H; q.ldo = H; q.col_off = 0;
            q.bias_out = g->b_hh_stack + (int64_t)(l - 1) * 3 * H;
        }
    }
    {
        TnProblem &q = pr[n++];
        RowView x; x.base = DO; x.batch_stride = (int64_t)T * NO; x.row_stride = NO; x.rows_per_batch = T; x.shift = 0;
        x.col_split = NO; x.col_skip = 0; x.dtype = 0;
        q.X = x; q.NX = NO; q.Y = actview(L - 1, 0); q.NY = H; q.out = g->out_weight; q.ldo = H; q.col_off = 0;
        q.bias_out = g->out_bias;
    }
    return n;
}

static BwdLayout bwd_layout(const vsde_head_dims *d) {
    const int NO = d->S + d->S * (d->S + 1) / 2;
    BwdLayout o; size_t off = 0;
    o.packB = off; off += align256((size_t)(2 * d->L - 1) * kMatF4 * sizeof(float4));
    o.WcT = off; off += align256((size_t)3 * d->H * d->C * sizeof(float));
    o.D4 = off; off += align256((size_t)d->B * d->T * d->L * 4 * d->H * sizeof(float));
    o.DO = off; off += align256((size_t)d->B * d->T * NO * sizeof(float));
    o.frags = off; off += mp_bwd_applicable(d->H, d->L, d->S) ? align256(mp_bwd_frag_bytes()) : 0;   // vsde_head_mp.hip
    o.tn = off;
    // the TN plan only depends on shapes; build it with dummy pointers
    TnProblem pr[kMaxTnProblems];
    vsde_head_grads g = {}; vsde_context_view cv = {}; float dummy = 0.f;
    g.b_ih_l0 = g.b_hh_l0 = g.b_ih_stack = g.b_hh_stack = g.out_bias = &dummy;
    int n = build_tn(d, &cv, nullptr, nullptr, nullptr, nullptr, nullptr, &g, pr);
    off += align256(tn_workspace_bytes(pr, n, d->B * d->T));
    o.total = off;
    return o;
}
}  // namespace vsde

extern "C" size_t vsde_head_backward_workspace_bytes(const vsde_head_dims *d) {
    if (check_dims(d) != 0) return 0;
    return bwd_layout(d).total;
}

extern "C" int vsde_head_backward(const vsde_head_dims *d, const float *g_paths, const float *g_means,
                                  const float *g_chol, const vsde_context_view *ctx, const float *theta,
                                  const float *eps, const float *paths, const float *chol_raw,
                                  const float *acts, const vsde_head_weights *w, double time_step,
                                  double diag_min, const vsde_head_grads *g, void *workspace,
                                  size_t workspace_bytes, void *stream) {
    int rc = check_dims(d);
    if (rc) return rc;
    VSDE_CHECK_ARG(g_paths && g_means && g_chol && ctx && ctx->base && eps && paths && chol_raw && acts && w && g && workspace,
                   VSDE_E_BADARG, "NULL argument");
    VSDE_CHECK_ARG(g->x0 && g->context && g->W_ih_l0 && g->W_hh_l0 && g->b_ih_l0 && g->b_hh_l0 && g->out_weight && g->out_bias,
                   VSDE_E_BADARG, "NULL gradient output");
    VSDE_CHECK_ARG(d->P == 0 || (theta && g->theta), VSDE_E_BADARG, "theta / grad theta is NULL");
    VSDE_CHECK_ARG(d->L == 1 || (w->W_ih_stack && w->W_hh_stack && g->W_ih_stack && g->W_hh_stack && g->b_ih_stack && g->b_hh_stack),
                   VSDE_E_BADARG, "stacked layer tensors are NULL");
    const BwdLayout lay = bwd_layout(d);
    VSDE_CHECK_ARG(workspace_bytes >= lay.total, VSDE_E_WORKSPACE, "backward workspace too small: %zu < %zu", workspace_bytes, lay.total);
    hipStream_t s = (hipStream_t)stream;
    char *ws = (char *)workspace;
    const int NO = d->S + d->S * (d->S + 1) / 2, M = d->B * d->T;

    PackParams pk = {};
    pk.S = d->S; pk.P = d->P; pk.C = d->C; pk.H = d->H; pk.L = d->L; pk.NO = NO;
    pk.W_ih0 = w->W_ih_l0; pk.W_hh0 = w->W_hh_l0; pk.W_ih_st = w->W_ih_stack; pk.W_hh_st = w->W_hh_stack; pk.out_W = w->out_weight;
    if (!(d->H > kHP || NO > kWave)) pk.packB = (float4 *)(ws + lay.packB);
    pk.WcT = (float *)(ws + lay.WcT);
    prof_mark(3, 0, s);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(64), dim3(256), 0, s, pk);
    VSDE_CHECK_HIP(hipGetLastError());

    BwdParams p = {};
    p.B = d->B; p.T = d->T; p.S = d->S; p.P = d->P; p.C = d->C; p.H = d->H; p.NO = NO; p.ntril = NO - d->S;
    p.g_paths = g_paths; p.g_means = g_means; p.g_chol = g_chol; p.theta = theta; p.eps = eps; p.chol_raw = chol_raw; p.acts = acts;
    p.W_ih0 = w->W_ih_l0; p.out_W = w->out_weight; p.packB = pk.packB;
    p.W_hh0 = w->W_hh_l0; p.W_ih_st = w->W_ih_stack; p.W_hh_st = w->W_hh_stack;
    p.dt = (float)time_step; p.sqdt = (float)sqrt(time_step); p.diag_min = (float)diag_min;
    p.D4 = (float *)(ws + lay.D4); p.DO = (float *)(ws + lay.DO); p.g_x0 = g->x0; p.g_theta = g->theta;
    if (!g_force_v1 && use_mp_bwd(d)) {   // 4 / 8 paths per workgroup on the matrix cores
        MpBwdLaunch a = {};
        a.B = d->B; a.T = d->T; a.S = d->S; a.P = d->P; a.C = d->C;
        a.np = g_mp_mode >= 0 ? g_mp_mode : mp_env();
        a.g_paths = g_paths; a.g_means = g_means; a.g_chol = g_chol; a.eps = eps; a.chol_raw = chol_raw; a.acts = acts;
        a.W_ih0 = w->W_ih_l0; a.W_hh0 = w->W_hh_l0; a.W_ih_st = w->W_ih_stack; a.W_hh_st = w->W_hh_stack; a.out_W = w->out_weight;
        a.frags = ws + lay.frags;
        a.dt = p.dt; a.sqdt = p.sqdt; a.diag_min = p.diag_min;
        a.D4 = p.D4; a.DO = p.DO; a.g_x0 = p.g_x0; a.g_theta = p.g_theta;
        rc = launch_head_bwd_mp(a, s, prof_mark);
    } else if (d->H > kHP || NO > kWave) {  // generic kernels
        const int block = ((d->H > NO ? d->H : NO) + 63) / 64 * 64 > 1024 ? 1024 : ((d->H > NO ? d->H : NO) + 63) / 64 * 64;
        const size_t ldsw = (size_t)(4 * d->H + NO + d->S + d->S * d->H + 8) * sizeof(float);
        hipLaunchKernelGGL(head_bwd_wide_kernel, dim3(d->B), dim3(block), ldsw, s, p, d->L);
        VSDE_CHECK_HIP(hipGetLastError());
        rc = 0;
    } else if (d->L <= 2 && (d->H % 4) == 0 && (NO > 16 || d->S > kMaxSRegV2) && d->S <= 16 && !g_force_v1) {
        // wide variant: emission rows spread over the waves (NO <= 64 here), always with LDS-DMA activation chunks;
        // one workgroup per CU is acceptable when the staging does not fit in half the LDS
        p.wpb = 1;
        const int cand[3] = {16, 10, 8};
        int ch = 0;
        for (int pass = 0; pass < 2 && !ch; ++pass)
            for (int q = 0; q < 3 && !ch; ++q)
                if ((size_t)bwd_v2_lds(d->H, d->S, d->L, cand[q], true, true).total * sizeof(float) <= (pass == 0 ? 80u : 156u) * 1024) ch = cand[q];
        VSDE_CHECK_ARG(ch != 0, VSDE_E_STATE, "LDS budget exceeded for the wide backward kernel");
        const size_t lds2 = (size_t)bwd_v2_lds(d->H, d->S, d->L, ch, true, true).total * sizeof(float);
#define VSDE_LAUNCH_BWD_WIDE(LL, CC)                                                                                \
    do {                                                                                                            \
        VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)head_bwd_v2_kernel<LL, CC, 0, true, true>,                 \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));                  \
        prof_mark(1, 0, s);                                                                                         \
        hipLaunchKernelGGL((head_bwd_v2_kernel<LL, CC, 0, true, true>), dim3(d->B), dim3(256), lds2, s, p);         \
        prof_mark(1, 1, s);                                                                                         \
    } while (0)
#define VSDE_LAUNCH_BWD_WIDE_C(LL)                                                                                  \
    do {                                                                                                            \
        if (ch == 16) VSDE_LAUNCH_BWD_WIDE(LL, 16);                                                                 \
        else if (ch == 10) VSDE_LAUNCH_BWD_WIDE(LL, 10);                                                            \
        else VSDE_LAUNCH_BWD_WIDE(LL, 8);                                                                           \
    } while (0)
        if (d->L == 1) VSDE_LAUNCH_BWD_WIDE_C(1); else VSDE_LAUNCH_BWD_WIDE_C(2);
#undef VSDE_LAUNCH_BWD_WIDE_C
#undef VSDE_LAUNCH_BWD_WIDE
        VSDE_CHECK_HIP(hipGetLastError());
        rc = 0;
    } else if (d->L <= 2 && NO <= 16 && d->S <= kMaxSRegV2 && !g_force_v1) {
        p.wpb = 1;
        // DMA variant (16-byte aligned activation records): two activation buffers, chunk from {16, 10, 8}
        const bool dma = (d->H % 4) == 0;
        int ch = 16;
        if (dma) {
            const int cand[3] = {16, 10, 8};
            int pick = 0;
            for (int q = 0; q < 3 && !pick; ++q)
                if ((size_t)bwd_v2_lds(d->H, d->S, d->L, cand[q], true).total * sizeof(float) <= 80 * 1024) pick = cand[q];
            ch = pick;
        }
        if (dma && ch) {
            const size_t lds2 = (size_t)bwd_v2_lds(d->H, d->S, d->L, ch, true).total * sizeof(float);
#define VSDE_LAUNCH_BWD_DMA(LL, CC, SM)                                                                             \
    do {                                                                                                            \
        VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)head_bwd_v2_kernel<LL, CC, SM, true>,                      \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));                  \
        prof_mark(1, 0, s);                                                                                         \
        hipLaunchKernelGGL((head_bwd_v2_kernel<LL, CC, SM, true>), dim3(d->B), dim3(256), lds2, s, p);              \
        prof_mark(1, 1, s);                                                                                         \
    } while (0)
#define VSDE_LAUNCH_BWD_DMA_S(LL, CC)                                                                               \
    do {                                                                                                            \
        if (d->S == 1) VSDE_LAUNCH_BWD_DMA(LL, CC, 1);                                                              \
        else if (d->S == 2) VSDE_LAUNCH_BWD_DMA(LL, CC, 2);                                                         \
        else VSDE_LAUNCH_BWD_DMA(LL, CC, 0);                                                                        \
    } while (0)
#define VSDE_LAUNCH_BWD_DMA_C(LL)                                                                                   \
    do {                                                                                                            \
        if (ch == 16) VSDE_LAUNCH_BWD_DMA_S(LL, 16);                                                                \
        else if (ch == 10) VSDE_LAUNCH_BWD_DMA_S(LL, 10);                                                           \
        else VSDE_LAUNCH_BWD_DMA_S(LL, 8);                                                                          \
    } while (0)
            if (d->L == 1) VSDE_LAUNCH_BWD_DMA_C(1); else VSDE_LAUNCH_BWD_DMA_C(2);
#undef VSDE_LAUNCH_BWD_DMA_C
#undef VSDE_LAUNCH_BWD_DMA_S
#undef VSDE_LAUNCH_BWD_DMA
        } else {
        ch = 16;
        while (ch > 4 && (size_t)bwd_v2_lds(d->H, d->S, d->L, ch).total * sizeof(float) > 80 * 1024) --ch;
        const size_t lds2 = (size_t)bwd_v2_lds(d->H, d->S, d->L, ch).total * sizeof(float);
#define VSDE_LAUNCH_BWD_V2(LL, CC)                                                                                  \
    do {                                                                                                            \
        VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)head_bwd_v2_kernel<LL, CC, 0, false>,                      \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));                  \
        prof_mark(1, 0, s);                                                                                         \
        hipLaunchKernelGGL((head_bwd_v2_kernel<LL, CC, 0, false>), dim3(d->B), dim3(256), lds2, s, p);              \
        prof_mark(1, 1, s);                                                                                         \
    } while (0)
#define VSDE_LAUNCH_BWD_V2_L(LL)                                                                                    \
    do {                                                                                                            \
        switch (ch) {                                                                                               \
            case 16: VSDE_LAUNCH_BWD_V2(LL, 16); break;                                                             \
            case 15: VSDE_LAUNCH_BWD_V2(LL, 15); break;                                                             \
            case 14: VSDE_LAUNCH_BWD_V2(LL, 14); break;                                                             \
            case 13: VSDE_LAUNCH_BWD_V2(LL, 13); break;                                                             \
            case 12: VSDE_LAUNCH_BWD_V2(LL, 12); break;                                                             \
            default: ch = 8; VSDE_LAUNCH_BWD_V2(LL, 8); break;                                                      \
        }                                                                                                           \
    } while (0)
        if (ch < 12) ch = 8;
        if (d->L == 1) VSDE_LAUNCH_BWD_V2_L(1); else VSDE_LAUNCH_BWD_V2_L(2);
#undef VSDE_LAUNCH_BWD_V2_L
#undef VSDE_LAUNCH_BWD_V2
        }
        VSDE_CHECK_HIP(hipGetLastError());
        rc = 0;
    } else {
    const size_t lds_fixed = (size_t)fwd_lds_matrices(d->L) * kMatF4 * sizeof(float4) + (size_t)NO * kHP * sizeof(float);
    const size_t lds_wave = kBwdScratchPerWave * sizeof(float);
    p.wpb = pick_wpb(d->B, lds_fixed, lds_wave);
    const size_t lds = lds_fixed + p.wpb * lds_wave;
    VSDE_CHECK_ARG(lds <= 160 * 1024, VSDE_E_STATE, "LDS budget exceeded (%zu B)", lds);
    const int grid = (d->B + p.wpb - 1) / p.wpb, block = 64 * p.wpb;
    switch (d->L) {
        case 1: rc = launch_bwd_L<1>(p, grid, block, lds, s); break;
        case 2: rc = launch_bwd_L<2>(p, grid, block, lds, s); break;
        case 3: rc = launch_bwd_L<3>(p, grid, block, lds, s); break;
        default: rc = launch_bwd_L<4>(p, grid, block, lds, s); break;
    }
    }
    if (rc) return rc;

    // grad_context[m][c] = sum_n d_pre0(m, n) W_ih_l0[n][S + c]   (backward.py:550-564)
    RowView dv; dv.base = p.D4; dv.batch_stride = (int64_t)d->T * d->L * 4 * d->H; dv.row_stride = (int64_t)d->L * 4 * d->H;
    dv.rows_per_batch = d->T; dv.shift = 0; dv.col_split = 3 * d->H; dv.col_skip = 0; dv.dtype = 0;
    VSDE_CHECK_ARG(g->context_dtype == 0 || g->context_dtype == 1, VSDE_E_BADARG, "grads->context_dtype must be 0 (f32) or 1 (bf16)");
    VSDE_CHECK_ARG(g->context_batch_stride == 0 || g->context_batch_stride >= (int64_t)d->T * d->C, VSDE_E_BADARG,
                   "grads->context_batch_stride smaller than T*C");
    prof_mark(5, 0, s);
    // the grouped weight-gradient reduction below runs after this GEMM on the same stream: its workspace doubles as plane scratch
    rc = (proj_fast_path() && g->context_dtype == 1)
             ? launch_proj_bwd_bf16(dv, M, 3 * d->H, pk.WcT, 3 * d->H, d->C, g->context, d->C, g->context_batch_stride ? d->T : 0,
                                    g->context_batch_stride, ws + lay.tn, lay.total - lay.tn, s) : 0;
    if (rc < 0) return rc;
    if (rc == 0) rc = launch_gemm_nt(dv, M, 3 * d->H, pk.WcT, 3 * d->H, d->C, nullptr, g->context, d->C, s,
                                     g->context_batch_stride ? d->T : 0, g->context_batch_stride, g->context_dtype);
    else rc = 0;
    prof_mark(5, 1, s);
    if (rc) return rc;

    TnProblem pr[kMaxTnProblems];
    int n = build_tn(d, ctx, theta, paths, acts, p.D4, p.DO, g, pr);
    prof_mark(6, 0, s);
    rc = launch_tn_grouped(pr, n, M, ws + lay.tn, lay.total - lay.tn, s);
    prof_mark(6, 1, s);
    prof_mark(3, 1, s);
    return rc;
}
