// Multi-path GRU path sampler on the matrix cores (round 4): 16 sample paths per workgroup, split-precision f16 MFMA.
//
// Reference semantics: kernels/forward.py:137-375 (the same time loop as head_fwd_v2_kernel in vsde_head.hip).
//
// Why a second forward kernel.  head_fwd_v2_kernel gives one sample path four wavefronts and multiplies on the VALU: with
// 2 paths per CU every time step is a ~2,700-cycle dependency chain (LDS round trips, barriers, 276 VALU issues) and the
// design saturates at 512 paths per GPU -- from there on time doubles with the batch (1.0 M sampled paths/s at any batch,
// 0.027 of the HBM roofline for the no-grad sampling launch).  The recurrent products W h are tiny matrix-vector products per
// path; over a GROUP of paths they are GEMMs with N = paths, which is what the matrix pipe wants:
//
//   * workgroup = 16 paths; 4 waves per GRU layer (roles: see the kernel).  Wave w of a layer owns the hidden units 16 w .. 16 w + 15 of every gate; its three A tiles of a
//     recurrent matrix are the r / u / n rows of those units, the B operand is h^T [64 x 16 paths].  v_mfma_f32_16x16x32_f16
//     leaves lane (q = lane >> 4, p = lane & 15) with D[rows 4 q .. 4 q + 3][column p]: the r, u and n pre-activations of FOUR
//     units of ONE path in one lane -- the whole gate algebra is lane-local, no cross-lane traffic.
//   * fp32-equivalent results from f16 operands: every operand is split x = hi + lo / 2048 with hi = f16(x), lo = f16((x - hi)
//     * 2048) (22 mantissa bits; the 2^11 scale keeps lo out of the f16 subnormals, which the matrix pipe flushes) and a product is three MFMAs,
//     hi*hi into one accumulator, hi*lo + lo*hi into a second one that is folded in with one v_fma (x 2^-11); the dropped
//     lo*lo term is 2^-22 relative.  Accumulation is fp32.  Gate rows are pre-scaled into the exp2 domain like the v2 kernel's.
//   * all recurrent weights live in VGPRs as ready-made A fragments (built once per launch by mp_prep_kernel): 48 VGPRs per
//     matrix, 144 + 32 (emission rows) for two layers; the time loop issues no weight loads.
//   * the hidden state crosses the four waves as f16 hi / lo planes through 4 KB of LDS per layer, stored by its owners in
//     B-fragment order (each lane one ds_write_b64 per plane, each reader one conflict-free ds_read_b128 per k-step and plane),
//     one workgroup barrier per layer.
//   * the emission rows are replicated four times down their A tile (tile row i holds out_proj row 4 tile + (i & 3)), so EVERY
//     lane ends up with all S + S(S+1)/2 emission values of its path: the Euler-Maruyama update and the state-input term of the
//     next step are lane-local as well (z_t is fp32 on the VALU: it is not bounded like h).
//   * the products a step does not need at once -- W_hh h_t of both layers, used by step t + 1 -- are issued behind the ones
//     on the critical path, so they run on the matrix pipe while the VALU does the next gate block.
//   * global traffic: the projected context record G[b, t, 3H] is read straight into registers one step ahead (16 B per lane
//     and gate), saved activations leave as 16-byte lanes (four waves complete each 256-byte row), outputs from 16 lanes each.
//
// Scope: hidden_dim 64, 1 or 2 layers, state_dim 1 or 2 (the OU and Lotka-Volterra heads); everything else keeps the v2 / v1 /
// generic kernels.  Weights must fit f16 range after the exp2 scaling (|W| < 2.2e4; a GRU gate saturates long before).
#include "vsde_common.h"

namespace vsde {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr float kMpLo = 2048.0f, kMpLoInv = 1.0f / 2048.0f;
constexpr float kMpSr = -1.4426950408889634f, kMpSn = 2.8853900817779268f, kMpInvSn = 1.0f / 2.8853900817779268f;
constexpr int kMpMatFrags = 4 * 3 * 2 * 2 * 64;   // f16x8 fragments of one recurrent matrix: [wave][gate][k-step][plane][lane]

// x = hi + lo / 2048.  The matrix pipe flushes f16 DENORMAL inputs to zero (measured: rows of W holding an element below
// 2^-14 lost it entirely, 2e-5 .. 7e-5 on their dot products), so a value whose hi would be subnormal goes into lo alone
// (lo = 2048 x is normal down to |x| = 2^-25; below that the flush costs < 3e-8).
__device__ __forceinline__ void mp_split(float v, _Float16 &hi, _Float16 &lo) {
    const float h = fabsf(v) < 6.103515625e-5f ? 0.0f : (float)(_Float16)v;
    hi = (_Float16)h;
    lo = (_Float16)((v - h) * kMpLo);
}

// ---------------------------------------------------------------------------------------------------------------------------
// A fragments of the recurrent matrices and the emission rows, in the register order of the main kernel.
// k-slot (lane group q, element e) of k-step ks stands for reduction index k = 32 ks + 8 q + e on BOTH operands (the hardware
// contracts A slot with B slot; any consistent labelling sums over all k).
struct MpPrep {
    int nm, no, nto;
    const float *W[3];     // W_hh_l0, W_ih_l1, W_hh_l1  ([192][64], nn.GRU layout)
    const float *out_W;    // [no][64]
    f16x8 *frags;
};

__global__ void __launch_bounds__(256) mp_prep_kernel(MpPrep q) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int nmat = q.nm * 1536;                     // (wave, gate, k-step, lane) tuples per matrix
    if (i < nmat) {
        const int m = i / 1536, r = i - m * 1536;
        const int lane = r & 63, ks = (r >> 6) & 1, wg = r >> 7, g = wg % 3, w = wg / 3;
        const int row = g * 64 + 16 * w + (lane & 15), k0 = 32 * ks + 8 * (lane >> 4);
        const float sc = g < 2 ? kMpSr : kMpSn;
        const float *W = q.W[m] + (int64_t)row * 64 + k0;
        f16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) { _Float16 a, b; mp_split(sc * W[e], a, b); hi[e] = a; lo[e] = b; }
        f16x8 *dst = q.frags + (int64_t)m * kMpMatFrags + (((w * 3 + g) * 2 + ks) * 2) * 64 + lane;
        dst[0] = hi; dst[64] = lo;
    } else if (i - nmat < q.nto * 128) {
        const int r = i - nmat, lane = r & 63, ks = (r >> 6) & 1, tl = r >> 7;
        const int orow = 4 * tl + (lane & 3), k0 = 32 * ks + 8 * (lane >> 4);
        f16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            _Float16 a, b;
            mp_split(orow < q.no ? q.out_W[(int64_t)orow * 64 + k0 + e] : 0.0f, a, b);
            hi[e] = a; lo[e] = b;
        }
        f16x8 *dst = q.frags + (int64_t)q.nm * kMpMatFrags + ((tl * 2 + ks) * 2) * 64 + lane;
        dst[0] = hi; dst[64] = lo;
    }
}

struct MpParams {
    int B, T, P, C;
    const float *x0, *theta, *eps, *G;
    const float *W_ih0;                          // [192][S + C + P]: state and theta columns are read here
    const float *b_hh0, *b_ih1, *b_hh1, *out_b;
    const f16x8 *frags;
    float dt, sqdt, diag_min;
    float *paths, *means, *chol, *chol_raw, *acts;
};

__device__ __forceinline__ f32x4 mp_mfma(const f16x8 &a, const f16x8 &b, const f32x4 &c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

template <int N> __device__ __forceinline__ float mp_row_shl(float v) {   // lane i <- lane i + N of its 16-lane row
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x100 + N, 0xf, 0xf, true));
}

// [W (three gate tiles of this wave's units)] x [h^T of the group's paths] -> R (fp32-equivalent), three MFMA products per tile:
//   NP == 16: the B operand has 16 path columns per plane: hi*hi -> A1, hi*lo' + lo'*hi -> A2, R = A1 + A2 / 2048;
//   NP < 16 : ONE B operand holds both planes side by side -- columns [0, NP) the hi parts of the NP paths, [NP, 2 NP) their lo'
//             parts -- so W_hi x B gives hi*hi (column c) AND hi*lo' (column c + NP) in one MFMA and W_lo' x B adds lo'*hi: two
//             MFMAs per tile and k-step instead of three; lane c pulls the hi*lo' sum from lane c + NP of its row (one DPP op).
template <int NP>
__device__ __forceinline__ void mp_matmul(const f16x8 (&wf)[3][2][2], const f16x8 (&hb)[2][2], f32x4 (&R)[3]) {
    f32x4 A1[3], A2[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) { A1[g] = f32x4{0.f, 0.f, 0.f, 0.f}; A2[g] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int g = 0; g < 3; ++g) A1[g] = mp_mfma(wf[g][ks][0], hb[0][ks], A1[g]);
        if (NP == 16) {
#pragma unroll
            for (int g = 0; g < 3; ++g) A2[g] = mp_mfma(wf[g][ks][0], hb[1][ks], A2[g]);
        }
#pragma unroll
        for (int g = 0; g < 3; ++g) A2[g] = mp_mfma(wf[g][ks][1], hb[0][ks], A2[g]);
    }
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            R[g][r] = NP == 16 ? fmaf(A2[g][r], kMpLoInv, A1[g][r]) : fmaf(mp_row_shl<NP & 15>(A1[g][r]) + A2[g][r], kMpLoInv, A1[g][r]);
}

// Roles.  One layer: four waves (wave w = units 16 w ..).  Two layers: EIGHT waves, two per SIMD -- waves 0-3 are layer 0 (W_hh_l0
// and the emission rows in registers, the Euler-Maruyama update, the context record), waves 4-7 are layer 1 (W_ih_l1, W_hh_l1).
// With all three matrices in one wave the kernel needed 432 registers and hipcc moved ~180 values per step between the
// accumulator and the vector half of the file; split by layer each role stays below 256 registers, and the products a step
// does not need at once (W_hh h_t, consumed by step t + 1) run on one role's matrix pipe while the other role's VALU does gates.
//   step t:  [L0: gates -> h0_t]  barrier A  [L1: W_ih1 h0_t, gates -> h1_t | L0: W_hh0 h0_t]  barrier B
//            [L0: emission rows x h1_t, z_{t+1}, outputs, then gates of step t + 1 | L1: W_hh1 h1_t]
template <int L, bool SAVE, int S, int NP>
__global__ void __launch_bounds__(256 * L, 3 - L) head_fwd_mp_kernel(MpParams p) {
    constexpr int NTRIL = S * (S + 1) / 2, NO = S + NTRIL, NTO = (NO + 3) / 4, PL = NP == 16 ? 2 : 1;
    static_assert(L >= 1 && L <= 2 && S >= 1 && S <= 2 && (NP == 16 || NP == 8 || NP == 4), "multi-path kernel: L <= 2, state_dim <= 2");
    // hidden state in B-fragment order: [step parity][layer][plane][k-step][lane group][column] x 8 f16 (NP < 16: one plane, the hi parts
    // in columns [0, NP), the lo' parts in [NP, 2 NP), the rest zero)
    __shared__ __attribute__((aligned(16))) f16x8 hbuf[2][L][PL][2][4][16];
    const int tid = threadIdx.x, wv = tid >> 6, role = wv >> 2, w = wv & 3, lane = tid & 63, q = lane >> 4, pp = lane & 15;
    const int b_raw = blockIdx.x * NP + (pp & (NP - 1));
    const bool owner = pp < NP;                    // NP < 16: lanes of the other columns run along and store nothing
    const bool live = owner && b_raw < p.B;
    const int b = b_raw < p.B ? b_raw : p.B - 1;   // lanes beyond the batch recompute the last path and store nothing
    const int j0 = 16 * w + 4 * q, T = p.T, I = S + p.C + p.P;
    if (NP < 16) {
        for (int e = tid; e < 2 * L * PL * 2 * 4 * 16; e += 256 * L) (&hbuf[0][0][0][0][0][0])[e] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
        __syncthreads();
    }

    // the 4 owned units' new state -> f16 hi / lo' in B-fragment order (unit j = 16 w + 4 q + r -> k-step j >> 5, lane group (j >> 3) & 3,
    // element j & 7)
    auto publish = [&](const float (&h)[4], int t, int l) {
        f16x4 hi, lo;
#pragma unroll
        for (int r = 0; r < 4; ++r) { _Float16 a, c; mp_split(h[r], a, c); hi[r] = a; lo[r] = c; }
        const int par = t & 1;
        if (NP == 16) {
            *((f16x4 *)&hbuf[par][l][0][w >> 1][2 * (w & 1) + (q >> 1)][pp] + (q & 1)) = hi;
            *((f16x4 *)&hbuf[par][l][PL - 1][w >> 1][2 * (w & 1) + (q >> 1)][pp] + (q & 1)) = lo;
        } else if (owner) {
            *((f16x4 *)&hbuf[par][l][0][w >> 1][2 * (w & 1) + (q >> 1)][pp] + (q & 1)) = hi;
            *((f16x4 *)&hbuf[par][l][0][w >> 1][2 * (w & 1) + (q >> 1)][pp + (NP & 15)] + (q & 1)) = lo;
        }
    };
    // saved activations acts[b][t][l][{h, r, z, n, n_hh}][64] (kernels/weights.py:11-23): 16-byte lanes, issued BEHIND the barrier that
    // publishes the state -- in the slack where this role waits for the other one, not on the step's critical path
    auto save_acts = [&](const float (&h)[4], const float (&rg)[4], const float (&ug)[4], const float (&ng)[4], const float (&cn)[4],
                         int t, int l) {
        if (SAVE && live) {
            float *ab = p.acts + (((int64_t)b * T + t) * L + l) * 320 + j0;
            *(f32x4 *)(ab) = f32x4{h[0], h[1], h[2], h[3]};
            *(f32x4 *)(ab + 64) = f32x4{rg[0], rg[1], rg[2], rg[3]};
            *(f32x4 *)(ab + 128) = f32x4{ug[0], ug[1], ug[2], ug[3]};
            *(f32x4 *)(ab + 192) = f32x4{ng[0], ng[1], ng[2], ng[3]};
            *(f32x4 *)(ab + 256) = f32x4{cn[0] * kMpInvSn, cn[1] * kMpInvSn, cn[2] * kMpInvSn, cn[3] * kMpInvSn};
        }
    };
    // gate block of one layer for the 4 owned units (exp2 domain: r = 1 / (1 + 2^x_r), n = 1 - 2 / (1 + 2^(a_n + r c_n)))
    auto gates = [&](const float (&ar)[4], const float (&au)[4], const float (&an)[4], const f32x4 (&c)[3],
                     const float (&bn)[4], float (&h)[4], float (&rg)[4], float (&ug)[4], float (&ng)[4], float (&cn)[4], int t, int l) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            cn[r] = bn[r] + c[2][r];
            rg[r] = fast_rcp(1.0f + fast_exp2(ar[r] + c[0][r]));
            ug[r] = fast_rcp(1.0f + fast_exp2(au[r] + c[1][r]));
            ng[r] = fmaf(-2.0f, fast_rcp(1.0f + fast_exp2(fmaf(rg[r], cn[r], an[r]))), 1.0f);
            h[r] = fmaf(ug[r], h[r] - ng[r], ng[r]);           // (1 - u) n + u h
        }
        publish(h, t, l);
    };
    auto barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto read_state = [&](int t, int l, f16x8 (&hb)[2][2]) {
        const int par = t & 1;
#pragma unroll
        for (int pl = 0; pl < PL; ++pl)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) hb[pl][ks] = hbuf[par][l][pl][ks][q][pp];
    };
    auto load_matrix = [&](int m, f16x8 (&wf)[3][2][2]) {
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    wf[g][ks][pl] = p.frags[(int64_t)m * kMpMatFrags + ((((w * 3 + g) * 2 + ks) * 2) + pl) * 64 + lane];
    };

    if (role == 1) {
        // =================================================================== layer-1 waves (two-layer heads only)
        if (L > 1) {
            f16x8 wi[3][2][2], wh[3][2][2];
            load_matrix(1, wi);
            load_matrix(2, wh);
            float k1[3][4], bn1[4], h1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = g * 64 + j0 + r;
                    const float sc = g < 2 ? kMpSr : kMpSn;
                    k1[g][r] = sc * p.b_ih1[row] + (g < 2 ? sc * p.b_hh1[row] : 0.f);
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) bn1[r] = kMpSn * p.b_hh1[128 + j0 + r];
            f32x4 c1[3];                               // W_hh^1 h^1_{t-1}: h_{-1} = 0
#pragma unroll
            for (int g = 0; g < 3; ++g) c1[g] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int t = 0; t < T; ++t) {
                barrier();                             // A: h^0_t published
                f16x8 hb[2][2];
                read_state(t, 0, hb);
                f32x4 a1[3];
                mp_matmul<NP>(wi, hb, a1);             // W_ih^1 h^0_t
                float ar[4], au[4], an[4], rg[4], ug[4], ng[4], cn[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { ar[r] = k1[0][r] + a1[0][r]; au[r] = k1[1][r] + a1[1][r]; an[r] = k1[2][r] + a1[2][r]; }
                gates(ar, au, an, c1, bn1, h1, rg, ug, ng, cn, t, L - 1);
                barrier();                             // B: h^1_t published
                read_state(t, L - 1, hb);
                mp_matmul<NP>(wh, hb, c1);             // W_hh^1 h^1_t: consumed by step t + 1
                save_acts(h1, rg, ug, ng, cn, t, L - 1);
            }
        }
        return;
    }

    // ======================================================================= layer-0 waves
    f16x8 wf[3][2][2], of[NTO][2][2];
    load_matrix(0, wf);
#pragma unroll
    for (int tl = 0; tl < NTO; ++tl)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
                of[tl][ks][pl] = p.frags[(int64_t)(2 * L - 1) * kMpMatFrags + (((tl * 2 + ks) * 2) + pl) * 64 + lane];

    // per-lane constants of the 4 owned units (exp2 domain): biases, hoisted theta projection, state columns of W_ih_l0
    float k0[3][4], bn0[4], wx[S][3][4], ob[NO];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = g * 64 + j0 + r;
            const float sc = g < 2 ? kMpSr : kMpSn;
            float th = 0.f;   // theta term (forward.py:157-175)
            for (int e = 0; e < p.P; ++e) th = fmaf(p.theta[(int64_t)b * p.P + e], p.W_ih0[(int64_t)row * I + S + p.C + e], th);
            k0[g][r] = sc * th + (g < 2 ? sc * p.b_hh0[row] : 0.f);
#pragma unroll
            for (int i = 0; i < S; ++i) wx[i][g][r] = sc * p.W_ih0[(int64_t)row * I + i];
        }
#pragma unroll
    for (int r = 0; r < 4; ++r) bn0[r] = kMpSn * p.b_hh0[128 + j0 + r];
#pragma unroll
    for (int r = 0; r < NO; ++r) ob[r] = p.out_b[r];

    float z[S], h0[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4 c0[3];                                       // W_hh^0 h^0_{t-1}: h_{-1} = 0
#pragma unroll
    for (int g = 0; g < 3; ++g) c0[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < S; ++i) z[i] = p.x0[(int64_t)b * S + i];
    if (w == 0 && q == 0 && live) {
#pragma unroll
        for (int i = 0; i < S; ++i) p.paths[(int64_t)b * (T + 1) * S + i] = z[i];
    }

    // the projected context record and eps, one step ahead in registers (the wait for them at the top of a step is a vmcnt(0))
    const float *Gb = p.G + (int64_t)b * T * 192 + j0;
    const float *eb = p.eps + (int64_t)b * T * S;
    f32x4 gq[3];
    float ev[S];
    auto fetch = [&](int t) {
        const int tc = t < T ? t : T - 1;
#pragma unroll
        for (int g = 0; g < 3; ++g) gq[g] = *(const f32x4 *)(Gb + (int64_t)tc * 192 + g * 64);
#pragma unroll
        for (int i = 0; i < S; ++i) ev[i] = eb[(int64_t)tc * S + i];
    };
    fetch(0);

    // Outputs of step t are stored during step t + 1, in the slack behind barrier A (where the layer-0 waves wait for layer 1):
    // hipcc's wait for the prefetched context record at the top of the loop is a vmcnt(0) (stores sit in conditional blocks it
    // cannot count), and a store issued just before it would put a full store round trip on every step.
    float pz[S], pmu[S], pL[S][S], praw[NTRIL];
    auto store_outputs = [&](int t) {     // paths[b, t + 1], means[b, t], chol[b, t], chol_raw[b, t]: one wave each
        if (q == 0 && live) {
            const int64_t bt = (int64_t)b * T + t;
            if (w == 0) {
#pragma unroll
                for (int i = 0; i < S; ++i) p.paths[(bt + b + 1) * S + i] = pz[i];
            } else if (w == 1) {
#pragma unroll
                for (int i = 0; i < S; ++i) p.means[bt * S + i] = pmu[i];
            } else if (w == 2) {
#pragma unroll
                for (int i = 0; i < S; ++i)
#pragma unroll
                    for (int c = 0; c < S; ++c) p.chol[bt * S * S + i * S + c] = pL[i][c];
            } else if (SAVE) {
#pragma unroll
                for (int r = 0; r < NTRIL; ++r) p.chol_raw[bt * NTRIL + r] = praw[r];
            }
        }
    };

    for (int t = 0; t < T; ++t) {
        f32x4 g0 = gq[0], g1 = gq[1], g2 = gq[2];
        float e[S];
#pragma unroll
        for (int i = 0; i < S; ++i) e[i] = ev[i];
        fetch(t + 1);
        // ---- layer 0: a = G_t (context projection + b_ih) + theta term + z_t W_x   (forward.py:195-219)
        float ar[4], au[4], an[4], rg[4], ug[4], ng[4], cn[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ar[r] = fmaf(g0[r], kMpSr, k0[0][r]); au[r] = fmaf(g1[r], kMpSr, k0[1][r]); an[r] = fmaf(g2[r], kMpSn, k0[2][r]);
#pragma unroll
            for (int i = 0; i < S; ++i) {
                ar[r] = fmaf(z[i], wx[i][0][r], ar[r]); au[r] = fmaf(z[i], wx[i][1][r], au[r]); an[r] = fmaf(z[i], wx[i][2][r], an[r]);
            }
        }
        gates(ar, au, an, c0, bn0, h0, rg, ug, ng, cn, t, 0);
        barrier();                                     // A: h^0_t published
        f16x8 hb[2][2];
        read_state(t, 0, hb);
        if (L > 1) {
            mp_matmul<NP>(wf, hb, c0);                 // W_hh^0 h^0_t: consumed by step t + 1, runs beside layer 1's gates
            save_acts(h0, rg, ug, ng, cn, t, 0);
            if (t > 0) store_outputs(t - 1);
            __builtin_amdgcn_sched_barrier(0);         // (register-only MFMAs are not ordered by the barrier's "memory" clobber)
            barrier();                                 // B: h^1_t published
            read_state(t, L - 1, hb);
        }
        // ---- emission (forward.py:314-375): every owner lane ends up with all NO values of its path
        f32x4 O1[NTO], O2[NTO];
#pragma unroll
        for (int tl = 0; tl < NTO; ++tl) { O1[tl] = f32x4{0.f, 0.f, 0.f, 0.f}; O2[tl] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int tl = 0; tl < NTO; ++tl) {
                O1[tl] = mp_mfma(of[tl][ks][0], hb[0][ks], O1[tl]);
                if (NP == 16) O2[tl] = mp_mfma(of[tl][ks][0], hb[PL - 1][ks], O2[tl]);
                O2[tl] = mp_mfma(of[tl][ks][1], hb[0][ks], O2[tl]);
            }
        if (L == 1) {
            mp_matmul<NP>(wf, hb, c0);
            save_acts(h0, rg, ug, ng, cn, t, 0);
            if (t > 0) store_outputs(t - 1);
        }
        float o[NO];
#pragma unroll
        for (int r = 0; r < NO; ++r) {
            const float a1 = O1[r >> 2][r & 3], a2 = O2[r >> 2][r & 3];
            o[r] = ob[r] + (NP == 16 ? fmaf(a2, kMpLoInv, a1) : fmaf(mp_row_shl<NP & 15>(a1) + a2, kMpLoInv, a1));
        }
        float mu[S], Lc[S][S];
#pragma unroll
        for (int i = 0; i < S; ++i) {
            mu[i] = o[i];
#pragma unroll
            for (int c = 0; c < S; ++c) {
                if (c > i) { Lc[i][c] = 0.f; continue; }
                const float v = o[S + i * (i + 1) / 2 + c];
                Lc[i][c] = (c == i && v < p.diag_min) ? p.diag_min : v;   // NaN propagates (torch.max semantics)
            }
        }
#pragma unroll
        for (int i = 0; i < S; ++i) {
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c <= i; ++c) acc = fmaf(Lc[i][c], e[c], acc);
            z[i] = z[i] + mu[i] * p.dt + acc * p.sqdt;
        }
        // outputs leave one step late (store_outputs): kept in registers until the next step's barrier A
#pragma unroll
        for (int i = 0; i < S; ++i) { pz[i] = z[i]; pmu[i] = mu[i]; }
#pragma unroll
        for (int i = 0; i < S; ++i)
#pragma unroll
            for (int c = 0; c < S; ++c) pL[i][c] = Lc[i][c];
#pragma unroll
        for (int r = 0; r < NTRIL; ++r) praw[r] = o[S + r];
    }
    store_outputs(T - 1);
}

// ---------------------------------------------------------------------------------------------------------------------------
size_t mp_frag_bytes(int L, int S) {
    const int no = S + S * (S + 1) / 2, nto = (no + 3) / 4;
    return ((size_t)(2 * L - 1) * kMpMatFrags + (size_t)nto * 2 * 2 * 64) * sizeof(f16x8);
}

bool mp_applicable(int H, int L, int S) { return H == 64 && L >= 1 && L <= 2 && S >= 1 && S <= 2; }

int launch_head_fwd_mp(const MpLaunch &a, hipStream_t s, void (*mark)(int, int, hipStream_t)) {
    const int no = a.S + a.S * (a.S + 1) / 2, nto = (no + 3) / 4, nm = 2 * a.L - 1;
    MpPrep q = {};
    q.nm = nm; q.no = no; q.nto = nto;
    q.W[0] = a.W_hh0; q.W[1] = a.W_ih_st; q.W[2] = a.W_hh_st; q.out_W = a.out_W;
    q.frags = (f16x8 *)a.frags;
    const int nthr = nm * 1536 + nto * 128;
    hipLaunchKernelGGL(mp_prep_kernel, dim3((nthr + 255) / 256), dim3(256), 0, s, q);
    MpParams p = {};
    p.B = a.B; p.T = a.T; p.P = a.P; p.C = a.C;
    p.x0 = a.x0; p.theta = a.theta; p.eps = a.eps; p.G = a.G; p.W_ih0 = a.W_ih0;
    p.b_hh0 = a.b_hh0; p.b_ih1 = a.b_ih_st; p.b_hh1 = a.b_hh_st; p.out_b = a.out_b;
    p.frags = (const f16x8 *)a.frags;
    p.dt = a.dt; p.sqdt = a.sqdt; p.diag_min = a.diag_min;
    p.paths = a.paths; p.means = a.means; p.chol = a.chol; p.chol_raw = a.chol_raw; p.acts = a.acts;
    // paths per workgroup: 16 fills the matrix pipe (large batches); a small batch takes 4 or 8 so that its groups spread over more CUs --
    // the time of a launch is T x one step's latency whatever the group size, and ONE CU's vector-memory pipe would have to carry the
    // saved activations of all its paths (41 KB per step for 16 paths: +30 % at 512 paths, profiles/r04_head_mp.txt)
    int np = a.np;
    if (np != 4 && np != 8 && np != 16) np = a.B <= 1024 ? 4 : (a.B <= 2048 ? 8 : 16);
    const dim3 grid((a.B + np - 1) / np), block(256 * a.L);
    if (mark) mark(0, 0, s);
#define VSDE_MP_LAUNCH_(LL, SV, SS, NN) hipLaunchKernelGGL((head_fwd_mp_kernel<LL, SV, SS, NN>), grid, block, 0, s, p)
#define VSDE_MP_LAUNCH(LL, SV, SS)                                  \
    do {                                                            \
        if (np == 4) VSDE_MP_LAUNCH_(LL, SV, SS, 4);                \
        else if (np == 8) VSDE_MP_LAUNCH_(LL, SV, SS, 8);           \
        else VSDE_MP_LAUNCH_(LL, SV, SS, 16);                       \
    } while (0)
    if (a.L == 1) {
        if (a.save) { if (a.S == 1) VSDE_MP_LAUNCH(1, true, 1); else VSDE_MP_LAUNCH(1, true, 2); }
        else { if (a.S == 1) VSDE_MP_LAUNCH(1, false, 1); else VSDE_MP_LAUNCH(1, false, 2); }
    } else {
        if (a.save) { if (a.S == 1) VSDE_MP_LAUNCH(2, true, 1); else VSDE_MP_LAUNCH(2, true, 2); }
        else { if (a.S == 1) VSDE_MP_LAUNCH(2, false, 1); else VSDE_MP_LAUNCH(2, false, 2); }
    }
#undef VSDE_MP_LAUNCH
#undef VSDE_MP_LAUNCH_
    if (mark) mark(0, 1, s);
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace vsde
