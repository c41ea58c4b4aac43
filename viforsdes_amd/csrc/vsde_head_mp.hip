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
#include <stdlib.h>

#include "vsde_common.h"

namespace vsde {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr float kMpLo = 2048.0f, kMpLoInv = 1.0f / 2048.0f;
constexpr float kMpSr = -1.4426950408889634f, kMpSn = 2.8853900817779268f, kMpInvSn = 1.0f / 2.8853900817779268f;
template <int V> struct MpSlot { static constexpr int value = V; };
constexpr int kMpMatFrags = 4 * 3 * 2 * 2 * 64;   // f16x8 fragments of one recurrent matrix: [wave][gate][k-step][plane][lane]

// x = hi + lo / 2048.  The matrix pipe flushes f16 DENORMAL inputs to zero (measured: rows of W holding an element below
// 2^-14 lost it entirely, 2e-5 .. 7e-5 on their dot products), so a value whose hi would be subnormal goes into lo alone
// (lo = 2048 x is normal down to |x| = 2^-25; below that the flush costs < 3e-8).
__device__ __forceinline__ void mp_split(float v, _Float16 &hi, _Float16 &lo) {
    const float h = fabsf(v) < 6.103515625e-5f ? 0.0f : (float)(_Float16)v;
    hi = (_Float16)h;
    lo = (_Float16)((v - h) * kMpLo);
}
// The time loops run with the f16 denormal mode of the wave set to FLUSH (mp_flush_f16_denormals, MODE.fp_denorm[3:2] = 0): the
// conversions then produce the zero the matrix pipe would see anyway, and the split needs no compare / select (7 -> 4.5 VALU
// operations per value; the backward splits 16 values per lane and layer on its critical path).
__device__ __forceinline__ void mp_flush_f16_denormals() {
    __builtin_amdgcn_s_setreg(1 | (6 << 6) | (1 << 11), 0);   // hwreg(HW_REG_MODE, offset 6, 2 bits): f16 / f64 denormals -> flush
}
__device__ __forceinline__ void mp_split_fast(float v, _Float16 &hi, _Float16 &lo) {
    hi = (_Float16)v;
    lo = (_Float16)((v - (float)hi) * kMpLo);
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
    int *overflow;         // host-mapped sticky flag: a scaled weight left the f16 range (see mp_weights_overflowed)
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
        bool big = false;
#pragma unroll
        for (int e = 0; e < 8; ++e) { _Float16 a, b; mp_split(sc * W[e], a, b); hi[e] = a; lo[e] = b; big = big || !(fabsf(sc * W[e]) < 65504.0f); }
        if (big && q.overflow) *q.overflow = 1;
        f16x8 *dst = q.frags + (int64_t)m * kMpMatFrags + (((w * 3 + g) * 2 + ks) * 2) * 64 + lane;
        dst[0] = hi; dst[64] = lo;
    } else if (i - nmat < q.nto * 128) {
        const int r = i - nmat, lane = r & 63, ks = (r >> 6) & 1, tl = r >> 7;
        const int orow = 4 * tl + (lane & 3), k0 = 32 * ks + 8 * (lane >> 4);
        f16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            _Float16 a, b;
            const float wv = orow < q.no ? q.out_W[(int64_t)orow * 64 + k0 + e] : 0.0f;
            mp_split(wv, a, b);
            hi[e] = a; lo[e] = b;
            if (!(fabsf(wv) < 65504.0f) && q.overflow) *q.overflow = 1;
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
    int abl;   // timing-only ablation (VSDE_MP_FWD_ABL; wrong results): 1 = layer 0 does not store its activations, 2 = layer 1 does not, 4 = no output stores
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
//   UPL < 4 ("spread", NP = 4 UPL): both planes again (three MFMAs), but the 16 columns of an operand hold the NP paths 16 / NP times
//             over -- every replica lane gets the four rows of its lane group and keeps only ITS UPL of them (replica index rr): the
//             gate block of a step then costs a lane UPL units' worth of transcendentals instead of four (the step is a latency
//             chain: the idle columns of a 4- or 8-path group were free).
//   UPL < 4, NP = 2 UPL: the same with the planes side by side again (two MFMAs): columns = [hi of the NP paths | their lo' parts] x
//             4 / UPL replicas; the lanes of the hi columns own the paths.  Per SIMD and step the two-layer kernel then issues 44
//             MFMAs instead of 66 -- and the SIMD's issue slots are what a step costs: its two waves' MFMAs and VALU operations
//             do not overlap (profiles/r05_mfma_valu_overlap.txt), "slack" products of one role delay the other role's chain.
// the MFMAs of one product, accumulated into A1 / A2 (which may carry another product's sums: see mp_fold)
template <int NP, int UPL>
__device__ __forceinline__ void mp_matmul_acc(const f16x8 (&wf)[3][2][2], const f16x8 (&hb)[2][2], f32x4 (&A1)[3], f32x4 (&A2)[3]) {
    constexpr bool P2 = NP == 4 * UPL;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int g = 0; g < 3; ++g) A1[g] = mp_mfma(wf[g][ks][0], hb[0][ks], A1[g]);
        if (P2) {
#pragma unroll
            for (int g = 0; g < 3; ++g) A2[g] = mp_mfma(wf[g][ks][0], hb[1][ks], A2[g]);
        }
#pragma unroll
        for (int g = 0; g < 3; ++g) A2[g] = mp_mfma(wf[g][ks][1], hb[0][ks], A2[g]);
    }
}
// this lane's UPL rows of a tile's four (replica rr), planes folded.  Selecting costs 6 v_cndmask + 3 more instructions per value: sums
// that a step needs only TOGETHER (W_ih h^0_t + W_hh h^1_{t-1} of the r and u gates) are therefore accumulated in ONE pair of
// accumulators across the two products and folded once.
template <int NP, int UPL>
__device__ __forceinline__ void mp_fold(const f32x4 &A1, const f32x4 &A2, int rr, float (&R)[UPL]) {
    constexpr bool P2 = NP == 4 * UPL;
    float s1[UPL], s2[UPL];
    if constexpr (UPL == 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[r] = A1[r]; s2[r] = A2[r]; }
    } else if constexpr (UPL == 2) {
        s1[0] = rr ? A1[2] : A1[0]; s1[1] = rr ? A1[3] : A1[1];
        s2[0] = rr ? A2[2] : A2[0]; s2[1] = rr ? A2[3] : A2[1];
    } else {
        const float l1 = (rr & 1) ? A1[1] : A1[0], h1 = (rr & 1) ? A1[3] : A1[2];
        const float l2 = (rr & 1) ? A2[1] : A2[0], h2 = (rr & 1) ? A2[3] : A2[2];
        s1[0] = (rr & 2) ? h1 : l1; s2[0] = (rr & 2) ? h2 : l2;
    }
#pragma unroll
    for (int r = 0; r < UPL; ++r)
        R[r] = P2 ? fmaf(s2[r], kMpLoInv, s1[r]) : fmaf(mp_row_shl<NP & 15>(s1[r]) + s2[r], kMpLoInv, s1[r]);
}
template <int NP, int UPL>
__device__ __forceinline__ void mp_matmul(const f16x8 (&wf)[3][2][2], const f16x8 (&hb)[2][2], float (&R)[3][UPL], int rr) {
    f32x4 A1[3], A2[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) { A1[g] = f32x4{0.f, 0.f, 0.f, 0.f}; A2[g] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    mp_matmul_acc<NP, UPL>(wf, hb, A1, A2);
#pragma unroll
    for (int g = 0; g < 3; ++g) mp_fold<NP, UPL>(A1[g], A2[g], rr, R[g]);
}
// UPL consecutive floats
template <int UPL> __device__ __forceinline__ void mp_ldu(float (&d)[UPL], const float *src) {
    if constexpr (UPL == 4) { const f32x4 v = *(const f32x4 *)src; d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3]; }
    else if constexpr (UPL == 2) { const float2 v = *(const float2 *)src; d[0] = v.x; d[1] = v.y; }
    else d[0] = *src;
}
template <int UPL> __device__ __forceinline__ void mp_stu(float *dst, const float (&v)[UPL], float sc = 1.0f) {
    if constexpr (UPL == 4) *(f32x4 *)dst = f32x4{v[0] * sc, v[1] * sc, v[2] * sc, v[3] * sc};
    else if constexpr (UPL == 2) *(float2 *)dst = make_float2(v[0] * sc, v[1] * sc);
    else *dst = v[0] * sc;
}

// Roles.  One layer: four waves (wave w = units 16 w ..).  Two layers: EIGHT waves, two per SIMD -- waves 0-3 are layer 0 (W_hh_l0
// and the emission rows in registers, the Euler-Maruyama update, the context record), waves 4-7 are layer 1 (W_ih_l1, W_hh_l1).
// With all three matrices in one wave the kernel needed 432 registers and hipcc moved ~180 values per step between the
// accumulator and the vector half of the file; split by layer each role stays below 256 registers, and the products a step
// does not need at once (W_hh h_t, consumed by step t + 1) run on one role's matrix pipe while the other role's VALU does gates.
//   step t:  [L0: gates -> h0_t]  barrier A  [L1: W_ih1 h0_t, gates -> h1_t | L0: W_hh0 h0_t]  barrier B
//            [L0: emission rows x h1_t, z_{t+1}, outputs, then gates of step t + 1 | L1: W_hh1 h1_t]
template <int L, bool SAVE, int S, int NP, int UPL = 4>
__global__ void __launch_bounds__(256 * L, 3 - L) head_fwd_mp_kernel(MpParams p) {
    constexpr bool SPREAD = UPL != 4, P2 = NP == 4 * UPL;   // SPREAD: replicated path columns, UPL units per lane; P2: one operand per plane (see mp_matmul)
    constexpr int NTRIL = S * (S + 1) / 2, NO = S + NTRIL, NTO = (NO + 3) / 4, PL = P2 ? 2 : 1;
    constexpr int CW = P2 ? NP : 2 * NP;                    // columns of one replica
    static_assert(L >= 1 && L <= 2 && S >= 1 && S <= 2 && (NP == 16 || NP == 8 || NP == 4 || NP == 2), "multi-path kernel: L <= 2, state_dim <= 2");
    static_assert(UPL == 4 || NP == 4 * UPL || NP == 2 * UPL, "spread form: 4 / UPL replicas of every path's column(s)");
    static_assert(NP >= 4 || SPREAD, "groups of 2 paths exist in the spread form only");
    // hidden state in B-fragment order: [step parity][layer][plane][k-step][lane group][column] x 8 f16 (NP < 16: one plane, the hi parts
    // in columns [0, NP), the lo' parts in [NP, 2 NP), the rest zero)
    __shared__ __attribute__((aligned(16))) f16x8 hbuf[2][L][PL][2][4][16];
    // Training launch, two layers: the layer-1 role does not store its saved-activation records itself -- five 16-byte stores per lane
    // and step cost ~250 issue cycles in its short slack (VSDE_MP_FWD_ABL=2: 30 us of 510).  It leaves them in LDS ([step parity]
    // [path][5][64] fp32: five ds_write_b128) and the layer-0 waves, which have the longer slack, copy the previous step's record out
    // as whole 16-byte lanes of consecutive addresses (one or two instructions per wave) behind their barrier A.
    constexpr bool STASH = SAVE && L > 1;
    constexpr int SPITCH = 324;                           // floats per path: 320 + 4, so that the 8-lane groups of a ds_write_b128 (paths
                                                          // 0..7 of one row group) fall on different banks (320 = 0 mod 32: 4-way conflicts)
    // (measured and dropped, round 5: the layer-0 role leaving its record there too, both copied out as 16-byte lanes -- 420 vs 407-417 us
    //  at 512 paths on one box: its five 4-byte stores per lane are not what the step waits for.  Nor is their number: with the idle
    //  lo'-column lanes taking two of the five stores (three instructions instead of five, values handed over by DPP) the training launch
    //  went 425 -> 437 us, the reverse sweep with two D4 stores instead of four 527 -> 534 us: tools/ab_lib.sh + tools/head_ab.py)
    constexpr int SREC = STASH ? NP * SPITCH : 4;         // floats per record (all paths of the group)
    __shared__ __attribute__((aligned(16))) float srec[2][SREC];
    const int tid = threadIdx.x, wv = tid >> 6, role = wv >> 2, w = wv & 3, lane = tid & 63, q = lane >> 4, pp = lane & 15;
    const int pc = pp & (NP - 1), rr = SPREAD ? pp / CW : 0;   // path column; replica index (SPREAD: this lane's units are rows rr UPL .. of its group's four)
    const int b_raw = blockIdx.x * NP + pc;
    const bool owner = SPREAD ? (P2 || (pp & (CW - 1)) < NP) : pp < NP;   // side-by-side planes: the lanes of the lo' columns (and, not spread, of the unused ones) run along and store nothing
    const bool live = owner && b_raw < p.B;
    const bool first = live && rr == 0;            // the lane that stores what exists once per path
    const int b = b_raw < p.B ? b_raw : p.B - 1;   // lanes beyond the batch recompute the last path and store nothing
    const int j0 = 16 * w + 4 * q + rr * UPL, T = p.T, I = S + p.C + p.P;
    mp_flush_f16_denormals();
    if (NP < 16) {
        for (int e = tid; e < 2 * L * PL * 2 * 4 * 16; e += 256 * L) (&hbuf[0][0][0][0][0][0])[e] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
        __syncthreads();
    }

    // the 4 owned units' new state -> f16 hi / lo' in B-fragment order (unit j = 16 w + 4 q + r -> k-step j >> 5, lane group (j >> 3) & 3,
    // element j & 7)
    auto publish = [&](const float (&h)[UPL], int t, int l) {
        const int par = t & 1;
        if constexpr (SPREAD) {   // UPL f16 per plane, element 4 (q & 1) + rr UPL of the path's column (the readers replicate)
            if (!owner) return;
            _Float16 *dh = (_Float16 *)&hbuf[par][l][0][w >> 1][2 * (w & 1) + (q >> 1)][pc] + 4 * (q & 1) + rr * UPL;
            _Float16 *dl = (_Float16 *)&hbuf[par][l][PL - 1][w >> 1][2 * (w & 1) + (q >> 1)][P2 ? pc : pc + NP] + 4 * (q & 1) + rr * UPL;
            _Float16 a[UPL], c[UPL];
#pragma unroll
            for (int r = 0; r < UPL; ++r) mp_split_fast(h[r], a[r], c[r]);
            if constexpr (UPL == 2) {
                typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
                *(f16x2 *)dh = f16x2{a[0], a[1]}; *(f16x2 *)dl = f16x2{c[0], c[1]};
            } else { *dh = a[0]; *dl = c[0]; }
            return;
        }
        f16x4 hi, lo;
#pragma unroll
        for (int r = 0; r < UPL; ++r) { _Float16 a, c; mp_split_fast(h[r], a, c); hi[r & 3] = a; lo[r & 3] = c; }
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
    auto save_acts = [&](const float (&h)[UPL], const float (&rg)[UPL], const float (&ug)[UPL], const float (&ng)[UPL], const float (&cn)[UPL],
                         int t, int l) {
        if (SAVE && live && !(p.abl & (1 << l))) {
            float *ab = p.acts + (((int64_t)b * T + t) * L + l) * 320 + j0;
            mp_stu<UPL>(ab, h); mp_stu<UPL>(ab + 64, rg); mp_stu<UPL>(ab + 128, ug); mp_stu<UPL>(ab + 192, ng); mp_stu<UPL>(ab + 256, cn, kMpInvSn);
        }
    };
    // gate block of one layer for the 4 owned units (exp2 domain: r = 1 / (1 + 2^x_r), n = 1 - 2 / (1 + 2^(a_n + r c_n)))
    auto gates = [&](const float (&ar)[UPL], const float (&au)[UPL], const float (&an)[UPL], const float (&c)[3][UPL],
                     const float (&bn)[UPL], float (&h)[UPL], float (&rg)[UPL], float (&ug)[UPL], float (&ng)[UPL], float (&cn)[UPL], int t, int l) {
#pragma unroll
        for (int r = 0; r < UPL; ++r) {
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
            for (int ks = 0; ks < 2; ++ks) hb[pl][ks] = hbuf[par][l][pl][ks][q][SPREAD ? (pp & (CW - 1)) : pp];
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
            float k1[3][UPL], bn1[UPL], h1[UPL];
#pragma unroll
            for (int r = 0; r < UPL; ++r) h1[r] = 0.f;
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int r = 0; r < UPL; ++r) {
                    const int row = g * 64 + j0 + r;
                    const float sc = g < 2 ? kMpSr : kMpSn;
                    k1[g][r] = sc * p.b_ih1[row] + (g < 2 ? sc * p.b_hh1[row] : 0.f);
                }
#pragma unroll
            for (int r = 0; r < UPL; ++r) bn1[r] = kMpSn * p.b_hh1[128 + j0 + r];
            // W_hh^1 h^1_{t-1} (h_{-1} = 0): the r / u tiles' sums stay in their accumulators (C1 / C2) and W_ih^1 h^0_t is accumulated on
            // top of them; only the n tile's c_n is needed by itself (c1[2])
            float c1[3][UPL];
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int r = 0; r < UPL; ++r) c1[g][r] = 0.f;
            f32x4 C1[3], C2[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) { C1[g] = f32x4{0.f, 0.f, 0.f, 0.f}; C2[g] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            for (int t = 0; t < T; ++t) {
                barrier();                             // A: h^0_t published
                f16x8 hb[2][2];
                read_state(t, 0, hb);
                float a1[3][UPL];
                C1[2] = f32x4{0.f, 0.f, 0.f, 0.f}; C2[2] = f32x4{0.f, 0.f, 0.f, 0.f};
                mp_matmul_acc<NP, UPL>(wi, hb, C1, C2);   // W_ih^1 h^0_t (+ the carried r / u sums)
#pragma unroll
                for (int g = 0; g < 3; ++g) mp_fold<NP, UPL>(C1[g], C2[g], rr, a1[g]);
                float ar[UPL], au[UPL], an[UPL], rg[UPL], ug[UPL], ng[UPL], cn[UPL];
#pragma unroll
                for (int r = 0; r < UPL; ++r) { ar[r] = k1[0][r] + a1[0][r]; au[r] = k1[1][r] + a1[1][r]; an[r] = k1[2][r] + a1[2][r]; }
                gates(ar, au, an, c1, bn1, h1, rg, ug, ng, cn, t, L - 1);
                barrier();                             // B: h^1_t published
                if (STASH && owner && !(p.abl & 2)) {      // the record of step t for the layer-0 waves (copied out behind barrier A of step t + 1)
                    float *rec = &srec[t & 1][0] + pc * SPITCH + j0;
                    mp_stu<UPL>(rec, h1); mp_stu<UPL>(rec + 64, rg); mp_stu<UPL>(rec + 128, ug); mp_stu<UPL>(rec + 192, ng); mp_stu<UPL>(rec + 256, cn, kMpInvSn);
                }
                read_state(t, L - 1, hb);
#pragma unroll
                for (int g = 0; g < 3; ++g) { C1[g] = f32x4{0.f, 0.f, 0.f, 0.f}; C2[g] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                mp_matmul_acc<NP, UPL>(wh, hb, C1, C2);   // W_hh^1 h^1_t: consumed by step t + 1
                mp_fold<NP, UPL>(C1[2], C2[2], rr, c1[2]);
            }
            if (STASH) barrier();                      // the last record is complete: the layer-0 waves copy it out
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
    float k0[3][UPL], bn0[UPL], wx[S][3][UPL], ob[NO];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < UPL; ++r) {
            const int row = g * 64 + j0 + r;
            const float sc = g < 2 ? kMpSr : kMpSn;
            float th = 0.f;   // theta term (forward.py:157-175)
            for (int e = 0; e < p.P; ++e) th = fmaf(p.theta[(int64_t)b * p.P + e], p.W_ih0[(int64_t)row * I + S + p.C + e], th);
            k0[g][r] = sc * th + (g < 2 ? sc * p.b_hh0[row] : 0.f);
#pragma unroll
            for (int i = 0; i < S; ++i) wx[i][g][r] = sc * p.W_ih0[(int64_t)row * I + i];
        }
#pragma unroll
    for (int r = 0; r < UPL; ++r) bn0[r] = kMpSn * p.b_hh0[128 + j0 + r];
#pragma unroll
    for (int r = 0; r < NO; ++r) ob[r] = p.out_b[r];

    float z[S], h0[UPL];
    float c0[3][UPL];                                  // W_hh^0 h^0_{t-1}: h_{-1} = 0
#pragma unroll
    for (int r = 0; r < UPL; ++r) { h0[r] = 0.f; c0[0][r] = 0.f; c0[1][r] = 0.f; c0[2][r] = 0.f; }
#pragma unroll
    for (int i = 0; i < S; ++i) z[i] = p.x0[(int64_t)b * S + i];
    if (w == 0 && q == 0 && first) {
#pragma unroll
        for (int i = 0; i < S; ++i) p.paths[(int64_t)b * (T + 1) * S + i] = z[i];
    }

    // the projected context record and eps, one step ahead in registers (the wait for them at the top of a step is a vmcnt(0)).
    // (Round 5, measured and dropped: both records by LDS-DMA from the layer-1 waves -- whose vector-memory queue is otherwise empty, so a
    // counted vmcnt(1) is exact -- into a ring of four slots, the layer-0 waves reading them out of LDS behind barrier B: five loads per
    // lane and step less on the layer-0 role, 425-437 / 356-366 us against 432-436 / 364 us (training / sampling launch): the sampling
    // launch's 1,640 cycles per step ARE its 44 MFMAs x 16 + ~190 VALU x 2.6 + 12 transcendentals x 12 per SIMD; loads issue elsewhere.)
    const float *Gb = p.G + (int64_t)b * T * 192 + j0;
    const float *eb = p.eps + (int64_t)b * T * S;
    float gq[3][UPL];
    float ev[S];
    auto fetch = [&](int t) {
        const int tc = t < T ? t : T - 1;
#pragma unroll
        for (int g = 0; g < 3; ++g) mp_ldu<UPL>(gq[g], Gb + (int64_t)tc * 192 + g * 64);
#pragma unroll
        for (int i = 0; i < S; ++i) ev[i] = eb[(int64_t)tc * S + i];
    };
    fetch(0);

    // Outputs of step t are stored during step t + 1, in the slack behind barrier A (where the layer-0 waves wait for layer 1):
    // hipcc's wait for the prefetched context record at the top of the loop is a vmcnt(0) (stores sit in conditional blocks it
    // cannot count), and a store issued just before it would put a full store round trip on every step.
    float pz[S], pmu[S], pL[S][S], praw[NTRIL];
    auto store_outputs = [&](int t) {     // paths[b, t + 1], means[b, t], chol[b, t], chol_raw[b, t]: one wave each
        if (q == 0 && first && !(p.abl & 4)) {
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

    // layer 1's record of step t: NP x 80 float4, lane = consecutive addresses (path f / 80, offset f % 80)
    auto copy_l1_record = [&](int t) {
        if (STASH && !(p.abl & 2)) {
            constexpr int NF = NP * 80;
            const f32x4 *src = (const f32x4 *)&srec[t & 1][0];
#pragma unroll
            for (int k = 0; k < (NF + 255) / 256; ++k) {
                const int f = k * 256 + w * 64 + lane;
                const int path = f / 80, off = f - path * 80;
                const int bb = blockIdx.x * NP + path;
                if (f < NF && bb < p.B) *(f32x4 *)(p.acts + (((int64_t)bb * T + t) * L + (L - 1)) * 320 + off * 4) = src[path * (SPITCH / 4) + off];
            }
        }
    };
    for (int t = 0; t < T; ++t) {
        float g0[UPL], g1[UPL], g2[UPL];
#pragma unroll
        for (int r = 0; r < UPL; ++r) { g0[r] = gq[0][r]; g1[r] = gq[1][r]; g2[r] = gq[2][r]; }
        float e[S];
#pragma unroll
        for (int i = 0; i < S; ++i) e[i] = ev[i];
        fetch(t + 1);
        // ---- layer 0: a = G_t (context projection + b_ih) + theta term + z_t W_x   (forward.py:195-219)
        float ar[UPL], au[UPL], an[UPL], rg[UPL], ug[UPL], ng[UPL], cn[UPL];
#pragma unroll
        for (int r = 0; r < UPL; ++r) {
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
            mp_matmul<NP, UPL>(wf, hb, c0, rr);        // W_hh^0 h^0_t: consumed by step t + 1, runs beside layer 1's gates
            save_acts(h0, rg, ug, ng, cn, t, 0);
            if (t > 0) { store_outputs(t - 1); copy_l1_record(t - 1); }
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
                if (P2) O2[tl] = mp_mfma(of[tl][ks][0], hb[PL - 1][ks], O2[tl]);
                O2[tl] = mp_mfma(of[tl][ks][1], hb[0][ks], O2[tl]);
            }
        if (L == 1) {
            mp_matmul<NP, UPL>(wf, hb, c0, rr);
            save_acts(h0, rg, ug, ng, cn, t, 0);
            if (t > 0) store_outputs(t - 1);
        }
        float o[NO];
#pragma unroll
        for (int r = 0; r < NO; ++r) {
            const float a1 = O1[r >> 2][r & 3], a2 = O2[r >> 2][r & 3];
            o[r] = ob[r] + (P2 ? fmaf(a2, kMpLoInv, a1) : fmaf(mp_row_shl<NP & 15>(a1) + a2, kMpLoInv, a1));
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
    if (STASH) { barrier(); copy_l1_record(T - 1); }
    store_outputs(T - 1);
}

// ===========================================================================================================================
// Reverse-time sweep on the matrix cores (reference kernels/backward.py:208-624), two GRU layers, 4 or 8 paths
// per workgroup.  Same machinery as the forward kernel, transposed: the products are W^T d with d = the gate gradients of a layer
// (K = 192 = six k-steps; a wave owns 16 OUTPUT units, one A tile per k-step), the B operand carries the hi parts of the group's
// paths in columns [0, NP) and the lo' parts in [NP, 2 NP).
//   layer-1 waves (4): W_hh_l1^T, the state columns of W_ih_l0 (transposed, replicated down their tile like the forward's emission
//       rows, so every lane gets the S components of d z_t), out_proj^T.
//       [W_x^T pi0(t+1) -> dz; dO_t; out_proj^T dO_t (through a wave-private LDS tile); layer-1 gate gradients] barrier 2
//       [W_hh_l1^T ph1 -> carried dh1 (slack)] barrier 3
//   layer-0 waves (4): W_ih_l1^T, W_hh_l0^T.
//       barrier 2 [W_ih_l1^T pi1 -> gradient of layer 0's output; layer-0 gate gradients] barrier 3 [W_hh_l0^T ph0 -> carried dh0]
// Range: gradients have no natural scale (a GradScaler multiplies the loss by 65536), f16 does.  The sweep is linear in the
// upstream gradients, so it runs on g / Sg with Sg = 2^(1 + exponent of max |g|) (mp_absmax_kernel; |g| / Sg in [1/2, 1): a gate
// gradient may grow to 6e4 x the largest upstream gradient before an f16 operand overflows, values below 2^-14 of that scale keep 11
// bits through the lo' plane alone -- with 2^5 more headroom a fifth of the gate gradients fell there and the gradients were 2e-5
// .. 9e-5 off) and every stored record is multiplied by Sg -- a power of two, exact.
struct MpBwdPrep {
    int S, no, I;
    const float *W_hh0, *W_ih1, *W_hh1, *W_ih0, *out_W;
    f16x8 *frags;
};
constexpr int kMpBwdWx = 3 * kMpMatFrags, kMpBwdOut = kMpBwdWx + 6 * 2 * 64, kMpBwdTotal = kMpBwdOut + 4 * 2 * 64;

__global__ void __launch_bounds__(256) mp_bwd_prep_kernel(MpBwdPrep q) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    f16x8 hi, lo;
    if (i < 3 * 1536) {                               // W^T tiles: A[i = unit 16 w + (lane & 15)][k = 32 ks + 8 (lane >> 4) + e] = W[k][unit]
        const int m = i / 1536, r = i - m * 1536, lane = r & 63, ks = (r >> 6) % 6, w = r / 384;
        const float *W = m == 0 ? q.W_hh0 : (m == 1 ? q.W_ih1 : q.W_hh1);
        const int unit = 16 * w + (lane & 15), k0 = 32 * ks + 8 * (lane >> 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) { _Float16 a, b; mp_split(W[(int64_t)(k0 + e) * 64 + unit], a, b); hi[e] = a; lo[e] = b; }
        f16x8 *dst = q.frags + (int64_t)m * kMpMatFrags + ((w * 6 + ks) * 2) * 64 + lane;
        dst[0] = hi; dst[64] = lo;
    } else if (i < 3 * 1536 + 384) {                  // state columns of W_ih_l0: tile row i holds column i & 3 (zero beyond S)
        const int r = i - 3 * 1536, lane = r & 63, ks = r >> 6, col = lane & 3, k0 = 32 * ks + 8 * (lane >> 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) { _Float16 a, b; mp_split(col < q.S ? q.W_ih0[(int64_t)(k0 + e) * q.I + col] : 0.f, a, b); hi[e] = a; lo[e] = b; }
        f16x8 *dst = q.frags + kMpBwdWx + (ks * 2) * 64 + lane;
        dst[0] = hi; dst[64] = lo;
    } else if (i < 3 * 1536 + 384 + 256) {            // out_proj^T: A[i = unit][k = emission row] (one k-step, rows >= NO zero)
        const int r = i - 3 * 1536 - 384, lane = r & 63, w = r >> 6, unit = 16 * w + (lane & 15), k0 = 8 * (lane >> 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) { _Float16 a, b; mp_split(k0 + e < q.no ? q.out_W[(int64_t)(k0 + e) * 64 + unit] : 0.f, a, b); hi[e] = a; lo[e] = b; }
        f16x8 *dst = q.frags + kMpBwdOut + (w * 2) * 64 + lane;
        dst[0] = hi; dst[64] = lo;
    }
}

// max |g| over the three upstream gradient tensors, as the bits of a non-negative float (order-independent: deterministic)
__global__ void __launch_bounds__(256) mp_absmax_kernel(const float *a, int64_t na, const float *b, int64_t nb, const float *c, int64_t nc,
                                                        unsigned *out) {
    __shared__ unsigned red[4];
    unsigned m = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < na + nb + nc; i += (int64_t)gridDim.x * 256) {
        const float v = i < na ? a[i] : (i < na + nb ? b[i - na] : c[i - na - nb]);
        const unsigned u = __float_as_uint(v) & 0x7fffffffu;
        m = u > m ? u : m;     // NaN / inf bit patterns compare above every finite value: the scale saturates, the sweep stays NaN
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const unsigned o = __shfl_xor(m, off, 64); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) m = red[w] > m ? red[w] : m;
        atomicMax(out, m);
    }
}

struct MpBwdParams {
    int B, T, P, C;
    const float *g_paths, *g_means, *g_chol, *eps, *chol_raw, *acts;
    const float *W_ih0;
    const f16x8 *frags;
    const unsigned *absmax;
    float dt, sqdt, diag_min;
    float *D4, *DO, *g_x0, *g_theta;
};

// R[r] = sum_k A[.][k] B[k][path] for the wave's 16 output units (lane: 4 of them), six k-steps, hi / lo' planes
template <int NP, bool SKIP = false>
__device__ __forceinline__ void mp_matmul_t(const f16x8 (&af)[6][2], const f16x8 (&bf)[6], f32x4 &R) {
    f32x4 A1 = {0.f, 0.f, 0.f, 0.f}, A2 = {0.f, 0.f, 0.f, 0.f};
    if (SKIP) { A1[0] = (float)bf[0][0]; A2[1] = (float)af[1][0][0]; }
#pragma unroll
    for (int ks = 0; ks < (SKIP ? 0 : 6); ++ks) {
        A1 = mp_mfma(af[ks][0], bf[ks], A1);
        A2 = mp_mfma(af[ks][1], bf[ks], A2);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) R[r] = fmaf(mp_row_shl<NP>(A1[r]) + A2[r], kMpLoInv, A1[r]);
}

// ABL: timing-only ablations (wrong results; VSDE_MP_BWD_ABL, S = 2 / NP = 4 only): 1 = no D4 / DO stores, 2 = no activation / upstream
// loads inside the loop, 4 = gate gradients published without the f16 split arithmetic, 8 = no matrix products
template <int S, int NP, int ABL = 0>
__global__ void __launch_bounds__(512, 1) head_bwd_mp_kernel(MpBwdParams p) {
    constexpr int L = 2, NTRIL = S * (S + 1) / 2, NO = S + NTRIL;
    static_assert(S >= 1 && S <= 2 && (NP == 4 || NP == 8), "multi-path backward: two layers, state_dim <= 2, 4 or 8 paths per group");
    // gate gradients in B-fragment order: [step parity][layer][block dr / du / dn / dc_n][k-step][lane group][column] x 8 f16
    __shared__ __attribute__((aligned(16))) f16x8 dbuf[2][L][4][2][4][16];
    __shared__ __attribute__((aligned(16))) f16x8 obuf[4][4][16];         // per layer-1 wave: dO as a one-k-step B operand
    const int tid = threadIdx.x, wv = tid >> 6, role = wv >> 2, w = wv & 3, lane = tid & 63, q = lane >> 4, pp = lane & 15;
    const int b_raw = blockIdx.x * NP + (pp & (NP - 1));
    const bool owner = pp < NP, live = owner && b_raw < p.B;
    const int b = b_raw < p.B ? b_raw : p.B - 1;
    const int j0 = 16 * w + 4 * q, T = p.T, I = S + p.C + p.P;
    mp_flush_f16_denormals();
    for (int e = tid; e < 2 * L * 4 * 2 * 4 * 16; e += 512) (&dbuf[0][0][0][0][0][0])[e] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    for (int e = tid; e < 4 * 4 * 16; e += 512) (&obuf[0][0][0])[e] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    __syncthreads();
    // scale of the sweep: Sg = 2^(E - 127 + 1), E the biased exponent of max |g| (0 -> everything is zero: any scale)
    const unsigned am = *p.absmax;
    int E = (int)(am >> 23);
    E = E < 1 ? 126 : (E > 250 ? 250 : E);
    const float Sg = __uint_as_float((unsigned)(E + 1) << 23), inv = __uint_as_float((unsigned)(253 - E) << 23);

    auto barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto load_t = [&](int m, f16x8 (&af)[6][2]) {
#pragma unroll
        for (int ks = 0; ks < 6; ++ks)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) af[ks][pl] = p.frags[(int64_t)m * kMpMatFrags + (((w * 6 + ks) * 2) + pl) * 64 + lane];
    };
    // gate gradients of the 4 owned units (backward.py:59-67) from d = dL/dh^l_t; returns the carried u * d
    auto gate_grads = [&](const f32x4 &d, const f32x4 &r, const f32x4 &u, const f32x4 &n, const f32x4 &cn, const f32x4 &hp, f32x4 &dr,
                          f32x4 &du, f32x4 &dn, f32x4 &dcn, f32x4 &carry) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float dnn = (1.0f - u[e]) * d[e], duu = (hp[e] - n[e]) * d[e];
            dn[e] = dnn * (1.0f - n[e] * n[e]);
            du[e] = duu * (u[e] * (1.0f - u[e]));
            dcn[e] = dn[e] * r[e];
            dr[e] = (dn[e] * cn[e]) * (r[e] * (1.0f - r[e]));
            carry[e] = u[e] * d[e];
        }
    };
    auto publish = [&](const f32x4 &v, int par, int l, int blk) {
        f16x4 hi, lo;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            _Float16 a, c;
            if (ABL & 4) { a = (_Float16)v[r]; c = a; } else mp_split_fast(v[r], a, c);
            hi[r] = a; lo[r] = c;
        }
        if (owner) {
            *((f16x4 *)&dbuf[par][l][blk][w >> 1][2 * (w & 1) + (q >> 1)][pp] + (q & 1)) = hi;
            *((f16x4 *)&dbuf[par][l][blk][w >> 1][2 * (w & 1) + (q >> 1)][pp + NP] + (q & 1)) = lo;
        }
    };
    auto read_d = [&](int par, int l, bool hh, f16x8 (&bf)[6]) {    // (dr, du, dn) or, hh, (dr, du, dc_n)
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) bf[ks] = dbuf[par][l][(ks >> 1) == 2 ? (hh ? 3 : 2) : (ks >> 1)][ks & 1][q][pp];
    };
    auto store_d4 = [&](const f32x4 &dr, const f32x4 &du, const f32x4 &dn, const f32x4 &dcn, int t, int l) {
        if (live && !(ABL & 1)) {   // D4[b][t][l][{dr, du, dn, dc_n}][64]
            float *o = p.D4 + (((int64_t)b * T + t) * L + l) * 256 + j0;
            *(f32x4 *)(o) = dr * Sg; *(f32x4 *)(o + 64) = du * Sg; *(f32x4 *)(o + 128) = dn * Sg; *(f32x4 *)(o + 192) = dcn * Sg;
        }
    };
    // saved activations of (t, layer): r, u, n, n_hh, and h of step t - 1 (zero before the first step)
    const float *ab = p.acts + (int64_t)b * T * L * 320 + j0;
    auto load_acts = [&](int t, int l, f32x4 (&a)[5]) {
        if ((ABL & (2 | 32)) && t < T - 2) return;
        const int tc = t < 0 ? 0 : t;
        const float *o = ab + ((int64_t)tc * L + l) * 320;
#pragma unroll
        for (int k = 1; k < 5; ++k) a[k] = *(const f32x4 *)(o + 64 * k);
        // h of step t - 1; step 0 has none: the CONSUMER puts the zero there (a select here reads the register the load has just been
        // issued for -- an s_waitcnt vmcnt on a two-steps-ahead load, one HBM round trip per step: 370 of the sweep's 1019 us in round 4)
        a[0] = *(const f32x4 *)(o - (tc > 0 ? L * 320 : 0));
    };

    if (role == 1) {
        // =================================================================== layer-1 waves
        f16x8 whh[6][2], wxs[6][2], wo[2];
        load_t(2, whh);
#pragma unroll
        for (int ks = 0; ks < 6; ++ks)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) wxs[ks][pl] = p.frags[kMpBwdWx + ((ks * 2) + pl) * 64 + lane];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) wo[pl] = p.frags[kMpBwdOut + ((w * 2) + pl) * 64 + lane];
        float dx[S];
#pragma unroll
        for (int i = 0; i < S; ++i) dx[i] = 0.f;
        f32x4 dh1 = {0.f, 0.f, 0.f, 0.f};
        // saved activations and the step's upstream gradients / noise / raw Cholesky entries travel TWO steps ahead in two register
        // sets (the records were written by the forward launch long ago: HBM latency, more than one step of this loop)
        f32x4 act[2][5];
        float gp[2][S], gm[2][S], gl[2][S][S], ee[2][S], raw[2][NTRIL];
        auto load_up = [&](int t, auto slot_c) {
            constexpr int sl = decltype(slot_c)::value;
            if ((ABL & (2 | 16)) && t < T - 2) return;
            const int tc = t < 0 ? 0 : t;
            const int64_t bt = (int64_t)b * T + tc;
#pragma unroll
            for (int i = 0; i < S; ++i) {
                gp[sl][i] = p.g_paths[(bt + b + 1) * S + i]; gm[sl][i] = p.g_means[bt * S + i]; ee[sl][i] = p.eps[bt * S + i];
#pragma unroll
                for (int c = 0; c < S; ++c) gl[sl][i][c] = p.g_chol[bt * S * S + i * S + c];
            }
#pragma unroll
            for (int r = 0; r < NTRIL; ++r) raw[sl][r] = p.chol_raw[bt * NTRIL + r];
        };
        load_acts(T - 1, 1, act[0]); load_up(T - 1, MpSlot<0>{});
        load_acts(T - 2, 1, act[1]); load_up(T - 2, MpSlot<1>{});
        auto step = [&](int t, auto slot_c) {
            constexpr int sl = decltype(slot_c)::value;
            const int par = t & 1;
            if (t < T - 1) {   // d z_{t+1} through layer 0's input of step t + 1: W_x^T pi0  (backward.py:494-509)
                f16x8 bf[6];
                read_d(par ^ 1, 0, false, bf);
                f32x4 dxd;
                mp_matmul_t<NP, (ABL & 8) != 0>(wxs, bf, dxd);
#pragma unroll
                for (int i = 0; i < S; ++i) dx[i] += dxd[i];
            }
            f32x4 a_r = act[sl][1], a_u = act[sl][2], a_n = act[sl][3], a_cn = act[sl][4], a_hp = t > 0 ? act[sl][0] : f32x4{0.f, 0.f, 0.f, 0.f};
            float cgp[S], cgm[S], cgl[S][S], ce[S], craw[NTRIL];
#pragma unroll
            for (int i = 0; i < S; ++i) {
                cgp[i] = gp[sl][i]; cgm[i] = gm[sl][i]; ce[i] = ee[sl][i];
#pragma unroll
                for (int c = 0; c < S; ++c) cgl[i][c] = gl[sl][i][c];
            }
#pragma unroll
            for (int r = 0; r < NTRIL; ++r) craw[r] = raw[sl][r];
            load_acts(t - 2, 1, act[sl]);
            load_up(t - 2, slot_c);
            // ---- dO_t  (backward.py:278-334)
            float dO[NO];
#pragma unroll
            for (int i = 0; i < S; ++i) { dx[i] = fmaf(cgp[i], inv, dx[i]); dO[i] = fmaf(dx[i], p.dt, cgm[i] * inv); }
            {
                int k = 0;
#pragma unroll
                for (int i = 0; i < S; ++i)
#pragma unroll
                    for (int c = 0; c <= i; ++c, ++k) {
                        float dL = fmaf(dx[i] * ce[c], p.sqdt, cgl[i][c] * inv);
                        if (i == c && !(craw[k] >= p.diag_min || dL < 0.f)) dL = 0.f;    // bounds.py:20
                        dO[S + k] = dL;
                    }
            }
            if (w == 0 && q == 0 && live && !(ABL & 1)) {
#pragma unroll
                for (int r = 0; r < NO; ++r) p.DO[((int64_t)b * T + t) * NO + r] = dO[r] * Sg;
            }
            // dO as a B operand (k = emission row, in lane group 0), through this wave's private tile
            if (q == 0 && owner) {
                f16x8 hi = {0, 0, 0, 0, 0, 0, 0, 0}, lo = hi;
#pragma unroll
                for (int r = 0; r < NO; ++r) { _Float16 a, c; mp_split_fast(dO[r], a, c); hi[r] = a; lo[r] = c; }
                obuf[w][0][pp] = hi; obuf[w][0][pp + NP] = lo;
            }
            wave_lds_fence();
            const f16x8 ob = obuf[w][q][pp];
            f32x4 dcur;
            {
                f32x4 A1 = {0.f, 0.f, 0.f, 0.f}, A2 = A1;
                A1 = mp_mfma(wo[0], ob, A1); A2 = mp_mfma(wo[1], ob, A2);
#pragma unroll
                for (int r = 0; r < 4; ++r) dcur[r] = fmaf(mp_row_shl<NP>(A1[r]) + A2[r], kMpLoInv, A1[r]);   // out_proj^T dO  (:296-349)
            }
            f32x4 dr, du, dn, dcn, carry;
            gate_grads(dcur + dh1, a_r, a_u, a_n, a_cn, a_hp, dr, du, dn, dcn, carry);
            publish(dr, par, 1, 0); publish(du, par, 1, 1); publish(dn, par, 1, 2); publish(dcn, par, 1, 3);
            barrier();                                 // 2: layer 1's gate gradients published
            {
                f16x8 bf[6];
                read_d(par, 1, true, bf);
                f32x4 rec;
                mp_matmul_t<NP, (ABL & 8) != 0>(whh, bf, rec);         // W_hh_l1^T ph1  (:96-105)
                dh1 = carry + rec;
            }
            store_d4(dr, du, dn, dcn, t, 1);
            __builtin_amdgcn_sched_barrier(0);
            barrier();                                 // 3: layer 0's gate gradients published
        };
        for (int t = T - 1; t >= 0; t -= 2) {
            step(t, MpSlot<0>{});
            if (t >= 1) step(t - 1, MpSlot<1>{});
        }
        {   // the step-0 term of d z_0, then grad x0  (:620-624)
            f16x8 bf[6];
            read_d(0, 0, false, bf);
            f32x4 dxd;
            mp_matmul_t<NP, (ABL & 8) != 0>(wxs, bf, dxd);
            if (w == 0 && q == 0 && live) {
#pragma unroll
                for (int i = 0; i < S; ++i) p.g_x0[(int64_t)b * S + i] = (dx[i] + dxd[i] + p.g_paths[(int64_t)b * (T + 1) * S + i] * inv) * Sg;
            }
        }
        barrier();                                     // X: the gate-gradient tiles are free (layer 0 puts its theta sums there)
        barrier();                                     // Y: pairs with the layer-0 waves' barrier in front of the grad-theta sums
        return;
    }

    // ======================================================================= layer-0 waves
    f16x8 wih1[6][2], whh0[6][2];
    load_t(1, wih1);
    load_t(0, whh0);
    f32x4 dh0 = {0.f, 0.f, 0.f, 0.f}, ths[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) ths[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 act[2][5];
    load_acts(T - 1, 0, act[0]);
    load_acts(T - 2, 0, act[1]);
    auto step = [&](int t, auto slot_c) {
        constexpr int sl = decltype(slot_c)::value;
        const int par = t & 1;
        f32x4 a_r = act[sl][1], a_u = act[sl][2], a_n = act[sl][3], a_cn = act[sl][4], a_hp = t > 0 ? act[sl][0] : f32x4{0.f, 0.f, 0.f, 0.f};
        load_acts(t - 2, 0, act[sl]);
        barrier();                                     // 2
        f32x4 dcur;
        {
            f16x8 bf[6];
            read_d(par, 1, false, bf);
            mp_matmul_t<NP, (ABL & 8) != 0>(wih1, bf, dcur);           // W_ih_l1^T pi1: gradient of layer 0's output  (:83-94)
        }
        f32x4 dr, du, dn, dcn, carry;
        gate_grads(dcur + dh0, a_r, a_u, a_n, a_cn, a_hp, dr, du, dn, dcn, carry);
        publish(dr, par, 0, 0); publish(du, par, 0, 1); publish(dn, par, 0, 2); publish(dcn, par, 0, 3);
        ths[0] += dr; ths[1] += du; ths[2] += dn;      // sum_t pi0: grad theta = W_theta^T of it (:511-548)
        barrier();                                     // 3
        {
            f16x8 bf[6];
            read_d(par, 0, true, bf);
            f32x4 rec;
            mp_matmul_t<NP, (ABL & 8) != 0>(whh0, bf, rec);            // W_hh_l0^T ph0  (:566-573)
            dh0 = carry + rec;
        }
        store_d4(dr, du, dn, dcn, t, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int t = T - 1; t >= 0; t -= 2) {
        step(t, MpSlot<0>{});
        if (t >= 1) step(t - 1, MpSlot<1>{});
    }
    // grad theta[b][e] = sum_rows W_ih_l0[row][S + C + e] * sum_t pi0[row]: the sums go through LDS (the gate-gradient tiles are free now)
    float *tsum = (float *)&dbuf[0][0][0][0][0][0];    // [NP][192]
    barrier();                                         // X: the layer-1 waves have read dbuf[0][0] (step 0) for grad x0
    if (owner) {
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) tsum[pp * 192 + g * 64 + j0 + r] = ths[g][r];
    }
    barrier();                                         // Y
    for (int o = tid; o < NP * p.P; o += 256) {
        const int path = o / p.P, e = o - path * p.P, bb = blockIdx.x * NP + path;
        if (bb < p.B) {
            float acc = 0.f;
            for (int row = 0; row < 192; ++row) acc = fmaf(p.W_ih0[(int64_t)row * I + S + p.C + e], tsum[path * 192 + row], acc);
            p.g_theta[(int64_t)bb * p.P + e] = acc * Sg;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Round 5: the reverse-time sweep for SMALL groups -- 2 or 4 paths per workgroup (512 / 1024 paths on 256 CUs) -- in the spread form
// of the forward kernel (every path's [hi | lo'] columns 4 / 2 times across the operand, so a lane owns UPL = NP / 2 hidden units
// instead of four: a quarter / half of the gate-gradient and f16-split arithmetic per lane), and with the per-step records out of the
// wave's way:
//   * round 4's kernel loaded the saved activations (five 16-byte loads per lane, layer and step) and thirteen upstream scalars into
//     registers two steps ahead; hipcc's counter for them degenerates to s_waitcnt vmcnt(0) at the loop boundary and the loads cost
//     the sweep 330 + 130 of its 1,080 us (VSDE_MP_BWD_ABL = 32 / 16).  Here they arrive by LDS-DMA (global_load_lds: no registers, no
//     compiler-inserted waits): record r of a layer = NP x 1,280 contiguous bytes, one 1 KB piece per wave, THREE steps ahead into a ring
//     of four slots; the upstream scalars of the group (3 S + S^2 + S (S + 1) / 2 per path from five tensors) are ONE 4-byte DMA by wave 3
//     of the layer-0 role.  A wave waits for its own piece with a counted vmcnt in front of the step's last barrier (every
//     vector-memory instruction below is issued by every wave in every step -- a store is masked per lane, never skipped: each wave holds
//     owner lanes of its group's first path; the one-lane DO store goes to a sink where it has no target -- so the count is exact).
//   * h_{t-1} is the h part of record t - 1, which is in the ring anyway.
template <int S, int NP>
__global__ void __launch_bounds__(512, 1) head_bwd_mps_kernel(MpBwdParams p) {
    constexpr int L = 2, NTRIL = S * (S + 1) / 2, NO = S + NTRIL, UPL = NP / 2, CW = 2 * NP;
    constexpr int NUP = 3 * S + S * S + NTRIL;                       // upstream items per path and step
    constexpr int APIECES = (NP * 1280 + 1023) / 1024, APW = (APIECES + 3) / 4, ASLOT = APW * 4 * 1024;   // bytes per (ring slot, layer)
    static_assert(S >= 1 && S <= 2 && (NP == 2 || NP == 4) && NUP <= 16 && NP * 16 <= 64, "spread sweep: 2 or 4 paths per group");
    extern __shared__ __attribute__((aligned(16))) char mps_lds[];
    typedef f16x8 (*dbuf_t)[L][4][2][4][16];
    dbuf_t dbuf = (dbuf_t)mps_lds;                                   // [2][L][4][2][4][16] x 16 bytes = 32 KB: gate gradients in B-fragment order
    f16x8 (*obuf)[4][16] = (f16x8 (*)[4][16])(mps_lds + 32768);      // [4][4][16]: dO as a one-k-step B operand, per layer-1 wave
    char *abuf = mps_lds + 32768 + 4096;                             // [4 slots][L][ASLOT]: saved-activation records
    float *ubuf = (float *)(abuf + 4 * L * ASLOT);                   // [4 slots][64]: upstream records [path][16]
    const int tid = threadIdx.x, wv = tid >> 6, role = wv >> 2, w = wv & 3, lane = tid & 63, q = lane >> 4, pp = lane & 15;
    const int pc = pp & (NP - 1), rr = pp / CW, colr = pp & (CW - 1);
    const int b0 = blockIdx.x * NP, b_raw = b0 + pc;
    const bool owner = colr < NP, live = owner && b_raw < p.B, first = live && rr == 0;
    const int b = b_raw < p.B ? b_raw : p.B - 1;
    const int j0 = 16 * w + 4 * q + rr * UPL, T = p.T, I = S + p.C + p.P;
    mp_flush_f16_denormals();
    for (int e = tid; e < (32768 + 4096) / 16; e += 512) ((f16x8 *)mps_lds)[e] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned am = *p.absmax;
    int E = (int)(am >> 23);
    E = E < 1 ? 126 : (E > 250 ? 250 : E);
    const float Sg = __uint_as_float((unsigned)(E + 1) << 23), inv = __uint_as_float((unsigned)(253 - E) << 23);
    float *sink = (float *)((char *)p.frags + (size_t)kMpBwdTotal * sizeof(f16x8) + 256) + 2 * tid;   // 4 KB behind the absmax word (8-byte slots)

    auto barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto wait_vm = [&](int n) {   // wave-uniform n
        switch (n) {
            case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
            case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
            case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
    };
    // this wave's piece(s) of the layer's record r (clamped at 0: the last steps re-fetch record 0 into slots nobody reads)
    int a_path[APW], a_off[APW];
#pragma unroll
    for (int i = 0; i < APW; ++i) {
        const int c = (w + 4 * i) * 64 + lane, path = c / 80;
        const bool ok = w + 4 * i < APIECES && path < NP && b0 + path < p.B;
        a_path[i] = ok ? path : 0; a_off[i] = ok ? c - path * 80 : lane;
    }
    auto dma_acts = [&](int r, int l_) {
        const int rc = r < 0 ? 0 : r;
#pragma unroll
        for (int i = 0; i < APW; ++i) {
            const float *src = p.acts + ((((int64_t)b0 + a_path[i]) * T + rc) * L + l_) * 320 + a_off[i] * 4;
            __builtin_amdgcn_global_load_lds((const void *)src, (__attribute__((address_space(3))) void *)(abuf + ((r & 3) * L + l_) * ASLOT + (w + 4 * i) * 1024),
                                             16, 0, 0);
        }
    };
    // upstream record of step r: lane = [path][16 items]: g_paths[r + 1], g_means[r], eps[r] (S each), g_chol[r] (S S), chol_raw[r]
    const float *u_base; int u_stride;
    {
        const int path = lane >> 4, item = lane & 15;
        const bool ok = path < NP && item < NUP && b0 + path < p.B;
        const int64_t bb = ok ? b0 + path : b0;
        const int it = ok ? item : 0;
        if (it < S) { u_base = p.g_paths + (bb * (T + 1) + 1) * S + it; u_stride = S; }
        else if (it < 2 * S) { u_base = p.g_means + bb * T * S + (it - S); u_stride = S; }
        else if (it < 3 * S) { u_base = p.eps + bb * T * S + (it - 2 * S); u_stride = S; }
        else if (it < 3 * S + S * S) { u_base = p.g_chol + bb * T * S * S + (it - 3 * S); u_stride = S * S; }
        else { u_base = p.chol_raw + bb * T * NTRIL + (it - 3 * S - S * S); u_stride = NTRIL; }
    }
    auto dma_up = [&](int r) {
        const int rc = r < 0 ? 0 : r;
        __builtin_amdgcn_global_load_lds((const void *)(u_base + (int64_t)rc * u_stride), (__attribute__((address_space(3))) void *)(ubuf + (r & 3) * 64), 4, 0, 0);
    };
    auto load_t = [&](int m, f16x8 (&af)[6][2]) {
#pragma unroll
        for (int ks = 0; ks < 6; ++ks)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) af[ks][pl] = p.frags[(int64_t)m * kMpMatFrags + (((w * 6 + ks) * 2) + pl) * 64 + lane];
    };
    // this lane's UPL rows (replica rr) of a product's four, planes folded
    auto fold = [&](const f32x4 &A1, const f32x4 &A2, float (&R)[UPL]) {
        float s1[UPL], s2[UPL];
        if constexpr (UPL == 2) {
            s1[0] = rr ? A1[2] : A1[0]; s1[1] = rr ? A1[3] : A1[1]; s2[0] = rr ? A2[2] : A2[0]; s2[1] = rr ? A2[3] : A2[1];
        } else {
            const float l1 = (rr & 1) ? A1[1] : A1[0], h1 = (rr & 1) ? A1[3] : A1[2], l2 = (rr & 1) ? A2[1] : A2[0], h2 = (rr & 1) ? A2[3] : A2[2];
            s1[0] = (rr & 2) ? h1 : l1; s2[0] = (rr & 2) ? h2 : l2;
        }
#pragma unroll
        for (int r = 0; r < UPL; ++r) R[r] = fmaf(mp_row_shl<NP>(s1[r]) + s2[r], kMpLoInv, s1[r]);
    };
    auto matmul_sel = [&](const f16x8 (&af)[6][2], const f16x8 (&bf)[6], float (&R)[UPL]) {
        f32x4 A1 = {0.f, 0.f, 0.f, 0.f}, A2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) { A1 = mp_mfma(af[ks][0], bf[ks], A1); A2 = mp_mfma(af[ks][1], bf[ks], A2); }
        fold(A1, A2, R);
    };
    auto gate_grads = [&](const float (&d)[UPL], const float (&r)[UPL], const float (&u)[UPL], const float (&n)[UPL], const float (&cn)[UPL],
                          const float (&hp)[UPL], float (&dr)[UPL], float (&du)[UPL], float (&dn)[UPL], float (&dcn)[UPL], float (&carry)[UPL]) {
#pragma unroll
        for (int e = 0; e < UPL; ++e) {
            const float dnn = (1.0f - u[e]) * d[e], duu = (hp[e] - n[e]) * d[e];
            dn[e] = dnn * (1.0f - n[e] * n[e]);
            du[e] = duu * (u[e] * (1.0f - u[e]));
            dcn[e] = dn[e] * r[e];
            dr[e] = (dn[e] * cn[e]) * (r[e] * (1.0f - r[e]));
            carry[e] = u[e] * d[e];
        }
    };
    auto publish = [&](const float (&v)[UPL], int par, int l, int blk) {
        _Float16 a[UPL], c[UPL];
#pragma unroll
        for (int r = 0; r < UPL; ++r) mp_split_fast(v[r], a[r], c[r]);
        if (owner) {
            _Float16 *dh = (_Float16 *)&dbuf[par][l][blk][w >> 1][2 * (w & 1) + (q >> 1)][pc] + 4 * (q & 1) + rr * UPL;
            _Float16 *dl = (_Float16 *)&dbuf[par][l][blk][w >> 1][2 * (w & 1) + (q >> 1)][pc + NP] + 4 * (q & 1) + rr * UPL;
            if constexpr (UPL == 2) {
                typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
                *(f16x2 *)dh = f16x2{a[0], a[1]}; *(f16x2 *)dl = f16x2{c[0], c[1]};
            } else { *dh = a[0]; *dl = c[0]; }
        }
    };
    auto read_d = [&](int par, int l, bool hh, f16x8 (&bf)[6]) {    // (dr, du, dn) or, hh, (dr, du, dc_n)
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) bf[ks] = dbuf[par][l][(ks >> 1) == 2 ? (hh ? 3 : 2) : (ks >> 1)][ks & 1][q][colr];
    };
    auto store_d4 = [&](const float (&dr)[UPL], const float (&du)[UPL], const float (&dn)[UPL], const float (&dcn)[UPL], int t, int l) {
        // D4[b][t][l][{dr, du, dn, dc_n}][64]; four instructions, always ISSUED: every wave of every workgroup holds the owner lanes of
        // its group's first path, which exists (grid = ceil(B / NP)) -- the other lanes are masked, not redirected
        if (live) {
            float *o = p.D4 + (((int64_t)b * T + t) * L + l) * 256 + j0;
            mp_stu<UPL>(o, dr, Sg); mp_stu<UPL>(o + 64, du, Sg); mp_stu<UPL>(o + 128, dn, Sg); mp_stu<UPL>(o + 192, dcn, Sg);
        }
    };
    // saved activations of (t, layer) out of the ring: r, u, n, n_hh, and h of step t - 1 (zero before the first step)
    auto read_acts = [&](int t, int l, float (&a)[5][UPL]) {
        const float *rec = (const float *)(abuf + ((t & 3) * L + l) * ASLOT) + pc * 320 + j0;
        const float *prv = (const float *)(abuf + (((t - 1) & 3) * L + l) * ASLOT) + pc * 320 + j0;
#pragma unroll
        for (int k = 1; k < 5; ++k) mp_ldu<UPL>(a[k], rec + 64 * k);
        mp_ldu<UPL>(a[0], prv);
        if (t <= 0) {
#pragma unroll
            for (int r = 0; r < UPL; ++r) a[0][r] = 0.f;
        }
    };

    // records T - 1, T - 2, T - 3 before the first step
    for (int r = T - 1; r >= T - 3; --r) { dma_acts(r, role == 1 ? 1 : 0); if (role == 0 && w == 3) dma_up(r); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    if (role == 1) {
        // =================================================================== layer-1 waves
        f16x8 whh[6][2], wxs[6][2], wo[2];
        load_t(2, whh);
#pragma unroll
        for (int ks = 0; ks < 6; ++ks)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) wxs[ks][pl] = p.frags[kMpBwdWx + ((ks * 2) + pl) * 64 + lane];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) wo[pl] = p.frags[kMpBwdOut + ((w * 2) + pl) * 64 + lane];
        float dx[S], dh1[UPL];
#pragma unroll
        for (int i = 0; i < S; ++i) dx[i] = 0.f;
#pragma unroll
        for (int r = 0; r < UPL; ++r) dh1[r] = 0.f;
        // this wave's DO item: lane group q of waves 0 / 1 stores emission row 4 (w & 1) + q of its path (replica 0), everyone else the sink
        const int doi = 4 * (w & 1) + q;
        const bool do_ok = first && w < 2 && doi < NO;
        for (int t = T - 1; t >= 0; --t) {
            const int par = t & 1;
            dma_acts(t - 3, 1);
            if (t < T - 1) {   // d z_{t+1} through layer 0's input of step t + 1: W_x^T pi0  (backward.py:494-509)
                f16x8 bf[6];
                read_d(par ^ 1, 0, false, bf);
                f32x4 dxd;
                mp_matmul_t<NP>(wxs, bf, dxd);
#pragma unroll
                for (int i = 0; i < S; ++i) dx[i] += dxd[i];
            }
            float act[5][UPL];
            read_acts(t, 1, act);
            const float *ur = ubuf + (t & 3) * 64 + pc * 16;
            float cgp[S], cgm[S], cgl[S][S], ce[S], craw[NTRIL];
#pragma unroll
            for (int i = 0; i < S; ++i) {
                cgp[i] = ur[i]; cgm[i] = ur[S + i]; ce[i] = ur[2 * S + i];
#pragma unroll
                for (int c = 0; c < S; ++c) cgl[i][c] = ur[3 * S + i * S + c];
            }
#pragma unroll
            for (int r = 0; r < NTRIL; ++r) craw[r] = ur[3 * S + S * S + r];
            // ---- dO_t  (backward.py:278-334)
            float dO[NO];
#pragma unroll
            for (int i = 0; i < S; ++i) { dx[i] = fmaf(cgp[i], inv, dx[i]); dO[i] = fmaf(dx[i], p.dt, cgm[i] * inv); }
            {
                int k = 0;
#pragma unroll
                for (int i = 0; i < S; ++i)
#pragma unroll
                    for (int c = 0; c <= i; ++c, ++k) {
                        float dL = fmaf(dx[i] * ce[c], p.sqdt, cgl[i][c] * inv);
                        if (i == c && !(craw[k] >= p.diag_min || dL < 0.f)) dL = 0.f;    // bounds.py:20
                        dO[S + k] = dL;
                    }
            }
            {
                float v = dO[0];
#pragma unroll
                for (int r = 1; r < NO; ++r) v = doi == r ? dO[r] : v;
                float *dst = do_ok ? p.DO + ((int64_t)b * T + t) * NO + doi : sink;
                *dst = v * Sg;
            }
            if (q == 0 && owner && rr == 0) {   // dO as a B operand (k = emission row, in lane group 0), through this wave's private tile
                f16x8 hi = {0, 0, 0, 0, 0, 0, 0, 0}, lo = hi;
#pragma unroll
                for (int r = 0; r < NO; ++r) { _Float16 a, c; mp_split_fast(dO[r], a, c); hi[r] = a; lo[r] = c; }
                obuf[w][0][pc] = hi; obuf[w][0][pc + NP] = lo;
            }
            wave_lds_fence();
            const f16x8 ob = obuf[w][q][colr];
            float dcur[UPL];
            {
                f32x4 A1 = {0.f, 0.f, 0.f, 0.f}, A2 = A1;
                A1 = mp_mfma(wo[0], ob, A1); A2 = mp_mfma(wo[1], ob, A2);
                fold(A1, A2, dcur);                    // out_proj^T dO  (:296-349)
            }
            float d[UPL], dr[UPL], du[UPL], dn[UPL], dcn[UPL], carry[UPL];
#pragma unroll
            for (int r = 0; r < UPL; ++r) d[r] = dcur[r] + dh1[r];
            gate_grads(d, act[1], act[2], act[3], act[4], act[0], dr, du, dn, dcn, carry);
            publish(dr, par, 1, 0); publish(du, par, 1, 1); publish(dn, par, 1, 2); publish(dcn, par, 1, 3);
            barrier();                                 // 2: layer 1's gate gradients published
            {
                f16x8 bf[6];
                read_d(par, 1, true, bf);
                float rec[UPL];
                matmul_sel(whh, bf, rec);              // W_hh_l1^T ph1  (:96-105)
#pragma unroll
                for (int r = 0; r < UPL; ++r) dh1[r] = carry[r] + rec[r];
            }
            store_d4(dr, du, dn, dcn, t, 1);
            __builtin_amdgcn_sched_barrier(0);
            // record t - 2 (issued at step t + 1) has landed once at most the instructions issued after it are in flight:
            // DO + D4 of step t + 1, this step's DMA(s), DO and D4
            wait_vm(10 + APW);
            barrier();                                 // 3: layer 0's gate gradients published; records t - 2 visible
        }
        {   // the step-0 term of d z_0, then grad x0  (:620-624)
            f16x8 bf[6];
            read_d(0, 0, false, bf);
            f32x4 dxd;
            mp_matmul_t<NP>(wxs, bf, dxd);
            if (w == 0 && q == 0 && first) {
#pragma unroll
                for (int i = 0; i < S; ++i) p.g_x0[(int64_t)b * S + i] = (dx[i] + dxd[i] + p.g_paths[(int64_t)b * (T + 1) * S + i] * inv) * Sg;
            }
        }
        barrier();                                     // X: the gate-gradient tiles are free (layer 0 puts its theta sums there)
        barrier();                                     // Y: pairs with the layer-0 waves' barrier in front of the grad-theta sums
        return;
    }

    // ======================================================================= layer-0 waves
    f16x8 wih1[6][2], whh0[6][2];
    load_t(1, wih1);
    load_t(0, whh0);
    float dh0[UPL], ths[3][UPL];
#pragma unroll
    for (int r = 0; r < UPL; ++r) { dh0[r] = 0.f; ths[0][r] = 0.f; ths[1][r] = 0.f; ths[2][r] = 0.f; }
    for (int t = T - 1; t >= 0; --t) {
        const int par = t & 1;
        dma_acts(t - 3, 0);
        if (w == 3) dma_up(t - 3);
        float act[5][UPL];
        read_acts(t, 0, act);
        barrier();                                     // 2
        float dcur[UPL];
        {
            f16x8 bf[6];
            read_d(par, 1, false, bf);
            matmul_sel(wih1, bf, dcur);                // W_ih_l1^T pi1: gradient of layer 0's output  (:83-94)
        }
        float d[UPL], dr[UPL], du[UPL], dn[UPL], dcn[UPL], carry[UPL];
#pragma unroll
        for (int r = 0; r < UPL; ++r) d[r] = dcur[r] + dh0[r];
        gate_grads(d, act[1], act[2], act[3], act[4], act[0], dr, du, dn, dcn, carry);
        publish(dr, par, 0, 0); publish(du, par, 0, 1); publish(dn, par, 0, 2); publish(dcn, par, 0, 3);
#pragma unroll
        for (int r = 0; r < UPL; ++r) { ths[0][r] += dr[r]; ths[1][r] += du[r]; ths[2][r] += dn[r]; }   // sum_t pi0: grad theta = W_theta^T of it (:511-548)
        // records t - 2 (issued at step t + 1): behind them this wave issued [the upstream DMA,] D4 of step t + 1 and this step's DMA(s)
        wait_vm(w == 3 ? 5 + APW : 4 + APW);
        barrier();                                     // 3
        {
            f16x8 bf[6];
            read_d(par, 0, true, bf);
            float rec[UPL];
            matmul_sel(whh0, bf, rec);                 // W_hh_l0^T ph0  (:566-573)
#pragma unroll
            for (int r = 0; r < UPL; ++r) dh0[r] = carry[r] + rec[r];
        }
        store_d4(dr, du, dn, dcn, t, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    // grad theta[b][e] = sum_rows W_ih_l0[row][S + C + e] * sum_t pi0[row]: the sums go through LDS (the gate-gradient tiles are free now)
    float *tsum = (float *)mps_lds;                    // [NP][192]
    barrier();                                         // X: the layer-1 waves have read dbuf[0][0] (step 0) for grad x0
    if (owner) {
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int r = 0; r < UPL; ++r) tsum[pc * 192 + g * 64 + j0 + r] = ths[g][r];
    }
    barrier();                                         // Y
    for (int o = tid; o < NP * p.P; o += 256) {
        const int path = o / p.P, e = o - path * p.P, bb = blockIdx.x * NP + path;
        if (bb < p.B) {
            float acc = 0.f;
            for (int row = 0; row < 192; ++row) acc = fmaf(p.W_ih0[(int64_t)row * I + S + p.C + e], tsum[path * 192 + row], acc);
            p.g_theta[(int64_t)bb * p.P + e] = acc * Sg;
        }
    }
}
template <int S, int NP> static constexpr size_t mps_lds_bytes() {
    constexpr int APIECES = (NP * 1280 + 1023) / 1024, APW = (APIECES + 3) / 4, ASLOT = APW * 4 * 1024;
    return 32768 + 4096 + (size_t)4 * 2 * ASLOT + 4 * 64 * 4;
}

// ---------------------------------------------------------------------------------------------------------------------------
size_t mp_frag_bytes(int L, int S) {
    const int no = S + S * (S + 1) / 2, nto = (no + 3) / 4;
    return ((size_t)(2 * L - 1) * kMpMatFrags + (size_t)nto * 2 * 2 * 64) * sizeof(f16x8);
}

bool mp_applicable(int H, int L, int S) { return H == 64 && L >= 1 && L <= 2 && S >= 1 && S <= 2; }
bool mp_bwd_applicable(int H, int L, int S) { return H == 64 && L == 2 && S >= 1 && S <= 2; }
size_t mp_bwd_frag_bytes(void) { return (size_t)kMpBwdTotal * sizeof(f16x8) + 256 + 4096 + 64; }   // fragments + the absmax word + the spread sweep's store sink (512 x 8 bytes)

int launch_head_bwd_mp(const MpBwdLaunch &a, hipStream_t s, void (*mark)(int, int, hipStream_t)) {
    const int no = a.S + a.S * (a.S + 1) / 2;
    f16x8 *frags = (f16x8 *)a.frags;
    unsigned *absmax = (unsigned *)(frags + kMpBwdTotal);
    MpBwdPrep q = {};
    q.S = a.S; q.no = no; q.I = a.S + a.C + a.P;
    q.W_hh0 = a.W_hh0; q.W_ih1 = a.W_ih_st; q.W_hh1 = a.W_hh_st; q.W_ih0 = a.W_ih0; q.out_W = a.out_W; q.frags = frags;
    hipLaunchKernelGGL(mp_bwd_prep_kernel, dim3((3 * 1536 + 384 + 256 + 255) / 256), dim3(256), 0, s, q);
    VSDE_CHECK_HIP(hipMemsetAsync(absmax, 0, sizeof(unsigned), s));
    const int64_t n1 = (int64_t)a.B * (a.T + 1) * a.S, n2 = (int64_t)a.B * a.T * a.S, n3 = (int64_t)a.B * a.T * a.S * a.S;
    int blocks = (int)((n1 + n2 + n3 + 1023) / 1024);
    blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
    hipLaunchKernelGGL(mp_absmax_kernel, dim3(blocks), dim3(256), 0, s, a.g_paths, n1, a.g_means, n2, a.g_chol, n3, absmax);
    MpBwdParams p = {};
    p.B = a.B; p.T = a.T; p.P = a.P; p.C = a.C;
    p.g_paths = a.g_paths; p.g_means = a.g_means; p.g_chol = a.g_chol; p.eps = a.eps; p.chol_raw = a.chol_raw; p.acts = a.acts;
    p.W_ih0 = a.W_ih0; p.frags = frags; p.absmax = absmax;
    p.dt = a.dt; p.sqdt = a.sqdt; p.diag_min = a.diag_min;
    p.D4 = a.D4; p.DO = a.DO; p.g_x0 = a.g_x0; p.g_theta = a.g_theta;
    // groups of 2 (up to 512 paths) / 4 (up to 1024): the spread sweep (head_bwd_mps_kernel); VSDE_MP_BWD_SPREAD=0: round 4's kernel (A/B runs)
    static int spread = -1;
    if (spread < 0) spread = (int)vsde_knob("VSDE_MP_BWD_SPREAD", 1);
    int np = a.np == 2 || a.np == 4 || a.np == 8 ? a.np : (a.np == 16 ? 8 : (a.B <= 512 ? 2 : (a.B <= 1024 ? 4 : 8)));
    if (!spread && np == 2) np = 4;
    const dim3 grid((a.B + np - 1) / np), block(512);
    if (mark) mark(1, 0, s);
    if (spread && (np == 2 || np == 4)) {
#define VSDE_MPS_LAUNCH(SS, NN)                                                                                                       \
        do {                                                                                                                           \
            auto kern = head_bwd_mps_kernel<SS, NN>;                                                                                   \
            const size_t lds = mps_lds_bytes<SS, NN>();                                                                                \
            VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));            \
            hipLaunchKernelGGL(kern, grid, block, lds, s, p);                                                                          \
        } while (0)
        if (a.S == 1) { if (np == 2) VSDE_MPS_LAUNCH(1, 2); else VSDE_MPS_LAUNCH(1, 4); }
        else { if (np == 2) VSDE_MPS_LAUNCH(2, 2); else VSDE_MPS_LAUNCH(2, 4); }
#undef VSDE_MPS_LAUNCH
        if (mark) mark(1, 1, s);
        VSDE_CHECK_HIP(hipGetLastError());
        return 0;
    }
#ifdef VSDE_ABLATIONS
    static int abl = -1;
    if (abl < 0) abl = ablation_env("VSDE_MP_BWD_ABL");
    if (abl && a.S == 2 && np == 4) {
        switch (abl) {
            case 1: hipLaunchKernelGGL((head_bwd_mp_kernel<2, 4, 1>), grid, block, 0, s, p); break;
            case 2: hipLaunchKernelGGL((head_bwd_mp_kernel<2, 4, 2>), grid, block, 0, s, p); break;
            case 3: hipLaunchKernelGGL((head_bwd_mp_kernel<2, 4, 3>), grid, block, 0, s, p); break;
            case 4: hipLaunchKernelGGL((head_bwd_mp_kernel<2, 4, 4>), grid, block, 0, s, p); break;
            case 8: hipLaunchKernelGGL((head_bwd_mp_kernel<2, 4, 8>), grid, block, 0, s, p); break;
            case 7: hipLaunchKernelGGL((head_bwd_mp_kernel<2, 4, 7>), grid, block, 0, s, p); break;
            case 16: hipLaunchKernelGGL((head_bwd_mp_kernel<2, 4, 16>), grid, block, 0, s, p); break;   // 16: upstream-gradient loads only
            case 32: hipLaunchKernelGGL((head_bwd_mp_kernel<2, 4, 32>), grid, block, 0, s, p); break;   // 32: saved-activation loads only
            default: hipLaunchKernelGGL((head_bwd_mp_kernel<2, 4, 15>), grid, block, 0, s, p); break;
        }
    } else
    if (a.S == 1) { if (np == 4) hipLaunchKernelGGL((head_bwd_mp_kernel<1, 4>), grid, block, 0, s, p); else hipLaunchKernelGGL((head_bwd_mp_kernel<1, 8>), grid, block, 0, s, p); }
    else { if (np == 4) hipLaunchKernelGGL((head_bwd_mp_kernel<2, 4>), grid, block, 0, s, p); else hipLaunchKernelGGL((head_bwd_mp_kernel<2, 8>), grid, block, 0, s, p); }
#else   // the shipped library: groups of 8 only get here (2 / 4 took the spread sweep above)
    if (a.S == 1) hipLaunchKernelGGL((head_bwd_mp_kernel<1, 8>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((head_bwd_mp_kernel<2, 8>), grid, block, 0, s, p);
#endif
    if (mark) mark(1, 1, s);
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}

// The multi-path kernels hold the recurrent weights as split f16 fragments: a (scaled) weight beyond the f16 range -- |W| > 2.2e4
// for gate rows, 6.5e4 for the emission rows; only a diverged run gets there -- would become inf in the products.  mp_prep_kernel
// raises a sticky flag in host-mapped memory when it meets one; the dispatcher (vsde_head.hip::use_mp) reads it WITHOUT a device
// synchronisation before every launch and routes to the fp32 four-waves-per-path kernels from then on (the launch that raised it
// has already produced non-finite outputs, which the trainer's non-finite check / GradScaler skip as for any diverged step).
static int *g_mp_overflow = nullptr;
static int *mp_overflow_flag() {
    if (!g_mp_overflow) {
        if (hipHostMalloc((void **)&g_mp_overflow, sizeof(int), hipHostMallocMapped) != hipSuccess) { g_mp_overflow = nullptr; return nullptr; }
        *g_mp_overflow = 0;
    }
    return g_mp_overflow;
}
void mp_clear_overflow() {
    if (g_mp_overflow) *(volatile int *)g_mp_overflow = 0;
}
bool mp_weights_overflowed() {
    static bool told = false;
    const bool o = g_mp_overflow != nullptr && *(volatile int *)g_mp_overflow != 0;
    if (o && !told) {
        told = true;
        fprintf(stderr, "libvsde_hip: a GRU head weight left the f16 range of the multi-path MFMA kernels (|W| > 2.2e4): "
                        "the fp32 four-waves-per-path kernels take over for the rest of the process\n");
    }
    return o;
}

int launch_head_fwd_mp(const MpLaunch &a, hipStream_t s, void (*mark)(int, int, hipStream_t)) {
    const int no = a.S + a.S * (a.S + 1) / 2, nto = (no + 3) / 4, nm = 2 * a.L - 1;
    MpPrep q = {};
    q.nm = nm; q.no = no; q.nto = nto;
    q.W[0] = a.W_hh0; q.W[1] = a.W_ih_st; q.W[2] = a.W_hh_st; q.out_W = a.out_W;
    q.frags = (f16x8 *)a.frags;
    q.overflow = mp_overflow_flag();
    const int nthr = nm * 1536 + nto * 128;
    hipLaunchKernelGGL(mp_prep_kernel, dim3((nthr + 255) / 256), dim3(256), 0, s, q);
    MpParams p = {};
    p.B = a.B; p.T = a.T; p.P = a.P; p.C = a.C;
    p.x0 = a.x0; p.theta = a.theta; p.eps = a.eps; p.G = a.G; p.W_ih0 = a.W_ih0;
    p.b_hh0 = a.b_hh0; p.b_ih1 = a.b_ih_st; p.b_hh1 = a.b_hh_st; p.out_b = a.out_b;
    p.frags = (const f16x8 *)a.frags;
    p.dt = a.dt; p.sqdt = a.sqdt; p.diag_min = a.diag_min;
    p.paths = a.paths; p.means = a.means; p.chol = a.chol; p.chol_raw = a.chol_raw; p.acts = a.acts;
    { static int abl = -1; if (abl < 0) abl = ablation_env("VSDE_MP_FWD_ABL"); p.abl = abl; }
    // paths per workgroup: 16 fills the matrix pipe (large batches); a small batch takes 4 or 8 so that its groups spread over more CUs --
    // the time of a launch is T x one step's latency whatever the group size, and ONE CU's vector-memory pipe would have to carry the
    // saved activations of all its paths (41 KB per step for 16 paths: +30 % at 512 paths, profiles/r04_head_mp.txt)
    static int spread = -1;   // VSDE_MP_SPREAD: see below
    if (spread < 0) spread = (int)vsde_knob("VSDE_MP_SPREAD", 1);
    int np = a.np;
    if (np != 2 && np != 4 && np != 8 && np != 16) np = (a.B <= 512 && spread) ? 2 : (a.B <= 1024 ? 4 : (a.B <= 2048 ? 8 : 16));
    const dim3 grid((a.B + np - 1) / np), block(256 * a.L);
    if (mark) mark(0, 0, s);
    // groups of 4 paths: the spread form (every path 4 times across the operand columns, ONE unit per lane; see mp_matmul): 519 -> 442 us
    // (training) and 479 -> 394 us (sampling) at 512 paths.  Groups of 8 (two units per lane) measure 3 % slower than the side-by-side
    // planes (842 vs 817 us at 2048 paths: a third more MFMAs for half the gate work) and stay as they were.
    // VSDE_MP_SPREAD=0: the side-by-side planes of round 4 everywhere, =2: the spread form for groups of 8 as well (A/B runs)
#define VSDE_MP_LAUNCH_(LL, SV, SS, NN, UU) hipLaunchKernelGGL((head_fwd_mp_kernel<LL, SV, SS, NN, UU>), grid, block, 0, s, p)
#ifdef VSDE_ABLATIONS
#define VSDE_MP_LAUNCH(LL, SV, SS)                                  \
    do {                                                            \
        if (np == 2) VSDE_MP_LAUNCH_(LL, SV, SS, 2, 1);             \
        else if (np == 4 && spread == 3) VSDE_MP_LAUNCH_(LL, SV, SS, 4, 2); \
        else if (np == 4 && spread) VSDE_MP_LAUNCH_(LL, SV, SS, 4, 1);   \
        else if (np == 4) VSDE_MP_LAUNCH_(LL, SV, SS, 4, 4);        \
        else if (np == 8 && spread >= 2) VSDE_MP_LAUNCH_(LL, SV, SS, 8, 2); \
        else if (np == 8) VSDE_MP_LAUNCH_(LL, SV, SS, 8, 4);        \
        else VSDE_MP_LAUNCH_(LL, SV, SS, 16, 4);                    \
    } while (0)
#else   // the shipped library: the forms the dispatcher takes (spread for groups of 2 / 4, side-by-side planes for 8 / 16)
#define VSDE_MP_LAUNCH(LL, SV, SS)                                  \
    do {                                                            \
        if (np == 2) VSDE_MP_LAUNCH_(LL, SV, SS, 2, 1);             \
        else if (np == 4) VSDE_MP_LAUNCH_(LL, SV, SS, 4, 1);        \
        else if (np == 8) VSDE_MP_LAUNCH_(LL, SV, SS, 8, 4);        \
        else VSDE_MP_LAUNCH_(LL, SV, SS, 16, 4);                    \
    } while (0)
#endif
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
