// Self-attention core of the SiT observation encoder on bf16 MFMA (gfx950), for the encoder's shape class:
// short sequences (N <= 576 tokens: the observation grid + 1), head_dim 64, many (batch, head) pairs.
//
//   O[b,n,h,:] = softmax_j(scale * <q[b,n,h,:], k[b,j,h,:]>) v[b,j,h,:]        (reference: primitives/attn.py:104-106,
//                                                                               F.scaled_dot_product_attention)
//
// The library flash kernels tile for long sequences and reach ~10 % of the MFMA peak here.  At these lengths the whole
// K and V of one (batch, head) fit in LDS (N = 401: 114 KB), so one workgroup owns one (b, h):
//   * K is staged row-major [key][d] (the B^T... A operand of S^T = K Q^T wants d contiguous per lane: no transpose),
//     V is staged transposed [d][key] through an in-register 8x8 bf16 transpose (O^T = V^T P^T wants keys contiguous);
//   * the products are computed "swapped" (S^T = K Q^T, O^T = V^T P^T) so a lane owns ONE query column: the softmax
//     statistics are in-register reductions plus a single exchange between lane l and l+32, and P^T leaves the first
//     MFMA already in the B-operand layout of the second (no LDS round trip for the probabilities);
//   * softmax without a running-maximum rescale chain: the shift is the Cauchy-Schwarz bound |q_i| max_j |k_j| >= max_j
//     <q_i, k_j> (softmax is shift-invariant, so this is exact), available before the first product; only if that
//     bound is so loose that every term could underflow (scale |q||k| > 27, never the case behind the encoder's
//     QK-RMS-norm) a first pass computes the true row maxima.
// Token-major layout [B][N][H][64] for q, k, v, o (what qk_norm_rope writes and gate_merge reads).
#include <stdlib.h>

#include "vsde_common.h"

namespace vsde {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16v2 __attribute__((ext_vector_type(2)));
typedef float f32v2 __attribute__((ext_vector_type(2)));

constexpr int AT_D = 64;     // head dim
constexpr int AT_KLD = 72;   // LDS row stride of K in bf16 (144 B: conflict-free ds_read_b128 fragments)
constexpr int AT_MAXN = 544; // forward: 544*272 + 512 B of LDS; backward: 2*544*144 + 8*544 B  (<= 160 KB)

struct AttnParams {
    const uint16_t *q, *k, *v;  // [B][N][H][64] bf16
    uint16_t *o;                // [B][N][H][64] bf16
    float *lse;                 // [B][H][N] natural-log sum-exp of the scaled scores
    int N, H;
    int npad;                   // N rounded up to 32
    int vld;                    // LDS row stride of V^T in bf16: npad + 4 (stride/2 dwords = 2*odd mod 64: conflict-free b64)
    float scale, scale_log2e;
    const uint16_t *gate;       // optional output gate [B*N][ldg] (64 factors per token, shared by the heads, = rnd(sigmoid(logit))): o *= gate
    int64_t ldg;
    int64_t pairs;              // B * H (the persistent kernel's loop bound)
    int dbg;                    // ring kernel, timing-only ablations (VSDE_ATTN_RING_DBG): 1 = the producer loads only the first pair, 2 = no tile-step
                                // barriers, 4 = the consumers skip their tile steps, 8 = no epilogue stores
    long long *trace;           // ABL = 16 (vsde_attn_debug_trace): per-wave phase cycle sums of workgroup 0, [12 waves][4] + pairs
};

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    f32v2 f = {a, b};
    bf16v2 r = __builtin_convertvector(f, bf16v2);  // v_cvt_pk_bf16_f32
    return *(uint32_t *)&r;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float bfl(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bfh(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

typedef short f8bf4 __attribute__((ext_vector_type(4)));
// ds_read_b64_tr_b16: within a 16-lane group, lane m supplies the address of 4 contiguous bf16 = row m / 4, columns 4 (m % 4) .. of a
// [4][16] block, and lane i receives column i (rows 0..3)
__device__ __forceinline__ uint2 f8_read_tr(const uint16_t *ptr) {
    f8bf4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) f8bf4 *)ptr);
    return *(uint2 *)&r;
}

// the staging loads of one (batch, head) pair into registers: K and V rows as 16-byte chunks (8 adjacent lanes = one 128-byte row);
// staging thread pt of PT; base_ = element offset of the pair's first row
template <int KIT>
__device__ __forceinline__ void fwd_request(uint4 (&kreg)[KIT], uint4 (&vreg)[KIT], const AttnParams &p, int64_t base_, int pt, int PT,
                                            bool stager, int N, int npad, int64_t ts) {
    const uint16_t *kb = p.k + base_, *vb = p.v + base_;
#pragma unroll
    for (int it = 0; it < KIT; ++it) {
        const int i = pt + it * PT, n = i >> 3, c = i & 7;
        const bool ok = stager && i < npad * 8 && n < N;
        kreg[it] = ok ? *(const uint4 *)(kb + n * ts + c * 8) : make_uint4(0, 0, 0, 0);
        vreg[it] = ok ? *(const uint4 *)(vb + n * ts + c * 8) : make_uint4(0, 0, 0, 0);
    }
}

// PERSIST (N <= 416, i.e. at most one wave with a second query block): the grid is one workgroup per CU and a workgroup walks
// the (batch, head) pairs blockIdx.x, blockIdx.x + gridDim.x, ...  A wave that has finished its last query block of the current
// pair requests the K / V rows of the NEXT pair into its staging registers right away (they are free: the block's accumulators have
// been stored) and parks at the barrier; the wave that owns the 13th block is still computing then, so the HBM latency and part of
// the transfer of the next pair's operands run behind it instead of in front of everybody.  Only the waves without a second
// block stage (704 of 768 threads at N = 401).
// ... and their way into LDS, both row-major as they are ([key][AT_KLD]; round 5: V used to go through an in-register 8 x 8 transpose
// into a V^T image -- the PV product now reads its V^T fragments out of the row-major tile with ds_read_b64_tr_b16); the wave's
// maximum squared key norm goes to kred[wave] (the softmax shift needs max_j |k_j|)
template <int KIT>
__device__ __forceinline__ void fwd_commit(const uint4 (&kreg)[KIT], const uint4 (&vreg)[KIT], uint16_t *Ks, uint16_t *Vs, float *kred, int pt,
                                           int PT, bool stager, int npad, int lane, int wave) {
    float kss_max = 0.f;  // max_j |k_j|^2 (8 adjacent lanes hold one key row)
#pragma unroll
    for (int it = 0; it < KIT; ++it) {
        const int i = pt + it * PT, n = i >> 3, c = i & 7;
        if (stager && i < npad * 8) {
            *(uint4 *)(Ks + n * AT_KLD + c * 8) = kreg[it];
            *(uint4 *)(Vs + n * AT_KLD + c * 8) = vreg[it];
        }
        const uint32_t w[4] = {kreg[it].x, kreg[it].y, kreg[it].z, kreg[it].w};
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float lo = __uint_as_float(w[e] << 16), hi = __uint_as_float(w[e] & 0xffff0000u);
            ss = fmaf(lo, lo, fmaf(hi, hi, ss));
        }
        ss += xor_lane<1>(ss); ss += xor_lane<2>(ss); ss += xor_lane<4>(ss);
        kss_max = fmaxf(kss_max, ss);
    }
    kss_max = fmaxf(kss_max, xor_lane<8>(kss_max)); kss_max = fmaxf(kss_max, xor_lane<16>(kss_max));   // (8 adjacent lanes already agree)
    {   // lanes l and l ^ 32 without a lane index (a __shfl_xor's index registers, set up once, are spilled across the tile loop)
        const uint32_t u = __float_as_uint(kss_max);
        const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        kss_max = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    if (lane == 0) kred[wave] = kss_max;
}

// ABL: timing-only ablations of the tile loop's VALU work (results are wrong): 1 = exponentials of the raw scores (no scale / shift
// fma), 2 = no row-sum adds, 4 = no exponentials at all (tools/attn_core_bench.py with VSDE_ATTN_FWD_ABL)
// NT = 256 (non-persistent, N <= 128): four-wave workgroups, three per CU (see AT_BT_SMALL)
template <bool PERSIST, int ABL = 0, int NT = 768>
__global__ void __launch_bounds__(NT, 768 / NT) attn_fwd_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) uint16_t asmem[];
    uint16_t *Ks = asmem;                      // [npad][AT_KLD]
    uint16_t *Vs = asmem + p.npad * AT_KLD;    // [npad][AT_KLD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = p.N, npad = p.npad;
    const int64_t ts = (int64_t)p.H * AT_D;    // token stride in elements
    constexpr int NW = NT / 64, MAXN = NT == 768 ? AT_MAXN : 128;
    static_assert(NT == 768 || !PERSIST, "the persistent form is built for 12-wave workgroups");
    const int two_round = PERSIST ? ((npad >> 5) > NW ? (npad >> 5) - NW : 0) : 0;   // waves 0 .. two_round - 1 own two query blocks
    int pt = tid - 64 * two_round;                                                    // staging thread index ...
    const int PT = NT - 64 * two_round;                                               // ... and count
    const bool stager = pt >= 0;
    // ---- staging of K [key][d] and V^T [d][key]: every global load is issued before the first use; the staging registers live
    // only inside VSDE_FWD_STAGE (a loop-carried register set is spilled across the tile loop by hipcc)
    constexpr int KIT = PERSIST ? (416 * 8 + 703) / 704 : (MAXN * 8 + NT - 1) / NT;
    __shared__ float kred[NW];
    // the wave's first query block of a pair: one dword of each of its 64-byte half rows is touched together with the pair's K / V
    // requests, so that the fragment loads after the barrier are served by the cache (the fragments themselves, requested here,
    // are spilled: 16 more registers do not fit beside the staging's)
    uint32_t qtouch = 0u;
    auto touch_q = [&](int64_t base_) {
        int ln = tid & 63;
        asm volatile("" : "+v"(ln));
        const int pq = wave * 32 + (ln & 31);
        qtouch = *(const uint32_t *)(p.q + base_ + (pq < N ? pq : N - 1) * ts + (ln >> 5) * 32);
    };
    // (Round 5, tools/attn_trace.py: per pair of ~36 k cycles the tile loops take 16 k, the staging 11 k, the blocks' prologues -- the
    //  row-strided q fragment loads in front of the first MFMA -- 4-7 k, the epilogues 1-3 k.  Requesting the wave's q fragments behind
    //  the staging commit, in front of the barrier every wave waits at anyway, cost 26 spilled registers and 20 % of the kernel: the 16
    //  registers do not fit anywhere around the tile loop.  Second form: the q rows by LDS-DMA into 4 KB per wave behind K and V^T (a
    //  swizzled [32][8 x 16 B] image, conflict-free fragment reads), requested when the wave is done with the previous pair: 19 spilled
    //  registers for its addressing, 203 vs 175 us.  The kernel sits at its 168-register cap; both dropped.  Third attempt, after V
    //  went row-major into LDS (40 staging registers instead of 52 + 32 for the transposes): fragments requested behind the commit and
    //  carried into the block loop -- 14 spilled registers around the staging phase, 192 vs 168 us.)
#define VSDE_FWD_STAGE(base_, first_)                                                                          \
    do {                                                                                                       \
        uint4 kreg[KIT], vreg[KIT];                                                                            \
        asm volatile("" : "+v"(pt));   /* staging addresses are recomputed per pair, not kept across the tile loop */ \
        touch_q(base_);                                                                                        \
        fwd_request<KIT>(kreg, vreg, p, (base_), pt, PT, stager, N, npad, ts);                                 \
        if (!(first_)) __syncthreads();   /* everyone is done with the previous pair's K / V */               \
        fwd_commit<KIT>(kreg, vreg, Ks, Vs, kred, pt, PT, stager, npad, lane, wave);                           \
        __syncthreads();                                                                                       \
        asm volatile("" ::"v"(qtouch));                                                                        \
    } while (0)
    const int64_t nheads = p.pairs;              // total (batch, head) pairs
    int64_t head = blockIdx.x;
    // (batch, head) of the pair at hand: stepped by gridDim.x, no 64-bit division per pair and wave
    int b = (int)(head / p.H), hh = (int)(head - (int64_t)b * p.H);
    const int step_b = PERSIST ? (int)(gridDim.x / p.H) : 0, step_h = PERSIST ? (int)(gridDim.x - (unsigned)step_b * p.H) : 0;
    long long ph[4] = {0, 0, 0, 0}, last_ = 0, npairs_ = 0;
    if constexpr ((ABL & 16) != 0) last_ = __builtin_readcyclecounter();
#define VSDE_AT_STAMP(k_) do { if constexpr ((ABL & 16) != 0) { const long long now_ = __builtin_readcyclecounter(); ph[k_] += now_ - last_; last_ = now_; } } while (0)
    VSDE_FWD_STAGE(((int64_t)b * N * p.H + hh) * AT_D, true);
    VSDE_AT_STAMP(0);
  while (true) {
    const int64_t base = ((int64_t)b * N * p.H + hh) * AT_D;
    const uint16_t *qb = p.q + base;
    float kss_max = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) kss_max = fmaxf(kss_max, kred[w]);
    const float kmax = sqrtf(kss_max);
    const int64_t next = head + gridDim.x;
    const bool has_next = PERSIST && next < nheads;
    int nb = b + step_b, nh = hh + step_h;
    if (nh >= p.H) { nh -= p.H; ++nb; }

    int fr = lane & 31, h2 = lane >> 5;
    // PERSIST: every LDS address of the tile loop below is invariant across the pair loop; hoisted out of it (LICM) they fill the
    // register file (194 spilled VGPRs).  An opaque copy of the lane coordinates per pair keeps them where they are used.
    if constexpr (PERSIST) asm volatile("" : "+v"(fr), "+v"(h2));
    const int nkt = npad >> 5, nqb = npad >> 5;
    const bool ragged = (N & 31) != 0;
    for (int qblk = wave; qblk < nqb; qblk += NW) {
        const int query = qblk * 32 + fr;
        const bool qok = query < N;
        // the first 8 bytes of this lane's share of the token's gate row, requested a whole round before the epilogue needs the
        // row: the 128-byte line is in the cache by then (the tile loop itself issues no global loads)
        uint2 gate0 = make_uint2(0u, 0u);
        if (p.gate != nullptr && qok) gate0 = *(const uint2 *)(p.gate + ((int64_t)b * N + query) * p.ldg + 4 * h2);
        bf16x8 qf[4];  // B operand of S^T = K Q^T: column = query, k = d
        float qss = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            uint4 t = make_uint4(0, 0, 0, 0);
            if (qok) t = *(const uint4 *)(qb + query * ts + ks * 16 + h2 * 8);
            qf[ks] = *(bf16x8 *)&t;
            const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float lo = __uint_as_float(w[e] << 16), hi = __uint_as_float(w[e] & 0xffff0000u);
                qss = fmaf(lo, lo, fmaf(hi, hi, qss));
            }
        }
        qss = sum_xor32(qss);
        // shift of the softmax: |q| max|k| >= every score of this query (x 1.0001 against rounding of the norms)
        float mx = sqrtf(qss) * kmax * 1.0001f;
        const bool exact = !__all(mx * p.scale_log2e <= 40.0f);  // wave-uniform
        // ---- pass 1 (rare): true maximum score of every query ----------------------------------------------
        if (exact) mx = -INFINITY;
        for (int kt = 0; exact && kt < nkt; ++kt) {
            f32x16 s = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            const uint16_t *krow = Ks + (kt * 32 + fr) * AT_KLD + h2 * 8;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8 *)(krow + ks * 16), qf[ks], s, 0, 0, 0);
            if (ragged && kt == nkt - 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2 >= N) s[r] = -INFINITY;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[r]);
        }
        if (exact) {
            const uint32_t u = __float_as_uint(mx);
            const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }  // the other half-wave holds the other 16 keys of every tile
        // ---- pass 2: P^T = exp2(c (S^T - max)), O^T += V^T P^T -------------------------------------------
        const float c2 = p.scale_log2e, mc = mx * c2;
        float lsum = 0.f, lsum2 = 0.f;
        f32x16 o0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, o1 = o0;
        // Software pipeline, two key tiles per trip (ping-pong score registers, no copies).  One step, for tile kt:
        //   1. issue the LDS reads of K tile kt+1 and of the V^T columns of tile kt,
        //   2. exponentiate score tile kt on the VALU while those reads are in flight,
        //   3. score tile kt+1 and the PV product of tile kt on the matrix pipe.
        auto step = [&](f32x16 &scur, f32x16 &snxt, int kt) {
            const int ktn = min(kt + 1, nkt - 1);  // the last step recomputes its own tile (result unused)
            const uint16_t *krow = Ks + (ktn * 32 + fr) * AT_KLD + h2 * 8;
            bf16x8 kf[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) kf[ks] = *(const bf16x8 *)(krow + ks * 16);
            // V^T fragments (row = channel dt * 32 + fr, k = the keys 4 h2 + {0..3, 8..11 | 16..19, 24..27} of the tile, the order of P's
            // registers) out of the row-major tile by the transposing read (see accumulate_transposed)
            uint2 vf[8];
            const uint16_t *vsrc = Vs + (kt * 32 + 4 * h2 + ((fr & 15) >> 2)) * AT_KLD + ((fr >> 4) & 1) * 16 + (fr & 3) * 4;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int x = 0; x < 4; ++x) vf[dt * 4 + x] = f8_read_tr(vsrc + dt * 32 + x * 8 * AT_KLD);
            __builtin_amdgcn_sched_barrier(0);
            if (ragged && kt == nkt - 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2 >= N) scur[r] = -INFINITY;
            }
            float pr[16];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                if constexpr (ABL & 4) { pr[r] = scur[r]; pr[r + 1] = scur[r + 1]; }
                else if constexpr (ABL & 1) { pr[r] = fast_exp2(scur[r]); pr[r + 1] = fast_exp2(scur[r + 1]); }
                else { pr[r] = fast_exp2(fmaf(scur[r], c2, -mc)); pr[r + 1] = fast_exp2(fmaf(scur[r + 1], c2, -mc)); }
                if constexpr (!(ABL & 2)) { lsum += pr[r]; lsum2 += pr[r + 1]; }
            }
            // registers 0..7 are keys {4h2..4h2+3, 8+4h2..11+4h2} of the tile, 8..15 the same + 16: used as the two
            // k-steps of the second product, with V^T read at exactly those key columns
            uint4 pw0 = make_uint4(pack_bf16(pr[0], pr[1]), pack_bf16(pr[2], pr[3]), pack_bf16(pr[4], pr[5]), pack_bf16(pr[6], pr[7]));
            uint4 pw1 = make_uint4(pack_bf16(pr[8], pr[9]), pack_bf16(pr[10], pr[11]), pack_bf16(pr[12], pr[13]), pack_bf16(pr[14], pr[15]));
            const bf16x8 pb0 = *(bf16x8 *)&pw0, pb1 = *(bf16x8 *)&pw1;
            __builtin_amdgcn_sched_barrier(0);
            f32x16 t = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], t, 0, 0, 0);
            snxt = t;
            uint4 aw;
            aw = make_uint4(vf[0].x, vf[0].y, vf[1].x, vf[1].y); o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8 *)&aw, pb0, o0, 0, 0, 0);
            aw = make_uint4(vf[4].x, vf[4].y, vf[5].x, vf[5].y); o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8 *)&aw, pb0, o1, 0, 0, 0);
            aw = make_uint4(vf[2].x, vf[2].y, vf[3].x, vf[3].y); o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8 *)&aw, pb1, o0, 0, 0, 0);
            aw = make_uint4(vf[6].x, vf[6].y, vf[7].x, vf[7].y); o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8 *)&aw, pb1, o1, 0, 0, 0);
        };
        f32x16 sa = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, sb = sa;
        VSDE_AT_STAMP(1);
        {
            const uint16_t *krow = Ks + fr * AT_KLD + h2 * 8;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8 *)(krow + ks * 16), qf[ks], sa, 0, 0, 0);
        }
        for (int kt = 0; kt < nkt; kt += 2) {
            step(sa, sb, kt);
            if (kt + 1 < nkt) step(sb, sa, kt + 1);
        }
        lsum += lsum2;
        lsum = sum_xor32(lsum);
        VSDE_AT_STAMP(2);
        if (qok) {
            const float inv = 1.0f / lsum;
            uint16_t *orow = p.o + base + query * ts;
            if (p.gate != nullptr) {   // sigmoid output gate + head merge (primitives/attn.py:107-113) folded into the store: o is the merged [B,N,(h d)] row
                const uint16_t *grow = p.gate + ((int64_t)b * N + query) * p.ldg;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d0 = 8 * g + 4 * h2;
                    const uint2 ga = g == 0 ? gate0 : *(const uint2 *)(grow + d0), gb = *(const uint2 *)(grow + 32 + d0);
                    const float sa[4] = {bfl(ga.x), bfh(ga.x), bfl(ga.y), bfh(ga.y)};   // the sigmoid was applied where the gate was produced
                    const float sb[4] = {bfl(gb.x), bfh(gb.x), bfl(gb.y), bfh(gb.y)};
#pragma unroll
                    for (int i = 0; i < 4; ++i) { o0[4 * g + i] *= sa[i]; o1[4 * g + i] *= sb[i]; }
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {  // registers 4g..4g+3 = rows d = 8g + 4h2 + 0..3 (o0) and 32 + ... (o1)
                const int d0 = 8 * g + 4 * h2;
                *(uint2 *)(orow + d0) = make_uint2(pack_bf16(o0[4 * g] * inv, o0[4 * g + 1] * inv), pack_bf16(o0[4 * g + 2] * inv, o0[4 * g + 3] * inv));
                *(uint2 *)(orow + 32 + d0) = make_uint2(pack_bf16(o1[4 * g] * inv, o1[4 * g + 1] * inv), pack_bf16(o1[4 * g + 2] * inv, o1[4 * g + 3] * inv));
            }
            if (h2 == 0) p.lse[((int64_t)b * p.H + hh) * N + query] = mx * p.scale + __logf(lsum);
        }
        VSDE_AT_STAMP(3);
    }
    ++npairs_;
    if constexpr (!PERSIST) break;
    if (!has_next) break;
    // this wave is done with the current pair: its share of the next pair's K / V is requested NOW (a wave with a second block
    // has pt < 0: no loads) and lands while the wave that owns the 13th block is still computing
    VSDE_FWD_STAGE(((int64_t)nb * N * p.H + nh) * AT_D, false);
    VSDE_AT_STAMP(0);
    head = next; b = nb; hh = nh;
  }
    if constexpr ((ABL & 16) != 0) {
        if (blockIdx.x == 0 && lane == 0 && p.trace) {
#pragma unroll
            for (int k = 0; k < 4; ++k) p.trace[wave * 5 + k] = ph[k];
            p.trace[wave * 5 + 4] = npairs_;
        }
    }
#undef VSDE_AT_STAMP
#undef VSDE_FWD_STAGE
}

// (rare path of attn_fwd8_kernel) true row maxima of the scaled scores of a query block, as pass 1 of attn_fwd_kernel
__device__ __forceinline__ float fwd8_true_max(const bf16x8 (&qf)[4], const uint16_t *Ks, int nkt, int N, bool ragged, int fr, int h2) {
    float mx = -INFINITY;
#pragma unroll 1
    for (int kt = 0; kt < nkt; ++kt) {
        f32x16 s = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const uint16_t *krow = Ks + (kt * 32 + fr) * AT_KLD + h2 * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8 *)(krow + ks * 16), qf[ks], s, 0, 0, 0);
        if (ragged && kt == nkt - 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2 >= N) s[r] = -INFINITY;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[r]);
    }
    const uint32_t u = __float_as_uint(mx);
    const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Round 5: the persistent forward on EIGHT waves for 385 .. 416 tokens (13 query blocks of 32: the benchmark's 401).  The phase stamps of
// the twelve-wave kernel (tools/attn_trace.py, profiles/r05_attn_fwd_trace.txt) put 30 % of a pair into staging the next pair's K / V --
// requested only when a wave is done, its HBM latency in front of everybody -- and 15 % into the blocks' q fragment loads; at 168
// registers (three waves per SIMD) there is no room to request anything earlier.  Here a wave has 256 registers:
//   * blocks: waves 0..3 own two query blocks (w, w + 8), waves 4..7 one (w) plus a QUARTER of the ragged 13th block's key tiles
//     (partial O and row sums in spare LDS; the softmax shift is the Cauchy-Schwarz bound, the same for every sharer, so the partials
//     just add) -- every SIMD runs 3.25 blocks;
//   * a wave requests its share of the NEXT pair's K / V rows (7 + 8 x 16 bytes per lane) at the start of its LAST block and carries
//     them through that block's tile loop; when the pair's barrier opens they have landed and go straight into LDS;
//   * the q fragments of a wave's second block are requested under its first block's tile loop, those of the next pair's first block
//     under the last one's;
//   * the shared block's partials are summed (fixed order) by wave 4 between the two barriers of the pair boundary, its epilogue runs
//     behind the second one.
#ifdef VSDE_ABLATIONS   // attn_fwd8_kernel: a measured, losing variant -- only in the tools' build (vsde_common.h)
template <bool TRACE>
__global__ void __launch_bounds__(512, 1) attn_fwd8_kernel(AttnParams p) {
    // TRACE (tools/attn_trace.py): per-wave cycle sums of workgroup 0: 0 = tile loops, 1 = block prologues (norms, requests),
    // 2 = waits for requested rows, 3 = epilogues / partial sums, 4 = the pair's first barrier, 5 = LDS commit + second barrier
    long long ph[7] = {0, 0, 0, 0, 0, 0, 0}, last_ = 0, npairs_ = 0;   // (6 = issue of the staging requests)
    if constexpr (TRACE) last_ = __builtin_readcyclecounter();
#define VSDE_F8_STAMP(k_) do { if constexpr (TRACE) { const long long now_ = __builtin_readcyclecounter(); ph[k_] += now_ - last_; last_ = now_; } } while (0)
    extern __shared__ __attribute__((aligned(16))) uint16_t asmem[];
    constexpr int NW = 8, PT = 512, KIT = (416 * 8 + PT - 1) / PT, NKT = 13;
    uint16_t *Ks = asmem;                        // [npad][AT_KLD]
    uint16_t *Vs = asmem + p.npad * AT_KLD;      // [npad][AT_KLD], row-major as in HBM: the PV product reads V^T with ds_read_b64_tr_b16
    float *part = (float *)(asmem + 2 * p.npad * AT_KLD);   // [4 sharers][34][64]: 32 accumulator registers, row sum, shift
    __shared__ float kred[NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = p.N, npad = p.npad;
    const int64_t ts = (int64_t)p.H * AT_D;
    const uint32_t tsb = (uint32_t)ts * 2u;   // token stride in bytes
    int fr = lane & 31, h2 = lane >> 5;
    const bool ragged = (N & 31) != 0;
    const float c2 = p.scale_log2e;
    const int64_t nheads = p.pairs;
    int64_t head = blockIdx.x;
    // (batch, head) of the pair at hand, stepped by gridDim.x without a division per pair
    int b = (int)(head / p.H), hh = (int)(head - (int64_t)b * p.H);
    const int step_b = (int)(gridDim.x / p.H), step_h = (int)(gridDim.x - (unsigned)step_b * p.H);

    // ---- one key-tile sweep of a query block: tiles kt0, kt0 + stride, ... (count of them), software-pipelined as in attn_fwd_kernel
    // (hook(i): called in front of steps i = 0, 2, 4, ...: the first block's sweep requests the next pair's rows there, one K and one V
    //  instruction per two steps -- 8 waves x 14 instructions at the top of the pair queue up in the memory pipeline and every wave
    //  stalls at issue behind them)
    auto sweep = [&](const bf16x8 (&qf)[4], float mc, int kt0, int stride, int count, f32x16 &o0, f32x16 &o1, float &lsum_out, auto &&hook) {
        f32x2 lsum2v = {0.f, 0.f};
        f32x16 sa = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, sb = sa;
        auto score = [&](int kt, f32x16 &s) {
            const uint16_t *krow = Ks + (kt * 32 + fr) * AT_KLD + h2 * 8;
            f32x16 t = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8 *)(krow + ks * 16), qf[ks], t, 0, 0, 0);
            s = t;
        };
        auto step = [&](f32x16 &scur, f32x16 &snxt, int kt, int ktn) {
            const uint16_t *krow = Ks + (ktn * 32 + fr) * AT_KLD + h2 * 8;
            bf16x8 kf[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) kf[ks] = *(const bf16x8 *)(krow + ks * 16);
            // V^T fragments (row = channel dt * 32 + fr, k = the keys 4 h2 + {0..3, 8..11 | 16..19, 24..27} of the tile, the order of P's
            // registers) out of the row-major tile by the transposing read: see accumulate_transposed below
            uint2 vf[8];
            const uint16_t *vsrc = Vs + (kt * 32 + 4 * h2 + ((fr & 15) >> 2)) * AT_KLD + ((fr >> 4) & 1) * 16 + (fr & 3) * 4;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int x = 0; x < 4; ++x) vf[dt * 4 + x] = f8_read_tr(vsrc + dt * 32 + x * 8 * AT_KLD);
            __builtin_amdgcn_sched_barrier(0);
            if (ragged && kt == NKT - 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2 >= N) scur[r] = -INFINITY;
            }
            float pr[16];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {   // v_pk_fma_f32 / v_pk_add_f32: two scores per VALU instruction
                const f32x2 arg = __builtin_elementwise_fma(f32x2{scur[r], scur[r + 1]}, f32x2{c2, c2}, f32x2{-mc, -mc});
                pr[r] = fast_exp2(arg[0]); pr[r + 1] = fast_exp2(arg[1]);
                lsum2v += f32x2{pr[r], pr[r + 1]};
            }
            uint4 pw0 = make_uint4(pack_bf16(pr[0], pr[1]), pack_bf16(pr[2], pr[3]), pack_bf16(pr[4], pr[5]), pack_bf16(pr[6], pr[7]));
            uint4 pw1 = make_uint4(pack_bf16(pr[8], pr[9]), pack_bf16(pr[10], pr[11]), pack_bf16(pr[12], pr[13]), pack_bf16(pr[14], pr[15]));
            const bf16x8 pb0 = *(bf16x8 *)&pw0, pb1 = *(bf16x8 *)&pw1;
            __builtin_amdgcn_sched_barrier(0);
            f32x16 t = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], t, 0, 0, 0);
            snxt = t;
            uint4 aw;
            aw = make_uint4(vf[0].x, vf[0].y, vf[1].x, vf[1].y); o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8 *)&aw, pb0, o0, 0, 0, 0);
            aw = make_uint4(vf[4].x, vf[4].y, vf[5].x, vf[5].y); o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8 *)&aw, pb0, o1, 0, 0, 0);
            aw = make_uint4(vf[2].x, vf[2].y, vf[3].x, vf[3].y); o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8 *)&aw, pb1, o0, 0, 0, 0);
            aw = make_uint4(vf[6].x, vf[6].y, vf[7].x, vf[7].y); o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8 *)&aw, pb1, o1, 0, 0, 0);
        };
        score(kt0, sa);
        for (int i = 0; i < count; i += 2) {
            const int kt = kt0 + i * stride, k1 = kt + stride, k2 = k1 + stride;
            hook(i);
            step(sa, sb, kt, i + 1 < count ? k1 : kt);          // (the last step recomputes its own tile: result unused)
            if (i + 1 < count) step(sb, sa, k1, i + 2 < count ? k2 : k1);
        }
        lsum_out = lsum2v[0] + lsum2v[1];
    };
    // q fragments of query block qblk of a pair (rows past N: zero).  asm loads, not counted by the compiler (see request_next below):
    // the caller waits by hand (landed()) before the first use
    auto load_q = [&](const uint16_t *qb, int qblk, bf16x8 (&qf)[4]) {
        // no branch around an asm load: a phi behind it makes the compiler COPY the registers the load has not written yet.  Rows past
        // N read row N - 1 (their results are never stored)
        const uint32_t off = (uint32_t)min(qblk * 32 + fr, N - 1) * tsb + (uint32_t)h2 * 16u;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qf[ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "+v"(qf[ks]) : "v"(off), "s"(qb), "n"(ks * 32) : "memory");
        }
    };
    // (behind them in the queue: the 2 x KIT staging loads of request())
    auto landed = [&](bf16x8 (&qf)[4]) { asm volatile("s_waitcnt vmcnt(14)" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3])); };
    auto q_shift = [&](const bf16x8 (&qf)[4], float kmax) {   // |q| max|k| x 1.0001 >= every score of this query (softmax shift)
        float qss = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const uint4 t = *(const uint4 *)&qf[ks];
            const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float lo = __uint_as_float(w[e] << 16), hi = __uint_as_float(w[e] & 0xffff0000u);
                qss = fmaf(lo, lo, fmaf(hi, hi, qss));
            }
        }
        qss = sum_xor32(qss);
        return sqrtf(qss) * kmax * 1.0001f;
    };
    // epilogue of one query block: gate, normalisation, head-merged store, log-sum-exp
    auto finish = [&](int b, int hh, int64_t base, int qblk, f32x16 &o0, f32x16 &o1, float lsum, float mx) {
        const int query = qblk * 32 + fr;
        lsum = sum_xor32(lsum);
        if (query < N) {
            const float inv = 1.0f / lsum;
            uint16_t *orow = p.o + base + query * ts;
            if (p.gate != nullptr) {
                const uint16_t *grow = p.gate + ((int64_t)b * N + query) * p.ldg;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d0 = 8 * g + 4 * h2;
                    const uint2 ga = *(const uint2 *)(grow + d0), gb = *(const uint2 *)(grow + 32 + d0);
                    const float sa[4] = {bfl(ga.x), bfh(ga.x), bfl(ga.y), bfh(ga.y)};
                    const float sb[4] = {bfl(gb.x), bfh(gb.x), bfl(gb.y), bfh(gb.y)};
#pragma unroll
                    for (int i = 0; i < 4; ++i) { o0[4 * g + i] *= sa[i]; o1[4 * g + i] *= sb[i]; }
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = 8 * g + 4 * h2;
                *(uint2 *)(orow + d0) = make_uint2(pack_bf16(o0[4 * g] * inv, o0[4 * g + 1] * inv), pack_bf16(o0[4 * g + 2] * inv, o0[4 * g + 3] * inv));
                *(uint2 *)(orow + 32 + d0) = make_uint2(pack_bf16(o1[4 * g] * inv, o1[4 * g + 1] * inv), pack_bf16(o1[4 * g + 2] * inv, o1[4 * g + 3] * inv));
            }
            if (h2 == 0) p.lse[((int64_t)b * p.H + hh) * N + query] = mx * p.scale + __logf(lsum);
        }
    };

    // ---- staging registers: K and V rows as 16-byte chunks (8 adjacent lanes = one 128-byte row), requested by asm loads the compiler
    // does not count (its own wait for a load with a tile loop between request and use is vmcnt(0) in FRONT of the loop), kept in place
    // until the hand-written wait (staged()).  No branch around an asm load: a phi behind it makes the compiler COPY registers the
    // load has not written yet; rows past N read row N - 1 (masked keys, never-stored queries)
    u32x4 kq[KIT], vq[KIT];
    bf16x8 qnext[4];   // the q fragments of the wave's first block of the next pair
    static_assert(KIT == 7, "the hand-written wait names the staging registers");
    // (addresses: wave-uniform pair base in SGPRs + a 32-bit byte offset per lane -- the launcher checks that a pair's rows span < 2 GB)
    auto request = [&](int64_t base_, int it) {   // instruction pair `it` of KIT: 16 bytes of a K row and of the same V row per lane
        int pt = tid;
        asm volatile("" : "+v"(pt));   // staging addresses are recomputed where they are used, not kept (spilled) across the tile loops
        const uint16_t *kb = p.k + base_, *vb = p.v + base_;
        const int i = pt + it * PT;
        const uint32_t off = (uint32_t)min(i >> 3, N - 1) * tsb + (uint32_t)(i & 7) * 16u;
#define VSDE_F8_REQ(IT_) case IT_: \
            kq[IT_] = u32x4{0u, 0u, 0u, 0u}; vq[IT_] = u32x4{0u, 0u, 0u, 0u}; \
            asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(kq[IT_]) : "v"(off), "s"(kb) : "memory"); \
            asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(vq[IT_]) : "v"(off), "s"(vb) : "memory"); break;
        switch (it) { VSDE_F8_REQ(0) VSDE_F8_REQ(1) VSDE_F8_REQ(2) VSDE_F8_REQ(3) VSDE_F8_REQ(4) VSDE_F8_REQ(5) VSDE_F8_REQ(6) default: break; }
#undef VSDE_F8_REQ
    };
    auto staged = [&]() {   // behind the wave's last tile loop, in front of its stores (which a vmcnt wait would also cover)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(kq[0]), "+v"(kq[1]), "+v"(kq[2]), "+v"(kq[3]), "+v"(kq[4]), "+v"(kq[5]), "+v"(kq[6]));
        asm volatile("" : "+v"(vq[0]), "+v"(vq[1]), "+v"(vq[2]), "+v"(vq[3]), "+v"(vq[4]), "+v"(vq[5]), "+v"(vq[6]));
        asm volatile("" : "+v"(qnext[0]), "+v"(qnext[1]), "+v"(qnext[2]), "+v"(qnext[3]));
    };
    // ... and their way into LDS; the wave's maximum squared key norm goes to kred[wave] (the softmax shift needs max_j |k_j|)
    auto commit = [&]() {
        int pt = tid;
        asm volatile("" : "+v"(pt));
        float kss_max = 0.f;
#pragma unroll
        for (int it = 0; it < KIT; ++it) {
            const int i = pt + it * PT, n = i >> 3, c = i & 7;
            if (i < npad * 8) {
                *(u32x4 *)(Ks + n * AT_KLD + c * 8) = kq[it];
                *(u32x4 *)(Vs + n * AT_KLD + c * 8) = vq[it];
            }
            float ss = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float lo = __uint_as_float(kq[it][e] << 16), hi = __uint_as_float(kq[it][e] & 0xffff0000u);
                ss = fmaf(lo, lo, fmaf(hi, hi, ss));
            }
            ss += xor_lane<1>(ss); ss += xor_lane<2>(ss); ss += xor_lane<4>(ss);
            kss_max = fmaxf(kss_max, ss);
        }
        kss_max = fmaxf(kss_max, xor_lane<8>(kss_max)); kss_max = fmaxf(kss_max, xor_lane<16>(kss_max));
        const uint32_t u = __float_as_uint(kss_max);
        const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        kss_max = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        if (lane == 0) kred[wave] = kss_max;
    };
    // ---- first pair
    {
#pragma unroll
        for (int it = 0; it < KIT; ++it) request(((int64_t)b * N * p.H + hh) * AT_D, it);
        load_q(p.q + ((int64_t)b * N * p.H + hh) * AT_D, wave, qnext);
        staged();
        commit();
        __syncthreads();
    }
    while (true) {
        const int64_t base = ((int64_t)b * N * p.H + hh) * AT_D;
        const uint16_t *qb = p.q + base;
        asm volatile("" : "+v"(fr), "+v"(h2));   // (see attn_fwd_kernel: keeps the tile loops' LDS addresses from being hoisted and spilled)
        float kss_max = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) kss_max = fmaxf(kss_max, kred[w]);
        const float kmax = sqrtf(kss_max);
        const int64_t next = head + gridDim.x;
        const bool has_next = next < nheads;
        int nb = b + step_b, nh = hh + step_h;
        if (nh >= p.H) { nh -= p.H; ++nb; }
        const int64_t nbase = has_next ? ((int64_t)nb * N * p.H + nh) * AT_D : base;
        // the next pair's K / V rows are requested at the TOP of the pair, by every wave, right behind the second block's q fragments
        // (which are needed first: loads return in order) -- the CU takes in ~13 bytes per clock, 103 KB are 8 k cycles that have to
        // run under the whole pair's tile loops, not under the last block's
        bf16x8 qf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = qnext[ks];
        const bool two = wave < 4;                           // waves 0..3: blocks w and w + 8; waves 4..7: block w and a quarter of block 12
        f32x16 o0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, o1 = o0;
        float lsum = 0.f;
        if (two) {
            // ---- first block; the second block's q fragments fly under its tile loop
            float mx = q_shift(qf, kmax);
            const bool exact = !__all(mx * c2 <= 40.0f);
            load_q(qb, wave + 8, qnext);
            VSDE_F8_STAMP(1);
#pragma unroll
            for (int it = 0; it < KIT; ++it) request(nbase, it);
            VSDE_F8_STAMP(6);
            if (exact) mx = fwd8_true_max(qf, Ks, NKT, N, ragged, fr, h2);
            VSDE_F8_STAMP(1);
            sweep(qf, mx * c2, 0, 1, NKT, o0, o1, lsum, [](int) {});
            VSDE_F8_STAMP(0);
            landed(qnext);
            VSDE_F8_STAMP(2);
            finish(b, hh, base, wave, o0, o1, lsum, mx);
            VSDE_F8_STAMP(3);
            // ---- second (last) block: the next pair's K / V and first q fragments fly under its tile loop
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) qf[ks] = qnext[ks];
            load_q(p.q + nbase, wave, qnext);
            mx = q_shift(qf, kmax);
            const bool exact2 = !__all(mx * c2 <= 40.0f);
            if (exact2) mx = fwd8_true_max(qf, Ks, NKT, N, ragged, fr, h2);
#pragma unroll
            for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; }
            VSDE_F8_STAMP(1);
            sweep(qf, mx * c2, 0, 1, NKT, o0, o1, lsum, [](int) {});
            VSDE_F8_STAMP(0);
            staged();
            VSDE_F8_STAMP(2);
            finish(b, hh, base, wave + 8, o0, o1, lsum, mx);
        } else {
            // ---- own block; the shared block's q fragments fly under its tile loop
            float mx = q_shift(qf, kmax);
            const bool exact = !__all(mx * c2 <= 40.0f);
            bf16x8 qs[4];
            load_q(qb, 12, qs);
            VSDE_F8_STAMP(1);
#pragma unroll
            for (int it = 0; it < KIT; ++it) request(nbase, it);
            VSDE_F8_STAMP(6);
            if (exact) mx = fwd8_true_max(qf, Ks, NKT, N, ragged, fr, h2);
            VSDE_F8_STAMP(1);
            sweep(qf, mx * c2, 0, 1, NKT, o0, o1, lsum, [](int) {});
            VSDE_F8_STAMP(0);
            landed(qs);
            VSDE_F8_STAMP(2);
            finish(b, hh, base, wave, o0, o1, lsum, mx);
            VSDE_F8_STAMP(3);
            // ---- a quarter of the ragged block's key tiles (tiles w - 4, w, w + 4, ...): partial sums into LDS.  Last: the next pair's
            // operands fly under it
            load_q(p.q + nbase, wave, qnext);
            float mxs = q_shift(qs, kmax);
            const bool exacts = !__all(mxs * c2 <= 40.0f);   // (the same for every sharer: same rows, same kmax)
            const int s4 = wave - 4;
#pragma unroll
            for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; }
            float ls = 0.f;
            if (exacts) {   // rare: one sharer takes the whole block with the true maxima, the others contribute zeros
                if (s4 == 0) { mxs = fwd8_true_max(qs, Ks, NKT, N, ragged, fr, h2); VSDE_F8_STAMP(1); sweep(qs, mxs * c2, 0, 1, NKT, o0, o1, ls, [](int) {}); VSDE_F8_STAMP(0); }
            } else {
                VSDE_F8_STAMP(1);
                sweep(qs, mxs * c2, s4, 4, (NKT - s4 + 3) / 4, o0, o1, ls, [](int) {});
                VSDE_F8_STAMP(0);
            }
            staged();
            VSDE_F8_STAMP(2);
            float *pp = part + (s4 * 34) * 64 + lane;
#pragma unroll
            for (int e = 0; e < 16; ++e) { pp[e * 64] = o0[e]; pp[(16 + e) * 64] = o1[e]; }
            pp[32 * 64] = ls;
            if (s4 == 0) pp[33 * 64] = mxs;
        }
        VSDE_F8_STAMP(3);
        __syncthreads();   // everyone is done with this pair's K / V; the partials are complete
        VSDE_F8_STAMP(4);
        if (has_next) {
            commit();
            __syncthreads();
        }
        VSDE_F8_STAMP(5);
        if (wave == 4) {   // fixed-order sum of the four partial tiles and the shared block's epilogue, behind the staging barrier (nobody
                           // writes a partial tile again before the end of the next pair's tile loops)
            f32x16 r0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, r1 = r0;
            float rl = 0.f;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const float *pp = part + (s4 * 34) * 64 + lane;
#pragma unroll
                for (int e = 0; e < 16; ++e) { r0[e] += pp[e * 64]; r1[e] += pp[(16 + e) * 64]; }
                rl += pp[32 * 64];
            }
            finish(b, hh, base, 12, r0, r1, rl, part[33 * 64 + lane]);
        }
        if constexpr (TRACE) ++npairs_;
        if (!has_next) break;
        head = next; b = nb; hh = nh;
    }
    if constexpr (TRACE) {
        VSDE_F8_STAMP(3);
        if (blockIdx.x == 0 && lane == 0 && p.trace) {
#pragma unroll
            for (int k = 0; k < 7; ++k) p.trace[wave * 8 + k] = ph[k];
            p.trace[wave * 8 + 7] = npairs_;
        }
    }
#undef VSDE_F8_STAMP
}
#endif  // VSDE_ABLATIONS (attn_fwd8_kernel)

// ===================================================================================== backward
// Two kernels, each the mirror image of the other; both recompute the probabilities from q, k and the saved log-sum-exp.
//   dq kernel   : K and V of the head resident in LDS (row-major); a wavefront owns 32 queries (q, dO fragments and the
//                 dQ^T accumulators in registers).  Lane = query orientation: S^T = K Q^T, dP^T = V dO^T,
//                 dQ^T += K^T dS^T.  Also emits D_i = <dO_i, O_i>.
//   dk/dv kernel: Q, dO, lse, D resident in LDS; a wavefront owns 32 keys (k, v fragments, dK^T / dV^T accumulators).
//                 Lane = key orientation: S = Q K^T, dP = dO V^T, dV^T += dO^T P, dK^T += Q^T dS.
// In both, the tile that comes out of the first MFMA pair already has the B-operand layout of the accumulating product
// (with its 32 contraction indices in the order 4h+{0..3}, 8+4h+{0..3}, ...), so P and dS never leave registers; the
// transposed operands (K^T, Q^T, dO^T) are never materialised: ds_read_b64_tr_b16 reads them out of the row-major tiles.
// No atomics: every output element is owned by exactly one lane (deterministic).

// Fused projection-side backward (training step, round 3): the attention's q, k are RoPE(rnd(RMS(x) w)) of the projection
// output x (primitives/attn.py:80-95) and v = lam v_raw + (1 - lam) v0.  The dq / dkv kernels own whole 64-wide head rows of
// dq^, dk^, dv, so the RoPE + RMS-norm + value-mix backward is lane-local epilogue math and the gradients leave straight as
// columns of dy, the gradient of the [q | k | v | gate] projection -- the separate qk_norm_rope_bwd pass (a full read of the raw
// projection + dq, dk, dv and a write of dy) does not exist.  The raw projection is not kept at all: with a = R^T y^ (= n w, n
// the normalised row) and c = mean(dy^ . y^) (rotations preserve the dot product),
//     dx = rinv (w R^T dy^  -  (a / w) c),            rinv = the row's inverse RMS saved by the projection kernel's epilogue.
struct QkBwd {
    uint16_t *dy; int64_t ldy;       // [M][ldy]: columns [0, 64 H) dq_raw, [64 H, 128 H) dk_raw, [128 H, 192 H) dv_raw
    const float *rinv;               // [M][2 H]: inverse RMS of the q heads, then of the k heads
    const float *cosT, *sinT;        // rotary tables [N][32]
    const float *wq, *wk;            // frozen RMS weights [64] (all non-zero: checked by the host)
    const float *lam;                // value-mix weight [1] (blocks that mix)
    const uint16_t *vdiff;           // [M][64 H] v_raw - v0 saved by the projection epilogue, or nullptr (no mixing in this block)
    uint16_t *dv0; int dv0_accumulate;   // [M][64 H] gradient of the residual values: = or += (1 - lam) dv
    const uint16_t *dv_extra;        // [M][64 H] added to dv first (the block that produced v0), or nullptr
    float *dlam_partial;             // [B H ntile] per (head, key block) partial sums of <dv, v_raw - v0>
    int dbg;                         // ablation (VSDE_ATTN_DEBUG): 1 / 2 / 4 = k / v / q rows of the direct epilogues stored without their math, 8 = every round takes the direct epilogues
};

struct AttnBwdParams {
    const uint16_t *q, *k, *v, *o, *dout;  // [B][N][H][64] bf16
    const float *lse;                      // [B][H][N]
    float *delta;                          // [B][H][N]  D_i = <dO_i, O_i>   (written by the dq kernel, read by dk/dv; FUSED: given)
    uint16_t *dq, *dk, *dv;                // [B][N][H][64] bf16
    int N, H, ntile;                       // ntile = ceil(N / 32)
    float scale, scale_log2e;
    QkBwd f;                               // FUSED kernels only
    int park_off;                          // FUSED dk/dv: byte offset of the LDS area where first-round tiles wait (0 = no room: direct epilogue)
    // FUSED, ntile = 13 (N = 401): the ragged 13th block is not a second round of wave 0 -- a lone wave on an otherwise idle CU --
    // but shared by waves 0..3 (one per SIMD), a quarter of the tile loop each, in front of their own block; the four partial
    // tiles (valid rows only, fp32, the staged row layout) wait in spare LDS at these byte offsets and are summed in a fixed
    // order by the wave(s) that finish the block after the workgroup barrier.  0 = off (does not fit / other tile counts).
    // While a sharing wave works on the ragged block its OWN block's fragments (requested before the staging, like everybody's)
    // wait in its partial-tile space -- two fragment sets and a tile loop do not fit the register file.
    int split_dq, split_dkv;
    int split_pitch_dq, split_pitch_dkv;   // bytes of LDS per sharing wave (>= 8 KB: the kept fragments)
    int nragged;                           // valid rows of the ragged block
};

__device__ __forceinline__ f32x16 tile_product(const uint16_t *arow, const bf16x8 (&bfrag)[4]) {
    f32x16 t = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8 *)(arow + ks * 16), bfrag[ks], t, 0, 0, 0);
    return t;
}

typedef short bf16x4s __attribute__((ext_vector_type(4)));
// ds_read_b64_tr_b16 (probed on gfx950, tools/probes/tr_b16_probe.hip): within a 16-lane group, lane m supplies the address
// of 4 contiguous bf16 = row m/4, columns 4(m%4).. of a [4][16] block, and lane i receives column i (rows 0..3).
__device__ __forceinline__ uint2 lds_read_tr(const uint16_t *ptr) {
    bf16x4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4s *)ptr);
    return *(uint2 *)&r;
}

// acc[dt] += X^T B for a row-major LDS tile X [32 tokens][64] (stride AT_KLD): the A operand (row = channel d = dt*32 + fr,
// contraction over the tokens 4h2 + {0..3, 8..11 | 16..19, 24..27}, the order in which B's registers hold them) is read
// with the hardware transpose -- no transposed copy of the tile exists anywhere.
__device__ __forceinline__ void accumulate_transposed(const uint16_t *tile, int lane, const bf16x8 &b0, const bf16x8 &b1,
                                                      f32x16 &acc0, f32x16 &acc1) {
    const int h2 = lane >> 5, m = lane & 15;
    const uint16_t *src = tile + (4 * h2 + (m >> 2)) * AT_KLD + ((lane >> 4) & 1) * 16 + (m & 3) * 4;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        const uint2 a0 = lds_read_tr(src + dt * 32), a1 = lds_read_tr(src + dt * 32 + 8 * AT_KLD);
        const uint2 a2 = lds_read_tr(src + dt * 32 + 16 * AT_KLD), a3 = lds_read_tr(src + dt * 32 + 24 * AT_KLD);
        uint4 w0 = make_uint4(a0.x, a0.y, a1.x, a1.y), w1 = make_uint4(a2.x, a2.y, a3.x, a3.y);
        if (dt == 0) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8 *)&w0, b0, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8 *)&w1, b1, acc0, 0, 0, 0);
        } else {
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8 *)&w0, b0, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8 *)&w1, b1, acc1, 0, 0, 0);
        }
    }
}

// ===================================================================================== forward, K / V streamed through an LDS ring
// The LDS-resident forward above is a chain per (batch, head) pair: stage K / V (HBM) -> barrier -> tile loop -> barrier, one
// workgroup per CU, so the HBM phases add to the tile loop (56 us of tile loop in 188).  Here the operands never stop streaming:
//   * 8 waves: 7 consumers + 1 PRODUCER.  The producer copies the key tiles of the NEXT pair global -> LDS with
//     global_load_lds_dwordx4 (no registers, no transposition, nothing in any consumer's LDS queue) while the consumers work on the
//     current pair, and computes the next pair's max |k|^2 (the softmax shift) from the rows it has just pulled through L2;
//   * a ring of 18 tile slots (32 keys x 128 bytes of K and of V each, 144 KB): the 13 tiles of the current pair + 5 of the next;
//     the next pair's tile t >= 5 takes the slot of the current tile t - 5 once every consumer has passed it in its LAST sweep;
//   * a consumer owns query block w, then block w + 7 (two sweeps over the pair's tiles per pair; 13 blocks: 4 / 4 / 3 / 2 per
//     SIMD as in the resident kernel), one workgroup barrier per tile step -- the barrier is what tells the producer a slot is free;
//   * rows are unpadded, 16-byte chunk c of row r sits at chunk c ^ f(r): conflict-free for the K row fragments (ds_read_b128)
//     AND for the V^T fragments, which come out of the row-major V tile through ds_read_b64_tr_b16 (no V^T staging);
//   * the next sweep's Q fragments are requested a sweep ahead.
// Same arithmetic in the same order as attn_fwd_kernel: the results are bit-identical.
constexpr int RG_SLOTS = 18, RG_TILE = 32 * AT_D, RG_NW = 8, RG_NC = 7;

__device__ __forceinline__ int rg_f(int row) { const int a = row >> 1; return ((a & 1) << 2) | ((a >> 1) & 3); }
__device__ __forceinline__ void rg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
typedef __attribute__((address_space(3))) void *rg_lds_ptr;

__device__ __forceinline__ uint2 rg_read_tr(const uint16_t *ptr) {
    bf16x4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4s *)ptr);
    return *(uint2 *)&r;
}

#ifdef VSDE_ABLATIONS   // attn_fwd_ring_kernel: a measured, losing variant -- only in the tools' build (vsde_common.h)
__global__ void __launch_bounds__(64 * RG_NW) attn_fwd_ring_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) uint16_t asmem[];
    uint16_t *Kr = asmem, *Vr = asmem + RG_SLOTS * RG_TILE;
    __shared__ float kmax_s[2][RG_NC];   // per pair parity: the consumers' shares of max_j |k_j|^2
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = p.N, nkt = p.npad >> 5;
    const int64_t ts = (int64_t)p.H * AT_D;
    const int64_t first = blockIdx.x, stride = gridDim.x;
    const int npairs = (int)((p.pairs - first + stride - 1) / stride);   // >= 1 (grid <= pairs)
    constexpr int STEPS = 2 * 13;   // tile steps per pair: two sweeps of up to 13 tiles (shorter sequences idle through the rest)

    if (wave == RG_NC) {
        // ------------------------------------------------------------------------------------------------ producer
        auto issue_tile = [&](int64_t head, int t, int slot) {
            if (t >= nkt) return;
            const int64_t hb = head / p.H, base_ = (hb * N * p.H + (head - hb * p.H)) * AT_D;
#pragma unroll
            for (int i = 0; i < 4; ++i) {   // 8 rows per instruction: lane = (row, physical chunk), 64 x 16 bytes land back to back
                const int row = i * 8 + (lane >> 3), c = (lane & 7) ^ rg_f(row);
                int key = t * 32 + row;
                key = key < N ? key : N - 1;   // rows past the sequence repeat the last one (masked in the softmax, finite for P = 0)
                const int64_t off = base_ + (int64_t)key * ts + c * 8;
                __builtin_amdgcn_global_load_lds((const void *)(p.k + off), (rg_lds_ptr)(Kr + slot * RG_TILE + i * 512), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const void *)(p.v + off), (rg_lds_ptr)(Vr + slot * RG_TILE + i * 512), 16, 0, 0);
            }
        };
        for (int t = 0; t < nkt; ++t) issue_tile(first, t, t);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        rg_barrier();
        int pbase = 0;   // ring slot of the current pair's tile 0
        for (int pi = 0; pi < npairs; ++pi) {
            const int64_t next = first + (int64_t)(pi + 1) * stride;
            const bool has_next = pi + 1 < npairs;
            int nbase = pbase + 13; nbase -= nbase >= RG_SLOTS ? RG_SLOTS : 0;
            rg_barrier();   // (the consumers exchange their shares of max |k|^2 here)
#pragma unroll 1
            for (int s = 0; s < STEPS; ++s) {
                if (has_next) {
                    // tiles 0..4 of the next pair go into the 5 slots the current pair does not use; tile t >= 5 into the slot of the
                    // current tile t - 5, free once every consumer is past barrier 13 + (t - 5) (its second sweep)
                    int t = -1;
                    if (s < 5) t = s;
                    else if (s >= 14 && s <= 21) t = s - 9;
                    if (t >= 0 && !(p.dbg & 1)) { int slot = nbase + t; slot -= slot >= RG_SLOTS ? RG_SLOTS : 0; issue_tile(next, t, slot); }
                }
                if (s == STEPS - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next pair is complete at the pair boundary
                if (!(p.dbg & 2) || s == STEPS - 1) rg_barrier();
            }
            pbase = nbase;
        }
        return;
    }
    // ---------------------------------------------------------------------------------------------------- consumers
    const int fr = lane & 31, h2 = lane >> 5, m = lane & 15, l4 = (lane >> 4) & 1;
    int kofs[4], vofs[2][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kofs[ks] = fr * AT_D + (((2 * ks + h2) ^ rg_f(fr)) * 8);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
            const int m3 = (m >> 3) & 1, m1 = (m >> 1) & 1;
            const int pc = 4 * (dt ^ m3) + 2 * (l4 ^ jp) + (m1 ^ h2);
            vofs[dt][jp] = (4 * h2 + (m >> 2)) * AT_D + pc * 8 + (m & 1) * 4;
        }
    const bool ragged = (N & 31) != 0;
    uint4 qn[4];   // fragments of the next sweep's query block
    auto request_q = [&](int64_t head, int qblk) {
        const int query = qblk * 32 + fr;
        const int64_t hb = head / p.H, base_ = (hb * N * p.H + (head - hb * p.H)) * AT_D;
        const uint16_t *row = p.q + base_ + (int64_t)(query < N ? query : N - 1) * ts + h2 * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qn[ks] = *(const uint4 *)(row + ks * 16);
    };
    request_q(first, wave);
    // qn is waited for at the END of a sweep's tile loop, before that sweep's output stores are issued: a wait at the next sweep's
    // start would sit behind those stores (the counter is in order and the stores are conditional: the compiler drains to zero)
    uint4 qcur[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qcur[ks] = qn[ks];
    rg_barrier();
    int pbase = 0;
    for (int pi = 0; pi < npairs; ++pi) {
        const int64_t head = first + (int64_t)pi * stride;
        const int b = (int)(head / p.H), hh = (int)(head - (int64_t)b * p.H);
        const int64_t base = ((int64_t)b * N * p.H + hh) * AT_D;
        // max_j |k_j|^2 from the resident tiles (wave w: tiles w and w + 7), in the arithmetic order of the resident kernel's staging
        // (8 adjacent lanes = one row, chunk by chunk), so that the shift -- and with it every result -- is bit-identical
        {
            float mxk = 0.f;
            for (int t = wave; t < nkt; t += RG_NC) {
                int sl = pbase + t; sl -= sl >= RG_SLOTS ? RG_SLOTS : 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = i * 8 + (lane >> 3), pc = (lane & 7) ^ rg_f(row);
                    const uint4 kv = *(const uint4 *)(Kr + sl * RG_TILE + row * AT_D + pc * 8);
                    const uint32_t w[4] = {kv.x, kv.y, kv.z, kv.w};
                    float ss = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float lo = __uint_as_float(w[e] << 16), hi = __uint_as_float(w[e] & 0xffff0000u);
                        ss = fmaf(lo, lo, fmaf(hi, hi, ss));
                    }
                    ss += xor_lane<1>(ss); ss += xor_lane<2>(ss); ss += xor_lane<4>(ss);
                    if (t * 32 + row < N) mxk = fmaxf(mxk, ss);
                }
            }
            mxk = fmaxf(mxk, xor_lane<8>(mxk)); mxk = fmaxf(mxk, xor_lane<16>(mxk));
            const uint32_t u = __float_as_uint(mxk);
            const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            mxk = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            if (lane == 0) kmax_s[pi & 1][wave] = mxk;
        }
        rg_barrier();
        float kss_max = 0.f;
#pragma unroll
        for (int w = 0; w < RG_NC; ++w) kss_max = fmaxf(kss_max, kmax_s[pi & 1][w]);
        const float kmax = sqrtf(kss_max);
#pragma unroll 1
        for (int sweep = 0; sweep < 2; ++sweep) {
            const int qblk = wave + RG_NC * sweep;
            if (qblk >= nkt) {   // no block in this sweep: keep the barrier count (and the next pair's first block on its way)
                if (sweep == 1 && pi + 1 < npairs) request_q(head + stride, wave);
#pragma unroll 1
                for (int s = 0; s < 13; ++s) if (!(p.dbg & 2) || s == 12) rg_barrier();
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) qcur[ks] = qn[ks];
                continue;
            }
            const int query = qblk * 32 + fr;
            const bool qok = query < N;
            uint2 gr[8];   // the lane's share of the token's gate row (index 2 g: channels 8 g + 4 h2 .., 2 g + 1: + 32), requested two tile steps before the epilogue
#pragma unroll
            for (int j = 0; j < 8; ++j) gr[j] = make_uint2(0u, 0u);
            bf16x8 qf[4];
            float qss = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                uint4 t = qok ? qcur[ks] : make_uint4(0, 0, 0, 0);
                qf[ks] = *(bf16x8 *)&t;
                const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = __uint_as_float(w[e] << 16), hi = __uint_as_float(w[e] & 0xffff0000u);
                    qss = fmaf(lo, lo, fmaf(hi, hi, qss));
                }
            }
            // the next sweep's block: this pair's second, or the next pair's first
            if (sweep == 0) { if (wave + RG_NC < nkt) request_q(head, wave + RG_NC); }
            else if (pi + 1 < npairs) request_q(head + stride, wave);
            qss = sum_xor32(qss);
            const float mx = sqrtf(qss) * kmax * 1.0001f;
            const float c2 = p.scale_log2e, mc = mx * c2;
            float lsum = 0.f, lsum2 = 0.f;
            f32x16 o0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, o1 = o0;
            auto slot_of = [&](int kt) { int sl = pbase + kt; return sl - (sl >= RG_SLOTS ? RG_SLOTS : 0); };
            // One tile step = phase A (LDS reads of K tile kt + 1 and V tile kt, softmax of score tile kt on the VALU) + phase B (the
            // score product of tile kt + 1 and the PV product of tile kt on the matrix pipe, registers only).  The per-step barrier
            // keeps the two consumers of a SIMD in lock step -- both on the VALU, then both on the matrix pipe -- so the second wave
            // of every SIMD (waves 4..6) runs half a step out of phase: B of the previous tile, then A, between two barriers.  All
            // LDS reads of a step still happen before its barrier: the producer's slot rule does not change.
            bf16x8 kf[4], pb0, pb1;
            uint2 va[2][4];
            auto phase_a = [&](f32x16 &scur, int kt) {
                const int ktn = min(kt + 1, nkt - 1);
                const uint16_t *kn_ = Kr + slot_of(ktn) * RG_TILE, *vt_ = Vr + slot_of(kt) * RG_TILE;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) kf[ks] = *(const bf16x8 *)(kn_ + kofs[ks]);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) va[dt][j] = rg_read_tr(vt_ + vofs[dt][j & 1] + j * 8 * AT_D);
                __builtin_amdgcn_sched_barrier(0);
                if (ragged && kt == nkt - 1) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2 >= N) scur[r] = -INFINITY;
                }
                float pr[16];
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    pr[r] = fast_exp2(fmaf(scur[r], c2, -mc)); pr[r + 1] = fast_exp2(fmaf(scur[r + 1], c2, -mc));
                    lsum += pr[r]; lsum2 += pr[r + 1];
                }
                uint4 pw0 = make_uint4(pack_bf16(pr[0], pr[1]), pack_bf16(pr[2], pr[3]), pack_bf16(pr[4], pr[5]), pack_bf16(pr[6], pr[7]));
                uint4 pw1 = make_uint4(pack_bf16(pr[8], pr[9]), pack_bf16(pr[10], pr[11]), pack_bf16(pr[12], pr[13]), pack_bf16(pr[14], pr[15]));
                pb0 = *(bf16x8 *)&pw0; pb1 = *(bf16x8 *)&pw1;
                __builtin_amdgcn_sched_barrier(0);
            };
            auto phase_b = [&](f32x16 &snxt) {
                f32x16 t = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], t, 0, 0, 0);
                snxt = t;
                uint4 aw;
                aw = make_uint4(va[0][0].x, va[0][0].y, va[0][1].x, va[0][1].y); o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8 *)&aw, pb0, o0, 0, 0, 0);
                aw = make_uint4(va[1][0].x, va[1][0].y, va[1][1].x, va[1][1].y); o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8 *)&aw, pb0, o1, 0, 0, 0);
                aw = make_uint4(va[0][2].x, va[0][2].y, va[0][3].x, va[0][3].y); o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8 *)&aw, pb1, o0, 0, 0, 0);
                aw = make_uint4(va[1][2].x, va[1][2].y, va[1][3].x, va[1][3].y); o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8 *)&aw, pb1, o1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            };
            f32x16 sa = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, sb = sa;
            {
                const uint16_t *k0_ = Kr + slot_of(0) * RG_TILE;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8 *)(k0_ + kofs[ks]), qf[ks], sa, 0, 0, 0);
            }
            const bool late = wave >= 4;   // the second consumer of its SIMD
            const bool nobar = (p.dbg & 2) != 0, nostep = (p.dbg & 4) != 0;
            auto tile_barrier = [&](int kt) { if (!nobar || kt == 12) rg_barrier(); };
            auto gate_request = [&](int kt) {
                if (kt == 10 && p.gate != nullptr) {
                    const uint16_t *grow = p.gate + ((int64_t)b * N + (qok ? query : N - 1)) * p.ldg + 4 * h2;
#pragma unroll
                    for (int j = 0; j < 8; ++j) gr[j] = *(const uint2 *)(grow + 8 * (j >> 1) + 32 * (j & 1));
                }
            };
            if (!late) {
#pragma unroll 1
                for (int kt = 0; kt < 13; kt += 2) {
                    gate_request(kt);
                    if (kt < nkt && !nostep) { phase_a(sa, kt); phase_b(sb); }
                    tile_barrier(kt);
                    if (kt + 1 < 13) {
                        if (kt + 1 < nkt && !nostep) { phase_a(sb, kt + 1); phase_b(sa); }
                        tile_barrier(kt + 1);
                    }
                }
            } else {
                // score tiles alternate sa (even) / sb (odd): B of tile kt - 1 writes the tile A of tile kt reads
#pragma unroll 1
                for (int kt = 0; kt < 13; kt += 2) {
                    gate_request(kt);
                    if (kt < nkt && !nostep) { if (kt > 0) phase_b(sa); phase_a(sa, kt); }
                    tile_barrier(kt);
                    if (kt + 1 < 13) {
                        if (kt + 1 < nkt && !nostep) { phase_b(sb); phase_a(sb, kt + 1); }
                        tile_barrier(kt + 1);
                    }
                }
                f32x16 unused;
                phase_b(unused);   // the last tile's PV product (its score product recomputes the last tile: not used)
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) qcur[ks] = qn[ks];   // (requested a sweep ago)
            asm volatile("" : "+v"(qcur[0].x), "+v"(qcur[1].x), "+v"(qcur[2].x), "+v"(qcur[3].x));   // keep the copy (and its wait) here
            lsum += lsum2;
            lsum = sum_xor32(lsum);
            if (qok && !(p.dbg & 8)) {
                const float inv = 1.0f / lsum;
                uint16_t *orow = p.o + base + query * ts;
                if (p.gate != nullptr) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const uint2 ga = gr[2 * g], gb = gr[2 * g + 1];
                        const float ga4[4] = {bfl(ga.x), bfh(ga.x), bfl(ga.y), bfh(ga.y)};
                        const float gb4[4] = {bfl(gb.x), bfh(gb.x), bfl(gb.y), bfh(gb.y)};
#pragma unroll
                        for (int i = 0; i < 4; ++i) { o0[4 * g + i] *= ga4[i]; o1[4 * g + i] *= gb4[i]; }
                    }
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d0 = 8 * g + 4 * h2;
                    *(uint2 *)(orow + d0) = make_uint2(pack_bf16(o0[4 * g] * inv, o0[4 * g + 1] * inv), pack_bf16(o0[4 * g + 2] * inv, o0[4 * g + 3] * inv));
                    *(uint2 *)(orow + 32 + d0) = make_uint2(pack_bf16(o1[4 * g] * inv, o1[4 * g + 1] * inv), pack_bf16(o1[4 * g + 2] * inv, o1[4 * g + 3] * inv));
                }
                if (h2 == 0) p.lse[((int64_t)b * p.H + hh) * N + query] = mx * p.scale + __logf(lsum);
            }
        }
        pbase += 13; pbase -= pbase >= RG_SLOTS ? RG_SLOTS : 0;
    }
}
#endif  // VSDE_ABLATIONS (attn_fwd_ring_kernel)

__device__ __forceinline__ void pack_tile(const float (&x)[16], bf16x8 &b0, bf16x8 &b1) {
    uint4 w0 = make_uint4(pack_bf16(x[0], x[1]), pack_bf16(x[2], x[3]), pack_bf16(x[4], x[5]), pack_bf16(x[6], x[7]));
    uint4 w1 = make_uint4(pack_bf16(x[8], x[9]), pack_bf16(x[10], x[11]), pack_bf16(x[12], x[13]), pack_bf16(x[14], x[15]));
    b0 = *(bf16x8 *)&w0; b1 = *(bf16x8 *)&w1;
}

// row-major [row][64] global rows (token stride ts) -> registers of the B operand "column = row index, k = d"
__device__ __forceinline__ void load_bfrag(const uint16_t *base, int64_t ts, int row, bool ok, int h2, bf16x8 (&f)[4]) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        uint4 t = make_uint4(0, 0, 0, 0);
        if (ok) t = *(const uint4 *)(base + row * ts + ks * 16 + h2 * 8);
        f[ks] = *(bf16x8 *)&t;
    }
}

// accumulator tile (rows d, column = owned token) -> token-major bf16 row, scaled
__device__ __forceinline__ void store_transposed(uint16_t *row, int h2, const f32x16 &a0, const f32x16 &a1, float mul) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int d0 = 8 * g + 4 * h2;
        *(uint2 *)(row + d0) = make_uint2(pack_bf16(a0[4 * g] * mul, a0[4 * g + 1] * mul), pack_bf16(a0[4 * g + 2] * mul, a0[4 * g + 3] * mul));
        *(uint2 *)(row + 32 + d0) = make_uint2(pack_bf16(a1[4 * g] * mul, a1[4 * g + 1] * mul), pack_bf16(a1[4 * g + 2] * mul, a1[4 * g + 3] * mul));
    }
}

// B-operand fragments (lane (fr, h2): channels 16 ks + 8 h2 + e of token fr) -> the accumulator's channel set of the same
// token (lo[4 g + i] = channel 8 g + 4 h2 + i, hi = + 32): each half-wave hands the other 4 of every 8 channels,
// v_permlane32_swap moves two dwords per instruction.  All 64 lanes must execute this.
__device__ __forceinline__ void frag_to_acc_layout(const bf16x8 (&f)[4], float (&lo)[16], float (&hi)[16]) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const uint4 w = *(const uint4 *)&f[ks];
        const auto s0 = __builtin_amdgcn_permlane32_swap(w.x, w.z, false, false);   // [0]: channels (i = 0, 1) of g, [1]: of g + 1
        const auto s1 = __builtin_amdgcn_permlane32_swap(w.y, w.w, false, false);   // (i = 2, 3)
        float *dst = ks < 2 ? lo : hi;
        const int g0 = 2 * (ks & 1);
        dst[4 * g0] = bfl(s0[0]); dst[4 * g0 + 1] = bfh(s0[0]); dst[4 * g0 + 2] = bfl(s1[0]); dst[4 * g0 + 3] = bfh(s1[0]);
        dst[4 * g0 + 4] = bfl(s0[1]); dst[4 * g0 + 5] = bfh(s0[1]); dst[4 * g0 + 6] = bfl(s1[1]); dst[4 * g0 + 7] = bfh(s1[1]);
    }
}

// Epilogue of a q (KIND 0) or k (KIND 1) head row: a0 / a1 x mul = gradient of the rotated, normalised row y^ (fragments yf) of
// token m (position n) -> gradient of the raw projection row, stored as bf16 into dy.  Called by all lanes (cross-lane ops).
template <int KIND>
__device__ __forceinline__ void norm_rope_bwd_store(const QkBwd &f, const f32x16 &a0, const f32x16 &a1, float mul, const bf16x8 (&yf)[4],
                                                    int64_t m, int n, int hh, int H, int lane, bool ok) {
    const int h2 = lane >> 5;
    float yl[16], yh[16];
    frag_to_acc_layout(yf, yl, yh);
    float c = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) c = fmaf(a0[r], yl[r], fmaf(a1[r], yh[r], c));
    c = sum_xor32(c);
    c *= mul * (1.0f / 64.0f);
    if (!ok) return;
    const float rr = f.rinv[m * (2 * H) + KIND * H + hh];
    const float *w = KIND ? f.wk : f.wq;
    uint16_t *dst = f.dy + m * f.ldy + (KIND * H + hh) * 64;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int d0 = 8 * g + 4 * h2;
        const float4 c4 = *(const float4 *)(f.cosT + n * 32 + d0), s4 = *(const float4 *)(f.sinT + n * 32 + d0);
        const float4 wl4 = *(const float4 *)(w + d0), wh4 = *(const float4 *)(w + 32 + d0);
        const float cs[4] = {c4.x, c4.y, c4.z, c4.w}, sn[4] = {s4.x, s4.y, s4.z, s4.w};
        const float wl[4] = {wl4.x, wl4.y, wl4.z, wl4.w}, wh[4] = {wh4.x, wh4.y, wh4.z, wh4.w};
        float ol[4], oh[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 4 * g + i;
            const float gl = a0[r] * mul, gh = a1[r] * mul;
            const float ul = gl * cs[i] + gh * sn[i], uh = gh * cs[i] - gl * sn[i];            // R^T dy^
            const float al = yl[r] * cs[i] + yh[r] * sn[i], ah = yh[r] * cs[i] - yl[r] * sn[i];  // R^T y^ = n w
            ol[i] = rr * (wl[i] * ul - al * (c / wl[i]));
            oh[i] = rr * (wh[i] * uh - ah * (c / wh[i]));
        }
        *(uint2 *)(dst + d0) = make_uint2(pack_bf16(ol[0], ol[1]), pack_bf16(ol[2], ol[3]));
        *(uint2 *)(dst + 32 + d0) = make_uint2(pack_bf16(oh[0], oh[1]), pack_bf16(oh[2], oh[3]));
        __builtin_amdgcn_sched_barrier(0);   // one group's table loads at a time (hoisting all four costs 48 VGPRs: spills)
    }
}

// Epilogue of a v head row: a0 / a1 = gradient of the (mixed) values of token m.  Returns this lane's share of <dv, v_raw - v0>.
// All of the row's loads (v_raw - v0, the residual-gradient accumulator, the extra gradient: streamed data, HBM latency) are
// issued before the first use -- per-group loads cost four serialized memory round trips per wave (+200 us per launch).
__device__ __forceinline__ float value_bwd_store(const QkBwd &f, const f32x16 &a0, const f32x16 &a1, int64_t m, int hh, int H, int lane) {
    const int h2 = lane >> 5;
    const int64_t vo = m * ((int64_t)H * 64) + hh * 64 + 4 * h2;
    uint16_t *dst = f.dy + m * f.ldy + (2 * H + hh) * 64 + 4 * h2;
    const bool mix = f.vdiff != nullptr, acc = mix && f.dv0_accumulate, ext = f.dv_extra != nullptr;
    const float l = mix ? f.lam[0] : 1.0f;
    uint2 df[8], ya[8], ex[8];   // index 2 g + (0: channels 8 g + 4 h2 .., 1: + 32)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int off = 8 * (j >> 1) + 32 * (j & 1);
        df[j] = mix ? *(const uint2 *)(f.vdiff + vo + off) : make_uint2(0u, 0u);
        ya[j] = acc ? *(const uint2 *)(f.dv0 + vo + off) : make_uint2(0u, 0u);
        ex[j] = ext ? *(const uint2 *)(f.dv_extra + vo + off) : make_uint2(0u, 0u);
    }
    float dl = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int off = 8 * (j >> 1) + 32 * (j & 1), g = j >> 1;
        const f32x16 &a = (j & 1) ? a1 : a0;
        float gv[4] = {a[4 * g] + bfl(ex[j].x), a[4 * g + 1] + bfh(ex[j].x), a[4 * g + 2] + bfl(ex[j].y), a[4 * g + 3] + bfh(ex[j].y)};
        if (mix) {
            const float d[4] = {bfl(df[j].x), bfh(df[j].x), bfl(df[j].y), bfh(df[j].y)};
            const float y[4] = {bfl(ya[j].x), bfh(ya[j].x), bfl(ya[j].y), bfh(ya[j].y)};
            float z[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                dl = fmaf(gv[i], d[i], dl);
                z[i] = (1.0f - l) * gv[i] + y[i];
                gv[i] *= l;
            }
            *(uint2 *)(f.dv0 + vo + off) = make_uint2(pack_bf16(z[0], z[1]), pack_bf16(z[2], z[3]));
        }
        *(uint2 *)(dst + off) = make_uint2(pack_bf16(gv[0], gv[1]), pack_bf16(gv[2], gv[3]));
    }
    return dl;
}

// ---- the same epilogues through LDS (the last round of every wave, after the workgroup is done with its resident operands) ----
// In the accumulator layout a lane owns 4 consecutive channels of one token: every global access of the direct epilogues above
// is 64 lanes x 8 bytes in 32 different rows, i.e. one tag lookup per lane for 8 bytes -- the epilogues were bound by the
// texture-address unit (+100 us per launch without, +270 us with the value mix).  Staged through a per-wave LDS slice the tile
// is re-read as rows: 8 lanes x 16 bytes = one full 128-byte head row, 8 rows per instruction, and the row's loads (saved
// y^, tables, v_raw - v0, the residual accumulator) and stores have the same shape.
constexpr int AT_ELD = 68;                   // floats per staged row (64 channels + 4: conflict-free b128 writes and reads)
constexpr int AT_ESLICE = 32 * AT_ELD;       // floats per wave (8.5 KB)

__device__ __forceinline__ void stage_acc_tile(float *slice, const f32x16 &a0, const f32x16 &a1, float mul, int lane) {
    float *row = slice + (lane & 31) * AT_ELD + 4 * (lane >> 5);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        *(float4 *)(row + 8 * g) = make_float4(a0[4 * g] * mul, a0[4 * g + 1] * mul, a0[4 * g + 2] * mul, a0[4 * g + 3] * mul);
        *(float4 *)(row + 32 + 8 * g) = make_float4(a1[4 * g] * mul, a1[4 * g + 1] * mul, a1[4 * g + 2] * mul, a1[4 * g + 3] * mul);
    }
}

// ... of the lane's own row only if `ok` (a partial tile of the shared ragged block: rows past the sequence are not kept)
__device__ __forceinline__ void stage_acc_rows(float *tile, const f32x16 &a0, const f32x16 &a1, float mul, int lane, bool ok) {
    float *row = tile + (lane & 31) * AT_ELD + 4 * (lane >> 5);
    if (ok) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            *(float4 *)(row + 8 * g) = make_float4(a0[4 * g] * mul, a0[4 * g + 1] * mul, a0[4 * g + 2] * mul, a0[4 * g + 3] * mul);
            *(float4 *)(row + 32 + 8 * g) = make_float4(a1[4 * g] * mul, a1[4 * g + 1] * mul, a1[4 * g + 2] * mul, a1[4 * g + 3] * mul);
        }
    }
}

// two fragment sets of a wave <-> its private LDS space (lane-interleaved 16-byte pieces: conflict-free); the pointer is made
// opaque in between so that the values are not simply kept in registers
__device__ __forceinline__ void keep_frags(uint4 *keep, const bf16x8 (&a)[4], const bf16x8 (&b2)[4], int lane) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { keep[ks * 64 + lane] = *(const uint4 *)&a[ks]; keep[(4 + ks) * 64 + lane] = *(const uint4 *)&b2[ks]; }
}
__device__ __forceinline__ void restore_frags(const uint4 *keep, bf16x8 (&a)[4], bf16x8 (&b2)[4], int lane) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const uint4 x = keep[ks * 64 + lane], y = keep[(4 + ks) * 64 + lane];
        a[ks] = *(const bf16x8 *)&x; b2[ks] = *(const bf16x8 *)&y;
    }
}

// slice [32][AT_ELD] = part[0] + part[1] + part[2] + part[3] (tiles of `rows` rows, `pitch` floats apart; fixed order: deterministic),
// zero rows past `rows`; lane (row = lane / 8 + 8 pass, c = lane % 8) as in the staged epilogues below
__device__ __forceinline__ void sum_partial_tiles(float *slice, const float *part, int pitch, int rows, int lane) {
    const int c = lane & 7;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int row = pass * 8 + (lane >> 3);
        float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
        if (row < rows) {
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float4 a = *(const float4 *)(part + w * pitch + row * AT_ELD + 8 * c), b2 = *(const float4 *)(part + w * pitch + row * AT_ELD + 8 * c + 4);
                lo.x += a.x; lo.y += a.y; lo.z += a.z; lo.w += a.w; hi.x += b2.x; hi.y += b2.y; hi.z += b2.z; hi.w += b2.w;
            }
        }
        *(float4 *)(slice + row * AT_ELD + 8 * c) = lo;
        *(float4 *)(slice + row * AT_ELD + 8 * c + 4) = hi;
    }
}

__device__ __forceinline__ void unpack8(const uint4 &w, float (&x)[8]) {
    x[0] = bfl(w.x); x[1] = bfh(w.x); x[2] = bfl(w.y); x[3] = bfh(w.y); x[4] = bfl(w.z); x[5] = bfh(w.z); x[6] = bfl(w.w); x[7] = bfh(w.w);
}
__device__ __forceinline__ void load8f(const float *p, float (&x)[8]) {
    const float4 a = *(const float4 *)p, b = *(const float4 *)(p + 4);
    x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
}

// The rows' streamed inputs are requested BEFORE the workgroup barrier that frees the LDS (waves 1..11 wait there for the
// wave that owns the 13th block): by the time the tile is staged they have arrived.  Lane (row = lane / 8 + 8 pass, c = lane % 8)
// owns channels 8 c .. 8 c + 7 of its rows and reads its rotary partner chunk c ^ 4 beside its own.
struct NormRowsIn { uint4 y[4], yp[4]; float rr[4]; };   // y^ chunks (own, partner) and the inverse RMS of the lane's 4 rows
struct ValueRowsIn { uint4 ex[4], df[4], ya[4]; };       // extra gradient, v_raw - v0, residual-gradient accumulator

template <int KIND>
__device__ __forceinline__ void norm_rows_request(const QkBwd &f, NormRowsIn &in, const uint16_t *yrows, int64_t ts, int64_t m0, int tok0,
                                                  int N, int hh, int H, int lane) {
    const int c = lane & 7;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int tok = tok0 + pass * 8 + (lane >> 3), tc = tok < N ? tok : N - 1;
        in.y[pass] = *(const uint4 *)(yrows + tc * ts + 8 * c);
        in.yp[pass] = *(const uint4 *)(yrows + tc * ts + 8 * (c ^ 4));
        in.rr[pass] = f.rinv[(m0 + tc) * (2 * H) + KIND * H + hh];
    }
}

// The y^ rows of a wave's OWN block are the B-operand fragments it has held since before the tile loop (lane (fr, h2): channels
// 16 ks + 8 h2 .. of token fr): through 4 KB of the wave's staging slice into the row layout of the staged epilogue -- instead of
// reading the rows from global memory once more (105 MB per launch, and in the dk/dv kernel a latency in front of the key epilogue).
// Only the inverse RMS still comes from memory (requested before the workgroup barrier).
template <int KIND>
__device__ __forceinline__ void norm_rinv_request(const QkBwd &f, NormRowsIn &in, int64_t m0, int tok0, int N, int hh, int H, int lane) {
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int tok = tok0 + pass * 8 + (lane >> 3), tc = tok < N ? tok : N - 1;
        in.rr[pass] = f.rinv[(m0 + tc) * (2 * H) + KIND * H + hh];
    }
}
__device__ __forceinline__ void norm_rows_from_frags(NormRowsIn &in, uint16_t *rows, const bf16x8 (&yf)[4], int lane) {
    const int fr = lane & 31, h2 = lane >> 5, c = lane & 7;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) *(uint4 *)(rows + fr * AT_D + (2 * ks + h2) * 8) = *(const uint4 *)&yf[ks];
    wave_lds_fence();
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const uint16_t *row = rows + (pass * 8 + (lane >> 3)) * AT_D;
        in.y[pass] = *(const uint4 *)(row + 8 * c);
        in.yp[pass] = *(const uint4 *)(row + 8 * (c ^ 4));
    }
    wave_lds_fence();   // (read before the slice is overwritten with the gradient tile)
}

__device__ __forceinline__ void value_rows_request(const QkBwd &f, ValueRowsIn &in, int64_t m0, int tok0, int N, int hh, int H, int lane) {
    const int c = lane & 7;
    const bool mix = f.vdiff != nullptr, acc = mix && f.dv0_accumulate, ext = f.dv_extra != nullptr;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int tok = tok0 + pass * 8 + (lane >> 3);
        const int64_t vo = (m0 + (tok < N ? tok : N - 1)) * ((int64_t)H * 64) + hh * 64 + 8 * c;
        in.ex[pass] = ext ? *(const uint4 *)(f.dv_extra + vo) : make_uint4(0u, 0u, 0u, 0u);
        in.df[pass] = mix ? *(const uint4 *)(f.vdiff + vo) : make_uint4(0u, 0u, 0u, 0u);
        in.ya[pass] = acc ? *(const uint4 *)(f.dv0 + vo) : make_uint4(0u, 0u, 0u, 0u);
    }
}

// The wave that finishes the shared ragged block asks for that block's rows only after its own epilogue (two row sets do not fit
// the register file next to an epilogue); before the workgroup barrier -- where it waits for the sharing waves anyway -- it
// touches one dword of every 64-byte half row so that the later request is served by the cache.
template <int KIND>
__device__ __forceinline__ void norm_rows_touch(const QkBwd &f, const uint16_t *yrows, int64_t ts, int64_t m0, int tok0, int N, int hh, int H, int lane) {
    const int tok = tok0 + (lane & 31), tc = tok < N ? tok : N - 1;
    const uint32_t a = *(const uint32_t *)(yrows + tc * ts + (lane >> 5) * 32);
    const float r = f.rinv[(m0 + tc) * (2 * H) + KIND * H + hh];
    asm volatile("" ::"v"(a), "v"(r));
}
__device__ __forceinline__ void value_rows_touch(const QkBwd &f, int64_t m0, int tok0, int N, int hh, int H, int lane) {
    const int tok = tok0 + (lane & 31);
    const int64_t vo = (m0 + (tok < N ? tok : N - 1)) * ((int64_t)H * 64) + hh * 64 + (lane >> 5) * 32;
    const bool mix = f.vdiff != nullptr, acc = mix && f.dv0_accumulate, ext = f.dv_extra != nullptr;
    const uint32_t a = ext ? *(const uint32_t *)(f.dv_extra + vo) : 0u, b2 = mix ? *(const uint32_t *)(f.vdiff + vo) : 0u;
    const uint32_t c = acc ? *(const uint32_t *)(f.dv0 + vo) : 0u;
    asm volatile("" ::"v"(a), "v"(b2), "v"(c));
}

// slice: gradient of the rotated, normalised rows y^ of the tile's 32 tokens (first token tok0 of batch row block m0 = b N)
template <int KIND>
__device__ __forceinline__ void staged_norm_rope_bwd(const QkBwd &f, const float *slice, const NormRowsIn &in, int64_t m0, int tok0, int N,
                                                     int hh, int H, int lane) {
    const int c = lane & 7, cp = c ^ 4;
    const float sgn = c < 4 ? 1.0f : -1.0f;
    float wv[8], iw[8], cs[4][8], sn[4][8];
    load8f((KIND ? f.wk : f.wq) + 8 * c, wv);
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {   // the rotary tables of all four rows in one batch (L2-resident)
        const int tok = tok0 + pass * 8 + (lane >> 3), tc = tok < N ? tok : N - 1;
        load8f(f.cosT + tc * 32 + 8 * (c & 3), cs[pass]); load8f(f.sinT + tc * 32 + 8 * (c & 3), sn[pass]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) iw[j] = 1.0f / wv[j];
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int row = pass * 8 + (lane >> 3), tok = tok0 + row;
        float g[8], gp[8], y[8], yp[8];
        load8f(slice + row * AT_ELD + 8 * c, g); load8f(slice + row * AT_ELD + 8 * cp, gp);
        unpack8(in.y[pass], y); unpack8(in.yp[pass], yp);
        float cc = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) cc = fmaf(g[j], y[j], cc);
        cc += xor_lane<1>(cc); cc += xor_lane<2>(cc); cc += xor_lane<4>(cc);
        cc *= 1.0f / 64.0f;
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float u = g[j] * cs[pass][j] + sgn * gp[j] * sn[pass][j];   // R^T dy^
            const float a = y[j] * cs[pass][j] + sgn * yp[j] * sn[pass][j];   // R^T y^ = n w
            o[j] = in.rr[pass] * (wv[j] * u - a * (cc * iw[j]));
        }
        if (tok < N)
            *(uint4 *)(f.dy + (m0 + tok) * f.ldy + (KIND * H + hh) * 64 + 8 * c) =
                make_uint4(pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3]), pack_bf16(o[4], o[5]), pack_bf16(o[6], o[7]));
    }
}

// slice: gradient of the (mixed) values of the tile's tokens; returns this lane's share of <dv, v_raw - v0>
__device__ __forceinline__ float staged_value_bwd(const QkBwd &f, const float *slice, const ValueRowsIn &in, int64_t m0, int tok0, int N,
                                                  int hh, int H, int lane) {
    const int c = lane & 7;
    const bool mix = f.vdiff != nullptr;
    const float l = mix ? f.lam[0] : 1.0f;
    float dl = 0.f;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int row = pass * 8 + (lane >> 3), tok = tok0 + row;
        const bool ok = tok < N;
        const int64_t vo = (m0 + tok) * ((int64_t)H * 64) + hh * 64 + 8 * c;
        float g[8], e[8];
        load8f(slice + row * AT_ELD + 8 * c, g);
        unpack8(in.ex[pass], e);
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] += e[j];
        if (mix) {
            float d[8], y[8], z[8];
            unpack8(in.df[pass], d); unpack8(in.ya[pass], y);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (ok) dl = fmaf(g[j], d[j], dl);
                z[j] = (1.0f - l) * g[j] + y[j];
                g[j] *= l;
            }
            if (ok) *(uint4 *)(f.dv0 + vo) = make_uint4(pack_bf16(z[0], z[1]), pack_bf16(z[2], z[3]), pack_bf16(z[4], z[5]), pack_bf16(z[6], z[7]));
        }
        if (ok)
            *(uint4 *)(f.dy + (m0 + tok) * f.ldy + (2 * H + hh) * 64 + 8 * c) =
                make_uint4(pack_bf16(g[0], g[1]), pack_bf16(g[2], g[3]), pack_bf16(g[4], g[5]), pack_bf16(g[6], g[7]));
        __builtin_amdgcn_sched_barrier(0);   // one pass at a time: interleaved, the four passes' unpacked rows need ~100 more registers
    }
    return dl;
}

// ---- staging of whole row-major operands [npad][64] -> LDS [npad][AT_KLD]; all global loads issued before first use ---
constexpr int AT_BT = 768;                                 // threads of a backward workgroup (12 waves: 3 per SIMD)
// Short sequences (N <= 128: at most four 32-token blocks, the OU example's 101 tokens): workgroups of FOUR waves (BT = 256).  Eight
// idle waves of a 12-wave workgroup still hold their registers, so a CU had room for one workgroup = one (batch, head) pair at a time;
// three four-wave workgroups per CU overlap each other's HBM phases and tile loops.
constexpr int AT_BT_SMALL = 256, AT_MAXN_SMALL = 128;

template <int BT, int MAXN>
__device__ __forceinline__ void stage_two(const uint16_t *a, const uint16_t *b2, int64_t ts, int N, int npad, int tid,
                                          uint16_t *sa, uint16_t *sb) {
    constexpr int AT_SIT = (MAXN * 8 + BT - 1) / BT;   // 16-byte chunks per thread and operand
    uint4 ra[AT_SIT], rb[AT_SIT];
#pragma unroll
    for (int it = 0; it < AT_SIT; ++it) {
        const int i = tid + it * BT, n = i >> 3, c = i & 7;
        const bool ok = i < npad * 8 && n < N;
        ra[it] = ok ? *(const uint4 *)(a + n * ts + c * 8) : make_uint4(0, 0, 0, 0);
        rb[it] = ok ? *(const uint4 *)(b2 + n * ts + c * 8) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int it = 0; it < AT_SIT; ++it) {
        const int i = tid + it * BT, n = i >> 3, c = i & 7;
        if (i < npad * 8) {
            *(uint4 *)(sa + n * AT_KLD + c * 8) = ra[it];
            *(uint4 *)(sb + n * AT_KLD + c * 8) = rb[it];
        }
    }
}

// dq: one workgroup per (batch, head); K and V resident in LDS, a wavefront owns 32 queries at a time.
// The per-wave fragments (B operands with "lane = owned token") are row-strided global loads (one 16-byte piece of a
// different 128-byte row per lane): slow to issue and long-latency.  They are therefore requested one round ahead -- the
// first round's before the operand staging, the next round's before the tile loop -- and only waited for at use.

// dq: one workgroup per (batch, head); K and V resident in LDS, a wavefront owns 32 queries at a time.
// FUSED: delta = <dO, O> is given (the gate backward computes it from the gated output), o is not read, and the result leaves
// through norm_rope_bwd_store as the q columns of dy.
// ONE (FUSED only): no wave has a second block (ntile <= 12, or the ragged 13th block is shared) and no ablation switch is set:
// the in-loop epilogue variants are not compiled -- their hoisted addresses would be spilled around the tile loop, and a kernel
// that uses scratch at all pays for it at every workgroup launch (the dk/dv kernel: 375 -> 442 us with 100 more spilled registers).
template <bool FUSED, bool ONE = false, int BT = AT_BT>
__global__ void __launch_bounds__(BT, AT_BT / BT) attn_bwd_dq_kernel(AttnBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) uint16_t asmem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 31, h2 = lane >> 5;
    const int b = blockIdx.x / p.H, hh = blockIdx.x - b * p.H, N = p.N, npad = p.ntile * 32;
    uint16_t *Ks = asmem, *Vs = asmem + npad * AT_KLD;
    const int64_t ts = (int64_t)p.H * AT_D, base = ((int64_t)b * N * p.H + hh) * AT_D;
    const int64_t srow = ((int64_t)b * p.H + hh) * N;
    bf16x8 qn[4], don[4], on[4];  // fragments of the NEXT round
    float lsen, deln = 0.f;
    auto request = [&](int qblk) {
        const int query = qblk * 32 + fr;
        const bool ok = qblk < p.ntile && query < N;
        load_bfrag(p.q + base, ts, query, ok, h2, qn);
        load_bfrag(p.dout + base, ts, query, ok, h2, don);
        if constexpr (!FUSED) load_bfrag(p.o + base, ts, query, ok, h2, on);
        else deln = ok ? p.delta[srow + query] : 0.f;
        lsen = ok ? p.lse[srow + query] : INFINITY;  // padded queries: P = 0
    };
    request(wave);
    constexpr int NWV = BT / 64;
    const bool split = FUSED && p.split_dq != 0;          // the ragged block NWV is shared by waves 0..3 (AttnBwdParams)
    const bool sharer = split && wave < 4;
    bf16x8 qx[4], dox[4];                                  // ... its fragments, requested with the wave's own before the staging
    float lsex = INFINITY, delx = 0.f;
    if constexpr (FUSED) {
        if (sharer) {
            const int query = NWV * 32 + fr;
            const bool ok = query < N;
            load_bfrag(p.q + base, ts, query, ok, h2, qx);
            load_bfrag(p.dout + base, ts, query, ok, h2, dox);
            delx = ok ? p.delta[srow + query] : 0.f;
            lsex = ok ? p.lse[srow + query] : INFINITY;
        }
    }
    stage_two<BT, (BT == AT_BT ? AT_MAXN : AT_MAXN_SMALL)>(p.k + base, p.v + base, ts, N, npad, tid, Ks, Vs);
    __syncthreads();
    const float c2 = p.scale_log2e;
    const bool ragged = (N & 31) != 0;
    f32x16 acc0, acc1;
    // acc += (K^T dS^T) over the key tiles [kt0, kt1) for the 32 queries whose fragments are qf / dof
    auto sweep = [&](const bf16x8 (&qf)[4], const bf16x8 (&dof)[4], float lse2, float dsum, int kt0, int kt1) {
#pragma unroll 1
        for (int kt = kt0; kt < kt1; ++kt) {
            const uint16_t *kt_ = Ks + kt * 32 * AT_KLD, *vt_ = Vs + kt * 32 * AT_KLD;
            const f32x16 stl = tile_product(kt_ + fr * AT_KLD + h2 * 8, qf);    // S^T  [key][query]
            const f32x16 dpt = tile_product(vt_ + fr * AT_KLD + h2 * 8, dof);   // dP^T [key][query]
            float ds[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) ds[r] = fast_exp2(fmaf(stl[r], c2, -lse2)) * (dpt[r] - dsum);
            if (ragged && kt == p.ntile - 1) {  // keys beyond N (zero rows of K: their P is not 0)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2 >= N) ds[r] = 0.f;
            }
            bf16x8 b0, b1;
            pack_tile(ds, b0, b1);
            accumulate_transposed(kt_, lane, b0, b1, acc0, acc1);  // dQ^T += K^T dS^T
        }
    };
    if constexpr (FUSED) {
        if (sharer) {   // this wave's quarter of the ragged block's key tiles -> a partial tile in spare LDS
            char *mine = (char *)asmem + p.split_dq + wave * p.split_pitch_dq;
            keep_frags((uint4 *)mine, qn, don, lane);
            asm volatile("" : "+v"(mine));
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
            sweep(qx, dox, lsex * 1.4426950408889634f, delx, (p.ntile * wave) >> 2, (p.ntile * (wave + 1)) >> 2);
            restore_frags((const uint4 *)mine, qn, don, lane);
            asm volatile("" : "+v"(mine));
            stage_acc_rows((float *)mine, acc0, acc1, p.scale, lane, NWV * 32 + fr < N);
        }
    }
    const int nblk = split ? NWV : p.ntile;
    int last = -1;   // FUSED: the wave's last round, whose epilogue goes through LDS after the workgroup barrier below
    bf16x8 ykeep[4];   // ONE: the q^ fragments of that block (the epilogue's y^ rows)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) ykeep[ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    for (int qblk = wave; qblk < nblk; qblk += NWV) {
        if (qblk != wave) request(qblk);  // later rounds are rare with 12 waves (N <= 384 needs none): fetch on demand
        const int query = qblk * 32 + fr;
        const bool qok = query < N;
        bf16x8 qf[4], dof[4];
        float dsum = 0.f;  // D_i = <dO_i, O_i>: this lane holds half of the 64 channels of its query
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qf[ks] = qn[ks]; dof[ks] = don[ks];
            if constexpr (!FUSED) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    dsum = fmaf(__uint_as_float(((uint32_t)(uint16_t)don[ks][e]) << 16), __uint_as_float(((uint32_t)(uint16_t)on[ks][e]) << 16), dsum);
            }
        }
        if constexpr (FUSED) dsum = deln;
        else dsum = sum_xor32(dsum);
        const float lse2 = lsen * 1.4426950408889634f;
        if constexpr (!FUSED) { if (qok && h2 == 0) p.delta[srow + query] = dsum; }
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
        sweep(qf, dof, lse2, dsum, 0, p.ntile);
        if constexpr (FUSED) {
            asm volatile("" ::: "memory");   // keep the epilogue's table loads out of the tile loop (loop-invariant: hoisted, they spill)
            if constexpr (ONE) {
                last = qblk;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) ykeep[ks] = qf[ks];
            } else if (qblk + NWV >= nblk && !(p.f.dbg & 8)) last = qblk;
            else if (p.f.dbg & 4) { if (qok) store_transposed(p.f.dy + ((int64_t)b * N + query) * p.f.ldy + hh * 64, h2, acc0, acc1, p.scale); }
            else norm_rope_bwd_store<0>(p.f, acc0, acc1, p.scale, qf, (int64_t)b * N + query, query, hh, p.H, lane, qok);
        }
        else if (qok) store_transposed(p.dq + base + query * ts, h2, acc0, acc1, p.scale);
    }
    if constexpr (FUSED) {
        NormRowsIn in;
        const bool finisher = split && wave == 4;   // sums the ragged block's partial tiles and runs its epilogue (waves 0..3 had the extra tiles)
        if constexpr (!ONE) { if (p.f.dbg & 64) last = -1; }
        if (last >= 0) {
            if constexpr (ONE) norm_rinv_request<0>(p.f, in, (int64_t)b * N, last * 32, N, hh, p.H, lane);
            else norm_rows_request<0>(p.f, in, p.q + base, ts, (int64_t)b * N, last * 32, N, hh, p.H, lane);
        }
        if (finisher) norm_rows_touch<0>(p.f, p.q + base, ts, (int64_t)b * N, NWV * 32, N, hh, p.H, lane);
        __syncthreads();   // every wave is done with K / V: their space becomes the waves' staging slices
        float *slice = (float *)asmem + wave * AT_ESLICE;
        if (last >= 0) {
            if constexpr (ONE) norm_rows_from_frags(in, (uint16_t *)slice, ykeep, lane);
            stage_acc_tile(slice, acc0, acc1, p.scale, lane);
            wave_lds_fence();
            staged_norm_rope_bwd<0>(p.f, slice, in, (int64_t)b * N, last * 32, N, hh, p.H, lane);
        }
        if (finisher) {
            norm_rows_request<0>(p.f, in, p.q + base, ts, (int64_t)b * N, NWV * 32, N, hh, p.H, lane);
            wave_lds_fence();
            sum_partial_tiles(slice, (const float *)((const char *)asmem + p.split_dq), p.split_pitch_dq / 4, p.nragged, lane);
            wave_lds_fence();
            staged_norm_rope_bwd<0>(p.f, slice, in, (int64_t)b * N, NWV * 32, N, hh, p.H, lane);
        }
    }
}

// ---- Round 5 prototype (VSDE_ATTN_DQ_WIDE=1, the unfused API only): dq with 64 queries per wave.  Eight waves at 256 registers, wave w
// owns the query blocks 2w and 2w + 1; a key tile's K rows, V rows and K^T fragments are read from LDS ONCE and feed the MFMAs of both
// blocks -- 12 KB of LDS fragment reads per 24 MFMAs instead of per 12 (on this chip a SIMD's LDS-read, MFMA and VALU time add:
// DESIGN 3.16).  No sharing of a ragged last unit, no fused epilogue: it exists to measure the tile loop of the wide form against
// attn_bwd_dq_kernel<false> at token counts where both are balanced (512: 16 blocks = 8 units = 2 per SIMD either way).
#ifdef VSDE_ABLATIONS   // attn_bwd_dq_wide_kernel: a measured, losing variant -- only in the tools' build (vsde_common.h)
template <bool FUSED>
__global__ void __launch_bounds__(512, 1) attn_bwd_dq_wide_kernel(AttnBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) uint16_t asmem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 31, h2 = lane >> 5;
    const int b = blockIdx.x / p.H, hh = blockIdx.x - b * p.H, N = p.N, npad = p.ntile * 32;
    uint16_t *Ks = asmem, *Vs = asmem + npad * AT_KLD;
    const int64_t ts = (int64_t)p.H * AT_D, base = ((int64_t)b * N * p.H + hh) * AT_D;
    const int64_t srow = ((int64_t)b * p.H + hh) * N;
    bf16x8 qf[2][4], dof[2][4];
    float lse2[2], dsum[2];
    bool qok[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const int qblk = 2 * wave + blk, query = qblk * 32 + fr;
        qok[blk] = qblk < p.ntile && query < N;
        load_bfrag(p.q + base, ts, query, qok[blk], h2, qf[blk]);
        load_bfrag(p.dout + base, ts, query, qok[blk], h2, dof[blk]);
        lse2[blk] = (qok[blk] ? p.lse[srow + query] : INFINITY) * 1.4426950408889634f;   // padded queries: P = 0
        if constexpr (FUSED) {
            dsum[blk] = qok[blk] ? p.delta[srow + query] : 0.f;   // D = <dO, O> from the gate backward
        } else {
            bf16x8 on[4];
            load_bfrag(p.o + base, ts, query, qok[blk], h2, on);
            float d = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    d = fmaf(__uint_as_float(((uint32_t)(uint16_t)dof[blk][ks][e]) << 16), __uint_as_float(((uint32_t)(uint16_t)on[ks][e]) << 16), d);
            dsum[blk] = sum_xor32(d);
            if (qok[blk] && h2 == 0) p.delta[srow + query] = dsum[blk];
        }
    }
    stage_two<512, 512>(p.k + base, p.v + base, ts, N, npad, tid, Ks, Vs);
    __syncthreads();
    const float c2 = p.scale_log2e;
    const bool ragged = (N & 31) != 0;
    f32x16 acc[2][2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc[blk][0][e] = 0.f; acc[blk][1][e] = 0.f; }
    if (2 * wave < p.ntile) {
        const int m = lane & 15;
#pragma unroll 1
        for (int kt = 0; kt < p.ntile; ++kt) {
            const uint16_t *kt_ = Ks + kt * 32 * AT_KLD, *vt_ = Vs + kt * 32 * AT_KLD;
            bf16x8 kr[4], vr[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                kr[ks] = *(const bf16x8 *)(kt_ + fr * AT_KLD + h2 * 8 + ks * 16);
                vr[ks] = *(const bf16x8 *)(vt_ + fr * AT_KLD + h2 * 8 + ks * 16);
            }
            // K^T fragments (see accumulate_transposed): rows = channels dt * 32 + fr, k = the tile's keys in the order of dS's registers
            const uint16_t *src = kt_ + (4 * h2 + (m >> 2)) * AT_KLD + ((lane >> 4) & 1) * 16 + (m & 3) * 4;
            uint4 kw[2][2];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const uint2 a0 = lds_read_tr(src + dt * 32), a1 = lds_read_tr(src + dt * 32 + 8 * AT_KLD);
                const uint2 a2 = lds_read_tr(src + dt * 32 + 16 * AT_KLD), a3 = lds_read_tr(src + dt * 32 + 24 * AT_KLD);
                kw[dt][0] = make_uint4(a0.x, a0.y, a1.x, a1.y); kw[dt][1] = make_uint4(a2.x, a2.y, a3.x, a3.y);
            }
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                f32x16 stl = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, dpt = stl;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) stl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kr[ks], qf[blk][ks], stl, 0, 0, 0);   // S^T  [key][query]
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vr[ks], dof[blk][ks], dpt, 0, 0, 0);  // dP^T [key][query]
                float ds[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) ds[r] = fast_exp2(fmaf(stl[r], c2, -lse2[blk])) * (dpt[r] - dsum[blk]);
                if (ragged && kt == p.ntile - 1) {  // keys beyond N (zero rows of K: their P is not 0)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2 >= N) ds[r] = 0.f;
                }
                bf16x8 b0, b1;
                pack_tile(ds, b0, b1);
                acc[blk][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8 *)&kw[0][0], b0, acc[blk][0], 0, 0, 0);   // dQ^T += K^T dS^T
                acc[blk][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8 *)&kw[0][1], b1, acc[blk][0], 0, 0, 0);
                acc[blk][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8 *)&kw[1][0], b0, acc[blk][1], 0, 0, 0);
                acc[blk][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8 *)&kw[1][1], b1, acc[blk][1], 0, 0, 0);
            }
        }
    }
    if constexpr (!FUSED) {
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const int query = (2 * wave + blk) * 32 + fr;
            if (qok[blk]) store_transposed(p.dq + base + query * ts, h2, acc[blk][0], acc[blk][1], p.scale);
        }
    } else {
        // the staged epilogue of attn_bwd_dq_kernel<true, true>, once per block: the q^ rows come from the fragments, the inverse RMS from
        // memory (requested in front of the barrier), the result leaves as the q columns of dy (norm_rope_bwd through the wave's slice)
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));   // the epilogue's addresses are computed here, not hoisted above the tile loop and spilled
        NormRowsIn in[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
            if ((2 * wave + blk) * 32 < N) norm_rinv_request<0>(p.f, in[blk], (int64_t)b * N, (2 * wave + blk) * 32, N, hh, p.H, lane_e);
        __syncthreads();   // every wave is done with K / V: their space becomes the waves' staging slices
        float *slice = (float *)asmem + wave * AT_ESLICE;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const int tok0 = (2 * wave + blk) * 32;
            if (tok0 < N) {   // wave-uniform
                norm_rows_from_frags(in[blk], (uint16_t *)slice, qf[blk], lane_e);
                stage_acc_tile(slice, acc[blk][0], acc[blk][1], p.scale, lane_e);
                wave_lds_fence();
                staged_norm_rope_bwd<0>(p.f, slice, in[blk], (int64_t)b * N, tok0, N, hh, p.H, lane_e);
                wave_lds_fence();
            }
        }
    }
}
#endif  // VSDE_ABLATIONS (attn_bwd_dq_wide_kernel)

// dk, dv: one workgroup per (batch, head); Q, dO, lse and delta resident in LDS, a wavefront owns 32 keys at a time.
template <bool FUSED, bool ONE = false, int BT = AT_BT>
__global__ void __launch_bounds__(BT, AT_BT / BT) attn_bwd_dkv_kernel(AttnBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) uint16_t asmem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 31, h2 = lane >> 5;
    const int b = blockIdx.x / p.H, hh = blockIdx.x - b * p.H, N = p.N, npad = p.ntile * 32;
    uint16_t *Qs = asmem, *Os = asmem + npad * AT_KLD;
    float *lse2s = (float *)(asmem + 2 * npad * AT_KLD), *dels = lse2s + npad;
    const int64_t ts = (int64_t)p.H * AT_D, base = ((int64_t)b * N * p.H + hh) * AT_D;
    const int64_t srow = ((int64_t)b * p.H + hh) * N;
    bf16x8 kn[4], vn[4];  // fragments of the NEXT round
    auto request = [&](int kblk) {
        const int key = kblk * 32 + fr;
        const bool ok = kblk < p.ntile && key < N;
        load_bfrag(p.k + base, ts, key, ok, h2, kn);
        load_bfrag(p.v + base, ts, key, ok, h2, vn);
    };
    request(wave);
    constexpr int NWV = BT / 64;
    const bool split = FUSED && p.split_dkv != 0;         // the ragged block NWV is shared by waves 0..3 (AttnBwdParams)
    const bool sharer = split && wave < 4;
    bf16x8 kx[4], vx[4];                                   // ... its fragments, requested with the wave's own before the staging
    if constexpr (FUSED) {
        if (sharer) {
            const int key = NWV * 32 + fr;
            load_bfrag(p.k + base, ts, key, key < N, h2, kx);
            load_bfrag(p.v + base, ts, key, key < N, h2, vx);
        }
    }
    stage_two<BT, (BT == AT_BT ? AT_MAXN : AT_MAXN_SMALL)>(p.q + base, p.dout + base, ts, N, npad, tid, Qs, Os);
    for (int i = tid; i < npad; i += BT) {  // padded queries: lse = +inf -> P = 0
        lse2s[i] = i < N ? p.lse[srow + i] * 1.4426950408889634f : INFINITY;
        dels[i] = i < N ? p.delta[srow + i] : 0.f;
    }
    __syncthreads();
    const float c2 = p.scale_log2e;
    f32x16 dk0, dk1, dv0, dv1;
    // dV^T += dO^T P, dK^T += Q^T dS over the query tiles [qt0, qt1) for the 32 keys whose fragments are kf / vf
    auto sweep = [&](const bf16x8 (&kf)[4], const bf16x8 (&vf)[4], int qt0, int qt1) {
#pragma unroll 1
        for (int qt = qt0; qt < qt1; ++qt) {
            const uint16_t *qt_ = Qs + qt * 32 * AT_KLD, *dot_ = Os + qt * 32 * AT_KLD;
            const f32x16 sc = tile_product(qt_ + fr * AT_KLD + h2 * 8, kf);     // S  [query][key]
            const f32x16 dp = tile_product(dot_ + fr * AT_KLD + h2 * 8, vf);    // dP [query][key]
            float pr[16], ds[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {  // registers 4g..4g+3 are queries 8g + 4h2 + 0..3 of the tile
                const float4 l4 = *(const float4 *)(lse2s + qt * 32 + 8 * g + 4 * h2);
                const float4 d4 = *(const float4 *)(dels + qt * 32 + 8 * g + 4 * h2);
                const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e;
                    pr[r] = fast_exp2(fmaf(sc[r], c2, -lv[e]));
                    ds[r] = pr[r] * (dp[r] - dv[e]);
                }
            }
            bf16x8 p0, p1, s0, s1;
            pack_tile(pr, p0, p1);
            pack_tile(ds, s0, s1);
            accumulate_transposed(dot_, lane, p0, p1, dv0, dv1);  // dV^T += dO^T P
            accumulate_transposed(qt_, lane, s0, s1, dk0, dk1);   // dK^T += Q^T dS
        }
    };
    if constexpr (FUSED) {
        if (sharer) {   // this wave's quarter of the ragged block's query tiles -> partial dV / dK tiles in spare LDS
            char *mine = (char *)asmem + p.split_dkv + wave * p.split_pitch_dkv;
            keep_frags((uint4 *)mine, kn, vn, lane);
            asm volatile("" : "+v"(mine));
#pragma unroll
            for (int e = 0; e < 16; ++e) { dk0[e] = 0.f; dk1[e] = 0.f; dv0[e] = 0.f; dv1[e] = 0.f; }
            sweep(kx, vx, (p.ntile * wave) >> 2, (p.ntile * (wave + 1)) >> 2);
            restore_frags((const uint4 *)mine, kn, vn, lane);
            asm volatile("" : "+v"(mine));
            const bool ok = NWV * 32 + fr < N;
            stage_acc_rows((float *)mine, dv0, dv1, 1.0f, lane, ok);
            stage_acc_rows((float *)mine + p.nragged * AT_ELD, dk0, dk1, p.scale, lane, ok);
        }
    }
    const int nblk = split ? NWV : p.ntile;
    int last = -1;   // FUSED: the wave's last round, whose epilogue goes through LDS after the workgroup barrier below
    int parked = -1; // FUSED: the wave's first round when a second one follows and spare LDS can hold its tiles
    bf16x8 ykeep[4];   // ONE: the k^ fragments of the wave's block (the key epilogue's y^ rows)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) ykeep[ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    for (int kblk = wave; kblk < nblk; kblk += NWV) {
        if (kblk != wave) request(kblk);
        const int key = kblk * 32 + fr;
        const bool kok = key < N;
        bf16x8 kf[4], vf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { kf[ks] = kn[ks]; vf[ks] = vn[ks]; }
#pragma unroll
        for (int e = 0; e < 16; ++e) { dk0[e] = 0.f; dk1[e] = 0.f; dv0[e] = 0.f; dv1[e] = 0.f; }
        sweep(kf, vf, 0, p.ntile);
        if constexpr (FUSED) {
            asm volatile("" ::: "memory");   // keep the epilogue's table loads out of the tile loop (they are loop-invariant: hoisted, they spill)
            if constexpr (ONE) {
                last = kblk;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) ykeep[ks] = kf[ks];
                continue;
            }
            if (kblk + NWV >= nblk && !(p.f.dbg & 8)) { last = kblk; continue; }
            if (p.park_off != 0 && kblk < NWV && !(p.f.dbg & 8)) {
                // a wave with one more round to go (the 13th block at N = 401): its epilogue would sit between its two rounds,
                // i.e. on the workgroup's critical path with nothing to hide its loads.  The tile waits in spare LDS instead
                // (already in the staged row layout) and is finished after the barrier below.
                float *park = (float *)((char *)asmem + p.park_off) + wave * (2 * AT_ESLICE);
                stage_acc_tile(park, dv0, dv1, 1.0f, lane);
                stage_acc_tile(park + AT_ESLICE, dk0, dk1, p.scale, lane);
                parked = kblk;
                continue;
            }
            const int64_t m = (int64_t)b * N + key;
            float dl = 0.f;
            if (p.f.dbg & 2) { if (kok) store_transposed(p.f.dy + m * p.f.ldy + (2 * p.H + hh) * 64, h2, dv0, dv1, 1.0f); }
            else dl = kok ? value_bwd_store(p.f, dv0, dv1, m, hh, p.H, lane) : 0.f;   // values first: their accumulators die here
            if (p.f.dlam_partial != nullptr) {   // every key block is owned by exactly one wave: deterministic partials
                dl = wave_sum(dl);
                if (lane == 0) p.f.dlam_partial[(int64_t)blockIdx.x * p.ntile + kblk] = dl;
            }
            if (p.f.dbg & 1) { if (kok) store_transposed(p.f.dy + m * p.f.ldy + (p.H + hh) * 64, h2, dk0, dk1, p.scale); }
            else norm_rope_bwd_store<1>(p.f, dk0, dk1, p.scale, kf, m, key, hh, p.H, lane, kok);
        } else if (kok) {
            store_transposed(p.dk + base + key * ts, h2, dk0, dk1, p.scale);
            store_transposed(p.dv + base + key * ts, h2, dv0, dv1, 1.0f);
        }
    }
    if constexpr (FUSED) {
        // opaque copies of the lane coordinates: the epilogue's addresses are then computed HERE -- hoisted above the tile loops
        // they would be spilled around them (and a kernel that uses scratch at all pays for it at every workgroup launch)
        int lane_e = lane, wave_e = wave;
        asm volatile("" : "+v"(lane_e), "+v"(wave_e));
        const int64_t m0 = (int64_t)b * N;
        NormRowsIn kin; ValueRowsIn vin;
        if constexpr (!ONE) { if (p.f.dbg & 64) last = -1; }
        if (last >= 0) {
            value_rows_request(p.f, vin, m0, last * 32, N, hh, p.H, lane_e);
            // the key rows are only touched here (cache) and requested once the value epilogue has freed its registers: four
            // accumulator tiles + both row sets + an epilogue's temporaries do not fit the register file
            if constexpr (ONE) norm_rinv_request<1>(p.f, kin, m0, last * 32, N, hh, p.H, lane_e);   // (the rows come from the fragments)
            else norm_rows_request<1>(p.f, kin, p.k + base, ts, m0, last * 32, N, hh, p.H, lane_e);
        }
        // the ragged block is finished by two waves that had no extra tiles: wave 4 its values, wave 5 its keys
        if (split && wave_e == 4) value_rows_touch(p.f, m0, NWV * 32, N, hh, p.H, lane_e);
        if (split && wave_e == 5) norm_rows_touch<1>(p.f, p.k + base, ts, m0, NWV * 32, N, hh, p.H, lane_e);
        __syncthreads();   // every wave is done with Q / dO / lse / delta: their space becomes the waves' staging slices
        if (last >= 0) {
            float *slice = (float *)asmem + wave_e * AT_ESLICE;
            // values: the own tile, then (wave 4 of a sharing workgroup) the ragged block's summed partial tiles -- one loop, so
            // that the value epilogue exists once (a second inlined copy costs ~150 spilled registers)
            const int vreps = (split && wave_e == 4) ? 2 : 1;
            stage_acc_tile(slice, dv0, dv1, 1.0f, lane_e);
#pragma unroll 1
            for (int rep = 0; rep < vreps; ++rep) {
                const int blk = rep == 0 ? last : NWV;
                if (rep != 0) {
                    value_rows_request(p.f, vin, m0, blk * 32, N, hh, p.H, lane_e);
                    sum_partial_tiles(slice, (const float *)((const char *)asmem + p.split_dkv), p.split_pitch_dkv / 4, p.nragged, lane_e);
                }
                wave_lds_fence();
                int lane_v = lane_e;   // opaque: the store addresses are recomputed here, not kept (spilled) from the request above
                asm volatile("" : "+v"(lane_v));
                float dl = (!ONE && (p.f.dbg & 32)) ? 0.f : staged_value_bwd(p.f, slice, vin, m0, blk * 32, N, hh, p.H, lane_v);
                if (p.f.dlam_partial != nullptr) {
                    dl = wave_sum(dl);
                    if (lane_e == 0) p.f.dlam_partial[(int64_t)blockIdx.x * p.ntile + blk] = dl;
                }
                wave_lds_fence();
            }
            int lane_k = lane_e;   // opaque: the key epilogue's addresses are computed after the value loop, not spilled around it
            asm volatile("" : "+v"(lane_k));
            if constexpr (ONE) norm_rows_from_frags(kin, (uint16_t *)slice, ykeep, lane_k);
            stage_acc_tile(slice, dk0, dk1, p.scale, lane_k);
            wave_lds_fence();
            if (ONE || !(p.f.dbg & 16)) staged_norm_rope_bwd<1>(p.f, slice, kin, m0, last * 32, N, hh, p.H, lane_k);
            if (split && wave_e == 5) {   // keys of the ragged block
                norm_rows_request<1>(p.f, kin, p.k + base, ts, m0, NWV * 32, N, hh, p.H, lane_k);
                wave_lds_fence();
                sum_partial_tiles(slice, (const float *)((const char *)asmem + p.split_dkv) + p.nragged * AT_ELD, p.split_pitch_dkv / 4, p.nragged, lane_k);
                wave_lds_fence();
                staged_norm_rope_bwd<1>(p.f, slice, kin, m0, NWV * 32, N, hh, p.H, lane_k);
            }
        }
        if (!ONE && parked >= 0) {   // wave-uniform; the parked tiles were written by this wave: in-order LDS, no barrier needed
            const float *park = (const float *)((char *)asmem + p.park_off) + wave_e * (2 * AT_ESLICE);
            value_rows_request(p.f, vin, m0, parked * 32, N, hh, p.H, lane_e);
            norm_rows_request<1>(p.f, kin, p.k + base, ts, m0, parked * 32, N, hh, p.H, lane_e);
            wave_lds_fence();
            float dl = staged_value_bwd(p.f, park, vin, m0, parked * 32, N, hh, p.H, lane_e);
            if (p.f.dlam_partial != nullptr) {
                dl = wave_sum(dl);
                if (lane_e == 0) p.f.dlam_partial[(int64_t)blockIdx.x * p.ntile + parked] = dl;
            }
            staged_norm_rope_bwd<1>(p.f, park + AT_ESLICE, kin, m0, parked * 32, N, hh, p.H, lane_e);
        }
    }
}

// Backward of the sigmoid output gate that the forward folded into the attention store (og = o s, s = rnd(sigmoid(logit)), one
// gate row per token shared by the heads; primitives/attn.py:107-113; `gate` holds s): dO = dout s, the gradient of the LOGITS
// dgate = (1 - s) sum_h dout og, and the attention backward's D = <dO, O> = <dout, og> per (token, head).
// 8 threads per token, 8 channels each.
__global__ void __launch_bounds__(256) gate_bwd_delta_kernel(const uint16_t *__restrict__ dout, const uint16_t *__restrict__ og,
                                                             const uint16_t *__restrict__ gate, int64_t ldg, uint16_t *__restrict__ dattn,
                                                             uint16_t *__restrict__ dgate, int64_t ldd, float *__restrict__ delta,
                                                             int64_t M, int N, int H) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = gid < M * 8;
    const int64_t m = ok ? gid >> 3 : M - 1;
    const int k = (int)(gid & 7) * 8;
    const int64_t b = m / N, n = m - b * N;
    const uint4 g4 = *(const uint4 *)(gate + m * ldg + k);
    const uint32_t gw[4] = {g4.x, g4.y, g4.z, g4.w};
    float s[8], acc[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { s[2 * e] = bfl(gw[e]); s[2 * e + 1] = bfh(gw[e]); }
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int hh = 0; hh < H; ++hh) {
        const int64_t off = (m * H + hh) * 64 + k;
        const uint4 d4 = *(const uint4 *)(dout + off), o4 = *(const uint4 *)(og + off);
        const uint32_t dw[4] = {d4.x, d4.y, d4.z, d4.w}, ow[4] = {o4.x, o4.y, o4.z, o4.w};
        uint32_t r[4];
        float dsum = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d0 = bfl(dw[e]), d1 = bfh(dw[e]), p0 = d0 * bfl(ow[e]), p1 = d1 * bfh(ow[e]);
            acc[2 * e] += p0; acc[2 * e + 1] += p1; dsum += p0 + p1;
            r[e] = pack_bf16(d0 * s[2 * e], d1 * s[2 * e + 1]);
        }
        dsum += __shfl_xor(dsum, 1, 64); dsum += __shfl_xor(dsum, 2, 64); dsum += __shfl_xor(dsum, 4, 64);
        if (ok) {
            *(uint4 *)(dattn + off) = make_uint4(r[0], r[1], r[2], r[3]);
            if (k == 0) delta[(b * H + hh) * N + n] = dsum;
        }
    }
    if (ok)
        *(uint4 *)(dgate + m * ldd + k) = make_uint4(pack_bf16(acc[0] * (1.0f - s[0]), acc[1] * (1.0f - s[1])), pack_bf16(acc[2] * (1.0f - s[2]), acc[3] * (1.0f - s[3])),
                                                     pack_bf16(acc[4] * (1.0f - s[4]), acc[5] * (1.0f - s[5])), pack_bf16(acc[6] * (1.0f - s[6]), acc[7] * (1.0f - s[7])));
}

}  // namespace vsde

using namespace vsde;

// VSDE_ATTN_STREAM=1: take the streamed kernels for every shape (A/B runs)
static bool force_stream() {
    static int f = -1;
    if (f < 0) f = (int)vsde_knob("VSDE_ATTN_STREAM", 0);
    return f != 0;
}

// VSDE_ATTN_PERSIST=0: one workgroup per (batch, head) pair for every shape (A/B runs)
static bool persist_enabled() {
    static int f = -1;
    if (f < 0) f = (int)vsde_knob("VSDE_ATTN_PERSIST", 1);
    return f != 0;
}
// VSDE_ATTN_RING=1: forward with K / V streamed through the LDS ring by a producer wave (attn_fwd_ring_kernel) where it applies.
// Opt-in (it is correct -- bit-identical to the resident kernel -- but slower: profiles/r04_attn_ring.txt); read at every launch so
// that a test can switch it inside one process.
#ifdef VSDE_ABLATIONS
static bool ring_enabled() {
    const char *e = getenv("VSDE_ATTN_RING");
    return e != nullptr && atoi(e) != 0;
}
#endif
// VSDE_ATTN_SMALL_WG=0: short sequences (N <= 128) on the 12-wave workgroups as before (A/B runs)
static bool small_wg_enabled() {
    static int f = -1;
    if (f < 0) f = (int)vsde_knob("VSDE_ATTN_SMALL_WG", 1);
    return f != 0;
}
static int attn_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return cus;
}

// forward launch: persistent workgroups (one per CU, next pair's operands requested behind the 13th query block) when a pair
// has at most 13 query blocks and there are at least two pairs per CU; one workgroup per pair otherwise
static long long *g_attn_trace = nullptr;
#ifdef VSDE_ABLATIONS
static bool fwd8_enabled() {   // VSDE_ATTN_FWD8=1: the eight-wave persistent forward for 385 .. 416 tokens (read per launch: tests toggle it)
    const char *e = getenv("VSDE_ATTN_FWD8");
    return e != nullptr && e[0] == '1';
}
static bool dq_wide_enabled() {   // VSDE_ATTN_DQ_WIDE=1: the dq kernel with 64 queries per wave (read per call: tests toggle it)
    const char *e = getenv("VSDE_ATTN_DQ_WIDE");
    return e != nullptr && e[0] == '1';
}
#endif
static int launch_attn_fwd(const AttnParams &p, int64_t pairs, size_t lds_kv, hipStream_t stream) {
    const int cus = attn_cus();
    const size_t lds = lds_kv, lds_p = lds_kv;
    AttnParams q = p;
    q.pairs = pairs;
    q.trace = g_attn_trace;
#ifdef VSDE_ABLATIONS
    if (g_attn_trace && !(fwd8_enabled() && p.npad == 416) && persist_enabled() && p.npad <= 416 && pairs >= 2 * (int64_t)cus) {   // phase stamps (tools/attn_trace.py)
        VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)attn_fwd_kernel<true, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p));
        hipLaunchKernelGGL((attn_fwd_kernel<true, 16>), dim3((unsigned)cus), dim3(768), lds_p, stream, q);
        VSDE_CHECK_HIP(hipGetLastError());
        return 0;
    }
    { static int dbg = -1; if (dbg < 0) dbg = ablation_env("VSDE_ATTN_RING_DBG"); q.dbg = dbg; }
    static int abl = -1;
    if (abl < 0) abl = ablation_env("VSDE_ATTN_FWD_ABL");
    if (abl && persist_enabled() && p.npad <= 416 && pairs >= 2 * (int64_t)cus) {   // timing-only variants (wrong results)
#define VSDE_ABL_LAUNCH(A_)                                                                                                      \
    do {                                                                                                                         \
        VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)attn_fwd_kernel<true, A_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p)); \
        hipLaunchKernelGGL((attn_fwd_kernel<true, A_>), dim3((unsigned)cus), dim3(768), lds_p, stream, q);                        \
    } while (0)
        if (abl == 1) VSDE_ABL_LAUNCH(1); else if (abl == 2) VSDE_ABL_LAUNCH(2); else if (abl == 3) VSDE_ABL_LAUNCH(3); else VSDE_ABL_LAUNCH(7);
#undef VSDE_ABL_LAUNCH
    } else
#endif
    if (small_wg_enabled() && p.npad <= 128) {
        VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)attn_fwd_kernel<false, 0, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((attn_fwd_kernel<false, 0, 256>), dim3((unsigned)pairs), dim3(256), lds, stream, q);
#ifdef VSDE_ABLATIONS
    } else if (ring_enabled() && p.npad >= 256 && p.npad <= 416 && pairs >= 2 * (int64_t)cus && pairs < (1LL << 31)) {
        // 8 .. 13 key tiles and query blocks (two sweeps of the seven consumer waves), at least two pairs per CU
        const size_t ring = (size_t)2 * RG_SLOTS * RG_TILE * sizeof(uint16_t);
        VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)attn_fwd_ring_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ring));
        hipLaunchKernelGGL(attn_fwd_ring_kernel, dim3((unsigned)cus), dim3(64 * RG_NW), ring, stream, q);
    } else if (fwd8_enabled() && p.npad == 416 && (int64_t)p.N * p.H * AT_D * 2 < (1LL << 31) && pairs >= 2 * (int64_t)cus && pairs < (1LL << 31)) {
        const size_t lds8 = (size_t)2 * p.npad * AT_KLD * sizeof(uint16_t) + (size_t)4 * 34 * 64 * sizeof(float);
        if (g_attn_trace) {
            VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)attn_fwd8_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8));
            hipLaunchKernelGGL(attn_fwd8_kernel<true>, dim3((unsigned)cus), dim3(512), lds8, stream, q);
        } else {
            VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)attn_fwd8_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8));
            hipLaunchKernelGGL(attn_fwd8_kernel<false>, dim3((unsigned)cus), dim3(512), lds8, stream, q);
        }
#endif
    } else if (persist_enabled() && p.npad <= 416 && pairs >= 2 * (int64_t)cus && pairs < (1LL << 31)) {
        VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)attn_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p));
        hipLaunchKernelGGL(attn_fwd_kernel<true>, dim3((unsigned)cus), dim3(768), lds_p, stream, q);
    } else {
        VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)attn_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(attn_fwd_kernel<false>, dim3((unsigned)pairs), dim3(768), lds, stream, q);
    }
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int vsde_attention_max_tokens(void) { return AT_MAXN; }
// debugging: the next persistent forward launches stamp their phases into buf ([12 waves][4 sums + pair count]); nullptr: off
extern "C" int vsde_attn_debug_trace(void *buf) { g_attn_trace = (long long *)buf; return 0; }

extern "C" int vsde_attention_fwd_bf16(const void *q, const void *k, const void *v, void *o, float *lse, int64_t B, int N, int H,
                                       int head_dim, double scale, void *stream) {
    VSDE_CHECK_ARG(q && k && v && o && lse && B > 0 && N > 0 && H > 0, VSDE_E_BADARG, "bad attention arguments");
    VSDE_CHECK_ARG(head_dim == 64 || head_dim == 128, VSDE_E_BADARG, "attention kernels are built for head_dim 64 and 128, got %d", head_dim);
    VSDE_CHECK_ARG(B * H < (1LL << 31), VSDE_E_BADARG, "too many (batch, head) pairs");
    if (head_dim != AT_D || N > AT_MAXN || force_stream())   // does not fit the LDS-resident kernel: K / V stream through LDS (vsde_attn_stream.hip)
        return launch_attention_stream_fwd(q, k, v, o, lse, B, N, H, head_dim, scale, (hipStream_t)stream);
    AttnParams p;
    p.q = (const uint16_t *)q; p.k = (const uint16_t *)k; p.v = (const uint16_t *)v; p.o = (uint16_t *)o; p.lse = lse;
    p.N = N; p.H = H; p.npad = (N + 31) & ~31; p.vld = p.npad + 4;
    p.scale = (float)scale; p.scale_log2e = (float)(scale * 1.4426950408889634);
    p.gate = nullptr; p.ldg = 0;
    const size_t lds = (size_t)2 * p.npad * AT_KLD * sizeof(uint16_t);   // K and V, row-major
    return launch_attn_fwd(p, B * H, lds, (hipStream_t)stream);
}

extern "C" int vsde_attention_bwd_bf16(const void *dout, const void *q, const void *k, const void *v, const void *o, const float *lse,
                                       void *dq, void *dk, void *dv, float *delta, int64_t B, int N, int H, int head_dim,
                                       double scale, void *stream) {
    VSDE_CHECK_ARG(dout && q && k && v && o && lse && dq && dk && dv && delta && B > 0 && N > 0 && H > 0, VSDE_E_BADARG,
                   "bad attention_bwd arguments");
    VSDE_CHECK_ARG(head_dim == 64 || head_dim == 128, VSDE_E_BADARG, "attention kernels are built for head_dim 64 and 128, got %d", head_dim);
    VSDE_CHECK_ARG(B * H < (1LL << 31), VSDE_E_BADARG, "too many (batch, head) pairs");
    if (head_dim != AT_D || N > AT_MAXN || force_stream())
        return launch_attention_stream_bwd(dout, q, k, v, o, lse, dq, dk, dv, delta, B, N, H, head_dim, scale, (hipStream_t)stream);
    AttnBwdParams p = {};
    p.q = (const uint16_t *)q; p.k = (const uint16_t *)k; p.v = (const uint16_t *)v; p.o = (const uint16_t *)o;
    p.dout = (const uint16_t *)dout; p.lse = lse; p.delta = delta;
    p.dq = (uint16_t *)dq; p.dk = (uint16_t *)dk; p.dv = (uint16_t *)dv;
    p.N = N; p.H = H; p.ntile = (N + 31) / 32;
    p.scale = (float)scale; p.scale_log2e = (float)(scale * 1.4426950408889634);
    const size_t lds_dq = (size_t)2 * p.ntile * 32 * AT_KLD * sizeof(uint16_t), lds_dkv = lds_dq + (size_t)2 * p.ntile * 32 * sizeof(float);
    VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)attn_bwd_dq_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dq));
    VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)attn_bwd_dkv_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dkv));
    hipStream_t s = (hipStream_t)stream;
#ifdef VSDE_ABLATIONS
    if (dq_wide_enabled() && p.ntile <= 16) {   // prototype: 64 queries per wave
        VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)attn_bwd_dq_wide_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dq));
        hipLaunchKernelGGL(attn_bwd_dq_wide_kernel<false>, dim3((unsigned)(B * H)), dim3(512), lds_dq, s, p);
    } else
#endif
    hipLaunchKernelGGL(attn_bwd_dq_kernel<false>, dim3((unsigned)(B * H)), dim3(AT_BT), lds_dq, s, p);   // also writes delta, read by the next kernel
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<false>, dim3((unsigned)(B * H)), dim3(AT_BT), lds_dkv, s, p);
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---- training step with the projection-side elementwise work folded into the attention kernels (head_dim 64, N <= AT_MAXN) ----
extern "C" int vsde_attention_fused_supported(int N, int head_dim) { return head_dim == AT_D && N > 0 && N <= AT_MAXN && !force_stream(); }

extern "C" int vsde_attention_fwd_gated_bf16(const void *q, const void *k, const void *v, const void *gate, int64_t ldg, void *o,
                                             float *lse, int64_t B, int N, int H, double scale, void *stream) {
    VSDE_CHECK_ARG(q && k && v && gate && o && lse && B > 0 && N > 0 && H > 0, VSDE_E_BADARG, "bad gated attention arguments");
    VSDE_CHECK_ARG(vsde_attention_fused_supported(N, AT_D), VSDE_E_BADARG, "gated attention runs the LDS-resident kernel: N <= %d", AT_MAXN);
    VSDE_CHECK_ARG(ldg >= 64 && ldg % 4 == 0 && ((uintptr_t)gate % 8) == 0 && B * H < (1LL << 31), VSDE_E_BADARG, "bad gate rows");
    AttnParams p;
    p.q = (const uint16_t *)q; p.k = (const uint16_t *)k; p.v = (const uint16_t *)v; p.o = (uint16_t *)o; p.lse = lse;
    p.N = N; p.H = H; p.npad = (N + 31) & ~31; p.vld = p.npad + 4;
    p.scale = (float)scale; p.scale_log2e = (float)(scale * 1.4426950408889634);
    p.gate = (const uint16_t *)gate; p.ldg = ldg;
    const size_t lds = (size_t)2 * p.npad * AT_KLD * sizeof(uint16_t);   // K and V, row-major
    return launch_attn_fwd(p, B * H, lds, (hipStream_t)stream);
}

extern "C" int vsde_gate_bwd_delta(const void *dout, const void *og, const void *gate, int64_t ldg, void *dattn, void *dgate, int64_t ldd,
                                   float *delta, int64_t B, int N, int H, void *stream) {
    VSDE_CHECK_ARG(dout && og && gate && dattn && dgate && delta && B > 0 && N > 0 && H > 0, VSDE_E_BADARG, "bad gate_bwd_delta arguments");
    VSDE_CHECK_ARG(ldg >= 64 && ldg % 8 == 0 && ldd >= 64 && ldd % 8 == 0 && ((uintptr_t)gate % 16) == 0 && ((uintptr_t)dgate % 16) == 0 &&
                   ((uintptr_t)dout % 16) == 0 && ((uintptr_t)og % 16) == 0 && ((uintptr_t)dattn % 16) == 0, VSDE_E_BADARG,
                   "gate_bwd_delta operands must be 16-byte aligned with row pitches that are multiples of 8 elements");
    const int64_t M = B * N, blocks = (M * 8 + 255) / 256;
    VSDE_CHECK_ARG(blocks < (1LL << 31), VSDE_E_BADARG, "too many rows");
    hipLaunchKernelGGL(gate_bwd_delta_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t *)dout,
                       (const uint16_t *)og, (const uint16_t *)gate, ldg, (uint16_t *)dattn, (uint16_t *)dgate, ldd, delta, M, N, H);
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int64_t vsde_attention_bwd_fused_partials(int64_t B, int N, int H) { return B * H * ((N + 31) / 32); }

extern "C" int vsde_attention_bwd_fused_bf16(const void *dattn, const void *q, const void *k, const void *v, const float *lse,
                                             const float *delta, const float *rinv, const float *cosT, const float *sinT,
                                             const float *wq, const float *wk, const void *vdiff, const float *lam, void *dv0,
                                             int dv0_accumulate, const void *dv_extra, void *dy, int64_t ldy, float *dlam_partial,
                                             int64_t B, int N, int H, double scale, void *stream) {
    VSDE_CHECK_ARG(dattn && q && k && v && lse && delta && rinv && cosT && sinT && wq && wk && dy && B > 0 && N > 0 && H > 0, VSDE_E_BADARG,
                   "bad attention_bwd_fused arguments");
    VSDE_CHECK_ARG(vsde_attention_fused_supported(N, AT_D), VSDE_E_BADARG, "attention_bwd_fused runs the LDS-resident kernels: N <= %d", AT_MAXN);
    VSDE_CHECK_ARG((!vdiff) == (!lam) && (!vdiff) == (!dv0) && (!vdiff) == (!dlam_partial), VSDE_E_BADARG,
                   "value mixing needs vdiff, lam, dv0 and dlam_partial together");
    VSDE_CHECK_ARG(ldy >= 3 * H * 64 && ldy % 4 == 0 && ((uintptr_t)dy % 8) == 0 && ((uintptr_t)cosT % 16) == 0 && ((uintptr_t)sinT % 16) == 0 &&
                   ((uintptr_t)wq % 16) == 0 && ((uintptr_t)wk % 16) == 0 && B * H < (1LL << 31), VSDE_E_BADARG, "bad attention_bwd_fused buffers");
    AttnBwdParams p = {};
    p.q = (const uint16_t *)q; p.k = (const uint16_t *)k; p.v = (const uint16_t *)v; p.o = nullptr;
    p.dout = (const uint16_t *)dattn; p.lse = lse; p.delta = (float *)delta;
    p.N = N; p.H = H; p.ntile = (N + 31) / 32;
    p.scale = (float)scale; p.scale_log2e = (float)(scale * 1.4426950408889634);
    p.f.dy = (uint16_t *)dy; p.f.ldy = ldy; p.f.rinv = rinv; p.f.cosT = cosT; p.f.sinT = sinT; p.f.wq = wq; p.f.wk = wk; p.f.lam = lam;
    p.f.vdiff = (const uint16_t *)vdiff; p.f.dv0 = (uint16_t *)dv0; p.f.dv0_accumulate = dv0_accumulate;
    p.f.dv_extra = (const uint16_t *)dv_extra; p.f.dlam_partial = dlam_partial;
    { static int dbg = -1; if (dbg < 0) dbg = ablation_env("VSDE_ATTN_DEBUG"); p.f.dbg = dbg; }
    const size_t stage = (size_t)(AT_BT / 64) * AT_ESLICE * sizeof(float);   // the waves' epilogue slices reuse the operand space
    size_t lds_dq = (size_t)2 * p.ntile * 32 * AT_KLD * sizeof(uint16_t), lds_dkv = lds_dq + (size_t)2 * p.ntile * 32 * sizeof(float);
    lds_dq = lds_dq > stage ? lds_dq : stage; lds_dkv = lds_dkv > stage ? lds_dkv : stage;
    {   // spare LDS past the operands / staging slices for the first-round tiles of the waves that run two rounds (N = 401: one)
        const int waves = AT_BT / 64, two_round = p.ntile > waves ? p.ntile - waves : 0;
        const size_t park = (size_t)two_round * 2 * AT_ESLICE * sizeof(float);
        if (two_round > 0 && p.ntile <= 2 * waves && lds_dkv + park <= 160 * 1024) { p.park_off = (int)lds_dkv; lds_dkv += park; }
    }
    {   // exactly one block more than waves (N = 385 .. 416): share it among four waves instead of a lone second round, if the
        // partial tiles fit behind the operands (VSDE_ATTN_SPLIT=0: the second round as before)
        static int on = -1;
        if (on < 0) on = (int)vsde_knob("VSDE_ATTN_SPLIT", 1);
        const int waves = AT_BT / 64;
        if (on && p.f.dbg == 0 && p.ntile == waves + 1) {
            p.nragged = N - waves * 32;
            const size_t tile = (size_t)p.nragged * AT_ELD * sizeof(float), keep = 8 * 64 * 16;   // a partial tile; two kept fragment sets
            const size_t pitch_dq = tile > keep ? tile : keep, pitch_dkv = 2 * tile > keep ? 2 * tile : keep;
            if (lds_dq + 4 * pitch_dq <= 160 * 1024) { p.split_dq = (int)lds_dq; p.split_pitch_dq = (int)pitch_dq; lds_dq += 4 * pitch_dq; }
            const size_t operands = lds_dkv - (p.park_off ? (size_t)2 * AT_ESLICE * sizeof(float) : 0);
            if (operands + 4 * pitch_dkv <= 160 * 1024) {
                p.split_dkv = (int)operands; p.split_pitch_dkv = (int)pitch_dkv; p.park_off = 0; lds_dkv = operands + 4 * pitch_dkv;
            }
        }
    }
    hipStream_t s = (hipStream_t)stream;
    const int nwv = AT_BT / 64;
    if (p.f.dbg == 0 && small_wg_enabled() && p.ntile <= AT_BT_SMALL / 64) {   // N <= 128: four-wave workgroups, three per CU
        const size_t stage_s = (size_t)(AT_BT_SMALL / 64) * AT_ESLICE * sizeof(float);
        size_t l_dq = (size_t)2 * p.ntile * 32 * AT_KLD * sizeof(uint16_t), l_dkv = l_dq + (size_t)2 * p.ntile * 32 * sizeof(float);
        l_dq = l_dq > stage_s ? l_dq : stage_s; l_dkv = l_dkv > stage_s ? l_dkv : stage_s;
        p.park_off = 0; p.split_dq = 0; p.split_dkv = 0;
        VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)attn_bwd_dq_kernel<true, true, AT_BT_SMALL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l_dq));
        VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)attn_bwd_dkv_kernel<true, true, AT_BT_SMALL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l_dkv));
        hipLaunchKernelGGL((attn_bwd_dq_kernel<true, true, AT_BT_SMALL>), dim3((unsigned)(B * H)), dim3(AT_BT_SMALL), l_dq, s, p);
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<true, true, AT_BT_SMALL>), dim3((unsigned)(B * H)), dim3(AT_BT_SMALL), l_dkv, s, p);
        VSDE_CHECK_HIP(hipGetLastError());
        return 0;
    }
#ifdef VSDE_ABLATIONS
    if (p.f.dbg == 0 && dq_wide_enabled() && p.ntile > 4 && p.ntile <= 16) {   // 64 queries per wave (attn_bwd_dq_wide_kernel)
        VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)attn_bwd_dq_wide_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dq));
        hipLaunchKernelGGL(attn_bwd_dq_wide_kernel<true>, dim3((unsigned)(B * H)), dim3(512), lds_dq, s, p);
    } else
#endif
    if (p.f.dbg == 0 && (p.ntile <= nwv || p.split_dq != 0)) {   // one block per wave: the lean instantiation
        VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)attn_bwd_dq_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dq));
        hipLaunchKernelGGL((attn_bwd_dq_kernel<true, true>), dim3((unsigned)(B * H)), dim3(AT_BT), lds_dq, s, p);
    } else {
        VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)attn_bwd_dq_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dq));
        hipLaunchKernelGGL(attn_bwd_dq_kernel<true>, dim3((unsigned)(B * H)), dim3(AT_BT), lds_dq, s, p);
    }
    if (p.f.dbg == 0 && (p.ntile <= nwv || p.split_dkv != 0)) {
        VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)attn_bwd_dkv_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dkv));
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<true, true>), dim3((unsigned)(B * H)), dim3(AT_BT), lds_dkv, s, p);
    } else {
        VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)attn_bwd_dkv_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dkv));
        hipLaunchKernelGGL(attn_bwd_dkv_kernel<true>, dim3((unsigned)(B * H)), dim3(AT_BT), lds_dkv, s, p);
    }
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}
