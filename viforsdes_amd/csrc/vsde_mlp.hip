// The SwiGLU feed-forward of a SiT block as ONE kernel per direction on bf16 MFMA (gfx950):
//     y = W_out (silu(a) * b) + b_out,   [a | b] = W_in x + b_in                       (reference: primitives/mlp.py:50-54)
// Round 5.  Until round 4 this was two GEMM launches per direction (csrc/vsde_linear.hip rows kernel with the SwiGLU math in its
// epilogue + a library GEMM over the deep reduction), with the pre-activation u [M, 2H] and s [M, H] written and re-read in
// between.  Here neither exists in the forward: a wave keeps its 32 rows' x AND their 32 x C output accumulators in registers and
// walks the hidden dimension in tiles of 16 units:
//     G1   u-tile (32 rows of W_in: 16 a-rows + 16 b-rows) = C / 16 k-steps of v_mfma_f32_32x32x16_bf16 over the resident x
//     E    s-tile = silu(a) * b in registers -- by the row order of the W_in image the 8 values a lane ends up with ARE the B
//          operand (k = 8 h .. 8 h + 7) of the next product, no cross-lane traffic
//     G2   y += s-tile (K = 16: ONE k-step) x the W_out tile [C][16]: C / 32 MFMAs into the resident accumulators.
// Workgroup = 8 waves x 32 rows = a 256-row stripe, two waves per SIMD (<= 256 registers: x 64 + y 128 + u 16 + fragments).  The
// weight tiles arrive by LDS-DMA (global_load_lds_dwordx4: no staging registers) into a ring of four slots, three tiles ahead of
// their use (measured, profiles/r05_mlp_fwd.txt: a CU pulls tiles out of L2 at ~42 B/clk at best -- with 128-row workgroups the
// tile stream alone took 80 us of the launch; 256 rows halve it, the ring hides it).
// A wave issues in order: its MFMAs only overlap what the OTHER wave of its SIMD does meanwhile.  Waves 4..7 therefore run one
// phase behind waves 0..3 (two workgroup barriers per tile): while one wave of a SIMD runs G1 (16 dependent MFMAs) the other runs
// E + G2 (~100 VALU + 8 MFMAs) -- in lockstep both would fight for the matrix pipe, then both for the VALU.
// The weight operands are IMAGES prepared on the host side (primitives/fused.py::MlpImages, refreshed with the packs):
//     W1 image  [T][32 rows][C + 8] bf16   row rho = 8 g + 4 h + i of tile t holds  (g < 2 ? a : b) unit 16 t + 8 h + 4 (g & 1) + i
//                                           (the MFMA result layout then gives lane (r, h) a_j, b_j for j = 8 h + 0..7);
//                                           16 bytes of padding per row: conflict-free ds_read_b128 with immediate offsets
//     W2 image  [T][2 h][C][8]     bf16   W_out[n][16 t + 8 h + 0..7]: lane (n, h) reads its A fragment at h * 16 C + 16 n
//     b1 image  [T][64]            fp32   b_in in the W1 image's row order (the u accumulators are initialised with it)
// Training additionally writes s [M, H] (natural unit order) for the weight gradient of W_out; u is NOT kept: the backward
// recomputes it from x with the same products in the same order (bit-identical), see mlp_bwd_kernel.
#include <stdlib.h>

#include "vsde_common.h"

namespace vsde {
namespace mlp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 hwbf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pack2(float lo, float hi) {   // v_cvt_pk_bf16_f32 (round to nearest even)
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hwbf16x2));
}
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ float sigm(float x) { return fast_rcp(1.0f + __expf(-x)); }

constexpr int NW = 8;     // waves per workgroup
constexpr int SLD = 72;   // staging row pitch (bf16): 64 columns + 16 bytes
constexpr int MODB = 4;   // block form: batch rows a 256-row stripe may touch (sequences of >= 86 tokens)
template <int C> struct Geo {
    static constexpr int W1_PITCH = 2 * C + 16;                                   // bytes per row of the W1 image
    static constexpr int W1_BYTES = (32 * W1_PITCH + 1023) / 1024 * 1024;         // padded to whole 1 KB DMA pieces
    static constexpr int B1_BYTES = 256;                                          // 64 floats (32 used)
    static constexpr int W2_BYTES = 32 * C;                                       // [2][C][16 bytes]
    static constexpr int BUF = W1_BYTES + B1_BYTES + W2_BYTES;                    // one slot: [W1 | b1 | W2]
    static constexpr int W1_PIECES = W1_BYTES / 1024, W2_PIECES = W2_BYTES / 1024;
    static constexpr int PER = (W1_PIECES + W2_PIECES + 1 + NW - 1) / NW;         // DMA instructions per wave and tile
    static constexpr int NSLOT = 4;
    static constexpr int KS = C / 16, CB = C / 32;
};

struct FwdParams {
    const uint16_t *X; int64_t ldx;     // activations [M][ldx] bf16
    const uint16_t *W1I, *W2I;          // weight images (see the header)
    const float *B1I;
    const uint16_t *b2;                 // [C] bf16 or nullptr
    uint16_t *Y; int64_t ldy;           // [M][ldy]
    uint16_t *S; int64_t lds_;          // training: s [M][lds_] (16 T columns), else nullptr
    int64_t M; int T;                   // T = tiles of 16 hidden units (a multiple of 4, >= 4)
    long long *trace;                   // DBG & 16: per-wave cycle stamps of workgroup 0, [8 waves][2 T + 2][2] (before / after each barrier)
    // BLOCK form (no-grad sampling, template flag BLK): the gated residual and modulated LayerNorm on either side of the MLP are
    // the kernel's prologue and epilogue (reference primitives/sit.py:112-128):
    //     x1 = x + ga * yin;   h = LN(x1) (1 + sc) + sh  -> the MLP input (X is unused);   out = x1 + gm * mlp(h)  -> TOK
    //     hn = LN(out) (1 + sn) + hn_shift -> HOUT   (SN == nullptr: last block, no next norm)
    // per-batch-row vectors [B][mp] bf16 (batch row of row m: m / tokens); roundings to bf16 where the unfused chain has them
    const uint16_t *R0, *R1, *GA, *SC, *SH, *GM, *SN, *HS;
    uint16_t *TOK, *HOUT;
    int64_t mp; int tokens; float eps, eps_next;
    int rotate;                         // 1: per-workgroup rotated tile order (default)
    int antiphase;                      // 2: lockstep, one barrier per tile (default); 1: waves 4..7 run one phase behind waves 0..3 (two barriers
                                        // per tile); 0: lockstep with two barriers (VSDE_MLP_ANTIPHASE, A/B runs)
};

// LDS-DMA of tile t = { W1 image, bias row, W2 image } into the slot at `buf`: 1 KB pieces round-robin over the 8 waves, EVERY wave
// issues exactly G::PER instructions per tile (surplus turns repeat the bias piece) so that one counted s_waitcnt vmcnt(N) serves
// all waves.  issue_piece = turn i of this wave.
template <int C>
__device__ __forceinline__ void issue_piece(const FwdParams &p, int t, char *buf, int wave, int lane, int i) {
    using G = Geo<C>;
    const int piece = wave + NW * i;   // wave-uniform
    if (piece < G::W1_PIECES)
        __builtin_amdgcn_global_load_lds((const void *)((const char *)p.W1I + (int64_t)t * G::W1_BYTES + lane * 16 + piece * 1024),
                                         (__attribute__((address_space(3))) void *)(buf + piece * 1024), 16, 0, 0);
    else if (piece < G::W1_PIECES + G::W2_PIECES)
        __builtin_amdgcn_global_load_lds((const void *)((const char *)p.W2I + (int64_t)t * G::W2_BYTES + lane * 16 + (piece - G::W1_PIECES) * 1024),
                                         (__attribute__((address_space(3))) void *)(buf + G::W1_BYTES + G::B1_BYTES + (piece - G::W1_PIECES) * 1024), 16, 0, 0);
    else
        __builtin_amdgcn_global_load_lds((const void *)((const char *)p.B1I + (int64_t)t * G::B1_BYTES + lane * 4),
                                         (__attribute__((address_space(3))) void *)(buf + G::W1_BYTES), 4, 0, 0);
}
template <int C>
__device__ __forceinline__ void issue_tile(const FwdParams &p, int t, char *buf, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < Geo<C>::PER; ++i) issue_piece<C>(p, t, buf, wave, lane, i);
}

// G1: u accumulators of one tile: bias, then KS k-steps over the resident x fragments; weight fragments are fetched a group of GK
// ahead of their MFMAs (two register sets; sched_barrier pins the order hipcc otherwise undoes by hoisting all 16 reads).
// The wave's DMA turns for tile `tn` (slot `nbuf`; tn < 0: none) sit between the MFMA groups: an LDS-DMA instruction occupies its
// wave for ~150 cycles (the CU's address path takes 64 x 16 bytes at 64 B/clk) -- issued in a burst behind a barrier all eight
// waves stood still for ~600 cycles per tile (tools/mlp_trace.py); here they hide in the shadow of the dependent MFMA chain.
template <int C>
__device__ __forceinline__ void gemm1(f32x16 &uacc, const bf16x8 (&xfr)[C / 16], const char *buf, int lane, const FwdParams &p, int tn,
                                      char *nbuf, int wave) {
    using G = Geo<C>;
    constexpr int GK = 2, NG = G::KS / GK;
    static_assert(G::PER <= NG, "one DMA turn per MFMA group at most");
    const int rho = lane & 31, h = lane >> 5;
    const char *src = buf + rho * G::W1_PITCH + 16 * h;
    bf16x8 bq[2][GK];
#pragma unroll
    for (int k = 0; k < GK; ++k) bq[0][k] = *(const bf16x8 *)(src + 32 * k);
    const float *bias = (const float *)(buf + G::W1_BYTES) + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 b4 = *(const f32x4 *)(bias + 8 * g);
#pragma unroll
        for (int i = 0; i < 4; ++i) uacc[4 * g + i] = b4[i];
    }
#pragma unroll
    for (int gk = 0; gk < NG; ++gk) {
        if (gk + 1 < NG)
#pragma unroll
            for (int k = 0; k < GK; ++k) bq[(gk + 1) & 1][k] = *(const bf16x8 *)(src + 32 * ((gk + 1) * GK + k));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < GK; ++k) uacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[gk & 1][k], xfr[gk * GK + k], uacc, 0, 0, 0);
        if ((gk & 1) && gk / 2 < G::PER / 2 && tn >= 0) issue_piece<C>(p, tn, nbuf, wave, lane, gk / 2);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// E: s = silu(a) * b from the bf16-rounded pre-activations, as the unfused chain computes it under autocast (mlp.py:21-24): quads
// g = 0, 1 of the accumulator are a_j (j = 8 h + 4 g + i), g = 2, 3 the matching b_j.  Written stage by stage over all 8 values:
// a wave issues in order and a dependent VALU result is ~2 issue slots away, so the per-value chains (convert, exp, add, rcp, mul,
// round, mul, round) must be interleaved 8 wide -- pair by pair the phase ran at one instruction per ~9 cycles.
__device__ __forceinline__ bf16x8 swiglu8(const f32x16 &u) {
    uint32_t aw[4], bw[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { aw[q] = pack2(u[2 * q], u[2 * q + 1]); bw[q] = pack2(u[8 + 2 * q], u[8 + 2 * q + 1]); }
    float a[8], e[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) { a[2 * q] = bf_lo(aw[q]); a[2 * q + 1] = bf_hi(aw[q]); }
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = fast_exp2(-1.4426950408889634f * a[j]);
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = fast_rcp(1.0f + e[j]);
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] *= a[j];
    uint32_t tw[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) tw[q] = pack2(e[2 * q], e[2 * q + 1]);
    u32x4 out;
#pragma unroll
    for (int q = 0; q < 4; ++q) out[q] = pack2(bf_lo(tw[q]) * bf_lo(bw[q]), bf_hi(tw[q]) * bf_hi(bw[q]));
    return __builtin_bit_cast(bf16x8, out);
}

// G2: y += s-tile x W2 tile (one k-step per 32-column block of y).  The first half of the W2 fragments is requested before the
// SwiGLU arithmetic (w2a), the second half under the first MFMAs.
template <int C>
__device__ __forceinline__ void gemm2_prefetch(bf16x8 (&w2a)[C / 64], const char *buf, int lane) {
    using G = Geo<C>;
    const char *src = buf + G::W1_BYTES + G::B1_BYTES + (lane >> 5) * (16 * C) + (lane & 31) * 16;
#pragma unroll
    for (int cb = 0; cb < G::CB / 2; ++cb) w2a[cb] = *(const bf16x8 *)(src + cb * 512);
}
template <int C>
__device__ __forceinline__ void gemm2(f32x16 (&yacc)[C / 32], const bf16x8 &sfr, const bf16x8 (&w2a)[C / 64], const char *buf, int lane,
                                      const FwdParams &p, int tn, char *nbuf, int wave) {
    using G = Geo<C>;
    const char *src = buf + G::W1_BYTES + G::B1_BYTES + (lane >> 5) * (16 * C) + (lane & 31) * 16;
    bf16x8 w2b[G::CB / 2];
#pragma unroll
    for (int cb = 0; cb < G::CB / 2; ++cb) w2b[cb] = *(const bf16x8 *)(src + (G::CB / 2 + cb) * 512);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int cb = 0; cb < G::CB / 2; ++cb) yacc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2a[cb], sfr, yacc[cb], 0, 0, 0);
    if (tn >= 0) {   // the second half of the wave's DMA turns for tile tn
#pragma unroll
        for (int i = G::PER / 2; i < G::PER; ++i) issue_piece<C>(p, tn, nbuf, wave, lane, i);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int cb = 0; cb < G::CB / 2; ++cb)
        yacc[G::CB / 2 + cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2b[cb], sfr, yacc[G::CB / 2 + cb], 0, 0, 0);
}

// 32 rows x 64 columns of bf16 out of a wave's staging rows as full 128-byte row segments (non-temporal: never re-read here)
__device__ __forceinline__ void flush64(const uint16_t *stage, uint16_t *dst, int64_t ld, int64_t row0, int64_t M, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (lane >> 3) + 8 * i, c = lane & 7;
        const u32x4 v = *(const u32x4 *)(stage + row * SLD + c * 8);
        if (row0 + row < M) __builtin_nontemporal_store(v, (u32x4 *)(dst + (row0 + row) * ld + c * 8));
    }
}

// End of a phase (a workgroup barrier).  Tile t is read during the phases 2 t .. 2 t + 2 (G1 / E + G2 of waves 0..3, then of waves
// 4..7 one phase later); a wave issues its pieces of tile t + 2 inside its G1(t) -- slot (t + 2) % 4 was tile t - 2's, free since
// the barrier that ended phase 2 t - 2.  g1 (the phase was a G1): the wave's pieces of tile t + 1 must have landed before anyone
// starts G1(t + 1) -- at most the PER younger instructions (tile t + 2) may stay in flight (loads return in order; whatever else
// the counter holds -- the s stores -- only makes the wait longer); `last`: no younger tile was issued, wait for everything.
template <int C, int DBG = 0>
__device__ __forceinline__ void end_phase(const FwdParams &p, int n, bool g1, bool last, int wave, int lane) {
    using G = Geo<C>;
    static_assert(G::PER < 64, "vmcnt is a 6-bit counter");
    if constexpr ((DBG & 16) != 0) { if (blockIdx.x == 0 && lane == 0) p.trace[(wave * (2 * p.T + 2) + n) * 2] = __builtin_readcyclecounter(); }
    if (g1) {
        if (!last) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(G::PER / 2) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if constexpr ((DBG & 16) != 0) { if (blockIdx.x == 0 && lane == 0) p.trace[(wave * (2 * p.T + 2) + n) * 2 + 1] = __builtin_readcyclecounter(); }
}

// SAVE: 1 = also write s (training).  DBG: timing ablations (wrong results): 2 no SwiGLU arithmetic, 4 no y product, 8 no u product
template <int C, int SAVE, int DBG = 0, int BLK = 0>
__global__ void __launch_bounds__(64 * NW, 2) mlp_fwd_kernel(FwdParams p) {
    using G = Geo<C>;
    extern __shared__ __attribute__((aligned(16))) char lsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
    // LDS: the tile slots | the waves' staging rows | b_out
    uint16_t *stage = (uint16_t *)(lsm + G::NSLOT * G::BUF) + wave * (32 * SLD);
    uint16_t *b2row = (uint16_t *)(lsm + G::NSLOT * G::BUF) + NW * (32 * SLD);   // [C] bf16
    const int64_t row0 = ((int64_t)blockIdx.x * NW + wave) * 32;
    if ((int64_t)blockIdx.x * NW * 32 >= p.M) return;
    // tiles 0 and 1 are on their way while the activations load (T >= 4: the host checks)
    // block form: this lane's 4 rows of the row-segment layout; their x / yin chunks are requested before anything else (the only
    // HBM latency of the prologue that nothing can hide: one workgroup per CU)
    int64_t mrow[4]; int brow[4];
    u32x4 xr[BLK ? C / 64 : 1][4], yr[BLK ? C / 64 : 1][4];
    if constexpr (BLK != 0) {
        const int64_t b0 = ((int64_t)blockIdx.x * NW * 32) / p.tokens;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t m = row0 + (lane >> 3) + 8 * i;
            mrow[i] = m < p.M ? m : p.M - 1;
            brow[i] = (int)(mrow[i] / p.tokens - b0);
        }
#pragma unroll
        for (int q = 0; q < C / 64; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                xr[q][i] = *(const u32x4 *)(p.R0 + mrow[i] * C + 64 * q + 8 * (lane & 7));
                yr[q][i] = *(const u32x4 *)(p.R1 + mrow[i] * C + 64 * q + 8 * (lane & 7));
            }
    }
    // Tiles are visited in an order rotated per workgroup (in whole groups of 4 tiles = 64 columns of s): the 256 workgroups of a
    // round pull different lines of the images out of L2 at any moment instead of all queueing for the same ones.
    const int rot = p.rotate ? 4 * (int)((blockIdx.x * 5u) % (unsigned)(p.T / 4)) : 0;
    issue_tile<C>(p, rot % p.T, lsm, wave, lane);
    issue_tile<C>(p, (1 + rot) % p.T, lsm + G::BUF, wave, lane);
    if (tid < C / 8) *(u32x4 *)(b2row + 8 * tid) = p.b2 ? *(const u32x4 *)(p.b2 + 8 * tid) : (u32x4){0u, 0u, 0u, 0u};
    bf16x8 xfr[G::KS];
    // block form: the modulation vectors of the (at most MODB) batch rows this workgroup's 256 rows belong to, [MODB][6][C] bf16 in LDS
    // (ga, sc, sh, gm, sn, hs); per lane: the 4 rows of the row-segment layout and their batch-row slots
    const uint16_t *mods = (const uint16_t *)(lsm + G::NSLOT * G::BUF + NW * 32 * SLD * 2 + C * 2);
    if constexpr (BLK != 0) {
        const int64_t wg0 = (int64_t)blockIdx.x * NW * 32, b0 = wg0 / p.tokens;
        const uint16_t *const vecs[6] = {p.GA, p.SC, p.SH, p.GM, p.SN, p.HS};
        for (int idx = tid; idx < MODB * 6 * (C / 8); idx += 64 * NW) {
            const int ch = idx % (C / 8), v = (idx / (C / 8)) % 6, bb = idx / (6 * (C / 8));
            int64_t b = b0 + bb;
            const int64_t blast = (p.M - 1) / p.tokens;
            b = b < blast ? b : blast;
            u32x4 val = {0u, 0u, 0u, 0u};
            if (vecs[v] != nullptr) val = *(const u32x4 *)(vecs[v] + b * p.mp + ch * 8);
            *(u32x4 *)(const_cast<uint16_t *>(mods) + (bb * 6 + v) * C + ch * 8) = val;
        }
        __syncthreads();   // (before any DMA wait is counted: the tile DMAs above stay in flight -- __syncthreads drains vmcnt, which only costs the prologue some overlap)
    }
    if constexpr (BLK == 0) {
        const int64_t m = row0 + r < p.M ? row0 + r : p.M - 1;   // rows past the end repeat the last one (never stored)
        const uint16_t *src = p.X + m * p.ldx + 8 * h;
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) xfr[ks] = *(const bf16x8 *)(src + ks * 16);
    } else {
        // Prologue of the block form, in the layout of full row segments: lane -> rows (lane >> 3) + 8 i, columns 64 q + 8 c .. + 7
        // (128 bytes of a row per 8 lanes).  x1 = x + ga * yin stays in registers (packed), the row statistics are sums over the
        // lane's 4 chunks and the 8 lanes of a row; h = LN(x1) (1 + sc) + sh then goes through the wave's staging rows into the
        // MFMA fragment layout (lane (r, h): columns 16 ks + 8 h .. + 7 of row r).
        constexpr int NQ = C / 64;
        const int c = lane & 7;
        u32x4 x1[NQ][4];
        float s1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int col = 64 * q + 8 * c;
                const u32x4 xv = xr[q][i], yv = yr[q][i];
                const u32x4 gv = *(const u32x4 *)(mods + (brow[i] * 6 + 0) * C + col);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t t = pack2(bf_lo(gv[e]) * bf_lo(yv[e]), bf_hi(gv[e]) * bf_hi(yv[e]));
                    x1[q][i][e] = pack2(bf_lo(xv[e]) + bf_lo(t), bf_hi(xv[e]) + bf_hi(t));
                    s1[i] += bf_lo(x1[q][i][e]) + bf_hi(x1[q][i][e]);
                }
            }
        float mu[4], rs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float t = s1[i];
            t += xor_lane<1>(t); t += xor_lane<2>(t); t += xor_lane<4>(t);
            mu[i] = t * (1.0f / C);
            float qq = 0.f;
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d0 = bf_lo(x1[q][i][e]) - mu[i], d1 = bf_hi(x1[q][i][e]) - mu[i]; qq += d0 * d0 + d1 * d1; }
            qq += xor_lane<1>(qq); qq += xor_lane<2>(qq); qq += xor_lane<4>(qq);
            rs[i] = rsqrtf(qq * (1.0f / C) + p.eps);
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int col = 64 * q + 8 * c;
                const u32x4 cv = *(const u32x4 *)(mods + (brow[i] * 6 + 1) * C + col), hv = *(const u32x4 *)(mods + (brow[i] * 6 + 2) * C + col);
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    o[e] = pack2((bf_lo(x1[q][i][e]) - mu[i]) * rs[i] * (1.0f + bf_lo(cv[e])) + bf_lo(hv[e]),
                                 (bf_hi(x1[q][i][e]) - mu[i]) * rs[i] * (1.0f + bf_hi(cv[e])) + bf_hi(hv[e]));
                *(u32x4 *)(stage + ((lane >> 3) + 8 * i) * SLD + c * 8) = o;
            }
            wave_lds_fence();
#pragma unroll
            for (int j = 0; j < 4; ++j) xfr[4 * q + j] = *(const bf16x8 *)(stage + r * SLD + 16 * j + 8 * h);
            wave_lds_fence();
        }
    }
    f32x16 yacc[G::CB];
#pragma unroll
    for (int cb = 0; cb < G::CB; ++cb)
#pragma unroll
        for (int e = 0; e < 16; ++e) yacc[cb][e] = 0.f;
    // tile 0 has landed once at most tile 1's instructions are in flight (the x loads above are older still)
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(G::PER) : "memory");
    const int off = p.antiphase == 1 ? wave >> 2 : 0;   // phase offset of this wave
    int n = 0;
    if (off) { end_phase<C, DBG>(p, n, false, false, wave, lane); ++n; }
    f32x16 u;
    for (int t = 0; t < p.T; ++t) {
        char *slot = lsm + (t % G::NSLOT) * G::BUF, *nslot = lsm + ((t + 2) % G::NSLOT) * G::BUF;
        const int tn = t + 2 < p.T ? (t + 2 + rot) % p.T : -1;   // the tile this wave helps fetch during step t
        if (!(DBG & 8)) gemm1<C>(u, xfr, slot, lane, p, tn, nslot, wave);
        if (p.antiphase != 2) { end_phase<C, DBG>(p, n, true, t + 2 >= p.T, wave, lane); ++n; }
        bf16x8 sfr, w2a[G::CB / 2];
        if (!(DBG & 4)) gemm2_prefetch<C>(w2a, slot, lane);
        if (!(DBG & 2)) sfr = swiglu8(u);
        if constexpr (SAVE == 1) {
            *(bf16x8 *)(stage + r * SLD + (t & 3) * 16 + 8 * h) = sfr;
            if ((t & 3) == 3) {
                wave_lds_fence();
                flush64(stage, p.S + ((t - 3 + rot) % p.T) * 16, p.lds_, row0, p.M, lane);
                wave_lds_fence();
            }
        }
        if (!(DBG & 4)) gemm2<C>(yacc, sfr, w2a, slot, lane, p, tn, nslot, wave);
        if (p.antiphase == 2) {   // lockstep, ONE barrier per tile: the next tile has landed, everyone is done with this one
            if (t + 2 < p.T) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(G::PER) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else if (t + 1 < p.T || !off) { end_phase<C, DBG>(p, n, false, false, wave, lane); ++n; }   // (2 T barriers per wave either way)
    }
    // y = acc + b_out, 64 columns at a time through the wave's staging rows
    if constexpr (BLK == 0) {
#pragma unroll
        for (int q = 0; q < G::CB / 2; ++q) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const f32x16 &a = yacc[2 * q + half];
                const uint16_t *bias32 = b2row + 64 * q + 32 * half;
                uint16_t *dst = stage + r * SLD + 32 * half;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const uint2 bb = *(const uint2 *)(bias32 + 8 * g + 4 * h);
                    *(uint2 *)(dst + 8 * g + 4 * h) = make_uint2(pack2(a[4 * g + 0] + bf_lo(bb.x), a[4 * g + 1] + bf_hi(bb.x)),
                                                                 pack2(a[4 * g + 2] + bf_lo(bb.y), a[4 * g + 3] + bf_hi(bb.y)));
                }
            }
            wave_lds_fence();
            flush64(stage, p.Y + 64 * q, p.ldy, row0, p.M, lane);
            wave_lds_fence();
        }
    } else {
        // epilogue of the block form, in the layout of the row-segment stores: lane -> rows (lane >> 3) + 8 i, columns 64 q + 8 c .. + 7;
        // x and yin of chunk q + 1 are requested before chunk q is worked on
        constexpr int NQ = G::CB / 2;
        const int c = lane & 7;
        u32x4 tk[NQ][4];
        float s1[4] = {0.f, 0.f, 0.f, 0.f};
        u32x4 xn[4], yn[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { xn[i] = *(const u32x4 *)(p.R0 + mrow[i] * C + 8 * c); yn[i] = *(const u32x4 *)(p.R1 + mrow[i] * C + 8 * c); }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const f32x16 &a = yacc[2 * q + half];
                const uint16_t *bias32 = b2row + 64 * q + 32 * half;
                uint16_t *dst = stage + r * SLD + 32 * half;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const uint2 bb = *(const uint2 *)(bias32 + 8 * g + 4 * h);
                    *(uint2 *)(dst + 8 * g + 4 * h) = make_uint2(pack2(a[4 * g + 0] + bf_lo(bb.x), a[4 * g + 1] + bf_hi(bb.x)),
                                                                 pack2(a[4 * g + 2] + bf_lo(bb.y), a[4 * g + 3] + bf_hi(bb.y)));
                }
            }
            u32x4 xv[4], yv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { xv[i] = xn[i]; yv[i] = yn[i]; }
            if (q + 1 < NQ) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    xn[i] = *(const u32x4 *)(p.R0 + mrow[i] * C + 64 * (q + 1) + 8 * c);
                    yn[i] = *(const u32x4 *)(p.R1 + mrow[i] * C + 64 * (q + 1) + 8 * c);
                }
            }
            wave_lds_fence();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int col = 64 * q + 8 * c;
                const u32x4 ml = *(const u32x4 *)(stage + ((lane >> 3) + 8 * i) * SLD + c * 8);
                const u32x4 gv = *(const u32x4 *)(mods + (brow[i] * 6 + 0) * C + col), mv = *(const u32x4 *)(mods + (brow[i] * 6 + 3) * C + col);
                u32x4 t;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t gy = pack2(bf_lo(gv[e]) * bf_lo(yv[i][e]), bf_hi(gv[e]) * bf_hi(yv[i][e]));
                    const uint32_t x1 = pack2(bf_lo(xv[i][e]) + bf_lo(gy), bf_hi(xv[i][e]) + bf_hi(gy));
                    const uint32_t gm = pack2(bf_lo(mv[e]) * bf_lo(ml[e]), bf_hi(mv[e]) * bf_hi(ml[e]));
                    t[e] = pack2(bf_lo(x1) + bf_lo(gm), bf_hi(x1) + bf_hi(gm));
                    s1[i] += bf_lo(t[e]) + bf_hi(t[e]);
                }
                tk[q][i] = t;
                if (row0 + (lane >> 3) + 8 * i < p.M) *(u32x4 *)(p.TOK + mrow[i] * C + col) = t;
            }
            wave_lds_fence();
        }
        if (p.SN != nullptr) {   // workgroup-uniform: the next block's modulated LayerNorm of the rows just written
            float mu[4], rs[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float t = s1[i];
                t += xor_lane<1>(t); t += xor_lane<2>(t); t += xor_lane<4>(t);
                mu[i] = t * (1.0f / C);
                float qq = 0.f;
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float d0 = bf_lo(tk[q][i][e]) - mu[i], d1 = bf_hi(tk[q][i][e]) - mu[i]; qq += d0 * d0 + d1 * d1; }
                qq += xor_lane<1>(qq); qq += xor_lane<2>(qq); qq += xor_lane<4>(qq);
                rs[i] = rsqrtf(qq * (1.0f / C) + p.eps_next);
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int col = 64 * q + 8 * c;
                    const u32x4 cv = *(const u32x4 *)(mods + (brow[i] * 6 + 4) * C + col), hv = *(const u32x4 *)(mods + (brow[i] * 6 + 5) * C + col);
                    u32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        o[e] = pack2((bf_lo(tk[q][i][e]) - mu[i]) * rs[i] * (1.0f + bf_lo(cv[e])) + bf_lo(hv[e]),
                                     (bf_hi(tk[q][i][e]) - mu[i]) * rs[i] * (1.0f + bf_hi(cv[e])) + bf_hi(hv[e]));
                    if (row0 + (lane >> 3) + 8 * i < p.M) *(u32x4 *)(p.HOUT + mrow[i] * C + col) = o;
                }
        }
    }
}

template <int C> static size_t fwd_lds_bytes() {
    using G = Geo<C>;
    return (size_t)G::NSLOT * G::BUF + (size_t)NW * 32 * SLD * 2 + (size_t)C * 2 + (size_t)MODB * 6 * C * 2;
}

template <int C, int SAVE, int DBG = 0, int BLK = 0>
static int launch_fwd(const FwdParams &p, hipStream_t s) {
    const size_t lds = fwd_lds_bytes<C>();
    VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)mlp_fwd_kernel<C, SAVE, DBG, BLK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t stripes = (p.M + NW * 32 - 1) / (NW * 32);
    hipLaunchKernelGGL((mlp_fwd_kernel<C, SAVE, DBG, BLK>), dim3((unsigned)stripes), dim3(64 * NW), lds, s, p);
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace mlp
}  // namespace vsde

using namespace vsde;

static long long *g_mlp_trace = nullptr;
// VSDE_MLP_DEBUG=16: device buffer of >= 8 (2 T + 2) 2 int64 that receives workgroup 0's phase stamps (tools/mlp_trace.py)
extern "C" int vsde_mlp_debug_trace(void *buf) { g_mlp_trace = (long long *)buf; return 0; }

// Sizes (bytes) of the three weight images for width C per tile of 16 hidden units: what primitives/fused.py allocates
extern "C" int vsde_mlp_image_bytes(int C, int64_t *w1_tile, int64_t *w2_tile, int64_t *b1_tile) {
    VSDE_CHECK_ARG(C == 128 || C == 256, VSDE_E_BADARG, "fused SwiGLU MLP: width %d not built (128, 256)", C);
    if (C == 128) { *w1_tile = mlp::Geo<128>::W1_BYTES; *w2_tile = mlp::Geo<128>::W2_BYTES; *b1_tile = mlp::Geo<128>::B1_BYTES; }
    else { *w1_tile = mlp::Geo<256>::W1_BYTES; *w2_tile = mlp::Geo<256>::W2_BYTES; *b1_tile = mlp::Geo<256>::B1_BYTES; }
    return 0;
}

static void mlp_env(mlp::FwdParams &p) {
    static int anti = -1, rot = -1;   // VSDE_MLP_ANTIPHASE (see FwdParams), VSDE_MLP_ROTATE=0: every workgroup walks the tiles in the same order
    if (anti < 0) { const char *e = getenv("VSDE_MLP_ANTIPHASE"); anti = e ? atoi(e) : 2; }
    if (rot < 0) { const char *e = getenv("VSDE_MLP_ROTATE"); rot = e ? atoi(e) : 1; }
    p.antiphase = anti; p.rotate = rot; p.trace = g_mlp_trace;
}

// The block form (no-grad): tokens x, attention branch output yin, per-batch-row modulation vectors (row pitch mp, batch row of
// row m = m / tokens) -> tok = x1 + gm * mlp(LN(x1) (1 + sc) + sh) with x1 = x + ga * yin, and hnext = LN(tok) (1 + sn) + hs (sn, hs,
// hnext may be NULL: last block).  Replaces residual_ln_fwd + the MLP + residual_ln_fwd of primitives/sit.py's fused chain.
extern "C" int vsde_mlp_block_fwd_bf16(const void *x, const void *yin, const void *ga, const void *sc, const void *sh, const void *gm,
                                       const void *sn, const void *hs, int64_t mp, int tokens, double eps, double eps_next,
                                       const void *w1_img, const void *w2_img, const float *b1_img, const void *b2, void *tok, void *hnext,
                                       int64_t M, int C, int H, void *stream) {
    VSDE_CHECK_ARG(x && yin && ga && sc && sh && gm && w1_img && w2_img && b1_img && tok && M > 0 && tokens > 0, VSDE_E_BADARG, "bad mlp_block_fwd arguments");
    VSDE_CHECK_ARG((C == 128 || C == 256) && H >= 64 && H % 64 == 0, VSDE_E_BADARG,
                   "fused SwiGLU MLP is built for widths 128 / 256 and a hidden size that is a multiple of 64 (got %d, %d)", C, H);
    VSDE_CHECK_ARG((!sn) == (!hs) && (!sn) == (!hnext), VSDE_E_BADARG, "next-norm scale, shift and output go together");
    const void *ptrs[] = {x, yin, ga, sc, sh, gm, sn, hs, w1_img, w2_img, b1_img, b2, tok, hnext};
    for (const void *q : ptrs) VSDE_CHECK_ARG(((uintptr_t)q % 16) == 0, VSDE_E_BADARG, "mlp_block_fwd operands must be 16-byte aligned");
    VSDE_CHECK_ARG(mp >= C && mp % 8 == 0, VSDE_E_BADARG, "bad modulation row pitch");
    VSDE_CHECK_ARG(tokens >= 86, VSDE_E_BADARG, "mlp_block_fwd keeps the modulation vectors of %d batch rows per 256-row stripe: sequences of >= 86 tokens", mlp::MODB);
    mlp::FwdParams p = {};
    p.R0 = (const uint16_t *)x; p.R1 = (const uint16_t *)yin; p.GA = (const uint16_t *)ga; p.SC = (const uint16_t *)sc; p.SH = (const uint16_t *)sh;
    p.GM = (const uint16_t *)gm; p.SN = (const uint16_t *)sn; p.HS = (const uint16_t *)hs; p.TOK = (uint16_t *)tok; p.HOUT = (uint16_t *)hnext;
    p.mp = mp; p.tokens = tokens; p.eps = (float)eps; p.eps_next = (float)eps_next;
    p.W1I = (const uint16_t *)w1_img; p.W2I = (const uint16_t *)w2_img; p.B1I = b1_img; p.b2 = (const uint16_t *)b2; p.M = M; p.T = H / 16;
    mlp_env(p);
    return C == 256 ? mlp::launch_fwd<256, 0, 0, 1>(p, (hipStream_t)stream) : mlp::launch_fwd<128, 0, 0, 1>(p, (hipStream_t)stream);
}

extern "C" int vsde_mlp_fwd_bf16(const void *x, int64_t ldx, const void *w1_img, const void *w2_img, const float *b1_img, const void *b2,
                                 void *y, int64_t ldy, void *s_out, int64_t lds, int64_t M, int C, int H, void *stream) {
    VSDE_CHECK_ARG(x && w1_img && w2_img && b1_img && y && M > 0, VSDE_E_BADARG, "bad mlp_fwd arguments");
    VSDE_CHECK_ARG((C == 128 || C == 256) && H >= 64 && H % 64 == 0, VSDE_E_BADARG,
                   "fused SwiGLU MLP is built for widths 128 / 256 and a hidden size that is a multiple of 64 (got %d, %d)", C, H);
    VSDE_CHECK_ARG(ldx >= C && ldx % 8 == 0 && ldy >= C && ldy % 8 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 &&
                   ((uintptr_t)w1_img % 16) == 0 && ((uintptr_t)w2_img % 16) == 0 && ((uintptr_t)b1_img % 16) == 0 &&
                   (!b2 || ((uintptr_t)b2 % 16) == 0), VSDE_E_BADARG,
                   "mlp_fwd operands must be 16-byte aligned with row pitches that are multiples of 8 elements");
    VSDE_CHECK_ARG(!s_out || (lds >= H && lds % 8 == 0 && ((uintptr_t)s_out % 16) == 0), VSDE_E_BADARG, "bad mlp_fwd s buffer");
    mlp::FwdParams p = {};
    p.X = (const uint16_t *)x; p.ldx = ldx; p.W1I = (const uint16_t *)w1_img; p.W2I = (const uint16_t *)w2_img; p.B1I = b1_img;
    p.b2 = (const uint16_t *)b2; p.Y = (uint16_t *)y; p.ldy = ldy; p.S = (uint16_t *)s_out; p.lds_ = lds; p.M = M; p.T = H / 16;
    static int dbg = -1;   // VSDE_MLP_DEBUG: timing ablations (wrong results)
    if (dbg < 0) dbg = ablation_env("VSDE_MLP_DEBUG");
    mlp_env(p);
    hipStream_t st = (hipStream_t)stream;
    if (C == 256 && dbg && !s_out) {
        switch (dbg) {
            case 2: return mlp::launch_fwd<256, 0, 2>(p, st);
            case 4: return mlp::launch_fwd<256, 0, 4>(p, st);
            case 8: return mlp::launch_fwd<256, 0, 8>(p, st);
            case 14: return mlp::launch_fwd<256, 0, 14>(p, st);
            case 16: return mlp::launch_fwd<256, 0, 16>(p, st);
            default: break;
        }
    }
    if (C == 256) return s_out ? mlp::launch_fwd<256, 1>(p, st) : mlp::launch_fwd<256, 0>(p, st);
    return s_out ? mlp::launch_fwd<128, 1>(p, st) : mlp::launch_fwd<128, 0>(p, st);
}
