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
#include <type_traits>

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

// Between the last MFMA of an accumulation chain and the first VALU read of its result: the wait states of the 16-pass
// v_mfma_f32_32x32x16_bf16 (19), spelled out.  hipcc pads this hazard inside a basic block; with a branch in between (a run-time
// trace flag in the first version of mlp_bwd_kernel) it did not, and the last accumulator rows were read one k-step early.
__device__ __forceinline__ void mfma_result_guard() { asm volatile("s_nop 15\n\ts_nop 3" ::: "memory"); }

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
    // BLK == 2: W_o arrives in chunks of 4 k-steps (4 x [2][C][16 bytes]); three chunk regions fit in the tile slots
    static constexpr int WO_CHUNK_BYTES = 4 * 32 * C, WO_CHUNKS = KS / 4, WO_PER = WO_CHUNK_BYTES / 1024 / NW;
    static_assert(3 * WO_CHUNK_BYTES <= NSLOT * BUF && WO_CHUNK_BYTES % (1024 * NW) == 0 && WO_CHUNKS <= 4, "W_o chunk regions");
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
    // BLK == 2: the attention branch's out projection is the prologue's first product (reference primitives/attn.py:107-110):
    //     yin = (o * sigmoid(glog[:, k % 64])) W_o^T + b_o   from the attention output o [M][C] and the gate logits [M][ldg] (64 columns);
    // WOI = W_o as C / 16 tiles in the W2 image's format ([2 h][C][8]: W_o[n][16 t + 8 h + 0..7]).  R1 is unused; x1 is parked in TOK
    // between the prologue and the epilogue (same lanes, same addresses).
    const uint16_t *OA, *GL, *WOI, *BO; int64_t ldg;
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

// BLK == 2: chunk ck of the W_o image (4 k-steps) into `buf`, WO_PER DMA instructions per wave
template <int C>
__device__ __forceinline__ void issue_wo_chunk(const FwdParams &p, int ck, char *buf, int wave, int lane) {
    using G = Geo<C>;
#pragma unroll
    for (int i = 0; i < G::WO_PER; ++i) {
        const int piece = wave + NW * i;
        __builtin_amdgcn_global_load_lds((const void *)((const char *)p.WOI + (int64_t)ck * G::WO_CHUNK_BYTES + piece * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void *)(buf + piece * 1024), 16, 0, 0);
    }
}
// BLK == 2: yin accumulators += the 4 k-steps of one W_o chunk over the gated o fragments og[4 ck ..]; A fragments in two register
// sets of C / 64 (half a k-step), fetched one group ahead
template <int C>
__device__ __forceinline__ void gemm0_chunk(f32x16 (&yacc)[C / 32], const bf16x8 (&og)[C / 16], int ck, const char *buf, int lane) {
    using G = Geo<C>;
    constexpr int HC = G::CB / 2, NG = 8;   // groups: (k-step, half of the column blocks)
    const char *src = buf + (lane >> 5) * (16 * C) + (lane & 31) * 16;
    bf16x8 wa[2][HC];
#pragma unroll
    for (int cb = 0; cb < HC; ++cb) wa[0][cb] = *(const bf16x8 *)(src + cb * 512);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG)
#pragma unroll
            for (int cb = 0; cb < HC; ++cb)
                wa[(g + 1) & 1][cb] = *(const bf16x8 *)(src + ((g + 1) >> 1) * (32 * C) + (((g + 1) & 1) * HC + cb) * 512);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int cb = 0; cb < HC; ++cb)
            yacc[(g & 1) * HC + cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[g & 1][cb], og[4 * ck + (g >> 1)], yacc[(g & 1) * HC + cb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
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
    mfma_result_guard();   // the caller reads uacc on the VALU, possibly across a branch (see mfma_result_guard)
}

// E: s = silu(a) * b from the bf16-rounded pre-activations, as the unfused chain computes it under autocast (mlp.py:21-24): quads
// g = 0, 1 of the accumulator are a_j (j = 8 h + 4 g + i), g = 2, 3 the matching b_j.  Written stage by stage over all 8 values:
// a wave issues in order and a dependent VALU result is ~2 issue slots away, so the per-value chains (convert, exp, add, rcp, mul,
// round, mul, round) must be interleaved 8 wide -- pair by pair the phase ran at one instruction per ~9 cycles.
// (Measured and dropped for the no-grad kernels: one rounding at the end instead of the chain's three -- 36 of the 72 plain VALU
// instructions -- 304-311 vs 305 us, block form 421-427 vs 416-419 us: the instruction count of this phase is not what the tile waits for.)
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
    const int64_t wg0 = (int64_t)blockIdx.x * (NW * 32), row0 = wg0 + wave * 32;
    if (wg0 >= p.M) return;
    // tiles 0 and 1 are on their way while the activations load (T >= 4: the host checks)
    // block form: this lane's 4 rows of the row-segment layout; their x / yin chunks are requested before anything else (the only
    // HBM latency of the prologue that nothing can hide: one workgroup per CU)
    int64_t mrow[4]; int brow[4];
    u32x4 xr[BLK == 1 ? C / 64 : 1][4], yr[BLK == 1 ? C / 64 : 1][4];
    if constexpr (BLK != 0) {
        const int64_t b0 = wg0 / p.tokens;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t m = row0 + (lane >> 3) + 8 * i;
            mrow[i] = m < p.M ? m : p.M - 1;
            brow[i] = (int)(mrow[i] / p.tokens - b0);
        }
    }
    if constexpr (BLK == 1) {
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
    bf16x8 xfr[G::KS];
    u32x4 gl[4];
    if constexpr (BLK == 2) {
        // the out projection's operands first: this lane's fragments of o and of the gate logits (row r, columns 16 ks + 8 h .. + 7),
        // and W_o's first chunks (4 k-steps each) into the tile slots, which the MLP does not need yet
        const int64_t m = row0 + r < p.M ? row0 + r : p.M - 1;
        const uint16_t *osrc = p.OA + m * C + 8 * h;
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) xfr[ks] = *(const bf16x8 *)(osrc + ks * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) gl[j] = *(const u32x4 *)(p.GL + m * p.ldg + 16 * j + 8 * h);
#pragma unroll
        for (int ck = 0; ck < (G::WO_CHUNKS < 3 ? G::WO_CHUNKS : 3); ++ck) issue_wo_chunk<C>(p, ck, lsm + ck * G::WO_CHUNK_BYTES, wave, lane);
    } else {
        issue_tile<C>(p, rot % p.T, lsm, wave, lane);
        issue_tile<C>(p, (1 + rot) % p.T, lsm + G::BUF, wave, lane);
    }
    if (tid < C / 8) *(u32x4 *)(b2row + 8 * tid) = p.b2 ? *(const u32x4 *)(p.b2 + 8 * tid) : (u32x4){0u, 0u, 0u, 0u};
    uint16_t *borow = (uint16_t *)(lsm + G::NSLOT * G::BUF + NW * 32 * SLD * 2 + C * 2 + MODB * 6 * C * 2);   // [C] bf16 (BLK == 2)
    if constexpr (BLK == 2) { if (tid < C / 8) *(u32x4 *)(borow + 8 * tid) = p.BO ? *(const u32x4 *)(p.BO + 8 * tid) : (u32x4){0u, 0u, 0u, 0u}; }
    // block form: the modulation vectors of the (at most MODB) batch rows this workgroup's 256 rows belong to, [MODB][6][C] bf16 in LDS
    // (ga, sc, sh, gm, sn, hs); per lane: the 4 rows of the row-segment layout and their batch-row slots
    const uint16_t *mods = (const uint16_t *)(lsm + G::NSLOT * G::BUF + NW * 32 * SLD * 2 + C * 2);
    if constexpr (BLK != 0) {
        const int64_t b0 = wg0 / p.tokens;
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
    f32x16 yacc[G::CB];
#pragma unroll
    for (int cb = 0; cb < G::CB; ++cb)
#pragma unroll
        for (int e = 0; e < 16; ++e) yacc[cb][e] = 0.f;
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
        if constexpr (BLK == 2) {
            // out projection: og = o * rnd(sigmoid(gate logit)) rounded to bf16 (the operand csrc/vsde_linear.hip's gated load builds),
            // then C / 16 k-steps into the (still unused) y accumulators.  Chunks 0..2 of W_o landed before the __syncthreads above.
            u32x4 sg[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) sg[j][e] = pack2(sigm(bf_lo(gl[j][e])), sigm(bf_hi(gl[j][e])));
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) {
                const u32x4 a = __builtin_bit_cast(u32x4, xfr[ks]), g = sg[ks & 3];
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = pack2(bf_lo(a[e]) * bf_lo(g[e]), bf_hi(a[e]) * bf_hi(g[e]));
                xfr[ks] = __builtin_bit_cast(bf16x8, o);
            }
            gemm0_chunk<C>(yacc, xfr, 0, lsm, lane);
            if constexpr (G::WO_CHUNKS == 4) {   // region 0 is free once every wave is through chunk 0: the last chunk goes there
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                issue_wo_chunk<C>(p, 3, lsm, wave, lane);
            }
#pragma unroll
            for (int ck = 1; ck < (G::WO_CHUNKS < 3 ? G::WO_CHUNKS : 3); ++ck) gemm0_chunk<C>(yacc, xfr, ck, lsm + ck * G::WO_CHUNK_BYTES, lane);
            if constexpr (G::WO_CHUNKS == 4) {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                gemm0_chunk<C>(yacc, xfr, 3, lsm, lane);
            }
            mfma_result_guard();
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave is done with W_o: the slots are the MLP's
            issue_tile<C>(p, rot % p.T, lsm, wave, lane);
            issue_tile<C>(p, (1 + rot) % p.T, lsm + G::BUF, wave, lane);
            // yin = acc + b_o (bf16) through the staging rows into the row-segment layout; x1 = x + ga * yin is parked in TOK
            u32x4 xn[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) xn[i] = *(const u32x4 *)(p.R0 + mrow[i] * C + 8 * c);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const f32x16 &a = yacc[2 * q + half];
                    const uint16_t *bias32 = borow + 64 * q + 32 * half;
                    uint16_t *dst = stage + r * SLD + 32 * half;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const uint2 bb = *(const uint2 *)(bias32 + 8 * g + 4 * h);
                        *(uint2 *)(dst + 8 * g + 4 * h) = make_uint2(pack2(a[4 * g + 0] + bf_lo(bb.x), a[4 * g + 1] + bf_hi(bb.x)),
                                                                     pack2(a[4 * g + 2] + bf_lo(bb.y), a[4 * g + 3] + bf_hi(bb.y)));
                    }
                }
                u32x4 xv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) xv[i] = xn[i];
                if (q + 1 < NQ) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) xn[i] = *(const u32x4 *)(p.R0 + mrow[i] * C + 64 * (q + 1) + 8 * c);
                }
                wave_lds_fence();
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int col = 64 * q + 8 * c;
                    const u32x4 yv = *(const u32x4 *)(stage + ((lane >> 3) + 8 * i) * SLD + c * 8);
                    const u32x4 gv = *(const u32x4 *)(mods + (brow[i] * 6 + 0) * C + col);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const uint32_t t = pack2(bf_lo(gv[e]) * bf_lo(yv[e]), bf_hi(gv[e]) * bf_hi(yv[e]));
                        x1[q][i][e] = pack2(bf_lo(xv[i][e]) + bf_lo(t), bf_hi(xv[i][e]) + bf_hi(t));
                        s1[i] += bf_lo(x1[q][i][e]) + bf_hi(x1[q][i][e]);
                    }
                    if (row0 + (lane >> 3) + 8 * i < p.M) *(u32x4 *)(p.TOK + mrow[i] * C + col) = x1[q][i];
                }
                wave_lds_fence();
            }
#pragma unroll
            for (int cb = 0; cb < G::CB; ++cb)
#pragma unroll
                for (int e = 0; e < 16; ++e) yacc[cb][e] = 0.f;
        } else {
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
        const uint16_t *xsrc = BLK == 2 ? p.TOK : p.R0;   // BLK == 2: x1 itself, parked by the prologue
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            xn[i] = *(const u32x4 *)(xsrc + mrow[i] * C + 8 * c);
            if constexpr (BLK == 1) yn[i] = *(const u32x4 *)(p.R1 + mrow[i] * C + 8 * c);
        }
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
            for (int i = 0; i < 4; ++i) { xv[i] = xn[i]; if constexpr (BLK == 1) yv[i] = yn[i]; }
            if (q + 1 < NQ) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    xn[i] = *(const u32x4 *)(xsrc + mrow[i] * C + 64 * (q + 1) + 8 * c);
                    if constexpr (BLK == 1) yn[i] = *(const u32x4 *)(p.R1 + mrow[i] * C + 64 * (q + 1) + 8 * c);
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
                    uint32_t x1 = xv[i][e];
                    if constexpr (BLK == 1) {
                        const uint32_t gy = pack2(bf_lo(gv[e]) * bf_lo(yv[i][e]), bf_hi(gv[e]) * bf_hi(yv[i][e]));
                        x1 = pack2(bf_lo(xv[i][e]) + bf_lo(gy), bf_hi(xv[i][e]) + bf_hi(gy));
                    }
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
    return (size_t)G::NSLOT * G::BUF + (size_t)NW * 32 * SLD * 2 + (size_t)C * 2 + (size_t)MODB * 6 * C * 2 + (size_t)C * 2;
}

template <int C, int SAVE, int DBG = 0, int BLK = 0>
static int launch_fwd(const FwdParams &p, hipStream_t s) {
    const size_t lds = fwd_lds_bytes<C>();
    VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)mlp_fwd_kernel<C, SAVE, DBG, BLK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t stripes = (p.M + NW * 32 - 1) / (NW * 32);
    // (measured and dropped, profiles/r05_mlp_fwd.txt: half stripes -- waves 4..7 idle -- for the partly filled last round of one
    //  workgroup per CU: the wave-uniform "active" branches cost the tile loop 5 scratch accesses and 15 % of its speed, the round
    //  saved 3 %.  Second attempt: the last round's 34 stripes as a SECOND launch of four-wave workgroups (the wave count as a template
    //  parameter, 512 registers): 304.1 vs 304.2 us plain, 404 vs 409 us block form -- a four-wave workgroup takes ~0.9 of the
    //  eight-wave time, not the 0.6 its issue slots suggest, and the launch boundary eats the rest.  Dropped as well.)
    hipLaunchKernelGGL((mlp_fwd_kernel<C, SAVE, DBG, BLK>), dim3((unsigned)stripes), dim3(64 * NW), lds, s, p);
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}

// ================================================================================================= backward
// dx = (swiglu'(u) * (dy W_out)) W_in in ONE kernel (training step; reference primitives/mlp.py:50-54 differentiated): replaces the
// rows kernel with the SwiGLU derivative in its epilogue + the library GEMM over du (csrc/vsde_linear.hip EPI_SWIGLU_BWD, Cijk_*):
// du is written once (the weight gradient of W_in needs it) and never re-read, the 32 x C accumulators of dx stay in registers.
// Per wave (32 rows) and PAIR tile P (32 hidden units = 64 columns of u / du, the 16-row interleaved layout [a16 | b16 | a16 | b16]
// of primitives/fused.py::swiglu_packs(interleave=True)):
//     G1'  ds = dy W2T-tile      16 k-steps over the resident dy fragments; lane (r, h) gets ds for the units tau = 8 g + 4 h + i
//     E'   the u tile (prefetched one tile ahead as full 128-byte row segments) goes through the wave's staging rows into that
//          lane layout; da = ds b sg (1 + a (1 - sg)), db = ds a sg (sg = sigmoid(a)), rounded to bf16, back into the staging rows
//          (-> du leaves as full row segments) AND kept packed: by the k order of the W1 image they ARE the B fragments of G3
//     G3   dx += du W1-tile      4 k-steps (a units 0..15, a 16..31, b 0..15, b 16..31) x C / 32 column blocks.
// One wave per SIMD with the whole register file (dy 64 + dx 128 + tile staging 52 + u prefetch 16 ...): on this chip the matrix
// pipe and the VALU of a SIMD do not overlap (profiles/r05_mfma_valu_overlap.txt), a second wave per SIMD would only hide
// latencies -- and at <= 256 registers the resident operands alone take 192.  Weight tiles: global -> registers (requested a
// tile ahead) -> LDS after the barrier (two slots); an LDS-DMA instruction costs its wave ~150 cycles, 12 per tile and wave.
// Images (primitives/fused.py::MlpBwdImages), per pair tile P contiguous [W2T | W1]:
//     W2T  [32 units][C + 8] bf16   W_out[:, 32 P + tau]                     (A fragments by ds_read_b128, immediate offsets)
//     W1   [4 ks][2 h][C][8] bf16   ks = 2 ab + q, slot e = 4 g' + i  <->  W_in row of (ab, unit 32 P + 8 (2 q + g') + 4 h + i)
namespace bwd {
template <int C> struct Geo {
    static constexpr int W2_PITCH = 2 * C + 16;
    static constexpr int W2_BYTES = (32 * W2_PITCH + 1023) / 1024 * 1024;
    static constexpr int W1_BYTES = 4 * 2 * C * 16;
    static constexpr int BUF = (W2_BYTES + W1_BYTES + 4095) / 4096 * 4096;   // padded: every thread moves exactly NLD 16-byte chunks
    static constexpr int NLD = BUF / 4096;                                     // (a load under an exec mask derailed hipcc's vmcnt
                                                                               //  bookkeeping: a late arrival clobbered a reused register)
    static constexpr int KS = C / 16, CB = C / 32;
};
}  // namespace bwd

struct BwdParams {
    const uint16_t *DY; int64_t lddy;    // [M][lddy] bf16
    const uint16_t *U; int64_t ldu;      // saved pre-activations [M][ldu] (64 columns per pair tile)
    const char *IMG;                     // weight images, bwd::Geo<C>::BUF bytes per pair tile
    uint16_t *DU; int64_t lddu;          // [M][lddu]
    uint16_t *DX; int64_t lddx;          // [M][lddx]
    int64_t M; int TP;                   // TP = pair tiles (hidden / 32)
    long long *trace;                    // debugging (vsde_mlp_debug_trace): per-wave phase cycle sums of workgroup 0, [4 waves][8]
};

#ifdef VSDE_ABLATIONS   // mlp_bwd_kernel: measured 470-490 us against 435 us for rows kernel + library GEMM (profiles/r05_mlp_bwd.txt)
template <int C, int VAR = 3, bool TRACE = false>
__global__ void __launch_bounds__(256, 1) mlp_bwd_kernel(BwdParams p) {
    using G = bwd::Geo<C>;
    extern __shared__ __attribute__((aligned(16))) char lsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
    uint16_t *stage = (uint16_t *)(lsm + 2 * G::BUF) + wave * (32 * SLD);
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * 32;
    if ((int64_t)blockIdx.x * 128 >= p.M) return;
    // tiles in an order rotated per workgroup: the workgroups of a round pull different lines of the images out of L2
    const int rot = (int)((blockIdx.x * 5u) % (unsigned)p.TP);
#define VSDE_TILE(t_) (((t_) + rot) % p.TP)
    u32x4 wreg[G::NLD];   // the next tile's image on its way to LDS
    auto wload = [&](int tile) {
        const char *src = p.IMG + (int64_t)tile * G::BUF + tid * 16;
#pragma unroll
        for (int i = 0; i < G::NLD; ++i) wreg[i] = *(const u32x4 *)(src + i * 4096);
    };
    auto wstore = [&](char *slot) {
#pragma unroll
        for (int i = 0; i < G::NLD; ++i) *(u32x4 *)(slot + tid * 16 + i * 4096) = wreg[i];
    };
    // the slices of u of the next TWO tiles (rows (lane >> 3) + 8 i, 16-byte chunk lane & 7): HBM under load answers in 2-3 us, one
    // tile lasts ~2.5 us
    u32x4 ureg[2][4];
    auto uload = [&](u32x4 (&dst)[4], int tile) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t m = row0 + (lane >> 3) + 8 * i;
            dst[i] = *(const u32x4 *)(p.U + (m < p.M ? m : p.M - 1) * p.ldu + tile * 64 + (lane & 7) * 8);
        }
    };
    wload(VSDE_TILE(0));
    uload(ureg[0], VSDE_TILE(0));
    if (p.TP > 1) uload(ureg[1], VSDE_TILE(1));
    bf16x8 dyfr[G::KS];
    {
        const int64_t m = row0 + r < p.M ? row0 + r : p.M - 1;
        const uint16_t *src = p.DY + m * p.lddy + 8 * h;
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) dyfr[ks] = *(const bf16x8 *)(src + ks * 16);
    }
    f32x16 dx[G::CB];
#pragma unroll
    for (int cb = 0; cb < G::CB; ++cb)
#pragma unroll
        for (int e = 0; e < 16; ++e) dx[cb][e] = 0.f;
    wstore(lsm);
    if (p.TP > 1) wload(VSDE_TILE(1));
    // the resident fragments are complete BEFORE the loop: hipcc otherwise puts their vmcnt(15) .. vmcnt(0) ladder in front of the
    // loop's MFMAs, and a vmcnt(0) inside the loop also drains the u / image loads just issued for the next tile (the first version
    // spent 2,150 cycles per tile in the 16 MFMAs of G1')
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) asm volatile("" : "+v"(dyfr[ks]));
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // (TRACE is a template flag, not a run-time one: a branch between the last MFMA of a chain and the first read of its result made
    //  hipcc drop the MFMA -> VALU wait states -- the v_accvgpr_read of the last accumulator register then returned the value
    //  before the last k-step; see also mfma_result_guard)
#define VSDE_STAMP(k_) do { if constexpr (TRACE) { const long long now_ = __builtin_readcyclecounter(); ph[k_] += now_ - last_; last_ = now_; } } while (0)
    long long last_ = TRACE ? __builtin_readcyclecounter() : 0;
    for (int t0 = 0; t0 < p.TP; t0 += 2) {
#pragma unroll
      for (int par = 0; par < 2; ++par) {
        const int t = t0 + par;
        if (t >= p.TP) break;
        const char *slot = lsm + par * G::BUF;
        const int tile = VSDE_TILE(t);
        // the next tile's image: registers -> the other slot (its last readers finished before the barrier just passed), and the
        // registers are re-issued at once for the tile after it: a whole tile time to arrive
        if constexpr ((VAR & 1) != 0) {
            if (t + 1 < p.TP) wstore(lsm + (1 - par) * G::BUF);
            if (t + 2 < p.TP) wload(VSDE_TILE(t + 2));
        }
        VSDE_STAMP(6);
        // G1': ds = dy W2T tile
        f32x16 ds;
#pragma unroll
        for (int e = 0; e < 16; ++e) ds[e] = 0.f;
        {
            const char *src = slot + r * G::W2_PITCH + 16 * h;
            constexpr int GK = 4, NG = G::KS / GK;
            bf16x8 bq[2][GK];
#pragma unroll
            for (int k = 0; k < GK; ++k) bq[0][k] = *(const bf16x8 *)(src + 32 * k);
#pragma unroll
            for (int gk = 0; gk < NG; ++gk) {
                if (gk + 1 < NG)
#pragma unroll
                    for (int k = 0; k < GK; ++k) bq[(gk + 1) & 1][k] = *(const bf16x8 *)(src + 32 * ((gk + 1) * GK + k));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < GK; ++k) ds = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[gk & 1][k], dyfr[gk * GK + k], ds, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        mfma_result_guard();
        VSDE_STAMP(7);
        // u of this tile -> staging rows (full row segments); its register set is re-issued for the tile after the next one
#pragma unroll
        for (int i = 0; i < 4; ++i) *(u32x4 *)(stage + ((lane >> 3) + 8 * i) * SLD + (lane & 7) * 8) = ureg[par][i];
        if (t + 2 < p.TP) uload(ureg[par], VSDE_TILE(t + 2));
        VSDE_STAMP(0);
        wave_lds_fence();
        // E': du = swiglu'(u) ds, lane-local: quad g of ds <-> units tau = 8 g + 4 h + i <-> u columns 32 (g / 2) + 8 (g % 2) + 4 h + i (a), + 16 (b)
        uint32_t daw[4][2], dbw[4][2];   // [g][pair of i]: packed bf16
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint16_t *pa = stage + r * SLD + 32 * (g >> 1) + 8 * (g & 1) + 4 * h, *pb = pa + 16;
            const uint2 ua = *(const uint2 *)pa, ub = *(const uint2 *)pb;
            const float a[4] = {bf_lo(ua.x), bf_hi(ua.x), bf_lo(ua.y), bf_hi(ua.y)};
            const float b[4] = {bf_lo(ub.x), bf_hi(ub.x), bf_lo(ub.y), bf_hi(ub.y)};
            float da[4], db[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float gs = ds[4 * g + i], sg = fast_rcp(1.0f + fast_exp2(-1.4426950408889634f * a[i]));
                da[i] = gs * b[i] * sg * (1.0f + a[i] * (1.0f - sg));
                db[i] = gs * a[i] * sg;
            }
            daw[g][0] = pack2(da[0], da[1]); daw[g][1] = pack2(da[2], da[3]);
            dbw[g][0] = pack2(db[0], db[1]); dbw[g][1] = pack2(db[2], db[3]);
            *(uint2 *)pa = make_uint2(daw[g][0], daw[g][1]);
            *(uint2 *)pb = make_uint2(dbw[g][0], dbw[g][1]);
        }
        VSDE_STAMP(1);
        wave_lds_fence();
#pragma unroll
        for (int i = 0; i < 4; ++i) {   // du leaves as full 128-byte row segments
            const int row = (lane >> 3) + 8 * i;
            const u32x4 v = *(const u32x4 *)(stage + row * SLD + (lane & 7) * 8);
            if (row0 + row < p.M) __builtin_nontemporal_store(v, (u32x4 *)(p.DU + (row0 + row) * p.lddu + tile * 64 + (lane & 7) * 8));
        }
        VSDE_STAMP(2);
        // G3: dx += du W1 tile; k-step ks = 2 ab + q: this lane's B fragment = (ab ? db : da) of g = 2 q, 2 q + 1.  The 4 CB A fragments
        // are fetched a group of 4 ahead of their MFMAs (left to hipcc every MFMA waited for its own ds_read: 2,100 cycles for 32)
        {
            const char *src = slot + G::W2_BYTES + h * (C * 16) + r * 16;
            constexpr int GF = (VAR & 2) ? 4 : 1, NGR = 4 * G::CB / GF;   // fragment f = ks * CB + cb
            bf16x8 fq[2][GF];
#pragma unroll
            for (int k = 0; k < GF; ++k) fq[0][k] = *(const bf16x8 *)(src + (k / G::CB) * (2 * C * 16) + (k % G::CB) * 512);
#pragma unroll
            for (int gr = 0; gr < NGR; ++gr) {
                if (gr + 1 < NGR)
#pragma unroll
                    for (int k = 0; k < GF; ++k) {
                        const int f = (gr + 1) * GF + k;
                        fq[(gr + 1) & 1][k] = *(const bf16x8 *)(src + (f / G::CB) * (2 * C * 16) + (f % G::CB) * 512);
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < GF; ++k) {
                    const int f = gr * GF + k, ks = f / G::CB, cb = f % G::CB, ab = ks >> 1, q = ks & 1;
                    const u32x4 bw = ab ? (u32x4){dbw[2 * q][0], dbw[2 * q][1], dbw[2 * q + 1][0], dbw[2 * q + 1][1]}
                                        : (u32x4){daw[2 * q][0], daw[2 * q][1], daw[2 * q + 1][0], daw[2 * q + 1][1]};
                    dx[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq[gr & 1][k], __builtin_bit_cast(bf16x8, bw), dx[cb], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        VSDE_STAMP(3);
        if constexpr ((VAR & 1) == 0) {
            if (t + 1 < p.TP) wstore(lsm + (1 - par) * G::BUF);
            if (t + 2 < p.TP) wload(VSDE_TILE(t + 2));
        }
        VSDE_STAMP(4);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        VSDE_STAMP(5);
      }
    }
    if constexpr (TRACE) {
        if (p.trace != nullptr && blockIdx.x == 0 && lane == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) p.trace[wave * 8 + k] = ph[k];
        }
    }
#undef VSDE_STAMP
#undef VSDE_TILE
    // dx, 64 columns at a time through the wave's staging rows
#pragma unroll
    for (int q = 0; q < G::CB / 2; ++q) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const f32x16 &a = dx[2 * q + half];
            uint16_t *dst = stage + r * SLD + 32 * half;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(uint2 *)(dst + 8 * g + 4 * h) = make_uint2(pack2(a[4 * g + 0], a[4 * g + 1]), pack2(a[4 * g + 2], a[4 * g + 3]));
        }
        wave_lds_fence();
        flush64(stage, p.DX + 64 * q, p.lddx, row0, p.M, lane);
        wave_lds_fence();
    }
}

template <int C, int VAR = 3, bool TRACE = false>
static int launch_bwd(const BwdParams &p, hipStream_t s) {
    using G = bwd::Geo<C>;
    const size_t lds = (size_t)2 * G::BUF + (size_t)4 * 32 * SLD * 2;
    VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)mlp_bwd_kernel<C, VAR, TRACE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((mlp_bwd_kernel<C, VAR, TRACE>), dim3((unsigned)((p.M + 127) / 128)), dim3(256), lds, s, p);
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}

#endif  // VSDE_ABLATIONS (mlp_bwd_kernel)
}  // namespace mlp
}  // namespace vsde

using namespace vsde;

static long long *g_mlp_trace = nullptr;
// VSDE_MLP_DEBUG=16: device buffer of >= 8 (2 T + 2) 2 int64 that receives workgroup 0's phase stamps (tools/mlp_trace.py)
extern "C" int vsde_mlp_debug_trace(void *buf) { g_mlp_trace = (long long *)buf; return 0; }

// ===================================================================================================== deep reduction, 256 outputs
// (tools' build only -- VSDE_ABLATIONS: measured 10-15 % behind hipBLASLt at the LV shapes, profiles/r06_deep256_ablation.txt; the
// shipped library answers vsde_linear_deep256_bf16 with an error and primitives/fused.py never routes to it)
#ifdef VSDE_ABLATIONS
// y [M][256] = x [M][K] W^T (+ bias), K a multiple of 64 and DEEP (704 / 832 / 1408 at the LV shapes): the SwiGLU output projection and
// the two input-gradient GEMMs of a SiT block (primitives/mlp.py:54, and the backward of mlp.py:50 / attn.py:80-82) -- the three products
// the library runs.  The weight is an IMAGE: K / 16 k-step images [2 h][256 n][8 k] (the W2 image format), 32 KB per tile of 64 reduction
// indices.  History (profiles/r06_deep256_ablation.txt; K = 704 | 1408 | 832, hipBLASLt 96 | 167 | 113 us):
//   round 5   deep256_kernel: a wave's 32 rows x all 256 columns stationary, fragment-shaped activation loads, tile-synchronous: 125 | 223 | 144
//   round 6   the same with the activations as full 128-byte row segments through per-wave staging rows:                      118 | 200 | 129
//             deep256p_kernel (below): operands requested one k-step ahead through the tile barrier, 64-row tail launch:         113 | 196 | 129
//             deep256q_kernel (below): 64 x 128 per wave, both operands by LDS-DMA:                                             112 | 193 | 126
// Three schedules, one result: with the MFMAs removed the memory / LDS pipeline alone takes as long as hipBLASLt's whole kernel (the
// activation stream then runs at the HBM rate), with only the MFMAs left the loop takes half of that -- and together they ADD (one
// tile-synchronous workgroup per CU: whoever waits for memory holds the matrix pipe's only two waves per SIMD).  The strip-read probe
// (tools/probes/strip_read_probe.hip) says the access pattern itself streams at 6.2 TB/s.
namespace vsde {
namespace mlp {
struct DeepParams {
    const uint16_t *X; int64_t ldx;
    const uint16_t *WI;        // [T * 4 k-steps][2][256][8] bf16
    const uint16_t *bias;      // [256] bf16 or nullptr
    uint16_t *Y; int64_t ldy;
    int64_t M; int T;          // T = tiles of 64 reduction indices
    int rotate;
    int64_t row_begin;         // pipelined form: first row of this launch
};
constexpr int DEEP_TILE = 4 * 32 * 256;   // bytes of one tile's images
constexpr int DEEP_NSLOT = 4;

// ---------------------------------------------------------------------------------------------------------------- round 6: pipelined form
// deep256p_kernel<NWV>: the same product, software-pipelined ACROSS the tile boundary.  What held the first form at ~4,800 cycles per tile
// for 2,048 cycles of MFMAs per SIMD was its schedule: behind every tile barrier all eight waves issued their whole LDS burst (the first
// sixteen weight fragments + the activation shuffle: 24 KB per wave) and waited for all of it with the matrix pipe idle, and once more in
// the middle of the tile.  Here a k-step's operands are requested one k-step ahead THROUGH the barrier:
//   * k-step g = 4 t + ks computes on wa[g & 1] / af[g & 1] while the nine reads of k-step g + 1 (8 weight fragments + 1 activation
//     fragment) are in flight -- for ks = 3 those are the first operands of tile t + 1, whose DMA pieces are guaranteed one tile earlier
//     than before (the barrier that ends tile t guarantees tile t + 2);
//   * the activations arrive as full 128-byte row segments two tiles ahead in two register sets; a set is written to the wave's staging
//     rows at ks = 3 of the tile BEFORE its use (in-order LDS: the last fragment read of the current tile was issued at ks = 2), read from
//     there one fragment per k-step, and re-requested at once for three tiles later;
//   * the tile's DMA (into the slot of the tile that has just ended) is issued behind the barrier, as the last thing of the loop body.
// Vector-memory stream per wave: ... A(t+2) D(t+3) | A(t+3) D(t+4) | ...  (A = NA row-segment loads, D = PW DMA pieces), loads return in
// order: the set A(t+1) has landed once at most [D(t+2) A(t+2) D(t+3)] are in flight, the pieces D(t+2) once at most [A(t+2) D(t+3) A(t+3)].
// NWV = 8: 256-row workgroups (the bulk of a launch); NWV = 2: 64-row workgroups for the rows of the last, partly filled round of
// workgroups (802 stripes on 256 CUs are 3.13 rounds: the 34 stripes of the fourth round run as 136 small workgroups on otherwise idle CUs).
// ABL (timing-only ablations, VSDE_DEEP256_ABL, results wrong): 1 no MFMAs, 2 no activation loads, 4 no weight DMA, 8 no weight fragment
// reads, 16 no activation staging / fragment reads
template <int NWV, int ABL = 0>
__global__ void __launch_bounds__(64 * NWV, NWV == 8 ? 2 : 1) deep256p_kernel(DeepParams p) {
    constexpr int PW = 32 / NWV, NA = 4;
    extern __shared__ __attribute__((aligned(16))) char lsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
    const int64_t wg0 = p.row_begin + (int64_t)blockIdx.x * (32 * NWV), row0 = wg0 + wave * 32;
    if (wg0 >= p.M) return;
    const int T = p.T;
    const int rot = p.rotate ? (int)((blockIdx.x * 5u) % (unsigned)T) : 0;
    auto tile_of = [&](int t) { return ((t < T ? t : T - 1) + rot) % T; };   // (trips past the end re-request the last tile)
    auto issue = [&](int t, int slot) {
        const int tt = tile_of(t);
        if constexpr ((ABL & 4) != 0) { if (t > 3) return; }
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int piece = wave + NWV * i;
            __builtin_amdgcn_global_load_lds((const void *)((const char *)p.WI + (int64_t)tt * DEEP_TILE + piece * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void *)(lsm + slot * DEEP_TILE + piece * 1024), 16, 0, 0);
        }
    };
    // row segments: load i covers rows 8 i .. 8 i + 7 of the wave's 32, lane -> (row 8 i + lane / 8, k-chunk lane % 8); offsets held at the
    // operand's last row for rows past the end (never stored)
    const uint32_t voff0 = (uint32_t)((lane >> 3) * p.ldx * 2 + (lane & 7) * 16), step8 = (uint32_t)(8 * p.ldx * 2);
    const int64_t rbase = row0 < p.M ? row0 : p.M - 1;
    const int64_t left = p.M - 1 - rbase;
    const uint32_t vlast = (uint32_t)((left < 31 ? left : 31) * p.ldx * 2 + (lane & 7) * 16);
    const char *xwave = (const char *)p.X + rbase * p.ldx * 2;
    char *stg = lsm + DEEP_NSLOT * DEEP_TILE + wave * 4096;   // staging rows [32][128 bytes]: chunk c of row q at slot c ^ ((q >> 1) & 7)
    bf16x8 araw[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) araw[i][k] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    auto load_a = [&](int t, bf16x8 (&a)[4]) {
        if constexpr ((ABL & 2) != 0) { if (t > 2) return; }
        const char *base = xwave + tile_of(t) * 128;
        const uint32_t v1 = min(voff0 + step8, vlast), v2 = min(voff0 + 2 * step8, vlast), v3 = min(voff0 + 3 * step8, vlast);
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %4, %8\n\tglobal_load_dwordx4 %1, %5, %8\n\t"
                     "global_load_dwordx4 %2, %6, %8\n\tglobal_load_dwordx4 %3, %7, %8"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(min(voff0, vlast)), "v"(v1), "v"(v2), "v"(v3), "s"(base) : "memory");
    };
    auto stage_a = [&](const bf16x8 (&a)[4]) {
        if constexpr ((ABL & 16) != 0) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = 8 * i + (lane >> 3);
            *(bf16x8 *)(stg + q * 128 + (((lane & 7) ^ ((q >> 1) & 7)) << 4)) = a[i];
        }
        wave_lds_fence();
    };
    const char *afrag = stg + r * 128;
    const int asw = (r >> 1) & 7;
    auto read_af = [&](int ks) { return *(const bf16x8 *)(afrag + (((2 * ks + h) ^ asw) << 4)); };
    f32x16 yacc[8];
#pragma unroll
    for (int cb = 0; cb < 8; ++cb)
#pragma unroll
        for (int e = 0; e < 16; ++e) yacc[cb][e] = 0.f;
    // prologue: D(0) A(0) D(1) A(1) D(2); tile 0 and 1, set 0 landed -> barrier; set 0 -> staging; first operands; A(2) D(3)
    issue(0, 0);
    load_a(0, araw[0]); issue(1, 1);
    load_a(1, araw[1]); issue(2, 2);
    asm volatile("s_waitcnt vmcnt(%4)\n\ts_barrier" : "+v"(araw[0][0]), "+v"(araw[0][1]), "+v"(araw[0][2]), "+v"(araw[0][3]) : "n"(NA + PW) : "memory");
    stage_a(araw[0]);
    bf16x8 wa[2][8], af[2];
    {
        const char *s0 = lsm + h * (16 * 256) + r * 16;
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) wa[0][cb] = *(const bf16x8 *)(s0 + cb * 512);
        af[0] = read_af(0);
    }
    load_a(2, araw[0]);
    issue(3, 3);
    // tile t; `nxt`: the register set that holds A(t + 1) (requested two tiles ago) and takes A(t + 3)
    auto tile = [&](int t, bf16x8 (&nxt)[4]) {
        const char *slot = lsm + (t % DEEP_NSLOT) * DEEP_TILE + h * (16 * 256) + r * 16;
        const char *slot1 = lsm + ((t + 1) % DEEP_NSLOT) * DEEP_TILE + h * (16 * 256) + r * 16;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int cur = ks & 1, nb = cur ^ 1;
            if (ks < 3) {
                if constexpr ((ABL & 8) == 0) {
#pragma unroll
                    for (int cb = 0; cb < 8; ++cb) wa[nb][cb] = *(const bf16x8 *)(slot + (ks + 1) * 8192 + cb * 512);
                }
                if constexpr ((ABL & 16) == 0) af[nb] = read_af(ks + 1);
            } else {
                // A(t + 1) has landed once at most [D(t+2) A(t+2) D(t+3)] are in flight; the set is an operand of the wait so that nothing
                // below moves in front of it
                asm volatile("s_waitcnt vmcnt(%4)" : "+v"(nxt[0]), "+v"(nxt[1]), "+v"(nxt[2]), "+v"(nxt[3]) : "n"(NA + 2 * PW) : "memory");
                stage_a(nxt);
                if constexpr ((ABL & 8) == 0) {
#pragma unroll
                    for (int cb = 0; cb < 8; ++cb) wa[nb][cb] = *(const bf16x8 *)(slot1 + cb * 512);
                }
                if constexpr ((ABL & 16) == 0) af[nb] = read_af(0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr ((ABL & 1) == 0) {
#pragma unroll
                for (int cb = 0; cb < 8; ++cb) yacc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[cur][cb], af[cur], yacc[cb], 0, 0, 0);
            } else {
#pragma unroll
                for (int cb = 0; cb < 8; ++cb) asm volatile("" ::"v"(wa[cur][cb]), "v"(af[cur]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        load_a(t + 3, nxt);
        // D(t + 2) (issued behind the barrier of tile t - 2) has landed once at most [A(t+2) D(t+3) A(t+3)] are in flight
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * NA + PW) : "memory");
        issue(t + 4, t % DEEP_NSLOT);   // the slot everyone has just left
    };
    int t = 0;
    for (; t + 1 < T; t += 2) { tile(t, araw[1]); tile(t + 1, araw[0]); }
    if (t < T) tile(t, araw[1]);
    // the trailing requests must not outlive the tile slots' reuse below -- nor their destination registers' (kept alive up to here)
#pragma unroll
    for (int i = 0; i < 2; ++i) asm volatile("" ::"v"(araw[i][0]), "v"(araw[i][1]), "v"(araw[i][2]), "v"(araw[i][3]));
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int i = 0; i < 2; ++i) asm volatile("" ::"v"(araw[i][0]), "v"(araw[i][1]), "v"(araw[i][2]), "v"(araw[i][3]));
    mfma_result_guard();
    // y = acc + bias, 64 columns at a time through this wave's staging rows (the tile slots are free now)
    uint16_t *stage = (uint16_t *)lsm + wave * (32 * SLD);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const f32x16 &a = yacc[2 * q + half];
            uint16_t *dst = stage + r * SLD + 32 * half;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 bb = make_uint2(0u, 0u);
                if (p.bias != nullptr) bb = *(const uint2 *)(p.bias + 64 * q + 32 * half + 8 * g + 4 * h);
                *(uint2 *)(dst + 8 * g + 4 * h) = make_uint2(pack2(a[4 * g + 0] + bf_lo(bb.x), a[4 * g + 1] + bf_hi(bb.x)),
                                                             pack2(a[4 * g + 2] + bf_lo(bb.y), a[4 * g + 3] + bf_hi(bb.y)));
            }
        }
        wave_lds_fence();
        flush64(stage, p.Y + 64 * q, p.ldy, row0, p.M, lane);
        wave_lds_fence();
    }
}

// ---------------------------------------------------------------------------------------------------------- round 6: two-dimensional form
// deep256q_kernel: the bulk kernel of the deep reductions.  The ablations of deep256p_kernel (profiles/r06_deep256_ablation.txt) say that
// its matrix pipe (2,048 cycles per tile and SIMD) and its LDS / memory pipeline (~2,600 cycles per tile with the MFMAs removed) run one
// AFTER the other, and that the LDS is the busiest unit: every wave reads the whole 32 KB weight tile (1 KB of fragment per MFMA) and
// shuffles its activation rows through staging rows on top.  Here the workgroup's 256 x 256 output block is cut in TWO dimensions --
// wave (wr, wc) owns rows 64 wr .. + 63 and columns 128 wc .. + 127 (2 x 4 accumulator blocks, 128 registers as before) -- so that a k-step
// is 2 activation + 4 weight fragment reads for 8 MFMAs (0.75 KB per MFMA), and BOTH operands arrive by LDS-DMA: no staging registers, no
// staging writes.  Activation tiles [256 rows][128 bytes] are row-major with the 16-byte chunks of row q XOR-ed by (q >> 1) & 7 -- applied
// to the SOURCE address of the lane-linear DMA and to the fragment reads (conflict-free ds_read_b128) -- in a ring of three slots, two
// tiles ahead (HBM); weight tiles (the W2 image format) in a ring of two slots, one tile ahead (L2).  160 KB of LDS, one workgroup per CU.
// Vector-memory stream per wave (4 + 4 pieces per tile): the group [W(t+2) A(t+3)] is issued behind the barrier that ends tile t; the
// barrier that ends tile t + 1 is preceded by vmcnt(4): everything but A(t+3) has landed, i.e. W(t+2) and A(t+2).
constexpr int DQ_ASLOT = 256 * 128, DQ_NA = 3, DQ_NW = 2;
template <int ABL = 0>
__global__ void __launch_bounds__(512, 2) deep256q_kernel(DeepParams p) {
    extern __shared__ __attribute__((aligned(16))) char lsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    const int64_t wg0 = p.row_begin + (int64_t)blockIdx.x * 256;
    if (wg0 >= p.M) return;
    const int T = p.T;
    const int rot = p.rotate ? (int)((blockIdx.x * 5u) % (unsigned)T) : 0;
    auto tile_of = [&](int t) { return ((t < T ? t : T - 1) + rot) % T; };
    char *aring = lsm, *wring = lsm + DQ_NA * DQ_ASLOT;
    // activation pieces of this wave: piece j = wave + 8 i covers rows 8 j .. 8 j + 7; lane -> row 8 j + lane / 8, LDS chunk position lane % 8,
    // source chunk (lane % 8) ^ ((row >> 1) & 7).  Rows past the end read the last row (never stored).
    const char *asrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = 8 * (wave + 8 * i) + (lane >> 3);
        int64_t m = wg0 + q;
        m = m < p.M ? m : p.M - 1;
        asrc[i] = (const char *)p.X + m * p.ldx * 2 + ((((lane & 7) ^ ((q >> 1) & 7))) << 4);
    }
    auto issue_w = [&](int t, int slot) {
        if constexpr ((ABL & 4) != 0) { if (t > 1) return; }
        const int tt = tile_of(t);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = wave + 8 * i;
            __builtin_amdgcn_global_load_lds((const void *)((const char *)p.WI + (int64_t)tt * DEEP_TILE + piece * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void *)(wring + slot * DEEP_TILE + piece * 1024), 16, 0, 0);
        }
    };
    auto issue_a = [&](int t, int slot) {
        if constexpr ((ABL & 2) != 0) { if (t > 2) return; }
        const int tt = tile_of(t);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = wave + 8 * i;
            __builtin_amdgcn_global_load_lds((const void *)(asrc[i] + tt * 128),
                                             (__attribute__((address_space(3))) void *)(aring + slot * DQ_ASLOT + piece * 1024), 16, 0, 0);
        }
    };
    f32x16 yacc[2][4];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int e = 0; e < 16; ++e) yacc[rb][cb][e] = 0.f;
    // fragment addresses inside a slot: weight (k-step ks, column block 4 wc + cb) at ks * 8192 + h * 4096 + (32 (4 wc + cb) + r) * 16;
    // activation (row block rb, k-step ks): row q = 64 wr + 32 rb + r, chunk (2 ks + h) ^ ((q >> 1) & 7)
    const int woff = h * 4096 + (128 * wc + r) * 16;
    int aoff[2], asw[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) { const int q = 64 * wr + 32 * rb + r; aoff[rb] = q * 128; asw[rb] = (q >> 1) & 7; }
    // prologue: W(0) A(0) A(1) | W(1) A(2); tile 0 needs the first two groups: at most [A(1) W(1) A(2)] stay in flight
    issue_w(0, 0); issue_a(0, 0); issue_a(1, 1);
    issue_w(1, 1); issue_a(2, 2);
    asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");
    for (int t = 0; t < T; ++t) {
        const char *ws = wring + (t % DQ_NW) * DEEP_TILE + woff;
        const char *as = aring + (t % DQ_NA) * DQ_ASLOT;
        bf16x8 wa[2][4], af[2][2];
        auto fetch = [&](int ks, int b) {
            if constexpr ((ABL & 8) == 0) {
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) wa[b][cb] = *(const bf16x8 *)(ws + ks * 8192 + cb * 512);
            }
            if constexpr ((ABL & 16) == 0) {
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) af[b][rb] = *(const bf16x8 *)(as + aoff[rb] + (((2 * ks + h) ^ asw[rb]) << 4));
            }
        };
        if constexpr ((ABL & 24) != 0) {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) wa[b][cb] = bf16x8{1, 2, 3, 4, 5, 6, 7, 8};
                af[b][0] = af[b][1] = bf16x8{1, 2, 3, 4, 5, 6, 7, 8};
            }
        }
        fetch(0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int cur = ks & 1;
            if (ks < 3) fetch(ks + 1, cur ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr ((ABL & 1) == 0) {
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb)
                        yacc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[cur][cb], af[cur][rb], yacc[rb][cb], 0, 0, 0);
            } else {
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) asm volatile("" ::"v"(wa[cur][cb]), "v"(af[cur][0]), "v"(af[cur][1]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // W(t+1) and A(t+1) (and A(t+2)'s older pieces) have landed once only the four pieces of A(t+2)... see the header: vmcnt(4)
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        issue_w(t + 2, t % DQ_NW);
        issue_a(t + 3, t % DQ_NA);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    mfma_result_guard();
    // y = acc + bias, 64 columns at a time through this wave's staging rows (the rings are free now)
    uint16_t *stage = (uint16_t *)lsm + wave * (32 * SLD);
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const f32x16 &a = yacc[rb][2 * q + half];
                uint16_t *dst = stage + r * SLD + 32 * half;
                const int col0 = 128 * wc + 64 * q + 32 * half;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 bb = make_uint2(0u, 0u);
                    if (p.bias != nullptr) bb = *(const uint2 *)(p.bias + col0 + 8 * g + 4 * h);
                    *(uint2 *)(dst + 8 * g + 4 * h) = make_uint2(pack2(a[4 * g + 0] + bf_lo(bb.x), a[4 * g + 1] + bf_hi(bb.x)),
                                                                 pack2(a[4 * g + 2] + bf_lo(bb.y), a[4 * g + 3] + bf_hi(bb.y)));
                }
            }
            wave_lds_fence();
            flush64(stage, p.Y + 128 * wc + 64 * q, p.ldy, wg0 + 64 * wr + 32 * rb, p.M, lane);
            wave_lds_fence();
        }
}

// ----------------------------------------------------------------------------------------------- round 6: two resident workgroups per CU
// deep256r_kernel.  The ablations of the forms above say it plainly: a tile-synchronous workgroup that owns its CU runs its memory / LDS
// pipeline and its MFMAs one after the other (116 + 79 -> 163 us).  The rows kernels of csrc/vsde_linear.hip never had that problem --
// two independent four-wave workgroups per CU drift apart and cover each other's waits.  The same here: a workgroup is FOUR waves
// (2 x 2: a wave owns 64 rows x 128 columns, 2 x 4 accumulator blocks as in deep256q_kernel) = 128 rows, 80 KB of LDS:
//   activation tiles [128 rows][128 bytes] (K = 64), chunks XOR-swizzled as in deep256q_kernel: ring of TWO slots (32 KB);
//   weight HALF tiles (K = 32: two k-step images, 16 KB): ring of THREE slots (48 KB); one barrier per half tile (16 MFMAs per wave).
// Every request has two half tiles of flight.  Stream per wave (4 pieces each): behind the barrier that ends half tile j the group
// [W(j+3)] is issued, for odd j = 2 t + 1 also A(t+2) (tile t has just been left); the barrier that ends half tile j is preceded
// by vmcnt(4) for odd j (only W(j+2) may still fly) and vmcnt(8) for even j ([W(j+2) A(..)]): W(j+1) -- and for odd j the tile A((j+1)/2)
// that starts behind it -- have landed.  Each workgroup streams the whole weight: twice the L2 -> LDS weight traffic per CU, which is
// what the second workgroup costs.  Measured (profiles/r06_deep256_ablation.txt, section 6): 104 | 192 | 119 us (hipBLASLt 99 | 168 | 113) --
// the closest of the four forms, MFMAs alone 65 us -- but its memory pipeline alone (no MFMAs) takes 156 us: 1,604 workgroups x 720 KB of
// weight images + the activations = 1.7 GB through L2 -> LDS at ~11 TB/s, the same aggregate rate the one-workgroup form reaches with
// 1.2 GB.  The bound of every form is that LDS-DMA intake, not the schedule around it.
constexpr int DR_ASLOT = 128 * 128, DR_WSLOT = 2 * 32 * 256, DR_NA = 2, DR_NW = 3;
template <int ABL = 0>
__global__ void __launch_bounds__(256, 2) deep256r_kernel(DeepParams p) {
    extern __shared__ __attribute__((aligned(16))) char lsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    const int64_t wg0 = p.row_begin + (int64_t)blockIdx.x * 128;
    if (wg0 >= p.M) return;
    const int T = p.T, G = 2 * T;
    const int rot = p.rotate ? (int)((blockIdx.x * 5u) % (unsigned)T) : 0;
    auto tile_of = [&](int t) { return ((t < T ? t : T - 1) + rot) % T; };
    char *aring = lsm, *wring = lsm + DR_NA * DR_ASLOT;
    const char *asrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = 8 * (wave + 4 * i) + (lane >> 3);
        int64_t m = wg0 + q;
        m = m < p.M ? m : p.M - 1;
        asrc[i] = (const char *)p.X + m * p.ldx * 2 + ((((lane & 7) ^ ((q >> 1) & 7))) << 4);
    }
    auto issue_w = [&](int g, int slot) {   // half tile g = 2 t + u: bytes [u * 16 KB, ..) of tile t's image
        if constexpr ((ABL & 4) != 0) { if (g > 2) return; }
        const int gg = g < G ? g : G - 1;
        const int64_t off = (int64_t)tile_of(gg >> 1) * DEEP_TILE + (gg & 1) * DR_WSLOT;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = wave + 4 * i;
            __builtin_amdgcn_global_load_lds((const void *)((const char *)p.WI + off + piece * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void *)(wring + slot * DR_WSLOT + piece * 1024), 16, 0, 0);
        }
    };
    auto issue_a = [&](int t, int slot) {
        if constexpr ((ABL & 2) != 0) { if (t > 1) return; }
        const int tt = tile_of(t);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = wave + 4 * i;
            __builtin_amdgcn_global_load_lds((const void *)(asrc[i] + tt * 128),
                                             (__attribute__((address_space(3))) void *)(aring + slot * DR_ASLOT + piece * 1024), 16, 0, 0);
        }
    };
    f32x16 yacc[2][4];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int e = 0; e < 16; ++e) yacc[rb][cb][e] = 0.f;
    const int woff = h * 4096 + (128 * wc + r) * 16;
    int aoff[2], asw[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) { const int q = 64 * wr + 32 * rb + r; aoff[rb] = q * 128; asw[rb] = (q >> 1) & 7; }
    // prologue = the groups of the barriers "-3, -2, -1": [W(0) A(0)] [W(1)] [W(2) A(1)]; half tile 0 needs the first
    issue_w(0, 0); issue_a(0, 0);
    issue_w(1, 1);
    issue_w(2, 2); issue_a(1, 1);
    asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");
    auto half = [&](int g, auto odd_tag) {
        constexpr bool ODD = decltype(odd_tag)::value;
        const char *ws = wring + (g % DR_NW) * DR_WSLOT + woff;
        const char *as = aring + ((g >> 1) % DR_NA) * DR_ASLOT;
        bf16x8 wa[2][4], af[2][2];
        if constexpr ((ABL & 24) != 0) {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) wa[b][cb] = bf16x8{1, 2, 3, 4, 5, 6, 7, 8};
                af[b][0] = af[b][1] = bf16x8{1, 2, 3, 4, 5, 6, 7, 8};
            }
        }
        auto fetch = [&](int ksl, int b) {
            if constexpr ((ABL & 8) == 0) {
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) wa[b][cb] = *(const bf16x8 *)(ws + ksl * 8192 + cb * 512);
            }
            if constexpr ((ABL & 16) == 0) {
                const int ks = (ODD ? 2 : 0) + ksl;
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) af[b][rb] = *(const bf16x8 *)(as + aoff[rb] + (((2 * ks + h) ^ asw[rb]) << 4));
            }
        };
        fetch(0, 0);
        fetch(1, 1);
#pragma unroll
        for (int ksl = 0; ksl < 2; ++ksl) {
            __builtin_amdgcn_sched_barrier(0);
            if constexpr ((ABL & 1) == 0) {
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb)
                        yacc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[ksl][cb], af[ksl][rb], yacc[rb][cb], 0, 0, 0);
            } else {
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) asm volatile("" ::"v"(wa[ksl][cb]), "v"(af[ksl][0]), "v"(af[ksl][1]));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ODD) {
            asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            issue_w(g + 3, g % DR_NW);
            issue_a((g >> 1) + 2, ((g >> 1) + 2) % DR_NA);   // tile (g >> 1) has just been left: its slot takes tile + 2
        } else {
            asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            issue_w(g + 3, g % DR_NW);
        }
    };
    for (int t = 0; t < T; ++t) {
        half(2 * t, std::false_type{});
        half(2 * t + 1, std::true_type{});
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    mfma_result_guard();
    uint16_t *stage = (uint16_t *)lsm + wave * (32 * SLD);
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int half_ = 0; half_ < 2; ++half_) {
                const f32x16 &a = yacc[rb][2 * q + half_];
                uint16_t *dst = stage + r * SLD + 32 * half_;
                const int col0 = 128 * wc + 64 * q + 32 * half_;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 bb = make_uint2(0u, 0u);
                    if (p.bias != nullptr) bb = *(const uint2 *)(p.bias + col0 + 8 * g + 4 * h);
                    *(uint2 *)(dst + 8 * g + 4 * h) = make_uint2(pack2(a[4 * g + 0] + bf_lo(bb.x), a[4 * g + 1] + bf_hi(bb.x)),
                                                                 pack2(a[4 * g + 2] + bf_lo(bb.y), a[4 * g + 3] + bf_hi(bb.y)));
                }
            }
            wave_lds_fence();
            flush64(stage, p.Y + 128 * wc + 64 * q, p.ldy, wg0 + 64 * wr + 32 * rb, p.M, lane);
            wave_lds_fence();
        }
}
}  // namespace mlp
}  // namespace vsde

#endif  // VSDE_ABLATIONS (deep reduction kernels)

// y [M][256] = x [M][K] W^T (+ bias): w_img = W as K / 16 k-step images [2][256][8] bf16 (W[n][16 t + 8 h + 0..7]: the layout of w2_img),
// K % 64 == 0, K >= 256.  Replaces the library GEMMs of primitives/mlp.py:54 (forward) and of the input gradients of mlp.py:50 /
// attn.py:80-82 at the encoder's width 256.
extern "C" int vsde_linear_deep256_bf16(const void *x, int64_t ldx, const void *w_img, const void *bias, void *y, int64_t ldy, int64_t M, int K,
                                        void *stream) {
    VSDE_CHECK_ARG(x && w_img && y && M > 0, VSDE_E_BADARG, "bad linear_deep256 arguments");
    VSDE_CHECK_ARG(K >= 256 && K % 64 == 0, VSDE_E_BADARG, "linear_deep256: the reduction length must be a multiple of 64, at least 256 (got %d)", K);
    VSDE_CHECK_ARG(ldx >= K && ldx % 8 == 0 && ldy >= 256 && ldy % 8 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 &&
                   ((uintptr_t)w_img % 16) == 0 && (!bias || ((uintptr_t)bias % 8) == 0), VSDE_E_BADARG,
                   "linear_deep256 operands must be 16-byte aligned with row pitches that are multiples of 8 elements");
#ifndef VSDE_ABLATIONS
    vsde::set_error("vsde_linear_deep256_bf16: the own deep-reduction GEMM is only built into the tools' library (python -m viforsdes_amd.build "
                    "--ablations, VSDE_HIP_LIB): it is 10-15 %% slower than hipBLASLt at the shapes it was written for");
    return VSDE_E_BADARG;
#else
    mlp::DeepParams p = {};
    p.X = (const uint16_t *)x; p.ldx = ldx; p.WI = (const uint16_t *)w_img; p.bias = (const uint16_t *)bias; p.Y = (uint16_t *)y; p.ldy = ldy;
    p.M = M; p.T = K / 64;
    { static int rot = -1; if (rot < 0) rot = (int)vsde_knob("VSDE_MLP_ROTATE", 1); p.rotate = rot; }
    // VSDE_DEEP256_Q=0: the bulk on deep256p_kernel<8> (one-dimensional wave tiling); VSDE_DEEP256_TAIL=0: one launch of 256-row
    // workgroups also when the last round is mostly empty; VSDE_DEEP256_ABL: timing-only ablations (bits in the kernels' headers)
    static int qk = -1, tail = -1, abl = -1, rk = -1;
    if (rk < 0) rk = (int)vsde_knob("VSDE_DEEP256_R", 1);   // 1: the bulk on deep256r_kernel (two four-wave workgroups per CU)
    if (qk < 0) qk = (int)vsde_knob("VSDE_DEEP256_Q", 1);
    if (tail < 0) tail = (int)vsde_knob("VSDE_DEEP256_TAIL", 1);
    if (abl < 0) abl = ablation_env("VSDE_DEEP256_ABL");
    const size_t ring = (size_t)mlp::DEEP_NSLOT * mlp::DEEP_TILE;
    const int64_t stripes = (M + 255) / 256;
    static int cus = 0;
    if (!cus) { int dev = 0; hipDeviceProp_t pr; cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; }
    const int64_t rem = stripes % cus;
    // the last round of 256-row workgroups fills at most half of the CUs: its rows run as 64-row workgroups instead (four times as many,
    // a quarter of the work each)
    const int64_t bulk = (tail && stripes > cus && rem > 0 && 2 * rem <= cus) ? stripes - rem : stripes;
    const size_t lds8 = ring + 8 * 4096, lds2 = ring + 2 * 4096;
    const size_t ldsq = (size_t)mlp::DQ_NA * mlp::DQ_ASLOT + (size_t)mlp::DQ_NW * mlp::DEEP_TILE;
    p.row_begin = 0;
    mlp::DeepParams pb = p;
    if (bulk < stripes) pb.M = bulk * 256;
    hipStream_t st = (hipStream_t)stream;
#define VSDE_DEEP_P(A) case A: VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)mlp::deep256p_kernel<8, A>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8)); \
        hipLaunchKernelGGL((mlp::deep256p_kernel<8, A>), dim3((unsigned)bulk), dim3(512), lds8, st, pb); break;
#define VSDE_DEEP_Q(A) case A: VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)mlp::deep256q_kernel<A>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsq)); \
        hipLaunchKernelGGL((mlp::deep256q_kernel<A>), dim3((unsigned)bulk), dim3(512), ldsq, st, pb); break;
    if (rk) {
        // 128-row workgroups, two per CU: the rows of the last, at most half filled round of (2 x CUs) workgroups go to the tail launch
        const int64_t s128 = (M + 127) / 128, slots = 2 * (int64_t)cus, rem128 = s128 % slots;
        // (VSDE_DEEP256_TAIL=2 only: with two workgroups per CU the rounds overlap and the second launch costs more than the ragged
        //  round -- 110 | 196 | 127 us with it, 104 | 192 | 119 without)
        const int64_t bulk128 = (tail == 2 && s128 > slots && rem128 > 0 && 2 * rem128 <= slots) ? s128 - rem128 : s128;
        const size_t ldsr = (size_t)mlp::DR_NA * mlp::DR_ASLOT + (size_t)mlp::DR_NW * mlp::DR_WSLOT;
        mlp::DeepParams pr = p;
        if (bulk128 < s128) pr.M = bulk128 * 128;
#define VSDE_DEEP_R(A) case A: VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)mlp::deep256r_kernel<A>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsr)); \
        hipLaunchKernelGGL((mlp::deep256r_kernel<A>), dim3((unsigned)bulk128), dim3(256), ldsr, st, pr); break;
        switch (abl) { VSDE_DEEP_R(1) VSDE_DEEP_R(6) VSDE_DEEP_R(24) VSDE_DEEP_R(30) VSDE_DEEP_R(31) default: VSDE_DEEP_R(0) }
#undef VSDE_DEEP_R
        if (bulk128 < s128) {
            VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)mlp::deep256p_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
            p.row_begin = bulk128 * 128;
            hipLaunchKernelGGL((mlp::deep256p_kernel<2>), dim3((unsigned)((M - p.row_begin + 63) / 64)), dim3(128), lds2, st, p);
        }
        VSDE_CHECK_HIP(hipGetLastError());
        return 0;
    }
    if (qk) {
        switch (abl) { VSDE_DEEP_Q(1) VSDE_DEEP_Q(6) VSDE_DEEP_Q(24) VSDE_DEEP_Q(30) VSDE_DEEP_Q(31) default: VSDE_DEEP_Q(0) }
    } else {
        switch (abl) { VSDE_DEEP_P(1) VSDE_DEEP_P(2) VSDE_DEEP_P(4) VSDE_DEEP_P(8) VSDE_DEEP_P(16) VSDE_DEEP_P(6) VSDE_DEEP_P(24) VSDE_DEEP_P(30) VSDE_DEEP_P(31)
                       default: VSDE_DEEP_P(0) }
    }
#undef VSDE_DEEP_P
#undef VSDE_DEEP_Q
    if (bulk < stripes) {
        VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)mlp::deep256p_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
        p.row_begin = bulk * 256;
        hipLaunchKernelGGL((mlp::deep256p_kernel<2>), dim3((unsigned)((M - p.row_begin + 63) / 64)), dim3(128), lds2, st, p);
    }
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
#endif
}

// Sizes (bytes) of the three weight images for width C per tile of 16 hidden units: what primitives/fused.py allocates
extern "C" int vsde_mlp_image_bytes(int C, int64_t *w1_tile, int64_t *w2_tile, int64_t *b1_tile) {
    VSDE_CHECK_ARG(C == 128 || C == 256, VSDE_E_BADARG, "fused SwiGLU MLP: width %d not built (128, 256)", C);
    if (C == 128) { *w1_tile = mlp::Geo<128>::W1_BYTES; *w2_tile = mlp::Geo<128>::W2_BYTES; *b1_tile = mlp::Geo<128>::B1_BYTES; }
    else { *w1_tile = mlp::Geo<256>::W1_BYTES; *w2_tile = mlp::Geo<256>::W2_BYTES; *b1_tile = mlp::Geo<256>::B1_BYTES; }
    return 0;
}

static void mlp_env(mlp::FwdParams &p) {
    static int anti = -1, rot = -1;   // VSDE_MLP_ANTIPHASE (see FwdParams), VSDE_MLP_ROTATE=0: every workgroup walks the tiles in the same order
    if (anti < 0) anti = (int)vsde_knob("VSDE_MLP_ANTIPHASE", 2);
    if (rot < 0) rot = (int)vsde_knob("VSDE_MLP_ROTATE", 1);
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

extern "C" int vsde_mlp_attn_block_fwd_bf16(const void *x, const void *attn, const void *glog, int64_t ldg, const void *wo_img, const void *bo,
                                            const void *ga, const void *sc, const void *sh, const void *gm, const void *sn, const void *hs,
                                            int64_t mp, int tokens, double eps, double eps_next, const void *w1_img, const void *w2_img,
                                            const float *b1_img, const void *b2, void *tok, void *hnext, int64_t M, int C, int H, void *stream) {
    VSDE_CHECK_ARG(x && attn && glog && wo_img && ga && sc && sh && gm && w1_img && w2_img && b1_img && tok && M > 0 && tokens > 0, VSDE_E_BADARG,
                   "bad mlp_attn_block_fwd arguments");
    VSDE_CHECK_ARG((C == 128 || C == 256) && H >= 64 && H % 64 == 0, VSDE_E_BADARG,
                   "fused SwiGLU MLP is built for widths 128 / 256 and a hidden size that is a multiple of 64 (got %d, %d)", C, H);
    VSDE_CHECK_ARG((!sn) == (!hs) && (!sn) == (!hnext), VSDE_E_BADARG, "next-norm scale, shift and output go together");
    const void *ptrs[] = {x, attn, glog, wo_img, bo, ga, sc, sh, gm, sn, hs, w1_img, w2_img, b1_img, b2, tok, hnext};
    for (const void *q : ptrs) VSDE_CHECK_ARG(((uintptr_t)q % 16) == 0, VSDE_E_BADARG, "mlp_attn_block_fwd operands must be 16-byte aligned");
    VSDE_CHECK_ARG(mp >= C && mp % 8 == 0 && ldg >= 64 && ldg % 8 == 0, VSDE_E_BADARG, "bad modulation / gate row pitch");
    VSDE_CHECK_ARG(tok != x, VSDE_E_BADARG, "mlp_attn_block_fwd parks x1 in tok: it must not alias x");
    VSDE_CHECK_ARG(tokens >= 86, VSDE_E_BADARG, "mlp_attn_block_fwd keeps the modulation vectors of %d batch rows per 256-row stripe: sequences of >= 86 tokens", mlp::MODB);
    mlp::FwdParams p = {};
    p.R0 = (const uint16_t *)x; p.OA = (const uint16_t *)attn; p.GL = (const uint16_t *)glog; p.ldg = ldg; p.WOI = (const uint16_t *)wo_img;
    p.BO = (const uint16_t *)bo; p.GA = (const uint16_t *)ga; p.SC = (const uint16_t *)sc; p.SH = (const uint16_t *)sh;
    p.GM = (const uint16_t *)gm; p.SN = (const uint16_t *)sn; p.HS = (const uint16_t *)hs; p.TOK = (uint16_t *)tok; p.HOUT = (uint16_t *)hnext;
    p.mp = mp; p.tokens = tokens; p.eps = (float)eps; p.eps_next = (float)eps_next;
    p.W1I = (const uint16_t *)w1_img; p.W2I = (const uint16_t *)w2_img; p.B1I = b1_img; p.b2 = (const uint16_t *)b2; p.M = M; p.T = H / 16;
    mlp_env(p);
    return C == 256 ? mlp::launch_fwd<256, 0, 0, 2>(p, (hipStream_t)stream) : mlp::launch_fwd<128, 0, 0, 2>(p, (hipStream_t)stream);
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
#ifdef VSDE_ABLATIONS
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
#else
    (void)dbg;
#endif
    if (C == 256) return s_out ? mlp::launch_fwd<256, 1>(p, st) : mlp::launch_fwd<256, 0>(p, st);
    return s_out ? mlp::launch_fwd<128, 1>(p, st) : mlp::launch_fwd<128, 0>(p, st);
}

// bytes of the backward's weight image per pair tile (32 hidden units): [W2T | W1], see mlp_bwd_kernel
extern "C" int64_t vsde_mlp_bwd_image_bytes(int C) { return C == 128 ? mlp::bwd::Geo<128>::BUF : (C == 256 ? mlp::bwd::Geo<256>::BUF : 0); }

// du [M][2 H] = swiglu'(u) * (dy W_out) and dx [M][C] = du W_in in one pass (u, du: 16-row interleaved layout, 64 columns per 32 units)
extern "C" int vsde_mlp_bwd_bf16(const void *dy, int64_t lddy, const void *u, int64_t ldu, const void *img, void *du, int64_t lddu, void *dx,
                                 int64_t lddx, int64_t M, int C, int H, void *stream) {
    VSDE_CHECK_ARG(dy && u && img && du && dx && M > 0, VSDE_E_BADARG, "bad mlp_bwd arguments");
    VSDE_CHECK_ARG((C == 128 || C == 256) && H >= 64 && H % 64 == 0, VSDE_E_BADARG,
                   "fused SwiGLU MLP is built for widths 128 / 256 and a hidden size that is a multiple of 64 (got %d, %d)", C, H);
    VSDE_CHECK_ARG(lddy >= C && lddx >= C && ldu >= 2 * H && lddu >= 2 * H && lddy % 8 == 0 && lddx % 8 == 0 && ldu % 8 == 0 && lddu % 8 == 0 &&
                   ((uintptr_t)dy % 16) == 0 && ((uintptr_t)u % 16) == 0 && ((uintptr_t)img % 16) == 0 && ((uintptr_t)du % 16) == 0 &&
                   ((uintptr_t)dx % 16) == 0, VSDE_E_BADARG, "mlp_bwd operands must be 16-byte aligned with row pitches that are multiples of 8 elements");
#ifndef VSDE_ABLATIONS
    vsde::set_error("vsde_mlp_bwd_bf16: the fused SwiGLU backward is only built into the tools' library (python -m viforsdes_amd.build --ablations, "
                    "VSDE_HIP_LIB): it is slower than the two launches the training step uses");
    return VSDE_E_BADARG;
#else
    mlp::BwdParams p = {};
    p.DY = (const uint16_t *)dy; p.lddy = lddy; p.U = (const uint16_t *)u; p.ldu = ldu; p.IMG = (const char *)img;
    p.DU = (uint16_t *)du; p.lddu = lddu; p.DX = (uint16_t *)dx; p.lddx = lddx; p.M = M; p.TP = H / 32;
    p.trace = g_mlp_trace;
    static int var = -1;   // VSDE_MLP_BWD_VAR: bit 0 = tile staging right after the barrier, bit 1 = grouped fragment prefetch in the dx product (A/B runs)
    if (var < 0) var = (int)vsde_knob("VSDE_MLP_BWD_VAR", 3);
    if (C == 256 && g_mlp_trace != nullptr) return mlp::launch_bwd<256, 3, true>(p, (hipStream_t)stream);   // tools/mlp_bwd_trace.py
    if (C == 256) {
        switch (var & 3) {
            case 0: return mlp::launch_bwd<256, 0>(p, (hipStream_t)stream);
            case 1: return mlp::launch_bwd<256, 1>(p, (hipStream_t)stream);
            case 2: return mlp::launch_bwd<256, 2>(p, (hipStream_t)stream);
            default: return mlp::launch_bwd<256, 3>(p, (hipStream_t)stream);
        }
    }
    return mlp::launch_bwd<128>(p, (hipStream_t)stream);
#endif
}
