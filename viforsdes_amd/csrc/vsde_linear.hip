// The encoder's dense contractions on bf16 MFMA (gfx950): y = x W^T + b for the SiT blocks' Linears
// (reference: primitives/attn.py:46-47,54,104-113 qkv / gate / out projections, primitives/mlp.py:41-54 SwiGLU pair,
// primitives/sit.py:162-186 output projection; their input gradients in the backward), with the SwiGLU activation and its
// derivative fused into the GEMM epilogues.
//
// Shape class: M = batch x tokens ~ 2e5 rows, K and N in {128 .. 1536}: every weight is small enough to live in L2, so the
// activation matrix is what streams.  One workgroup = 8 wavefronts x 32 rows = a 256-row stripe; a wave keeps ITS rows'
// operand / accumulators in registers and the weight streams through LDS in 32 KB tiles (double-buffered, one barrier per
// tile, 32 v_mfma_f32_32x32x16_bf16 per wave and tile):
//   * rows kernel ("A-stationary", K in {128, 256}): the wave's 32 x K slice of x sits in VGPRs for the whole stripe,
//     the loop runs over 64-column tiles of W ([64][K]); x is read from HBM exactly once, y written once.
//   * cols kernel ("C-stationary", 256- or 128-column output tiles, any K % 64 == 0): the waves' accumulators stay in VGPRs,
//     the loop runs over 64-deep K chunks of W ([NT][64]) and of x (fragment loads straight to registers).
// Products are computed swapped (D = W_tile . x_tile^T): the lane that owns activation row r keeps it through the whole
// stripe, and each accumulator register quad is 4 consecutive output columns of that row -- bias, SwiGLU and its derivative
// are then lane-local register math, and the tile leaves through a per-wave LDS staging buffer as full 128-byte row segments.
// Epilogues:
//   EPI_PLAIN        y = acc + bias                                                       (bf16)
//   EPI_SWIGLU       u = acc + bias (kept for the backward), s = silu(a) * b   with [a | b] the two 16-column halves of a
//                    32-column tile: the packed weight interleaves the SwiGLU halves in blocks of 16 rows (primitives/fused.py)
//   EPI_SWIGLU_BWD   acc = ds (gradient of s); reads u, writes du = (da | db) in the same interleaved layout
//   EPI_QKNORM       no-grad attention projection [q | k | v | gate]: a pair of tiles is one 64-wide head; q / k heads are
//                    RMS-normalised and rotated (RoPE) in the wave's staging buffer, v is mixed with the residual values, and every
//                    head leaves straight in the attention kernels' layout -- the [M, 3C + d] intermediate and the separate
//                    qk_norm_rope pass of the training path do not exist here
//   EPI_GATE_BWD     input gradient of the attention output projection with the backward of the sigmoid output gate in its
//                    epilogue (training step): a tile pair is one head's 64 columns of d(merged); with og = o s (the merged rows) and
//                    the gate factors s it leaves as dO = d s, D = <d, og> per (token, head), and -- after the wave's last head --
//                    the gate logits' gradient (1 - s) sum_h d og.  d(merged) itself is never written
#include <stdlib.h>

#include "vsde_common.h"

namespace vsde {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // a 16-byte register quad (native vector: stays in VGPRs)

constexpr int EPI_PLAIN = 0, EPI_SWIGLU = 1, EPI_SWIGLU_BWD = 2, EPI_QKNORM = 3, EPI_GATE_BWD = 4;
constexpr int R2_THREADS = 256, R2_SLD = 88;   // rows kernel: 4 waves, staging rows of 64 + 16 (+ 8 pad) elements

struct LinParams {
    const uint16_t *A; int64_t lda;    // activations [M][lda] bf16 (row pitch in elements)
    const uint16_t *W;                 // weight [N][K] bf16, contiguous
    const uint16_t *bias;              // [N] bf16 or nullptr
    uint16_t *C; int64_t ldc;          // output [M][ldc]; EPI_SWIGLU: u [M][N] (may be nullptr); EPI_SWIGLU_BWD: du [M][2N]
    uint16_t *S; int64_t lds_;         // EPI_SWIGLU: s [M][N/2]
    const uint16_t *U; int64_t ldu;    // EPI_SWIGLU_BWD: saved u [M][2N]
    int64_t M; int N, K;
    int chunks;                        // rows kernel: column chunks per row stripe
    int tail_local, tail_chunks;       // rows kernel: workgroups with `local` >= tail_local own the LAST stripes, in tail_chunks column chunks
    // EPI_QKNORM: N = 3 heads*64 + gate columns; a tile pair = one 64-wide head of q / k / v (or the gate block)
    uint16_t *Qo, *Ko, *Vo, *Go; int64_t ldg;   // q, k, v [M][heads*64] token-major; gate logits [M][ldg]
    const float *cosT, *sinT, *wq, *wk, *lam;   // rotary tables [tokens][32], RMS weights [64], value-mix weight [1]
    const uint16_t *V0;                         // residual values [M][heads*64] or nullptr
    int heads, tokens; float eps;
    int gate_sigmoid;                           // training: the gate block leaves as rnd(sigmoid(logit)) (what the attention store multiplies by)
    // EPI_GATE_BWD: A = gradient of the projection output, W = its weight transposed, N = heads * 64 = width of the merged rows
    const uint16_t *Og; const uint16_t *Sg; int64_t ldsg;   // merged gated rows [M][N], gate factors s [M][ldsg]
    uint16_t *Dgate; int64_t lddg;                          // gradient of the gate logits [M][lddg] (64 columns)
    float *Delta;                                           // [B][heads][tokens]: <dO, O> of the attention backward
    float *Rinv;                                // training: inverse RMS of every q / k head row [M][2 heads], or nullptr
    uint16_t *Vdiff;                            // training: v_raw - v0 [M][heads*64] (what the value-mix weight's gradient needs), or nullptr
    // gated A operand (no-grad out projection): row m of A is multiplied by sigmoid(Gate[m][k % 64]) on its way into the registers
    const uint16_t *Gate; int64_t ldgate;
    int dbg;                           // ablation (VSDE_LIN_DEBUG): 1 = skip the output stores
    int xstage;                        // rows kernel: 1 = activations loaded as full row segments and redistributed through LDS
};

typedef __bf16 hwbf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {   // v_cvt_pk_bf16_f32: round to nearest even, two at a time
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hwbf16x2));
}
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ float rbf(float x) { return (float)(__bf16)x; }   // value of x after rounding to bf16
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also carries a release fence for GLOBAL memory, i.e. an
// s_waitcnt vmcnt(0): in these loops that would drain the tile's output stores (HBM write latency) and the operand loads
// issued for the next iterations at every tile.  LDS operations are tracked by lgkmcnt alone.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float sigm_f(float x) { return fast_rcp(1.0f + __expf(-x)); }

// R rows x 64 columns of bf16 out of a wave's staging buffer (row stride SLD elements) as full 128-byte row segments
template <int SLD, int R>
__device__ __forceinline__ void flush_rows64(const uint16_t *stage, uint16_t *dst, int64_t ld, int64_t row0, int64_t M, int lane) {
#pragma unroll
    for (int i = 0; i < R / 8; ++i) {
        const int row = (lane >> 3) + 8 * i, c = lane & 7;
        const u32x4 v = *(const u32x4 *)(stage + row * SLD + c * 8);
        // streaming output, never re-read by this kernel: a non-temporal store keeps it from evicting the weight tiles (and the
        // activation rows still to be read) out of L2
        if (row0 + row < M) __builtin_nontemporal_store(v, (u32x4 *)(dst + (row0 + row) * ld + c * 8));
    }
}
// R rows x 32 columns (64-byte row segments: 4 lanes per row)
template <int SLD, int R>
__device__ __forceinline__ void flush_rows32(const uint16_t *stage, uint16_t *dst, int64_t ld, int64_t row0, int64_t M, int lane) {
#pragma unroll
    for (int i = 0; i < R / 16; ++i) {
        const int row = (lane >> 2) + 16 * i, c = lane & 3;
        const u32x4 v = *(const u32x4 *)(stage + row * SLD + c * 8);
        if (row0 + row < M) __builtin_nontemporal_store(v, (u32x4 *)(dst + (row0 + row) * ld + c * 8));
    }
}

// acc (one 32x32 MFMA block: this lane holds row r, columns 8 g + 4 h + i) + bias -> bf16 into the staging row
// (all four bias reads first: bias row and staging row are both LDS, the compiler must assume they alias and would otherwise wait for
//  every read behind the previous store -- four LDS round trips per block instead of one)
__device__ __forceinline__ void stage_block(const f32x16 &acc, const uint16_t *bias32, uint16_t *dst_row, int h) {
    uint2 bb[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bb[g] = *(const uint2 *)(bias32 + 8 * g + 4 * h);
#pragma unroll
    for (int g = 0; g < 4; ++g)
        *(uint2 *)(dst_row + 8 * g + 4 * h) = make_uint2(pack_bf16x2(acc[4 * g + 0] + bf_lo(bb[g].x), acc[4 * g + 1] + bf_hi(bb[g].x)),
                                                        pack_bf16x2(acc[4 * g + 2] + bf_lo(bb[g].y), acc[4 * g + 3] + bf_hi(bb[g].y)));
}

// weight tile [ROWS][KW] (row pitch ldw in global memory) <-> registers <-> LDS rows of LDB elements, THREADS threads
template <int NLD, int KW, int THREADS>
__device__ __forceinline__ void wtile_load(u32x4 (&breg)[NLD], const uint16_t *W, int64_t ldw, int tid) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = tid + THREADS * i, row = idx / (KW / 8), c = idx % (KW / 8);
        breg[i] = *(const u32x4 *)(W + (int64_t)row * ldw + c * 8);
    }
}
template <int NLD, int KW, int LDB, int THREADS>
__device__ __forceinline__ void wtile_store(const u32x4 (&breg)[NLD], uint16_t *Bs, int tid) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = tid + THREADS * i, row = idx / (KW / 8), c = idx % (KW / 8);
        *(u32x4 *)(Bs + row * LDB + c * 8) = breg[i];
    }
}

// Epilogue of one 32-column tile (acc[rb]: the wave's two 32-row blocks) of the rows kernel.  PAR = parity of the tile: tiles
// leave in pairs (2q, 2q + 1) = 64 output columns.  bias32: LDS row with the tile's 32 bias values; n0: the tile's first column.
//   EPI_SWIGLU      the packed weight interleaves the SwiGLU halves in blocks of 16 rows, so a 32-column tile is [a_16 | b_16]:
//                   this lane's register quads g = 0, 1 hold a_j and g = 2, 3 the matching b_j; a pair of tiles gives 32 columns of s.
//   EPI_SWIGLU_BWD  acc = ds for 32 columns j; u / du tile: 64 interleaved columns [a_16 | b_16 | a_16 | b_16]; no pairing.
// packed s = silu(a) * b of one tile for row block rb: quads g = 0, 1 of acc are a_j, g = 2, 3 the matching b_j
__device__ __forceinline__ void swiglu_quads(const f32x16 &acc, const uint16_t *bias32, int h, uint32_t (&out)[4]) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const uint2 ba = *(const uint2 *)(bias32 + 8 * g + 4 * h), bb = *(const uint2 *)(bias32 + 16 + 8 * g + 4 * h);
        const float av[4] = {bf_lo(ba.x), bf_hi(ba.x), bf_lo(ba.y), bf_hi(ba.y)};
        const float bv[4] = {bf_lo(bb.x), bf_hi(bb.x), bf_lo(bb.y), bf_hi(bb.y)};
        float sv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {   // from the bf16-rounded u, as the unfused chain (mlp.py:21-24 under autocast)
            const float a = rbf(acc[4 * g + i] + av[i]), b = rbf(acc[4 * (g + 2) + i] + bv[i]);
            sv[i] = rbf(a * sigm_f(a)) * b;
        }
        out[2 * g] = pack_bf16x2(sv[0], sv[1]); out[2 * g + 1] = pack_bf16x2(sv[2], sv[3]);
    }
}

// EPI_SWIGLU: swiglu_quads and stage_block of the same tile in one pass.  The unfused chain computes s from the bf16 u, so the
// packed words that go to the staging row ARE the rounded a / b: unpacking them replaces a second conversion per value (the
// epilogue is the longer phase of this kernel, profiles/r04_pmc_linear_swiglu.txt: 4.5 -> 2 v_cvt_pk_bf16_f32 and 4 -> 3 adds per pair).
__device__ __forceinline__ void swiglu_stage(const f32x16 &acc, const uint16_t *bias32, uint16_t *dst_row, int h, uint32_t (&out)[4]) {
    uint32_t uw[4][2];
    uint2 bbv[4];   // all four bias reads in front of the first staging store (see stage_block)
#pragma unroll
    for (int g = 0; g < 4; ++g) bbv[g] = *(const uint2 *)(bias32 + 8 * g + 4 * h);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const uint2 bb = bbv[g];
        uw[g][0] = pack_bf16x2(acc[4 * g + 0] + bf_lo(bb.x), acc[4 * g + 1] + bf_hi(bb.x));
        uw[g][1] = pack_bf16x2(acc[4 * g + 2] + bf_lo(bb.y), acc[4 * g + 3] + bf_hi(bb.y));
        *(uint2 *)(dst_row + 8 * g + 4 * h) = make_uint2(uw[g][0], uw[g][1]);
    }
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            const float a0 = bf_lo(uw[g][w]), a1 = bf_hi(uw[g][w]), b0 = bf_lo(uw[g + 2][w]), b1 = bf_hi(uw[g + 2][w]);
            const uint32_t t = pack_bf16x2(a0 * sigm_f(a0), a1 * sigm_f(a1));
            out[2 * g + w] = pack_bf16x2(bf_lo(t) * b0, bf_hi(t) * b1);
        }
}

// EPI_QKNORM on the staged head (R = 32 rows x 64 columns of bf16 = the projection output after bias and rounding, exactly what
// the unfused chain hands to qk_norm_rope): lane (r, h) owns the rotary pairs i in [16 h, 16 h + 16) of row r, i.e. columns i
// and i + 32.  kind 0 / 1: x -> rnd(x rms w) rotated by the row's (cos, sin); kind 2 with residual values: lam v + (1 - lam) v0.
__device__ __forceinline__ void qknorm_head(const LinParams &p, uint16_t *stage, int kind, int hh, int64_t row0, int lane,
                                            const float (&cs)[16], const float (&sn)[16], const u32x4 (&v0r)[4], const float *wlds) {
    const int r = lane & 31, h = lane >> 5;
    uint16_t *px = stage + r * R2_SLD + 16 * h;
    const u32x4 l0 = *(const u32x4 *)px, l1 = *(const u32x4 *)(px + 8), h0 = *(const u32x4 *)(px + 32), h1 = *(const u32x4 *)(px + 40);
    const uint32_t lw[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w}, hw[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
    float xl[16], xh[16];
#pragma unroll
    for (int e = 0; e < 8; ++e) { xl[2 * e] = bf_lo(lw[e]); xl[2 * e + 1] = bf_hi(lw[e]); xh[2 * e] = bf_lo(hw[e]); xh[2 * e + 1] = bf_hi(hw[e]); }
    uint32_t ol[8], oh[8];
    if (kind <= 1) {
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) ss = fmaf(xl[e], xl[e], fmaf(xh[e], xh[e], ss));
        ss = sum_xor32(ss);
        const float rq = rsqrtf(ss * (1.0f / 64.0f) + p.eps);
        if (p.Rinv != nullptr && h == 0 && row0 + r < p.M) p.Rinv[(row0 + r) * (2 * p.heads) + kind * p.heads + hh] = rq;
        const float *w = wlds + 64 * kind;   // [wq | wk] staged in LDS by the kernel prologue
        float o0[16], o1[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float a0 = rbf(xl[e] * rq * w[16 * h + e]), a1 = rbf(xh[e] * rq * w[32 + 16 * h + e]);
            o0[e] = a0 * cs[e] - a1 * sn[e]; o1[e] = a0 * sn[e] + a1 * cs[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { ol[e] = pack_bf16x2(o0[2 * e], o0[2 * e + 1]); oh[e] = pack_bf16x2(o1[2 * e], o1[2 * e + 1]); }
    } else {
        const float l = p.lam[0];
        const uint32_t vw[16] = {v0r[0].x, v0r[0].y, v0r[0].z, v0r[0].w, v0r[1].x, v0r[1].y, v0r[1].z, v0r[1].w,
                                 v0r[2].x, v0r[2].y, v0r[2].z, v0r[2].w, v0r[3].x, v0r[3].y, v0r[3].z, v0r[3].w};
        if (p.Vdiff != nullptr) {   // wave-uniform: v_raw - v0 leaves through the staging rows first, then the mixed values below
            uint32_t dl[8], dh[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                dl[e] = pack_bf16x2(xl[2 * e] - bf_lo(vw[e]), xl[2 * e + 1] - bf_hi(vw[e]));
                dh[e] = pack_bf16x2(xh[2 * e] - bf_lo(vw[8 + e]), xh[2 * e + 1] - bf_hi(vw[8 + e]));
            }
            *(u32x4 *)px = (u32x4){dl[0], dl[1], dl[2], dl[3]}; *(u32x4 *)(px + 8) = (u32x4){dl[4], dl[5], dl[6], dl[7]};
            *(u32x4 *)(px + 32) = (u32x4){dh[0], dh[1], dh[2], dh[3]}; *(u32x4 *)(px + 40) = (u32x4){dh[4], dh[5], dh[6], dh[7]};
            wave_lds_fence();
            flush_rows64<R2_SLD, 32>(stage, p.Vdiff + hh * 64, (int64_t)p.heads * 64, row0, p.M, lane);
            wave_lds_fence();
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            ol[e] = pack_bf16x2(l * xl[2 * e] + (1.0f - l) * bf_lo(vw[e]), l * xl[2 * e + 1] + (1.0f - l) * bf_hi(vw[e]));
            oh[e] = pack_bf16x2(l * xh[2 * e] + (1.0f - l) * bf_lo(vw[8 + e]), l * xh[2 * e + 1] + (1.0f - l) * bf_hi(vw[8 + e]));
        }
    }
    *(u32x4 *)px = (u32x4){ol[0], ol[1], ol[2], ol[3]}; *(u32x4 *)(px + 8) = (u32x4){ol[4], ol[5], ol[6], ol[7]};
    *(u32x4 *)(px + 32) = (u32x4){oh[0], oh[1], oh[2], oh[3]}; *(u32x4 *)(px + 40) = (u32x4){oh[4], oh[5], oh[6], oh[7]};
}

template <int EPI, int PAR, int RB>
__device__ __forceinline__ void rows_epilogue(const LinParams &p, f32x16 (&acc)[RB], const uint16_t *bias32, uint16_t *stage,
                                              const u32x4 (&ureg)[4 * RB], int64_t row0, int n0, int lane,
                                              float (&cs)[16], float (&sn)[16], const float *wlds, bool last_tile) {
    const int r = lane & 31, h = lane >> 5;
    constexpr int SLD = R2_SLD, R = 32 * RB;
    if constexpr (EPI == EPI_PLAIN || EPI == EPI_SWIGLU || EPI == EPI_QKNORM || EPI == EPI_GATE_BWD) {
        uint32_t sq[RB][4];   // EPI_SWIGLU: this tile's s (second tile of a pair: kept until the pair's u has left the staging rows)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            if constexpr (EPI == EPI_SWIGLU) {
                swiglu_stage(acc[rb], bias32, stage + (rb * 32 + r) * SLD + PAR * 32, h, sq[rb]);
                if constexpr (PAR == 0) {   // the first tile's s waits in columns 64..79 of the staging row
#pragma unroll
                    for (int g = 0; g < 2; ++g) *(uint2 *)(stage + (rb * 32 + r) * SLD + 64 + 8 * g + 4 * h) = make_uint2(sq[rb][2 * g], sq[rb][2 * g + 1]);
                }
            } else {
                stage_block(acc[rb], bias32, stage + (rb * 32 + r) * SLD + PAR * 32, h);
            }
        }
        if constexpr (EPI == EPI_GATE_BWD && PAR == 1) {
            // the staged pair = d(merged) of head hh for the wave's 32 rows (bf16, as the unfused chain stores it); lane (r, h) owns
            // columns 16 h .. 16 h + 15 and 32 + 16 h .. of row r.  cs / sn (unused rotary registers of this epilogue) carry this
            // lane's 32 partial sums of d og over the heads visited so far.
            const int hh = (n0 - 32) >> 6;
            const int64_t m = row0 + r < p.M ? row0 + r : p.M - 1;
            // the gate factors of the row: 4 x 16 bytes in the lane's own column set (reloaded per head: L1 hits)
            const uint16_t *sp = p.Sg + m * p.ldsg + 16 * h;
            u32x4 sv[4];
#pragma unroll
            for (int part = 0; part < 4; ++part) sv[part] = *(const u32x4 *)(sp + (part >> 1) * 32 + (part & 1) * 8);
            wave_lds_fence();
            uint16_t *px = stage + r * SLD + 16 * h;
            u32x4 dv[4], ogv[4];
#pragma unroll
            for (int part = 0; part < 4; ++part) dv[part] = *(const u32x4 *)(px + (part >> 1) * 32 + (part & 1) * 8);
            wave_lds_fence();
            // ureg: the head's 32 x 64 slice of og, requested when the pair began as full 128-byte rows (8 rows per instruction; in
            // the lane's own column set every load would touch 32 rows for 16 bytes each and the texture path becomes the bound):
            // through the staging rows into the lane's column set
#pragma unroll
            for (int i = 0; i < 4; ++i) *(u32x4 *)(stage + ((lane >> 3) + 8 * i) * SLD + (lane & 7) * 8) = ureg[i];
            wave_lds_fence();
#pragma unroll
            for (int part = 0; part < 4; ++part) ogv[part] = *(const u32x4 *)(px + (part >> 1) * 32 + (part & 1) * 8);
            wave_lds_fence();
            float dsum = 0.f;
#pragma unroll
            for (int part = 0; part < 4; ++part) {
                const u32x4 d4 = dv[part];
                u32x4 o4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d0 = bf_lo(d4[e]), d1 = bf_hi(d4[e]);
                    const float p0 = d0 * bf_lo(ogv[part][e]), p1 = d1 * bf_hi(ogv[part][e]);
                    dsum += p0 + p1;
                    float *g = part < 2 ? cs : sn;
                    g[(part & 1) * 8 + 2 * e] += p0; g[(part & 1) * 8 + 2 * e + 1] += p1;
                    o4[e] = pack_bf16x2(d0 * bf_lo(sv[part][e]), d1 * bf_hi(sv[part][e]));
                }
                *(u32x4 *)(px + (part >> 1) * 32 + (part & 1) * 8) = o4;
            }
            dsum = sum_xor32(dsum);
            if (h == 0 && row0 + r < p.M) {
                const int64_t bb = (row0 + r) / p.tokens, nn = (row0 + r) - bb * p.tokens;
                p.Delta[(bb * p.heads + hh) * p.tokens + nn] = dsum;
            }
            wave_lds_fence();
            flush_rows64<SLD, R>(stage, p.C + hh * 64, p.ldc, row0, p.M, lane);
            wave_lds_fence();
            if (last_tile) {   // wave-uniform: every head of these rows has been visited
#pragma unroll
                for (int part = 0; part < 4; ++part) {
                    const float *g = part < 2 ? cs : sn;
                    u32x4 o4;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        o4[e] = pack_bf16x2(g[(part & 1) * 8 + 2 * e] * (1.0f - bf_lo(sv[part][e])), g[(part & 1) * 8 + 2 * e + 1] * (1.0f - bf_hi(sv[part][e])));
                    *(u32x4 *)(px + (part >> 1) * 32 + (part & 1) * 8) = o4;
                }
                wave_lds_fence();
                flush_rows64<SLD, R>(stage, p.Dgate, p.lddg, row0, p.M, lane);
                wave_lds_fence();
            }
        } else
        if constexpr (EPI == EPI_QKNORM && PAR == 1) {
            // ureg[0..3] (RB = 1): this lane's 4 x 16 bytes of the residual values of the head, requested when the pair began
            const int pp = (n0 - 32) >> 6, kind = pp / p.heads, hh = pp - kind * p.heads;
            wave_lds_fence();
            if (kind <= 1 || (kind == 2 && p.V0 != nullptr)) {
                const u32x4 v0r[4] = {ureg[0], ureg[1], ureg[2], ureg[3]};
                qknorm_head(p, stage, kind, hh, row0, lane, cs, sn, v0r, wlds);
                wave_lds_fence();
            } else if (kind == 3 && p.gate_sigmoid) {   // one sigmoid per (token, channel) here instead of one per head in every attention epilogue
                uint16_t *px = stage + r * SLD + 16 * h;
#pragma unroll
                for (int part = 0; part < 4; ++part) {
                    u32x4 *q4 = (u32x4 *)(px + (part >> 1) * 32 + (part & 1) * 8);
                    u32x4 v = *q4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = pack_bf16x2(sigm_f(bf_lo(v[e])), sigm_f(bf_hi(v[e])));
                    *q4 = v;
                }
                wave_lds_fence();
            }
            uint16_t *dst = kind == 0 ? p.Qo : (kind == 1 ? p.Ko : (kind == 2 ? p.Vo : p.Go));
            const int64_t ld = kind <= 2 ? (int64_t)p.heads * 64 : p.ldg;
            flush_rows64<SLD, R>(stage, dst + (kind <= 2 ? hh * 64 : 0), ld, row0, p.M, lane);
            wave_lds_fence();
        } else
        if constexpr (PAR == 1) {
            wave_lds_fence();
            if (EPI == EPI_PLAIN || p.C != nullptr) flush_rows64<SLD, R>(stage, p.C + (n0 - 32), p.ldc, row0, p.M, lane);
            if constexpr (EPI == EPI_SWIGLU) {   // second tile's s into columns 0..15, then 32 columns of s per row
                wave_lds_fence();
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
                    for (int g = 0; g < 2; ++g) *(uint2 *)(stage + (rb * 32 + r) * SLD + 8 * g + 4 * h) = make_uint2(sq[rb][2 * g], sq[rb][2 * g + 1]);
                }
                wave_lds_fence();
#pragma unroll
                for (int i = 0; i < R / 16; ++i) {   // 64 bytes of s per row: lanes c = 0, 1 take the first tile's half, c = 2, 3 the second's
                    const int row = (lane >> 2) + 16 * i, c = lane & 3;
                    const u32x4 v = *(const u32x4 *)(stage + row * SLD + (c < 2 ? 64 + 8 * c : 8 * (c - 2)));
                    if (row0 + row < p.M) __builtin_nontemporal_store(v, (u32x4 *)(p.S + (row0 + row) * p.lds_ + ((n0 - 32) >> 1) + c * 8));
                }
            }
            wave_lds_fence();
        }
    } else {
        // ureg: the wave's R x 64 slice of u for THIS tile (8 rows x 128 bytes per register quad), requested a tile ago
#pragma unroll
        for (int i = 0; i < 4 * RB; ++i) {
            const int row = (lane >> 3) + 8 * i, c = lane & 7;
            *(u32x4 *)(stage + row * SLD + c * 8) = ureg[i];
        }
        wave_lds_fence();
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            // the row's eight u quads first, then the stores (read-after-store through LDS pointers the compiler cannot tell apart:
            // one round trip per row block instead of four)
            uint2 uav[4], ubv[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const uint16_t *pa = stage + (rb * 32 + r) * SLD + 32 * (g >> 1) + 8 * (g & 1) + 4 * h;
                uav[g] = *(const uint2 *)pa; ubv[g] = *(const uint2 *)(pa + 16);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {   // ds column 8 g + 4 h + i  <->  u columns 32 (g / 2) + 8 (g % 2) + 4 h + i (a), + 16 (b)
                uint16_t *pa = stage + (rb * 32 + r) * SLD + 32 * (g >> 1) + 8 * (g & 1) + 4 * h, *pb = pa + 16;
                const uint2 ua = uav[g], ub = ubv[g];
                const float a[4] = {bf_lo(ua.x), bf_hi(ua.x), bf_lo(ua.y), bf_hi(ua.y)};
                const float b[4] = {bf_lo(ub.x), bf_hi(ub.x), bf_lo(ub.y), bf_hi(ub.y)};
                float da[4], db[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float gs = acc[rb][4 * g + i], sg = sigm_f(a[i]);
                    da[i] = gs * b[i] * sg * (1.0f + a[i] * (1.0f - sg));
                    db[i] = gs * a[i] * sg;
                }
                *(uint2 *)pa = make_uint2(pack_bf16x2(da[0], da[1]), pack_bf16x2(da[2], da[3]));
                *(uint2 *)pb = make_uint2(pack_bf16x2(db[0], db[1]), pack_bf16x2(db[2], db[3]));
            }
        }
        wave_lds_fence();
        flush_rows64<SLD, R>(stage, p.C + 2 * n0, p.ldc, row0, p.M, lane);
        wave_lds_fence();
    }
}

// ------------------------------------------------------------------------------------------------ rows kernel
// Workgroup = 4 waves x RB 32-row MFMA blocks (RB = 2: 256-row stripes; RB = 1: 128-row stripes), the waves' K-slices of x
// resident in VGPRs; the weight streams through LDS in tiles of 32 output columns ([32][KC] + a bias row, double-buffered, one
// barrier per tile, 16 RB MFMAs per wave and tile, every weight fragment feeds RB MFMAs).  <= 80 KB of LDS and <= 256 VGPRs: at
// least TWO workgroups per CU, whose phases (MFMA burst / epilogue / tile refill) drift apart and overlap -- one 8-wave
// workgroup in lockstep ran the matrix pipe at 25 %.
// Tiles are visited in a rotated order (first tile pair = workgroup index): the workgroups of a launch pull DIFFERENT tiles of W
// out of L2 at any moment; the next tile's loads are in flight during the current tile's MFMAs and epilogue.
// Outputs leave in pairs of tiles (64 columns = 128-byte row segments, non-temporal) through the wave's staging buffer.
// EPI_SWIGLU_BWD runs with RB = 1: the tile's slice of the saved u is requested a tile ahead into registers, so its HBM
// latency is hidden behind the MFMAs (with RB = 2 there are no registers left for that and every tile waited for its u).
template <int EPI, int NKH = 1> constexpr int rows_rb() { return (EPI == EPI_SWIGLU_BWD || EPI == EPI_QKNORM || EPI == EPI_GATE_BWD || NKH > 1) ? 1 : 2; }

// KC: depth of the LDS weight tile; the wave's resident operand holds KSTOT = NKH * KC / 16 fragments per row block and this call
// multiplies the slice [K0, K0 + KC / 16) of them (K = 512 runs as two k-halves of 256 through the same 17 KB tile buffers).
template <int KC, int RB, int KSTOT, int K0, bool ZERO>
__device__ __forceinline__ void rows_tile_mfma(f32x16 (&acc)[RB], const bf16x8 (&afr)[RB][KSTOT], const uint16_t *bsrc) {
    constexpr int KS = KC / 16, GK = RB == 1 ? 4 : 2, NG = KS / GK;
    if constexpr (ZERO) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[rb][e] = 0.f;
    }
    bf16x8 bq[2][GK];
#pragma unroll
    for (int k = 0; k < GK; ++k) bq[0][k] = *(const bf16x8 *)(bsrc + k * 16);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int gk = 0; gk < NG; ++gk) {
        if (gk + 1 < NG)
#pragma unroll
            for (int k = 0; k < GK; ++k) bq[(gk + 1) & 1][k] = *(const bf16x8 *)(bsrc + ((gk + 1) * GK + k) * 16);
        __builtin_amdgcn_sched_barrier(0);   // keep the next group's reads ahead of this group's MFMAs (hipcc sinks them otherwise)
#pragma unroll
        for (int k = 0; k < GK; ++k)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[gk & 1][k], afr[rb][K0 + gk * GK + k], acc[rb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
}

// NW = waves per workgroup: 4 (two workgroups per CU), or 8 = ONE 256-row workgroup per CU for the epilogues that only leave
// registers for 32-row waves (VSDE_ROWS_NW=4 restores four): a weight tile then still serves 256 rows per trip through LDS.
template <int KC, int EPI, int NKH, int NW>
__global__ void __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) lin_rows_kernel(LinParams p) {
    constexpr int RB = rows_rb<EPI, NKH>(), ROWS = 32 * NW * RB, THREADS = 64 * NW;
    constexpr int KS = NKH * KC / 16, KT = NKH * KC, LDB = KC + 8, TILE = 33 * LDB;   // 32 weight rows + 1 bias row
    constexpr int NLD = 32 * KC / 8 / THREADS;                   // 16-byte loads per thread and tile
    extern __shared__ __attribute__((aligned(16))) uint16_t lsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    uint16_t *stage = lsm + 2 * TILE + wave * (32 * RB * R2_SLD);
    // Workgroup id = 8 * local + xcd (consecutive ids go round-robin over the XCDs); the column chunks of one row stripe take
    // consecutive `local` on ONE XCD: they run together and share the stripe's activation rows through that XCD's L2.
    // The stripes of the last, partly filled round of resident workgroups are cut into more column chunks (launch_rows_nw): the
    // tail then runs on all CUs for a fraction of a stripe's time instead of on a few CUs for a whole one.
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const bool tail = local >= p.tail_local;
    const int nchunks = tail ? p.tail_chunks : p.chunks, lrel = tail ? local - p.tail_local : local;
    const int chunk = lrel % nchunks;
    const int64_t stripe = (int64_t)(lrel / nchunks + (tail ? p.tail_local / p.chunks : 0)) * 8 + xcd;
    if (stripe * ROWS >= p.M) return;
    const int64_t row0 = stripe * ROWS + wave * (32 * RB);

    bf16x8 afr[RB][KS];
    // Two ways to fill the resident operand.  (a) fragment-shaped: lane (r, h) loads its 16 bytes of every k-step straight from
    // global memory -- 32 rows x 32 bytes per instruction, i.e. every 128-byte line of x is requested by four instructions.
    // (b) `xstage` (K <= 256, round 3): the wave loads its 32 rows as WHOLE row segments (a row = KT / 8 lanes x 16 bytes, full
    // lines), parks them in its own 32 x KT slice of LDS (16-byte chunks XOR-swizzled by row & 15: conflict-free both ways) and
    // reads the fragments back; no workgroup barrier is involved until the slices are handed over to the weight tiles.
    const bool xst = NKH == 1 && p.xstage && !(EPI == EPI_PLAIN && p.Gate != nullptr);
    if (xst) {
        constexpr int CPR = KT / 8, RPI = 64 / CPR;          // 16-byte chunks per row, rows per load instruction
        char *xs = (char *)lsm + wave * (32 * KT * 2);
        const int lrow = lane / CPR, lc = lane % CPR;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            u32x4 t[KS];
#pragma unroll
            for (int i = 0; i < KS; ++i) {
                const int rl = i * RPI + lrow;
                const int64_t m = row0 + rb * 32 + rl < p.M ? row0 + rb * 32 + rl : p.M - 1;
                t[i] = *(const u32x4 *)(p.A + m * p.lda + lc * 8);
            }
            if (rb > 0) wave_lds_fence();
#pragma unroll
            for (int i = 0; i < KS; ++i) {
                const int rl = i * RPI + lrow;
                *(u32x4 *)(xs + rl * (KT * 2) + ((lc ^ (rl & 15)) << 4)) = t[i];
            }
            wave_lds_fence();
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) afr[rb][ks] = *(const bf16x8 *)(xs + r * (KT * 2) + (((h + 2 * ks) ^ (r & 15)) << 4));
        }
        lds_barrier();   // every wave has its fragments: the slices become the weight-tile buffers / staging rows
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        if (xst) break;
        const int64_t m = row0 + rb * 32 + r < p.M ? row0 + rb * 32 + r : p.M - 1;   // rows past the end repeat the last one (never stored)
        const uint16_t *src = p.A + m * p.lda + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) afr[rb][ks] = *(const bf16x8 *)(src + ks * 16);
        if (EPI == EPI_PLAIN && NKH == 1 && p.Gate != nullptr) {   // workgroup-uniform: gate_merge (primitives/attn.py:107-109) folded into the load:
            const uint16_t *gsrc = p.Gate + m * p.ldgate + 8 * h;   // a[k] * rnd(sigmoid(g[k % 64])), rounded to bf16 like the kernel's output
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const u32x4 a = __builtin_bit_cast(u32x4, afr[rb][ks]), g = *(const u32x4 *)(gsrc + (ks & 3) * 16);
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    o[e] = pack_bf16x2(bf_lo(a[e]) * rbf(sigm_f(bf_lo(g[e]))), bf_hi(a[e]) * rbf(sigm_f(bf_hi(g[e]))));
                afr[rb][ks] = __builtin_bit_cast(bf16x8, o);
            }
        }
    }
    __shared__ float wlds[128];   // EPI_QKNORM: [wq | wk]
    if constexpr (EPI == EPI_QKNORM) { if (tid < 128) wlds[tid] = tid < 64 ? p.wq[tid] : p.wk[tid - 64]; }   // visible after the first barrier
    float cs[16], sn[16];   // EPI_QKNORM: (cos, sin) of this lane's 16 rotary pairs of ITS row (token = row % tokens)
#pragma unroll
    for (int e = 0; e < 16; ++e) { cs[e] = EPI == EPI_GATE_BWD ? 0.f : 1.f; sn[e] = 0.f; }   // EPI_GATE_BWD: the gate gradient's partial sums
    if constexpr (EPI == EPI_QKNORM) {
        const int64_t m = row0 + r < p.M ? row0 + r : p.M - 1;
        const int64_t tok = m % p.tokens;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 c4 = *(const float4 *)(p.cosT + tok * 32 + 16 * h + 4 * q), s4 = *(const float4 *)(p.sinT + tok * 32 + 16 * h + 4 * q);
            cs[4 * q] = c4.x; cs[4 * q + 1] = c4.y; cs[4 * q + 2] = c4.z; cs[4 * q + 3] = c4.w;
            sn[4 * q] = s4.x; sn[4 * q + 1] = s4.y; sn[4 * q + 2] = s4.z; sn[4 * q + 3] = s4.w;
        }
    }
    // a workgroup owns one chunk of the output columns of its stripe (see launch_rows)
    // column chunks in whole tile pairs; the last chunk may be shorter (N = 704: 11 pairs = 6 + 5)
    const int pairs_all = p.N / 64, ppc = (pairs_all + nchunks - 1) / nchunks, pair0 = chunk * ppc;
    const int ntiles = 2 * (ppc < pairs_all - pair0 ? ppc : pairs_all - pair0), tile0 = 2 * pair0;
    if (ntiles <= 0) return;   // workgroup-uniform: an empty trailing chunk
    const int rot = 2 * ((int)stripe % (ntiles / 2));
    u32x4 breg[NLD];          // the next sub-tile on its way from L2 to LDS (loaded one iteration before its LDS store)
    uint32_t biasreg = 0u;
    u32x4 ureg[4 * RB];       // EPI_SWIGLU_BWD: the next tile's slice of the saved u (32 RB rows x 64 columns)
    // Sub-tile q = NKH * t + kh: k-half kh of the t-th visited tile, i.e. W[32 tile .. 32 tile + 32][KC kh .. KC kh + KC] (row pitch KT);
    // the LDS buffers alternate with q.
#define VSDE_TILE_LOAD(q_)                                                                                    \
    do {                                                                                                      \
        const int tile_ = tile0 + ((q_) / NKH + rot) % ntiles;                                                \
        wtile_load<NLD, KC, THREADS>(breg, p.W + (int64_t)tile_ * 32 * KT + ((q_) % NKH) * KC, KT, tid);      \
        if (tid < 16) biasreg = p.bias ? *(const uint32_t *)(p.bias + tile_ * 32 + 2 * (lane_t & 15)) : 0u;   \
    } while (0)
#define VSDE_TILE_STORE(Bs_)                                                                                  \
    do {                                                                                                      \
        wtile_store<NLD, KC, LDB, THREADS>(breg, (Bs_), tid);                                                 \
        if (tid < 16) *(uint32_t *)((Bs_) + 32 * LDB + 2 * tid) = biasreg;                                    \
    } while (0)
#define VSDE_U_LOAD(t_)                                                                                       \
    do {                                                                                                      \
        if constexpr (EPI == EPI_QKNORM) {   /* residual values of a v head: requested when its tile pair begins */ \
            const int tl_ = tile0 + ((t_) + rot) % ntiles, pp_ = tl_ >> 1, kind_ = pp_ / p.heads;              \
            if (p.V0 != nullptr && (tl_ & 1) == 0 && kind_ == 2) {                                            \
                const int64_t m_ = row0 + (lane_t & 31) < p.M ? row0 + (lane_t & 31) : p.M - 1;               \
                const uint16_t *v_ = p.V0 + m_ * ((int64_t)p.heads * 64) + (pp_ - 2 * p.heads) * 64 + 16 * (lane_t >> 5); \
                ureg[0] = *(const u32x4 *)v_; ureg[1] = *(const u32x4 *)(v_ + 8);                             \
                ureg[2] = *(const u32x4 *)(v_ + 32); ureg[3] = *(const u32x4 *)(v_ + 40);                     \
            }                                                                                                 \
        }                                                                                                     \
        if constexpr (EPI == EPI_GATE_BWD) {   /* the head's slice of og as full rows: requested when its tile pair begins */ \
            const int tl_ = tile0 + ((t_) + rot) % ntiles;                                                    \
            if ((tl_ & 1) == 0) {                                                                             \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                               \
                    const int row = (lane_t >> 3) + 8 * i, c = lane_t & 7;                                    \
                    const int64_t m_ = row0 + row < p.M ? row0 + row : p.M - 1;                               \
                    ureg[i] = *(const u32x4 *)(p.Og + m_ * (int64_t)p.N + (tl_ >> 1) * 64 + c * 8);           \
                }                                                                                             \
            }                                                                                                 \
        }                                                                                                     \
        if constexpr (EPI == EPI_SWIGLU_BWD) {                                                                \
            const int n0_ = (tile0 + ((t_) + rot) % ntiles) * 32;                                             \
            _Pragma("unroll") for (int i = 0; i < 4 * RB; ++i) {                                              \
                const int row = (lane >> 3) + 8 * i, c = lane & 7;                                            \
                const int64_t m_ = row0 + row < p.M ? row0 + row : p.M - 1;                                   \
                ureg[i] = *(const u32x4 *)(p.U + m_ * p.ldu + 2 * n0_ + c * 8);                               \
            }                                                                                                 \
        }                                                                                                     \
    } while (0)
// one sub-tile: MFMAs out of LDS buffer q & 1 into the tile's accumulators; after the last k-half the epilogue; then sub-tile
// q + 1 (in breg) goes to the other buffer.  t and the tile index have the same parity (rot is even): TPAR_ = 1 closes a pair of tiles.
#define VSDE_ROWS_BODY(t_, TPAR_, KH_)                                                                        \
    do {                                                                                                      \
        constexpr int q_par = (NKH * (TPAR_) + (KH_)) & 1;                                                    \
        const int q_ = NKH * (t_) + (KH_);                                                                    \
        if constexpr (EPI == EPI_QKNORM || EPI == EPI_GATE_BWD) asm volatile("" : "+v"(lane_t));              \
        const uint16_t *Bs = lsm + q_par * TILE;                                                              \
        rows_tile_mfma<KC, RB, KS, (KH_) * (KC / 16), (KH_) == 0>(acc, afr, Bs + r * LDB + 8 * h);            \
        if constexpr ((KH_) == NKH - 1)                                                                       \
            rows_epilogue<EPI, TPAR_, RB>(p, acc, Bs + 32 * LDB, stage, ureg, row0, (tile0 + ((t_) + rot) % ntiles) * 32, lane_t, cs, sn, wlds, (t_) + 1 == ntiles); \
        if (q_ + 1 < NKH * ntiles) VSDE_TILE_STORE(lsm + (1 - q_par) * TILE);                                 \
        lds_barrier();                                                                                        \
        if (q_ + 2 < NKH * ntiles) VSDE_TILE_LOAD(q_ + 2);                                                    \
        if constexpr ((KH_) == NKH - 1) { if ((t_) + 1 < ntiles) VSDE_U_LOAD((t_) + 1); }                     \
    } while (0)
    // The fused epilogues' global addresses (rows of V0 / og / rinv / the outputs) are loop-invariant per lane: hoisted out of the
    // tile loop they are spilled (256 VGPRs are in use), and every reload of a spilled address is followed by a full vmcnt(0) drain of
    // the weight-tile loads in flight.  An opaque copy of the lane index per tile makes them cheap recomputations instead.
    int lane_t = lane;
    VSDE_TILE_LOAD(0);
    VSDE_U_LOAD(0);
    VSDE_TILE_STORE(lsm);
    lds_barrier();
    VSDE_TILE_LOAD(1);
    f32x16 acc[RB];
    for (int nt = 0; nt < ntiles; nt += 2) {
        if constexpr (NKH == 1) {
            VSDE_ROWS_BODY(nt, 0, 0);
            VSDE_ROWS_BODY(nt + 1, 1, 0);
        } else {
            VSDE_ROWS_BODY(nt, 0, 0);
            VSDE_ROWS_BODY(nt, 0, 1);
            VSDE_ROWS_BODY(nt + 1, 1, 0);
            VSDE_ROWS_BODY(nt + 1, 1, 1);
        }
    }
#undef VSDE_ROWS_BODY
#undef VSDE_U_LOAD
#undef VSDE_TILE_LOAD
#undef VSDE_TILE_STORE
}

// ------------------------------------------------------------------------------------------------ cols kernel
// Deep reductions (any K % 64 == 0): workgroup = 4 waves x 32 rows x one output tile of NT = 32 NB columns (NB = 8: 256
// columns, or 4), whose accumulator blocks stay in VGPRs while K streams in chunks of 64: the weight chunk [NT][64] through LDS
// (double-buffered, one barrier per chunk, 4 NB MFMAs per wave and chunk), the activation fragments of the next chunk straight
// into registers during the MFMAs of the current one.  With N <= 256 the activations -- the big operand -- are read from HBM
// exactly once.  <= 74 KB of LDS, <= 256 VGPRs: two workgroups per CU whose phases overlap.
constexpr int C2_THREADS = 256, C2_ROWS = 128, C2_LDB = 72;

template <int NB>
__device__ __forceinline__ void cols_chunk_mfma(f32x16 (&acc)[NB], const bf16x8 (&afr)[4], const uint16_t *bsrc) {
    // the chunk's 4 (k-step) x NB (column block) weight fragments in order; a group of 4 is fetched while the previous group's
    // MFMAs run
    constexpr int G = 4, NGRP = 4 * NB / G;
    bf16x8 bq[2][G];
#pragma unroll
    for (int i = 0; i < G; ++i) bq[0][i] = *(const bf16x8 *)(bsrc + (i % NB) * 32 * C2_LDB + (i / NB) * 16);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int g = 0; g < NGRP; ++g) {
        if (g + 1 < NGRP)
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const int q = (g + 1) * G + i;
                bq[(g + 1) & 1][i] = *(const bf16x8 *)(bsrc + (q % NB) * 32 * C2_LDB + (q / NB) * 16);
            }
        __builtin_amdgcn_sched_barrier(0);   // keep the next group's reads ahead of this group's MFMAs (hipcc sinks them otherwise)
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int q = g * G + i;
            acc[q % NB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[g & 1][i], afr[q / NB], acc[q % NB], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
}

template <int NB>
__global__ void __launch_bounds__(C2_THREADS, 2) lin_cols_kernel(LinParams p) {
    constexpr int NT = 32 * NB, TILE = NT * C2_LDB;
    constexpr int NLD = NT * 64 / 8 / C2_THREADS;   // 16-byte loads per thread and chunk (8 or 4)
    extern __shared__ __attribute__((aligned(16))) uint16_t lsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int ntn = p.N / NT, stripe = blockIdx.x / ntn, nbase = (blockIdx.x - stripe * ntn) * NT;
    const int64_t row0 = (int64_t)stripe * C2_ROWS + wave * 32;
    const int64_t m = row0 + r < p.M ? row0 + r : p.M - 1;   // rows past the end repeat the last one (never stored)
    const uint16_t *asrc = p.A + m * p.lda + 8 * h;
    const int ktiles = p.K / 64;
    const int rot = stripe % ktiles;   // K chunks in a rotated order: concurrent workgroups pull different chunks of W out of L2
    const uint16_t *wsrc = p.W + (int64_t)nbase * p.K;
    u32x4 breg[NLD];
    bf16x8 afr[2][4];   // [set][k-step]
    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[nb][e] = 0.f;
#define VSDE_CHUNK(t_) ((((t_) + rot) % ktiles) * 64)
#define VSDE_A_LOAD(t_, set_) \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) afr[set_][ks] = *(const bf16x8 *)(asrc + VSDE_CHUNK(t_) + ks * 16);
#define VSDE_COLS_BODY(t_, PAR_)                                                                              \
    do {                                                                                                      \
        if ((t_) + 1 < ktiles) { VSDE_A_LOAD((t_) + 1, 1 - (PAR_)) }                                          \
        cols_chunk_mfma<NB>(acc, afr[PAR_], lsm + (PAR_) * TILE + r * C2_LDB + 8 * h);                        \
        if ((t_) + 1 < ktiles) wtile_store<NLD, 64, C2_LDB, C2_THREADS>(breg, lsm + (1 - (PAR_)) * TILE, tid); \
        lds_barrier();                                                                                        \
        if ((t_) + 2 < ktiles) wtile_load<NLD, 64, C2_THREADS>(breg, wsrc + VSDE_CHUNK((t_) + 2), p.K, tid);  \
    } while (0)
    wtile_load<NLD, 64, C2_THREADS>(breg, wsrc + VSDE_CHUNK(0), p.K, tid);
    VSDE_A_LOAD(0, 0)
    wtile_store<NLD, 64, C2_LDB, C2_THREADS>(breg, lsm, tid);
    lds_barrier();
    if (ktiles > 1) wtile_load<NLD, 64, C2_THREADS>(breg, wsrc + VSDE_CHUNK(1), p.K, tid);
    for (int kt = 0; kt < ktiles; kt += 2) {
        VSDE_COLS_BODY(kt, 0);
        if (kt + 1 < ktiles) VSDE_COLS_BODY(kt + 1, 1);
    }
#undef VSDE_COLS_BODY
#undef VSDE_A_LOAD
#undef VSDE_CHUNK
    // epilogue: the weight buffers are free now: bias row in front, then one 32-row staging area per wave
    uint16_t *brow = lsm, *stage = lsm + NT + wave * (32 * C2_LDB);
    if (tid < NT / 2) *(uint32_t *)(brow + 2 * tid) = p.bias ? *(const uint32_t *)(p.bias + nbase + 2 * tid) : 0u;
    lds_barrier();
#pragma unroll
    for (int q = 0; q < NB / 2; ++q) {   // 64 columns at a time
        stage_block(acc[2 * q], brow + 64 * q, stage + r * C2_LDB, h);
        stage_block(acc[2 * q + 1], brow + 64 * q + 32, stage + r * C2_LDB + 32, h);
        wave_lds_fence();
        flush_rows64<C2_LDB, 32>(stage, p.C + nbase + 64 * q, p.ldc, row0, p.M, lane);
        wave_lds_fence();
    }
}

// ------------------------------------------------------------------------------------------------ deep kernel
// Deep reductions with a narrow output (K = 768 / 832 / 1536 -> N = 256 at the LV encoder: SwiGLU output projection and the
// input gradients of the wide projections; K = 1408 / 1664 / 2816 -> N = 512 at the 512-wide encoder): 2 K flops per activation
// byte, i.e. right at the MFMA / HBM balance point, and an operand slice too deep to stay in registers.
//   * persistent workgroups, one per CU (8 waves = 2 per SIMD, 96 KB of LDS): workgroup i owns the CONTIGUOUS row range
//     [i R, (i + 1) R) with R = M / #CUs rounded up to 32 rows and walks it in 256 x 256 output tiles; the last tile of a
//     range is partial and waves without valid rows skip their MFMAs -- 802 row tiles on 256 CUs cost ~3.2 tile times instead
//     of the 4 rounds of a tile-per-workgroup grid;
//   * both operands stream through LDS in 32-deep K chunks (activations [256][32] + weights [256][32] = 32 KB per stage, three
//     stages -> FOUR since the ablation below) filled by global_load_lds_dwordx4 in full 64-byte row segments: no staging registers, no fragment-shaped global
//     loads (which cost the texture path 4 requests per 128-byte line); a row's four 16-byte chunks are stored XOR-swizzled by
//     (row >> 2) & 3 -- applied to the SOURCE address, the LDS image of a DMA is lane-linear -- so that the ds_read_b128 of a
//     32-row MFMA fragment touches every bank group once;
//   * counted waits: chunk t + 2 is requested right after the barrier that publishes chunk t, `s_waitcnt vmcnt(4)` (the four
//     DMA instructions of chunk t + 1 may stay in flight) + a raw s_barrier per chunk -- __syncthreads() would drain the queue;
//   * waves are laid out 4 (rows) x 2 (columns): a wave owns 64 rows x 128 columns = 2 x 4 accumulator blocks, 6 fragment reads
//     per 8 MFMAs; products are swapped (D = W_tile x^T) like in the rows kernel, so the epilogue (bias, bf16, 128-byte row
//     segments through the wave's staging buffer) is shared with it.
constexpr int DP_THREADS = 512, DP_BK = 64, DP_ASTAGES = 3, DP_WSTAGES = 2, DP_HALF = 256 * DP_BK * 2, DP_SLD = 72;   // 32 KB per operand and chunk
constexpr int DP_LDS_BYTES = (DP_ASTAGES + DP_WSTAGES) * DP_HALF;   // 160 KB

// Second version (round 3): 64-deep K chunks.  With 32-deep chunks every 128-byte line of both operands was requested as two
// 64-byte halves one chunk apart (activation stream 3.6 TB/s, weight chunks out of L2 at 5.8 TB/s: profiles/r03_deep_gemm_ablation.txt).
// A chunk is now [256 rows] x 128 bytes = 32 KB per operand; a DMA instruction fills 8 full rows (8 lanes x 16 bytes each); the
// eight 16-byte chunks of a row are stored XOR-swizzled by (row >> 1) & 7 (on the source address) so that a 32-row ds_read_b128
// fragment touches every bank once; the four k-steps of a chunk are software-pipelined inside a wave (fragments of k-step s + 1
// are requested before the eight MFMAs of k-step s, hand-counted lgkmcnt).  The two operand streams have their own rings: the
// activations (HBM: latency-bound with one chunk in flight, 3.4 TB/s) THREE stages = two chunks in flight, loaded by waves 0..3;
// the weight chunks (L2 hits) two stages, loaded by waves 4..7 -- 160 KB of LDS, each wave counts only its own stream.
#ifdef VSDE_ABLATIONS   // lin_deep_kernel: 30 % behind the tuned library solutions (profiles/r03_deep_gemm_ablation.txt): tools' build only
__global__ void __launch_bounds__(DP_THREADS, 2) lin_deep_kernel(LinParams p, int rows_per_wg) {
    extern __shared__ __attribute__((aligned(16))) char dlds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
    const int64_t range0 = (int64_t)blockIdx.x * rows_per_wg;
    const int64_t range1 = range0 + rows_per_wg < p.M ? range0 + rows_per_wg : p.M;
    if (range0 >= p.M) return;
    const int nk = p.K / DP_BK, ncol = p.N / 256;
    // ---- this lane's part of the DMA pattern: instruction q of wave w fills LDS bytes [(8 w + q) 1024, + 1024) of a stage = rows
    // 8 (8 (w & 3) + q) + lane / 8 of the activation half (waves 0..3) or the weight half (waves 4..7); LDS chunk slot lane % 8 of
    // row R holds the row's logical chunk (lane % 8) ^ ((R >> 1) & 7), and (R >> 1) & 7 = ((lane >> 4) + 4 (q & 1)) & 7
    const int cs0 = ((lane & 7) ^ ((lane >> 4) & 7)) * 8, cs1 = ((lane & 7) ^ (((lane >> 4) + 4) & 7)) * 8;   // element offsets, q even / odd
    const int lrow = lane >> 3;
    // ---- fragment addresses (bytes inside a stage): row block base + r * 128 + ((2 ks + h) ^ swz) * 16
    const int swz = (r >> 1) & 7;
    int fo[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fo[ks] = r * 128 + (((2 * ks + h) ^ swz) << 4);
    const int a_base = (wm * 64) * 128, w_base = DP_ASTAGES * DP_HALF + (wn * 128) * 128;

    for (int64_t trow = range0; trow < range1; trow += 256) {
        const bool active = trow + wm * 64 < range1;     // wave-uniform: rows of this wave inside the range
        const int64_t bound = range1;                    // rows >= bound belong to the next workgroup (or do not exist)
        for (int ct = 0; ct < ncol; ++ct) {
            // source pointers of this lane's eight DMA instructions (chunk 0 of K)
            const uint16_t *src[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int gi = (wave & 3) * 8 + q, cs = (q & 1) ? cs1 : cs0;
                if (wave < 4) {
                    int64_t m = trow + gi * 8 + lrow;
                    m = m < p.M ? m : p.M - 1;            // rows past the end repeat the last one (never stored)
                    src[q] = p.A + m * p.lda + cs;
                } else {
                    src[q] = p.W + (int64_t)(ct * 256 + gi * 8 + lrow) * p.K + cs;
                }
            }
            auto issue = [&](int kt) {   // this wave's eight rows-of-8 of chunk kt of ITS operand
                char *dst = wave < 4 ? dlds + (kt % DP_ASTAGES) * DP_HALF + wave * 8192
                                     : dlds + DP_ASTAGES * DP_HALF + (kt % DP_WSTAGES) * DP_HALF + (wave - 4) * 8192;
                if (p.dbg && kt > 0 && (((p.dbg == 2 || p.dbg == 7) && wave < 4) || ((p.dbg == 3 || p.dbg == 6) && wave >= 4) || p.dbg == 4)) return;   // ablation only
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    __builtin_amdgcn_global_load_lds((const void *)(src[q] + kt * DP_BK), (__attribute__((address_space(3))) void *)(dst + q * 1024),
                                                     16, 0, 0);
            };
            f32x16 acc[2][4];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[rb][nb][e] = 0.f;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the previous tile's output stores (stores and loads are not ordered against each other)
            issue(0);
            if (wave < 4 && nk > 1) issue(1);
            for (int kt = 0; kt < nk; ++kt) {
                // this wave's part of chunk kt has landed once at most its younger chunk's eight DMA instructions are in flight
                if (wave < 4 && kt + 1 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_barrier" ::: "memory");            // chunk kt is in LDS for everyone; everyone is done with chunk kt - 1
                if (wave < 4) { if (kt + 2 < nk) issue(kt + 2); }  // into the stage chunk kt - 1 occupied
                else if (kt + 1 < nk) issue(kt + 1);
                if (active && p.dbg < 5) {
                    const uint32_t sa = (uint32_t)(uintptr_t)(dlds + (kt % DP_ASTAGES) * DP_HALF + a_base);
                    const uint32_t sw = (uint32_t)(uintptr_t)(dlds + (kt % DP_WSTAGES) * DP_HALF + w_base);
                    bf16x8 fa[2][2], fw[2][4];   // [k-step parity][block]
                    // inline asm with hand-counted waits: with DMA / scalar loads in flight hipcc only ever emits lgkmcnt(0) here
#define VSDE_FRAG(dst_, addr_, off_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst_) : "v"(addr_), "i"(off_))
#define VSDE_FRAGS(par_, ks_)                                                                                  \
                    do {                                                                                       \
                        const uint32_t a_ = sa + fo[ks_], w_ = sw + fo[ks_];                                   \
                        VSDE_FRAG(fa[par_][0], a_, 0); VSDE_FRAG(fw[par_][0], w_, 0); VSDE_FRAG(fw[par_][1], w_, 4096);   \
                        VSDE_FRAG(fa[par_][1], a_, 4096); VSDE_FRAG(fw[par_][2], w_, 8192); VSDE_FRAG(fw[par_][3], w_, 12288); \
                    } while (0)
#define VSDE_MFMAS(par_)                                                                                       \
                    do {                                                                                       \
                        _Pragma("unroll") for (int nb = 0; nb < 4; ++nb)                                       \
                            _Pragma("unroll") for (int rb = 0; rb < 2; ++rb)                                   \
                                acc[rb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[par_][nb], fa[par_][rb], acc[rb][nb], 0, 0, 0); \
                    } while (0)
                    VSDE_FRAGS(0, 0);
                    VSDE_FRAGS(1, 1);
                    __builtin_amdgcn_s_setprio(1);
                    asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    VSDE_MFMAS(0);
                    __builtin_amdgcn_sched_barrier(0);
                    VSDE_FRAGS(0, 2);                                  // overwrites the k-step 0 fragments: their MFMAs have issued (and read them)
                    asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");   // k-step 1 has landed
                    __builtin_amdgcn_sched_barrier(0);
                    VSDE_MFMAS(1);
                    __builtin_amdgcn_sched_barrier(0);
                    VSDE_FRAGS(1, 3);
                    asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    VSDE_MFMAS(0);
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    VSDE_MFMAS(1);
                    __builtin_amdgcn_s_setprio(0);
#undef VSDE_MFMAS
#undef VSDE_FRAGS
#undef VSDE_FRAG
                }
            }
            // ---- epilogue: stage buffers are free once everyone has left the K loop
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            uint16_t *brow = (uint16_t *)dlds;                                   // 256 bias values
            uint16_t *stg = (uint16_t *)dlds + 256 + wave * (32 * DP_SLD);      // 32 rows x 64 columns per wave
            if (tid < 128) *(uint32_t *)(brow + 2 * tid) = p.bias ? *(const uint32_t *)(p.bias + ct * 256 + 2 * tid) : 0u;
            lds_barrier();
            if (active) {
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {   // 64 columns at a time
                        const int n0 = wn * 128 + pr * 64;
                        stage_block(acc[rb][2 * pr], brow + n0, stg + r * DP_SLD, h);
                        stage_block(acc[rb][2 * pr + 1], brow + n0 + 32, stg + r * DP_SLD + 32, h);
                        wave_lds_fence();
                        flush_rows64<DP_SLD, 32>(stg, p.C + ct * 256 + n0, p.ldc, trow + wm * 64 + rb * 32, bound, lane);
                        wave_lds_fence();
                    }
            }
            lds_barrier();   // staging area is stage 0 again
        }
    }
}
#endif  // VSDE_ABLATIONS (lin_deep_kernel)

static int rows_xstage() {   // VSDE_ROWS_XSTAGE=0: fragment-shaped activation loads in the rows kernel (A/B runs)
    static int v = -1;
    if (v < 0) v = (int)vsde_knob("VSDE_ROWS_XSTAGE", 1);
    return v;
}

template <int KC, int EPI, int NKH, int NW> static size_t rows_lds_bytes() {
    const size_t need = (size_t)(2 * 33 * (KC + 8) + NW * 32 * rows_rb<EPI, NKH>() * R2_SLD) * sizeof(uint16_t);
    const size_t xstage = NKH == 1 ? (size_t)NW * 32 * KC * 2 : 0;   // the waves' activation slices of the prologue
    return need > xstage ? need : xstage;
}
template <int NB> static size_t cols_lds_bytes() { return (size_t)(2 * 32 * NB * C2_LDB) * sizeof(uint16_t); }   // two weight buffers (the epilogue reuses them)

template <int KC, int EPI, int NKH, int NW>
static int launch_rows_nw(const LinParams &p, hipStream_t s) {
    const size_t lds = rows_lds_bytes<KC, EPI, NKH, NW>();
    VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)lin_rows_kernel<KC, EPI, NKH, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    constexpr int rows = 32 * NW * rows_rb<EPI, NKH>();
    constexpr int resident = NW == 4 ? 512 : 256;   // workgroups on the chip at a time
    // Column chunks per row stripe (a chunk is a whole number of tile pairs; its workgroup re-reads the stripe's rows, L2 hits):
    //  * few stripes (the OU example has 12.9 k tokens = 51 stripes on 256 CUs): enough chunks for about two workgroups per CU;
    //  * many stripes: the grid is a non-integer number of rounds of the 512 resident workgroups (802 stripes = 1.57 rounds:
    //    1.39 ns per row at M = 205 k against 1.22 at M = 393 k) -- finer workgroups round up less.
    const int64_t stripes = (p.M + rows - 1) / rows;
    const int pairs = p.N / 64;
    int chunks = 1;
    if ((EPI == EPI_SWIGLU || EPI == EPI_SWIGLU_BWD || EPI == EPI_QKNORM) && stripes < resident && pairs >= 2) {
        // these epilogues work pair by pair, so a stripe's chunks need not be equal (the kernel gives the last one what is left):
        // about one round of resident workgroups whatever the pair count (N = 1408 / 832 / 704: 22 / 13 / 11 pairs -- the
        // doubling rule below stops at 2 / 1 / 1 chunks, i.e. ~100 workgroups on 256 CUs at the OU example's 12.9 k tokens)
        int64_t c = (resident + stripes - 1) / stripes;
        if (c > pairs) c = pairs;
        const int64_t ppc = (pairs + c - 1) / c;
        chunks = (int)((pairs + ppc - 1) / ppc);   // no empty trailing chunk
    } else {
        while (stripes * chunks < resident && chunks * 2 <= pairs && pairs % (chunks * 2) == 0) chunks *= 2;
    }
    {
        static int big = -1;   // VSDE_ROWS_CHUNKS: chunks at large M (A/B runs; 0 = the default below)
        if (big < 0) big = (int)vsde_knob("VSDE_ROWS_CHUNKS", 0);
        // measured at M = 205 k: SwiGLU forward 290 | 277 | 272 | 307 us and backward 303 | 287 | 295 | 305 us for 1 | 2 | 4 | 8
        // chunks (every chunk reloads the stripe's rows and restarts the tile pipeline); the plain epilogue does not gain
        const bool uneven_ok = EPI == EPI_SWIGLU || EPI == EPI_SWIGLU_BWD;   // a shorter last chunk is fine for these epilogues
        // (the QK-norm epilogue would also allow whole-head chunks, but its workgroup prologue is heavy: 240 | 241 | 254 | 296 us for 1..4 chunks)
        int want = big > 0 ? big : (EPI == EPI_PLAIN ? 1 : 2);
        if (big <= 0 && uneven_ok) {
            // the chunk count that leaves the fullest last round of resident workgroups (802 stripes on 512 slots: 2 chunks = 3.13
            // rounds, 3 chunks = 4.70: LV step -0.08 ms), a little in favour of fewer chunks (every chunk re-reads the stripe's rows)
            double best = 0.0;
            for (int c = 2; c <= 4; ++c) {
                if (pairs < 2 * c) break;
                // (0.05 per chunk since round 4: the SwiGLU backward's 1,604 stripes of 128 rows took 4 chunks with 0.02 -- 302 us against
                //  284 / 293 / 307 us for 2 / 3 / 4 chunks stand-alone; the forward keeps 3)
                const double rounds = (double)stripes * c / resident, eff = rounds / (double)(int64_t)(rounds + 0.999999) - 0.05 * c;
                if (eff > best) { best = eff; want = c; }
            }
        }
        if (stripes >= resident && want > 1 && (pairs % want == 0 || (uneven_ok && pairs >= 2 * want))) chunks = want;
    }
    if (EPI == EPI_GATE_BWD) chunks = 1;   // a wave must visit every head of its rows (the gate gradient sums over them)
    LinParams q = p;
    q.chunks = chunks;
    q.xstage = rows_xstage();
    // tail: stripe groups (8 stripes, one per XCD) beyond the last full round of resident workgroups
    const int64_t groups = (stripes + 7) / 8, wg_main = groups * 8 * chunks;
    int64_t tail_groups = 0; int tail_chunks = chunks;
    {
        static int on = -1;   // VSDE_ROWS_TAIL=0: no finer chunks for the last round (A/B runs)
        if (on < 0) on = (int)vsde_knob("VSDE_ROWS_TAIL", 1);
        const int64_t full = wg_main / resident * resident, rest = wg_main - full;   // workgroups of the partly filled last round
        const bool can_chunk = EPI != EPI_GATE_BWD && (EPI == EPI_SWIGLU || EPI == EPI_SWIGLU_BWD || EPI == EPI_QKNORM || EPI == EPI_PLAIN);
        if (on && can_chunk && full > 0 && rest >= 8 * (int64_t)chunks && rest * 2 <= resident) {   // (at least one stripe group in the last round)
            const int64_t tg = (rest / chunks + 7) / 8;                 // stripe groups in the last round
            int tc = (int)(resident / (tg * 8));                         // chunks that fill the chip once
            if (tc > pairs) tc = pairs;
            const bool even_only = EPI == EPI_PLAIN;                     // plain epilogue: chunks in equal numbers of tile pairs
            while (tc > chunks && even_only && pairs % tc != 0) --tc;
            if (tc > chunks) { tail_groups = tg; tail_chunks = tc; }
        }
    }
    q.tail_local = (int)((groups - tail_groups) * chunks);
    q.tail_chunks = tail_chunks;
    const int64_t nwg = ((groups - tail_groups) * chunks + tail_groups * tail_chunks) * 8;
    hipLaunchKernelGGL((lin_rows_kernel<KC, EPI, NKH, NW>), dim3((unsigned)nwg), dim3(64 * NW), lds, s, q);
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}

static int rows_wide_wg() {   // VSDE_ROWS_NW=4: four-wave workgroups for every epilogue (A/B runs)
    static int v = -1;
    if (v < 0) v = (int)vsde_knob("VSDE_ROWS_NW", 8);
    return v;
}
template <int KC, int EPI, int NKH = 1>
static int launch_rows(const LinParams &p, hipStream_t s) {
    // Eight waves (one 256-row workgroup per CU) at K = 512 (two k-halves, 32-row waves for every epilogue) when M fills the chip:
    // synthetic step 146.0 -> 145.1 ms.  The QK-norm epilogue loses with them (242 -> 260 us training / 205 -> 221 no grad, whatever
    // the column-chunk count: its long epilogues want a second workgroup on the CU to hide behind), the SwiGLU backward at K = 256
    // (two four-wave workgroups per CU overlap better) loses 0.08 ms per LV step, the plain / SwiGLU-forward epilogues (64-row
    // waves, 512-row workgroups) 0.1-0.25 ms.  Until round 3 the gate-backward epilogue at K = 256 gained from them (109.5 -> 95 us).
    // Round 4: with the epilogue's per-lane addresses no longer spilled (opaque lane copy per tile) the gate-backward epilogue at
    // K = 256 is faster on two four-wave workgroups as well: 85 vs 96 us stand-alone, LV step 26.74 / 26.89 -> 26.60 / 26.67 ms.  (Its
    // eight waves in anti-phase -- waves 4..7 one phase behind waves 0..3, two barriers per tile, three tile buffers -- were tried
    // for every epilogue and lose 15-20 % to lockstep: waves released by one barrier skew by themselves, the first wave of a SIMD to
    // win the matrix pipe finishes its MFMAs early and runs its epilogue under the other one's.)
    if constexpr (NKH > 1 && EPI != EPI_QKNORM) {
        if (rows_wide_wg() == 8 && p.M >= 256 * 256) return launch_rows_nw<KC, EPI, NKH, 8>(p, s);
    }
    return launch_rows_nw<KC, EPI, NKH, 4>(p, s);
}

template <int EPI>
static int launch_rows_k(const LinParams &p, hipStream_t s) {
    if (p.K == 512) return launch_rows<256, EPI, 2>(p, s);   // two k-halves through the K = 256 tile buffers
    return p.K == 128 ? launch_rows<128, EPI>(p, s) : launch_rows<256, EPI>(p, s);
}

template <int NB>
static int launch_cols(const LinParams &p, hipStream_t s) {
    const size_t lds = cols_lds_bytes<NB>();
    VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)lin_cols_kernel<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t stripes = (p.M + C2_ROWS - 1) / C2_ROWS;
    hipLaunchKernelGGL((lin_cols_kernel<NB>), dim3((unsigned)(stripes * (p.N / (32 * NB)))), dim3(C2_THREADS), lds, s, p);
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}

#ifdef VSDE_ABLATIONS
static int launch_deep(const LinParams &p, hipStream_t s) {
    static int cus = 0;
    if (!cus) {
        int dev = 0; hipDeviceProp_t prop;
        VSDE_CHECK_HIP(hipGetDevice(&dev));
        VSDE_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const size_t lds = (size_t)DP_LDS_BYTES;
    VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)lin_deep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int64_t rows = (p.M + cus - 1) / cus;
    rows = (rows + 31) / 32 * 32;                      // contiguous row range per workgroup, whole 32-row blocks
    if (rows < 64) rows = 64;
    const int64_t wgs = (p.M + rows - 1) / rows;
    hipLaunchKernelGGL(lin_deep_kernel, dim3((unsigned)wgs), dim3(DP_THREADS), lds, s, p, (int)rows);
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}
#endif

// EPI_QKNORM: K = 256 (the LV / OU encoder width) and K = 128 (the width of the reference-generated parity fixtures)
static int launch_qknorm(const LinParams &p, hipStream_t s) {
    return p.K == 128 ? launch_rows<128, EPI_QKNORM>(p, s) : launch_rows<256, EPI_QKNORM>(p, s);
}

#ifdef VSDE_ABLATIONS
static bool deep_enabled() {
    static int v = -1;
    // off by default: 137 / 260 us at K = 768 / 1536 (M = 205 k) against 102 / 179 us of the tuned library solutions
    // (profiles/r03_deep_gemm_ablation.txt); VSDE_DEEP_GEMM=1 routes the deep reductions here (tests, A/B runs)
    if (v < 0) v = (int)vsde_knob("VSDE_DEEP_GEMM", 0);
    return v != 0;
}
#endif

// 1 = rows kernel, 2 = cols kernel, 3 = deep kernel (persistent, both operands through LDS), 0 = shape not covered
static int lin_variant(int64_t M, int N, int K, int epilogue) {
    // deep reductions with a narrow output at sizes that fill the chip (>= 128 rows per CU): the persistent kernel
#ifdef VSDE_ABLATIONS
    if (epilogue == EPI_PLAIN && deep_enabled() && K >= 512 && K % 64 == 0 && N % 256 == 0 && N <= K && M >= 32768) return 3;
#endif
    const bool rows_ok = (K == 128 || K == 256 || K == 512) && N % 64 == 0;   // K = 512: two k-halves per output tile
    const bool cols_ok = K % 64 == 0 && N % 128 == 0 && epilogue == EPI_PLAIN;
    if (epilogue != EPI_PLAIN) return rows_ok ? 1 : 0;
    // both fit: the rows kernel reads the activations once and suits wide outputs; the cols kernel suits deep reductions
    if (rows_ok && (!cols_ok || N >= K)) return 1;
    return cols_ok ? 2 : (rows_ok ? 1 : 0);
}

}  // namespace vsde

using namespace vsde;

extern "C" int vsde_linear_bf16_supported(int64_t M, int N, int K, int epilogue) {
    if (M <= 0 || N <= 0 || K <= 0 || epilogue < 0 || epilogue > 2) return 0;
    return lin_variant(M, N, K, epilogue);
}

extern "C" int vsde_linear_bf16(const void *x, int64_t ldx, const void *w, const void *bias, void *y, int64_t ldy, int64_t M, int N,
                                int K, int epilogue, void *s_out, int64_t lds, const void *u_in, int64_t ldu, void *stream) {
    VSDE_CHECK_ARG(x && w && M > 0 && N > 0 && K > 0, VSDE_E_BADARG, "bad linear arguments");
    VSDE_CHECK_ARG(epilogue >= 0 && epilogue <= 2, VSDE_E_BADARG, "unknown linear epilogue %d", epilogue);
    const int variant = lin_variant(M, N, K, epilogue);
    VSDE_CHECK_ARG(variant != 0, VSDE_E_BADARG, "linear shape N=%d K=%d epilogue=%d is not covered by the gfx950 kernels", N, K, epilogue);
    VSDE_CHECK_ARG(ldx >= K && ldx % 8 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0, VSDE_E_BADARG,
                   "linear operands must be 16-byte aligned with row pitches that are multiples of 8 elements");
    LinParams p = {};
    p.A = (const uint16_t *)x; p.lda = ldx; p.W = (const uint16_t *)w; p.bias = (const uint16_t *)bias;
    p.C = (uint16_t *)y; p.ldc = ldy; p.M = M; p.N = N; p.K = K;
    p.S = (uint16_t *)s_out; p.lds_ = lds; p.U = (const uint16_t *)u_in; p.ldu = ldu;
    { static int dbg = -1; if (dbg < 0) dbg = ablation_env("VSDE_LIN_DEBUG"); p.dbg = dbg; }
    hipStream_t st = (hipStream_t)stream;
    if (epilogue == EPI_PLAIN) {
        VSDE_CHECK_ARG(y && ldy >= N && ldy % 8 == 0 && ((uintptr_t)y % 16) == 0, VSDE_E_BADARG, "bad linear output");
        if (variant == 1) return launch_rows_k<EPI_PLAIN>(p, st);
#ifdef VSDE_ABLATIONS
        if (variant == 3) return launch_deep(p, st);
#endif
        return N % 256 == 0 ? launch_cols<8>(p, st) : launch_cols<4>(p, st);
    }
    if (epilogue == EPI_SWIGLU) {
        VSDE_CHECK_ARG(s_out && lds >= N / 2 && lds % 8 == 0 && ((uintptr_t)s_out % 16) == 0 && (!y || (ldy >= N && ldy % 8 == 0)),
                       VSDE_E_BADARG, "bad SwiGLU epilogue outputs");
        return launch_rows_k<EPI_SWIGLU>(p, st);
    }
    VSDE_CHECK_ARG(y && u_in && ldy >= 2 * N && ldu >= 2 * N && ldy % 8 == 0 && ldu % 8 == 0 && ((uintptr_t)y % 16) == 0 &&
                   ((uintptr_t)u_in % 16) == 0, VSDE_E_BADARG, "bad SwiGLU-backward epilogue buffers");
    return launch_rows_k<EPI_SWIGLU_BWD>(p, st);
}

extern "C" int vsde_linear_qknorm_bf16(const void *x, int64_t ldx, const void *w, const void *bias, int64_t M, int K, int heads,
                                       int gate_width, int tokens, const float *cosT, const float *sinT, const float *wq,
                                       const float *wk, const void *v0, const float *lam, double eps, void *q, void *k, void *v,
                                       void *gate, int64_t ldg, int gate_sigmoid, float *rinv, void *vdiff, void *stream) {
    VSDE_CHECK_ARG(x && w && q && k && v && cosT && sinT && wq && wk && M > 0 && heads > 0 && tokens > 0, VSDE_E_BADARG,
                   "bad linear_qknorm arguments");
    VSDE_CHECK_ARG((K == 256 || K == 128) && gate_width % 64 == 0 && gate_width >= 0 && (gate_width == 0 || (gate && ldg >= gate_width && ldg % 8 == 0)),
                   VSDE_E_BADARG, "linear_qknorm is built for K in {128, 256}, head_dim 64 and a gate block that is a multiple of 64 wide");
    VSDE_CHECK_ARG((!v0) == (!lam), VSDE_E_BADARG, "residual values and their mixing weight go together");
    VSDE_CHECK_ARG(ldx >= K && ldx % 8 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)q % 16) == 0 &&
                   ((uintptr_t)k % 16) == 0 && ((uintptr_t)v % 16) == 0 && ((uintptr_t)cosT % 16) == 0 && ((uintptr_t)sinT % 16) == 0,
                   VSDE_E_BADARG, "linear_qknorm operands must be 16-byte aligned");
    LinParams p = {};
    p.A = (const uint16_t *)x; p.lda = ldx; p.W = (const uint16_t *)w; p.bias = (const uint16_t *)bias;
    p.M = M; p.N = 3 * heads * 64 + gate_width; p.K = K;
    p.Qo = (uint16_t *)q; p.Ko = (uint16_t *)k; p.Vo = (uint16_t *)v; p.Go = (uint16_t *)gate; p.ldg = ldg;
    p.cosT = cosT; p.sinT = sinT; p.wq = wq; p.wk = wk; p.lam = lam; p.V0 = (const uint16_t *)v0;
    p.heads = heads; p.tokens = tokens; p.eps = (float)eps;
    VSDE_CHECK_ARG(!vdiff || (v0 && ((uintptr_t)vdiff % 16) == 0), VSDE_E_BADARG, "vdiff needs residual values (and 16-byte alignment)");
    p.Rinv = rinv; p.Vdiff = (uint16_t *)vdiff; p.gate_sigmoid = gate_sigmoid;
    return launch_qknorm(p, (hipStream_t)stream);
}

extern "C" int vsde_linear_gated_bf16(const void *attn, int64_t ldx, const void *gate, int64_t ldgate, const void *w, const void *bias,
                                      void *y, int64_t ldy, int64_t M, int N, int K, void *stream) {
    VSDE_CHECK_ARG(attn && gate && w && y && M > 0 && N > 0, VSDE_E_BADARG, "bad linear_gated arguments");
    VSDE_CHECK_ARG((K == 128 || K == 256) && N % 64 == 0, VSDE_E_BADARG, "linear_gated runs on the rows kernel: K in {128, 256}, N %% 64 == 0");
    VSDE_CHECK_ARG(ldx >= K && ldx % 8 == 0 && ldgate >= 64 && ldgate % 8 == 0 && ldy >= N && ldy % 8 == 0 && ((uintptr_t)attn % 16) == 0 &&
                   ((uintptr_t)gate % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)y % 16) == 0, VSDE_E_BADARG,
                   "linear_gated operands must be 16-byte aligned with row pitches that are multiples of 8 elements");
    LinParams p = {};
    p.A = (const uint16_t *)attn; p.lda = ldx; p.W = (const uint16_t *)w; p.bias = (const uint16_t *)bias;
    p.C = (uint16_t *)y; p.ldc = ldy; p.M = M; p.N = N; p.K = K; p.Gate = (const uint16_t *)gate; p.ldgate = ldgate;
    return launch_rows_k<EPI_PLAIN>(p, (hipStream_t)stream);
}

extern "C" int vsde_linear_gate_bwd_bf16(const void *dy, int64_t ldy, const void *w_t, const void *og, const void *s, int64_t lds,
                                         void *dattn, void *dgate, int64_t ldd, float *delta, int64_t M, int K, int heads, int tokens,
                                         void *stream) {
    VSDE_CHECK_ARG(dy && w_t && og && s && dattn && dgate && delta && M > 0 && heads > 0 && tokens > 0, VSDE_E_BADARG, "bad linear_gate_bwd arguments");
    VSDE_CHECK_ARG(K == 128 || K == 256, VSDE_E_BADARG, "linear_gate_bwd runs on the rows kernel: K in {128, 256}");
    VSDE_CHECK_ARG(ldy >= K && ldy % 8 == 0 && lds >= 64 && lds % 8 == 0 && ldd >= 64 && ldd % 8 == 0 && ((uintptr_t)dy % 16) == 0 &&
                   ((uintptr_t)w_t % 16) == 0 && ((uintptr_t)og % 16) == 0 && ((uintptr_t)s % 16) == 0 && ((uintptr_t)dattn % 16) == 0 &&
                   ((uintptr_t)dgate % 16) == 0, VSDE_E_BADARG,
                   "linear_gate_bwd operands must be 16-byte aligned with row pitches that are multiples of 8 elements");
    LinParams p = {};
    p.A = (const uint16_t *)dy; p.lda = ldy; p.W = (const uint16_t *)w_t; p.bias = nullptr;
    p.C = (uint16_t *)dattn; p.ldc = (int64_t)heads * 64; p.M = M; p.N = heads * 64; p.K = K;
    p.Og = (const uint16_t *)og; p.Sg = (const uint16_t *)s; p.ldsg = lds; p.Dgate = (uint16_t *)dgate; p.lddg = ldd; p.Delta = delta;
    p.heads = heads; p.tokens = tokens;
    return K == 128 ? launch_rows<128, EPI_GATE_BWD>(p, (hipStream_t)stream) : launch_rows<256, EPI_GATE_BWD>(p, (hipStream_t)stream);
}
