// Weight / bias gradients of the encoder's Linear layers on bf16 MFMA (gfx950).
//
//   dW[n][k] = sum_m dY[m][n] * X[m][k]        db[n] = sum_m dY[m][n]          (M = B * tokens ~ 2e5)
//
// hipBLASLt runs these "reduce over a huge M into a small N x K" GEMMs at 15-270 TF/s (0.45-0.6 ms each at the LV shapes
// whatever their size) and torch adds a separate bf16 column-sum kernel for the bias.  They are memory-bound (dY and X are
// read once from HBM, X once more per output tile row from L2), so the kernel is organised around the loads:
//
//   * M is split over the grid in interleaved 32-row blocks (split s owns blocks s, s + nsplit, ...), every workgroup keeps
//     a TN x 256 fp32 output tile in MFMA accumulators (v_mfma_f32_32x32x16_bf16; a wave owns 64 x 128) and writes it as a
//     partial; a second kernel sums the partials in a fixed order (deterministic; fp32 results).
//   * dy and x rows are staged exactly as they sit in memory ([m][cols], 32 rows per step, two LDS buffers, one LDS-only barrier
//     per step); the MFMA fragments -- which want the reduction index m contiguous per lane -- come out of LDS through
//     ds_read_b64_tr_b16, so the transpose is free.  Row pitch = cols + 32 bf16 (64 bytes past a multiple of 256): the 4 rows
//     x 2 column halves one transposing read touches per 32 lanes fall in 8 different 8-bank groups.
//   * Loads are requested TWO steps ahead into two register sets, branch-free (clamped addresses, zeroing at the LDS store):
//     a predicated load sits in its own exec branch and hipcc then drains vmcnt(0) around each of them.
//   * The workgroups of one split (same rows, different output tiles) sit on one XCD, so the shared X rows are served by that
//     XCD's L2.  TN = 128: 4 waves, two workgroups per CU; TN = 256: 8 waves, one workgroup per CU, half the X re-reads.
//   * The column sums of dY (bias gradient) are accumulated from the staging registers on the way to LDS.
#include <math.h>
#include <stdlib.h>

#include "vsde_common.h"

namespace vsde {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t w2u4 __attribute__((ext_vector_type(4)));
typedef short w2bf4 __attribute__((ext_vector_type(4)));

constexpr int W2_TK = 256, W2_BM = 32;
constexpr int W2_LDB = W2_TK + 32;   // LDS row pitch of the x tile (bf16)

template <int TN, int TH = 2 * TN> struct W2 {
    static constexpr int THREADS = TH;                  // waves: (TH / 128) along n x 2 along k; a wave owns (TN / (TH / 128)) x 128 of the tile:
                                                        // 64 x 128 (TH = 2 TN) or, TN = 256 on FOUR waves, 128 x 128 (see wgrad_tr_body)
    static constexpr int AB = TN / (TH / 128) / 32;     // 32-row blocks of the wave's tile along n (2 | 4)
    static constexpr int LDA = TN + 32;                  // LDS row pitch of the dy tile
    static constexpr int BUF = W2_BM * (LDA + W2_LDB);   // bf16 elements per buffer
    static constexpr int PART = TN * W2_TK + TN;         // fp32 partial tile + bias row
    static constexpr int YC = TN / 8;                    // 16-byte chunks per dy tile row
    static constexpr int YR = THREADS / YC;              // dy rows per staging pass (16)
    static constexpr int NY = W2_BM / YR;                // dy chunks per thread and step (2)
    static constexpr int XR = THREADS / 32;              // x rows per staging pass (8 | 16)
    static constexpr int NX = W2_BM / XR;                // x chunks per thread and step (4 | 2)
};

struct Wgrad2Params {
    const uint16_t *dy, *x;
    int64_t M;
    int N, K, tn, tiles_n, tiles_k, tiles, nsplit;
    int64_t chunks;           // 32-row blocks of the reduction index
    float *partial, *dW, *db;
    const int32_t *row_map;   // output row of product row n (dW[row_map[n]], db[row_map[n]]; < 0: dropped), or nullptr = identity
    long long *trace;         // debugging (vsde_wgrad_debug_trace): per-wave phase cycle sums of workgroup 0, [waves][4] + step count
};

__device__ __forceinline__ uint2 w2_read_tr(const uint16_t *ptr) {
    w2bf4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) w2bf4 *)ptr);
    return *(uint2 *)&r;
}
__device__ __forceinline__ void w2_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// fragment "row index = column c0 + (lane & 31) of the tile, 8 reduction rows" for the 16-row half `mc` of a 32-row tile
__device__ __forceinline__ bf16x8 w2_frag(const uint16_t *tile, int ld, int c0, int mc, int lane) {
    const int h2 = lane >> 5, m = lane & 15;
    const uint16_t *src = tile + (16 * mc + 4 * h2 + (m >> 2)) * ld + c0 + ((lane >> 4) & 1) * 16 + (m & 3) * 4;
    const uint2 a0 = w2_read_tr(src), a1 = w2_read_tr(src + 8 * ld);
    uint4 w = make_uint4(a0.x, a0.y, a1.x, a1.y);
    return *(bf16x8 *)&w;
}

// Several problems in ONE launch (the OU example's 12.9 k tokens: every weight gradient of the step is a ~20 us kernel + a ~10 us
// reduction, 25 of each per step; grouped, the small problems fill the chip together).  Workgroup ids [first[g], first[g + 1]) belong
// to problem g (counts are multiples of 8: the XCD of a workgroup id is unchanged), tiles [tile0[g], tile0[g + 1]) in the reduction.
constexpr int W2_MAXG = 24;
struct Wgrad2Group {
    Wgrad2Params g[W2_MAXG];
    int first[W2_MAXG + 1], tile0[W2_MAXG + 1];
    int n;
};

// TH = 256 with TN = 256 (round 5, opt-in, slower -- see wgrad2_four_waves): four waves, one per SIMD, each with a 128 x 128 block = 4 x 4
// MFMA tiles in 256 accumulator registers; a step's fragments are (AB + 4) KB per wave and 16-row half for 4 AB MFMAs: 0.75 KB per MFMA at
// AB = 2, 0.5 KB at AB = 4.
template <int TN, int TH = 2 * TN, bool TRACE = false>
__device__ __forceinline__ void wgrad_tr_body(const Wgrad2Params &p, const int bid) {
    using C = W2<TN, TH>;
    extern __shared__ __attribute__((aligned(16))) uint16_t w2s[];
    __shared__ float bred[C::YR][TN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // consecutive workgroup ids go round-robin over the 8 XCDs, so id = 8 * local + xcd
    const int xcd = bid & 7, local = bid >> 3;
    const int tile = local % p.tiles, split = (local / p.tiles) * 8 + xcd;
    if (split >= p.nsplit) return;
    const int n_blk = (tile / p.tiles_k) * TN, k_blk = (tile % p.tiles_k) * W2_TK;
    const bool want_bias = p.db != nullptr && k_blk == 0;
    // Split s owns the 32-row blocks s, s + nsplit, s + 2 nsplit, ...: at any moment the resident workgroups read one contiguous
    // window of rows.
    const int64_t m_end = p.M, m_last = p.M - 1;
#define W2_ROW0(t_) (((int64_t)(t_) * p.nsplit + split) * W2_BM)
    const int ya_c = tid % C::YC, ya_r = tid / C::YC;   // + YR rows per i
    const int xb_c = tid & 31, xb_r = tid >> 5;         // + XR rows per i
    const bool ya_ok = n_blk + ya_c * 8 < p.N, xb_ok = k_blk + xb_c * 8 < p.K;   // N, K multiples of 8
    const uint16_t *ysrc = p.dy + (ya_ok ? n_blk + ya_c * 8 : 0), *xsrc = p.x + (xb_ok ? k_blk + xb_c * 8 : 0);
    // NSET register sets: step t travels in set t % NSET, requested NSET steps before its LDS store.  Four sets for the eight-wave
    // TN = 256 form (round 5): the phase stamps (tools/wgrad_trace.py) showed ~1,000 of a step's 2,530 cycles in the LDS store waiting
    // for loads requested two steps = 5,000 cycles earlier -- at 32 KB per workgroup and step the requests of two steps are all a CU
    // has in flight, and HBM under this load answers in ~2.4 us.  (The four-wave forms have no registers left for it.)
    constexpr int NSET = (TN == 256 && TH == 512) ? 4 : 2;
    w2u4 ry[NSET][C::NY], rx[NSET][C::NX];
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#define W2_FETCH(set_, m0_)                                                                                              \
    do {                                                                                                                 \
        _Pragma("unroll") for (int i = 0; i < C::NY; ++i) {                                                              \
            int64_t m = (m0_) + ya_r + C::YR * i; m = m < m_last ? m : m_last; ry[set_][i] = *(const w2u4 *)(ysrc + m * p.N); }  \
        _Pragma("unroll") for (int i = 0; i < C::NX; ++i) {                                                              \
            int64_t m = (m0_) + xb_r + C::XR * i; m = m < m_last ? m : m_last; rx[set_][i] = *(const w2u4 *)(xsrc + m * p.K); }  \
    } while (0)
#define W2_COMMIT(set_, buf_, m0_)                                                                                       \
    do {                                                                                                                 \
        uint16_t *ay_ = (buf_), *bx_ = (buf_) + W2_BM * C::LDA;                                                          \
        const w2u4 z = {0u, 0u, 0u, 0u};                                                                                 \
        _Pragma("unroll") for (int i = 0; i < C::NY; ++i) {                                                              \
            if (!(ya_ok && (m0_) + ya_r + C::YR * i < m_end)) ry[set_][i] = z;                                           \
            *(w2u4 *)(ay_ + (ya_r + C::YR * i) * C::LDA + ya_c * 8) = ry[set_][i]; }                                     \
        _Pragma("unroll") for (int i = 0; i < C::NX; ++i) {                                                              \
            if (!(xb_ok && (m0_) + xb_r + C::XR * i < m_end)) rx[set_][i] = z;                                           \
            *(w2u4 *)(bx_ + (xb_r + C::XR * i) * W2_LDB + xb_c * 8) = rx[set_][i]; }                                     \
        if (want_bias) {                                                                                                 \
            _Pragma("unroll") for (int i = 0; i < C::NY; ++i)                                                            \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                          \
                    const uint32_t w = ry[set_][i][j];                                                                   \
                    bsum[2 * j] += __uint_as_float(w << 16); bsum[2 * j + 1] += __uint_as_float(w & 0xffff0000u);        \
                }                                                                                                        \
        }                                                                                                                \
    } while (0)
    constexpr int AB = C::AB;
    f32x16 acc[AB][4];
#pragma unroll
    for (int a = 0; a < AB; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    const int wn = (wave >> 1) * (32 * AB), wk = (wave & 1) * 128;
    const int nsteps = (int)((p.chunks - split + p.nsplit - 1) / p.nsplit);
// step s_: MFMAs out of LDS buffer PAR_, then step s + 1 (register set SET_ = (s + 1) % NSET) goes to the other buffer and that set is
// re-requested for step s + 1 + NSET
    long long ph[4] = {0, 0, 0, 0}, last_ = 0;
    if constexpr (TRACE) last_ = __builtin_readcyclecounter();
#define W2_STAMP(k_) do { if constexpr (TRACE) { const long long now_ = __builtin_readcyclecounter(); ph[k_] += now_ - last_; last_ = now_; } } while (0)
#define W2_BODY(s_, PAR_, SET_)                                                                                              \
    do {                                                                                                                 \
        const uint16_t *buf = w2s + (PAR_) * C::BUF, *ay = buf, *bx = buf + W2_BM * C::LDA;                              \
        _Pragma("unroll") for (int mc = 0; mc < 2; ++mc) {                                                               \
            bf16x8 af[AB], bf[4];                                                                                        \
            _Pragma("unroll") for (int a = 0; a < AB; ++a) af[a] = w2_frag(ay, C::LDA, wn + 32 * a, mc, lane);           \
            _Pragma("unroll") for (int b = 0; b < 4; ++b) bf[b] = w2_frag(bx, W2_LDB, wk + 32 * b, mc, lane);            \
            _Pragma("unroll") for (int a = 0; a < AB; ++a)                                                               \
                _Pragma("unroll") for (int b = 0; b < 4; ++b)                                                            \
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);               \
        }                                                                                                                \
        W2_STAMP(0);                                                                                                     \
        if ((s_) + 1 < nsteps) W2_COMMIT(SET_, w2s + (1 - (PAR_)) * C::BUF, W2_ROW0((s_) + 1));                          \
        W2_STAMP(1);                                                                                                     \
        w2_barrier();                                                                                                    \
        W2_STAMP(2);                                                                                                     \
        W2_FETCH(SET_, W2_ROW0((s_) + 1 + NSET));                                                                        \
        W2_STAMP(3);                                                                                                     \
    } while (0)
    if (nsteps > 0) {
        W2_FETCH(0, W2_ROW0(0));
        W2_COMMIT(0, w2s, W2_ROW0(0));
        w2_barrier();
        if constexpr (NSET == 4) { W2_FETCH(1, W2_ROW0(1)); W2_FETCH(2, W2_ROW0(2)); W2_FETCH(3, W2_ROW0(3)); W2_FETCH(0, W2_ROW0(4)); }
        else { W2_FETCH(1, W2_ROW0(1)); W2_FETCH(0, W2_ROW0(2)); }
    }
    // no condition on the later steps inside the loop: a merge of "taken / not taken" makes hipcc's vmcnt bookkeeping fall back to a
    // full drain at the first step
    int s = 0;
    if constexpr (NSET == 4) {
        for (; s + 3 < nsteps; s += 4) {
            W2_BODY(s, 0, 1);
            W2_BODY(s + 1, 1, 2);
            W2_BODY(s + 2, 0, 3);
            W2_BODY(s + 3, 1, 0);
        }
        if (s < nsteps) W2_BODY(s, 0, 1);
        if (s + 1 < nsteps) W2_BODY(s + 1, 1, 2);
        if (s + 2 < nsteps) W2_BODY(s + 2, 0, 3);
    } else {
        for (; s + 1 < nsteps; s += 2) {
            W2_BODY(s, 0, 1);
            W2_BODY(s + 1, 1, 0);
        }
        if (s < nsteps) W2_BODY(s, 0, 1);
    }
    if constexpr (TRACE) {
        if (bid == 0 && lane == 0 && p.trace) {
#pragma unroll
            for (int k = 0; k < 4; ++k) p.trace[wave * 5 + k] = ph[k];
            p.trace[wave * 5 + 4] = nsteps;
        }
    }
#undef W2_STAMP
#undef W2_BODY
#undef W2_COMMIT
#undef W2_FETCH
#undef W2_ROW0
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float *out = p.partial + ((int64_t)tile * p.nsplit + split) * C::PART;
    const int fr = lane & 31, fh = lane >> 5;
#pragma unroll
    for (int a = 0; a < AB; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = wn + 32 * a + (e & 3) + 8 * (e >> 2) + 4 * fh, col = wk + 32 * b + fr;
                if (n_blk + wn + 32 * a < p.N) out[row * W2_TK + col] = acc[a][b][e];   // (whole 32-row blocks past N: never read back)
            }
    if (want_bias) {   // YR threads share a column chunk
#pragma unroll
        for (int j = 0; j < 8; ++j) bred[ya_r][ya_c * 8 + j] = bsum[j];
        __syncthreads();
        if (tid < TN) {
            float t = 0.f;
#pragma unroll
            for (int g = 0; g < C::YR; ++g) t += bred[g][tid];
            out[TN * W2_TK + tid] = t;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Round 5: the same product with the operand rows brought in by LDS-DMA (global_load_lds_dwordx4) instead of registers + ds_write.
// The phase stamps of wgrad_tr_body (tools/wgrad_trace.py, dW[1408,256]) read, per 32-row step of 2,530 cycles: fragment reads + MFMAs
// 540 / 940 (first / second wave of a SIMD), staging registers -> LDS ~1,000 -- whether the loads were requested two or four steps ahead:
// it is the 4 x ds_write_b128 + zero-masking + bias adds of sixteen waves in lockstep, not load latency --, barrier 170, issue of the next
// loads 200-470.  Only the first item is work.  Here a step's 36 KB ([32 rows][256 + 32] of dy, then of x: the layout the transposing
// fragment reads want) arrive as 36 pieces of 1 KB, five DMA instructions per wave (the last four of the forty are dummies, so that
// every wave's vmcnt counts the same), three steps ahead into a ring of four LDS buffers; no staging registers, no ds_write, no masking:
//   * a lane's 16 bytes land at piece * 1024 + 16 * lane, so the lane computes which (row, 16-byte chunk) that is -- chunks 32..35 of a
//     row are the padding: the lane loads chunk 0 again, nobody reads it;
//   * columns past N / K load column 0 of their row instead of zeros: they only reach output rows / columns that are never read back;
//   * rows past M cannot be zeroed this way: the kernel takes M % 32 == 0 only (the encoder's B x tokens at the benchmark shapes);
//   * the bias column sums come from the dy fragments the MFMAs use anyway (v_dot2c_f32_bf16 against a pair of ones).
template <int TN>
__device__ __forceinline__ void wgrad_dma_body(const Wgrad2Params &p, const int bid) {
    static_assert(TN == 256, "DMA staging: the eight-wave 256 x 256 form");
    using C = W2<TN, 512>;
    constexpr int NBUF = 4, TILE_BYTES = W2_BM * C::LDA * 2, PIECES = 2 * TILE_BYTES / 1024, PW = (PIECES + 7) / 8;
    static_assert(C::LDA == W2_LDB && 2 * TILE_BYTES == C::BUF * 2 && 2 * TILE_BYTES % 1024 == 0, "one buffer = whole 1 KB pieces");
    extern __shared__ __attribute__((aligned(16))) uint16_t w2s[];   // NBUF buffers + 1 KB for the dummy pieces
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = bid & 7, local = bid >> 3;
    const int tile = local % p.tiles, split = (local / p.tiles) * 8 + xcd;
    if (split >= p.nsplit) return;
    const int n_blk = (tile / p.tiles_k) * TN, k_blk = (tile % p.tiles_k) * W2_TK;
    const bool want_bias = p.db != nullptr && k_blk == 0;
    const int nsteps = (int)((p.chunks - split + p.nsplit - 1) / p.nsplit);
    // this lane's source of piece wave + 8 i at step 0, and the per-step advance of that operand (elements)
    const uint16_t *src[PW];
    int64_t adv[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int q = wave + 8 * i, o = (q < PIECES ? q : 0) * 1024 + 16 * lane;
        const int op = o >= TILE_BYTES, o2 = o - op * TILE_BYTES, row = o2 / (C::LDA * 2), cb = (o2 - row * (C::LDA * 2)) / 16;
        const int ld = op ? p.K : p.N, col0 = op ? k_blk : n_blk, lim = op ? p.K : p.N;
        const int col = (cb < TN / 8 && col0 + cb * 8 < lim) ? col0 + cb * 8 : col0;   // padding chunk / past the operand's width: any valid column
        src[i] = (op ? p.x : p.dy) + ((int64_t)split * W2_BM + row) * ld + col;
        adv[i] = (int64_t)p.nsplit * W2_BM * ld;
    }
    char *ring = (char *)w2s;
    auto dma = [&](int t) {   // step t (clamped: the last steps re-request the last block into buffers nobody reads)
        const int tc = t < nsteps ? t : nsteps - 1;
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int q = wave + 8 * i;
            char *dst = q < PIECES ? ring + (t & (NBUF - 1)) * (2 * TILE_BYTES) + q * 1024 : ring + NBUF * 2 * TILE_BYTES;
            __builtin_amdgcn_global_load_lds((const void *)(src[i] + tc * adv[i]), (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        }
    };
    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    float bsum[2] = {0.f, 0.f};
    const int wn = (wave >> 1) * 64, wk = (wave & 1) * 128;
    const bool bias_wave = want_bias && (wave & 1) == 0;
    if (nsteps > 0) {
        dma(0); dma(1); dma(2);
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * PW) : "memory");   // step 0 has landed (everybody's pieces)
    }
    for (int s = 0; s < nsteps; ++s) {
        const uint16_t *buf = (const uint16_t *)(ring + (s & (NBUF - 1)) * (2 * TILE_BYTES)), *ay = buf, *bx = buf + W2_BM * C::LDA;
#pragma unroll
        for (int mc = 0; mc < 2; ++mc) {
            bf16x8 af[2], bf[4];
#pragma unroll
            for (int a = 0; a < 2; ++a) af[a] = w2_frag(ay, C::LDA, wn + 32 * a, mc, lane);
#pragma unroll
            for (int b = 0; b < 4; ++b) bf[b] = w2_frag(bx, W2_LDB, wk + 32 * b, mc, lane);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
            if (bias_wave) {   // wave-uniform
                const uint32_t ones = 0x3f803f80u;
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const w2u4 w = __builtin_bit_cast(w2u4, af[a]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(bsum[a]) : "v"(w[j]), "v"(ones));
                }
            }
        }
        dma(s + 3);   // into the buffer of step s - 1: everyone left it at the previous barrier.  (Issued BEHIND this step's fragment reads:
                      // in front of them hipcc orders the reads after the DMA -- s_waitcnt vmcnt(0) -- since it cannot tell the buffers apart.)
        // step s + 1 (requested at the end of step s - 2) has landed once at most the pieces of steps s + 2 and s + 3 are in flight;
        // everyone is done reading this step's buffer (lgkmcnt) before the next trip's DMA overwrites the one of step s - 1 ... s
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * PW) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing (clamped) requests must not outlive the workgroup's LDS
    float *out = p.partial + ((int64_t)tile * p.nsplit + split) * C::PART;
    const int fr = lane & 31, fh = lane >> 5;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = wn + 32 * a + (e & 3) + 8 * (e >> 2) + 4 * fh, col = wk + 32 * b + fr;
                if (n_blk + wn + 32 * a < p.N) out[row * W2_TK + col] = acc[a][b][e];
            }
    if (bias_wave) {   // lanes l and l + 32 hold the two halves of every 16-row k-step: one sum per column
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const uint32_t u = __float_as_uint(bsum[a]);
            const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            const float t = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
            if (fh == 0) out[TN * W2_TK + wn + 32 * a + fr] = t;
        }
    }
}
template <int TN>
__global__ void __launch_bounds__(512, 1) wgrad_dma_kernel(Wgrad2Params p) { wgrad_dma_body<TN>(p, (int)blockIdx.x); }

template <int TN, int TH = 2 * TN, bool TRACE = false>
__global__ void __launch_bounds__(TH, TN == 128 ? 2 : 1) wgrad_tr_kernel(Wgrad2Params p) { wgrad_tr_body<TN, TH, TRACE>(p, (int)blockIdx.x); }

template <int TN, int TH = 2 * TN>
__global__ void __launch_bounds__(TH, TN == 128 ? 2 : 1) wgrad_tr_group_kernel(Wgrad2Group G) {
    int g = 0;
    while (g + 1 < G.n && (int)blockIdx.x >= G.first[g + 1]) ++g;   // workgroup-uniform
    wgrad_tr_body<TN, TH>(G.g[g], (int)blockIdx.x - G.first[g]);
}


// Fixed-order sum of the split partials.  grid (TN * 256 / 4 / 64 + 1, tiles), 256 threads = 64 float4 columns x 4 groups of
// splits: group g adds splits g, g + 4, ... with four loads in flight, the four group sums are combined through LDS in order.  (One
// thread per float4 walking all splits leaves ~64 workgroups on the chip for a two-tile problem and is latency-bound: 57 us against
// 50 us for the product itself.)  The extra block (last blockIdx.x) sums the tile's bias row.
template <int TN>
__device__ __forceinline__ void wgrad_tr_reduce_body(const Wgrad2Params &p, const int tile) {
    constexpr int PART = W2<TN>::PART;
    __shared__ float4 comb[3][64];
    __shared__ float bcomb[256];
    const int n_blk = (tile / p.tiles_k) * TN, k_blk = (tile % p.tiles_k) * W2_TK;
    const float *src = p.partial + (int64_t)tile * p.nsplit * PART;
    if (blockIdx.x == gridDim.x - 1) {
        if (p.db == nullptr || k_blk != 0) return;
        constexpr int H = 256 / TN;   // groups of splits per column (2 | 1)
        const int c = threadIdx.x % TN, h = threadIdx.x / TN;
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        int sp = h;
        for (; sp + 3 * H < p.nsplit; sp += 4 * H) {
#pragma unroll
            for (int q = 0; q < 4; ++q) t[q] += src[(int64_t)(sp + H * q) * PART + TN * W2_TK + c];
        }
        for (; sp < p.nsplit; sp += H) t[0] += src[(int64_t)sp * PART + TN * W2_TK + c];
        const float sum = (t[0] + t[1]) + (t[2] + t[3]);
        bcomb[threadIdx.x] = sum;
        __syncthreads();
        if (h == 0 && n_blk + c < p.N) {
            const int r = p.row_map ? p.row_map[n_blk + c] : n_blk + c;
            if (r >= 0) p.db[r] = H == 2 ? sum + bcomb[(threadIdx.x + TN) & 255] : sum;
        }
        return;
    }
    const int col = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int e = (blockIdx.x * 64 + col) * 4;
    const int n = n_blk + e / W2_TK, k = k_blk + e % W2_TK;
    if (n >= p.N) return;   // a workgroup is one output row (64 columns x 4 = W2_TK): rows past N of the last tile are not summed at all
    float4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    int sp = g;
    for (; sp + 12 < p.nsplit; sp += 16) {   // 4 loads in flight, fixed association
        float4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = *(const float4 *)(src + (int64_t)(sp + 4 * q) * PART + e);
#pragma unroll
        for (int q = 0; q < 4; ++q) { acc[q].x += v[q].x; acc[q].y += v[q].y; acc[q].z += v[q].z; acc[q].w += v[q].w; }
    }
    for (; sp < p.nsplit; sp += 4) {
        const float4 v = *(const float4 *)(src + (int64_t)sp * PART + e);
        acc[0].x += v.x; acc[0].y += v.y; acc[0].z += v.z; acc[0].w += v.w;
    }
    float4 t;
    t.x = (acc[0].x + acc[1].x) + (acc[2].x + acc[3].x); t.y = (acc[0].y + acc[1].y) + (acc[2].y + acc[3].y);
    t.z = (acc[0].z + acc[1].z) + (acc[2].z + acc[3].z); t.w = (acc[0].w + acc[1].w) + (acc[2].w + acc[3].w);
    if (g > 0) comb[g - 1][col] = t;
    __syncthreads();
    if (g == 0) {
#pragma unroll
        for (int q = 0; q < 3; ++q) { const float4 u = comb[q][col]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
        if (n < p.N && k < p.K) {   // K % 8 == 0: all four or none
            const int r = p.row_map ? p.row_map[n] : n;
            if (r >= 0) *(float4 *)(p.dW + (int64_t)r * p.K + k) = t;
        }
    }
}

template <int TN>
__global__ void __launch_bounds__(256) wgrad_tr_reduce_kernel(Wgrad2Params p) { wgrad_tr_reduce_body<TN>(p, (int)blockIdx.y); }

template <int TN>
__global__ void __launch_bounds__(256) wgrad_tr_reduce_group_kernel(Wgrad2Group G) {
    int g = 0;
    while (g + 1 < G.n && (int)blockIdx.y >= G.tile0[g + 1]) ++g;
    wgrad_tr_reduce_body<TN>(G.g[g], (int)blockIdx.y - G.tile0[g]);
}


// TN = 256 when the output has at least three 256 x 256 tiles (half the re-reads of X; measured 155 | 227 | 126 us against
// 172 | 264 | 136 us for dW[832,256] | [1536,256] | [256,768] at M = 205k, and 94 against 77 us for the one-tile [256,256]:
// VSDE_WGRAD_TN=128|256 forces one for such comparisons).  The grid is one round of resident
// workgroups (512 four-wave or 256 eight-wave workgroups), whole splits per XCD.  Problems of one or two tiles run 256 workgroups:
// their partial-tile traffic (131 KB per workgroup) is otherwise as large as the operands.
static void wgrad2_plan(int64_t M, int N, int K, Wgrad2Params &p) {
    static int force_tn = -1;
    if (force_tn < 0) force_tn = (int)vsde_knob("VSDE_WGRAD_TN", 0);
    p.M = M; p.N = N; p.K = K;
    p.tn = force_tn ? force_tn : (((N + 255) / 256) * ((K + W2_TK - 1) / W2_TK) >= 3 ? 256 : 128);
    p.tiles_n = (N + p.tn - 1) / p.tn; p.tiles_k = (K + W2_TK - 1) / W2_TK; p.tiles = p.tiles_n * p.tiles_k;
    const int64_t chunks = (M + W2_BM - 1) / W2_BM;
    const int wgs = p.tn == 256 || p.tiles <= 2 ? 256 : 512;
    int64_t nsplit = (wgs / p.tiles) & ~7;
    if (nsplit < 8) nsplit = 8;
    if (nsplit > 256) nsplit = 256;
    if (nsplit > chunks) nsplit = chunks;
    // One round of workgroups can leave the chip badly filled when the tile count does not divide it (dW[2816, 512]: 22 tiles x 8
    // splits = 176 workgroups on 256 CUs).  Then finer splits in several rounds: rounds x (rows per split), plus the partial
    // tiles' share of the traffic (each workgroup writes, and the reduction reads, one fp32 tile against (TN + 256) x 2 bytes per row).
    if (p.tiles >= 3 && p.tiles * nsplit < (wgs * 85) / 100 && !vsde_knob("VSDE_WGRAD_ONE_ROUND", 0)) {
        double best = 1e30; int64_t pick = nsplit;
        for (int64_t ns = 8; ns <= 64 && ns <= chunks; ns += 8) {
            const double rounds = (double)((p.tiles * ns + wgs - 1) / wgs);
            const double partial = (double)p.tn * W2_TK * 8.0 / ((double)(p.tn + W2_TK) * 2.0 * 32.0 * (double)chunks / (double)ns);
            const double cost = rounds / (double)ns * (1.0 + partial);
            if (cost < best * 0.98) { best = cost; pick = ns; }
        }
        nsplit = pick;
    }
    {
        // Few rows (the OU example: 12.9 k tokens = 404 row blocks): with the split count of a full round every workgroup has a handful of
        // steps and the partial tiles (tiles x nsplit x 263 KB) outweigh the operands -- the fixed-order reduction then costs as much as
        // the product (17 us each).  Product time ~ a + b chunks / nsplit, reduction ~ c nsplit: nsplit ~ sqrt(2.5 chunks), measured
        // on the OU step: 8 | 16 | 24 | 32 | 48 | 64 splits = 5.16 | 4.59 | 4.44 | 4.40 | 4.54 | 4.59 ms.  (At LV sizes the cap is above 64.)
        int64_t cap = ((int64_t)sqrt(2.5 * (double)chunks) + 4) & ~(int64_t)7;
        if (cap < 8) cap = 8;
        if (nsplit > cap) nsplit = cap;
    }
    {
        static int force_ns = -1;   // VSDE_WGRAD_NSPLIT: split count for every problem (A/B runs)
        if (force_ns < 0) force_ns = (int)vsde_knob("VSDE_WGRAD_NSPLIT", 0);
        if (force_ns > 0) { nsplit = force_ns; if (nsplit > chunks) nsplit = chunks; }
    }
    p.nsplit = (int)nsplit; p.chunks = chunks;
}
static size_t wgrad2_workspace(const Wgrad2Params &p) {
    return (size_t)p.tiles * p.nsplit * (p.tn == 256 ? W2<256>::PART : W2<128>::PART) * sizeof(float);
}

// TN = 256: eight waves with 64 x 128 blocks (default), or four with 128 x 128 (VSDE_WGRAD_WAVES=4; A/B runs).  Measured, round 5, LV
// shapes, one box: four waves 194 | 275 | 143 us against 166 | 234 | 133 us for dW[832,256] | [1408,256] | [256,704] -- a third fewer LDS
// bytes per MFMA does not pay for one wave per SIMD having nobody to hide its LDS latency behind.
#ifdef VSDE_ABLATIONS
static bool wgrad2_four_waves() {
    static int w = -1;
    if (w < 0) w = vsde_knob("VSDE_WGRAD_WAVES", 8) == 4 ? 4 : 8;
    return w == 4;
}
#define VSDE_FOUR_WAVES() wgrad2_four_waves()
#define VSDE_FOUR_WAVES_GROUP(G_, n_, s_) wgrad2_launch_group<256, 256>(G_, n_, s_)
#else
#define VSDE_FOUR_WAVES() false
#define VSDE_FOUR_WAVES_GROUP(G_, n_, s_) 0
#endif
static long long *g_wgrad_trace = nullptr;
template <int TN, int TH = 2 * TN> static int wgrad2_launch(const Wgrad2Params &p_, hipStream_t s) {
    using C = W2<TN, TH>;
    const size_t lds = (size_t)2 * C::BUF * sizeof(uint16_t);
    Wgrad2Params p = p_;
    p.trace = g_wgrad_trace;
#ifdef VSDE_ABLATIONS
    auto kern = (g_wgrad_trace && TN == 256 && TH == 512) ? wgrad_tr_kernel<TN, TH, (TN == 256 && TH == 512)> : wgrad_tr_kernel<TN, TH>;
#else
    auto kern = wgrad_tr_kernel<TN, TH>;
#endif
    // VSDE_WGRAD_DMA=1: the LDS-DMA form (wgrad_dma_kernel; A/B runs).  Opt-in: measured 184 | 251 | 143 us against 169 | 232 | 131 us for
    // dW[832,256] | [1408,256] | [256,704] -- the ~1,000 cycles per step the register form spends storing to LDS are not what bounds
    // it: with them gone the step waits as long for its data (36 KB per CU and step arrive at ~13 B/clk either way, three or four steps
    // ahead make no difference: profiles/r05_wgrad.txt)
#ifdef VSDE_ABLATIONS
    static int use_dma = -1;
    if (use_dma < 0) use_dma = vsde_knob("VSDE_WGRAD_DMA", 0) == 1 ? 1 : 0;
    if constexpr (TN == 256 && TH == 512) {
        if (use_dma && !g_wgrad_trace && p.M % W2_BM == 0) {   // whole 32-row blocks only (rows past M cannot be zeroed on the way in)
            const size_t dlds = (size_t)4 * C::BUF * sizeof(uint16_t) + 1024;
            VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)wgrad_dma_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dlds));
            hipLaunchKernelGGL(wgrad_dma_kernel<256>, dim3((unsigned)(((p.nsplit + 7) / 8) * 8 * p.tiles)), dim3(512), dlds, s, p);
            VSDE_CHECK_HIP(hipGetLastError());
            hipLaunchKernelGGL(wgrad_tr_reduce_kernel<TN>, dim3(TN * W2_TK / 4 / 64 + 1, p.tiles), dim3(256), 0, s, p);
            VSDE_CHECK_HIP(hipGetLastError());
            return 0;
        }
    }
#endif
    VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)(((p.nsplit + 7) / 8) * 8 * p.tiles)), dim3(C::THREADS), lds, s, p);
    VSDE_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(wgrad_tr_reduce_kernel<TN>, dim3(TN * W2_TK / 4 / 64 + 1, p.tiles), dim3(256), 0, s, p);
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace vsde

// debugging: the next single-problem launches of the eight-wave TN = 256 kernel stamp their phases (tools/wgrad_trace.py); nullptr: off
extern "C" int vsde_wgrad_debug_trace(void *buf) { vsde::g_wgrad_trace = (long long *)buf; return 0; }

template <int TN, int TH = 2 * TN> static int wgrad2_launch_group(const vsde::Wgrad2Group &G, int wgs, hipStream_t s) {
    using namespace vsde;
    using C = W2<TN, TH>;
    const size_t lds = (size_t)2 * C::BUF * sizeof(uint16_t);
    auto kern = wgrad_tr_group_kernel<TN, TH>;
    VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(C::THREADS), lds, s, G);
    VSDE_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(wgrad_tr_reduce_group_kernel<TN>, dim3(TN * W2_TK / 4 / 64 + 1, (unsigned)G.tile0[G.n]), dim3(256), 0, s, G);
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}

// One item of a grouped weight-gradient launch (include/vsde_hip.h: VsdeWgradItem)
struct VsdeWgradItemC {
    const void *dy, *x;
    int64_t M;
    int32_t N, K;
    float *dW, *db;
    const int32_t *row_map;
};

extern "C" size_t vsde_linear_wgrad_group_workspace_bytes(int n, const void *items_) {
    const VsdeWgradItemC *items = (const VsdeWgradItemC *)items_;
    size_t total = 0;
    for (int i = 0; i < n; ++i) {
        if (items[i].M <= 0 || items[i].N <= 0 || items[i].K <= 0) return 0;
        vsde::Wgrad2Params p;
        vsde::wgrad2_plan(items[i].M, items[i].N, items[i].K, p);
        total += (vsde::wgrad2_workspace(p) + 255) & ~(size_t)255;
    }
    return total;
}

// dW_i = dy_i^T x_i, db_i = colsum(dy_i) for n problems in (at most) two launches per tile width (+ their reductions) instead of n;
// same arithmetic as vsde_linear_wgrad_bf16_rows problem by problem (same plan, same fixed-order sums: identical results)
extern "C" int vsde_linear_wgrad_group_bf16(int n, const void *items_, int group_plan, void *workspace, size_t workspace_bytes, void *stream) {
    using namespace vsde;
    const VsdeWgradItemC *items = (const VsdeWgradItemC *)items_;
    VSDE_CHECK_ARG(n > 0 && items && workspace, VSDE_E_BADARG, "bad linear_wgrad_group arguments");
    VSDE_CHECK_ARG(workspace_bytes >= vsde_linear_wgrad_group_workspace_bytes(n, items_), VSDE_E_WORKSPACE, "linear_wgrad_group workspace too small");
    char *ws = (char *)workspace;
    // group_plan: a problem's split count only has to fill the chip TOGETHER with the others -- about 3.5 rounds of resident
    // workgroups over all tiles of its tile-width class (every workgroup costs ~8 us of prologue / partial-tile write whatever its
    // share of the rows, and the reduction reads tiles x splits x 263 KB).  Never more splits than the problem's own plan (the
    // workspace is sized for that).  The sums are then taken in a different order than a single launch takes them.
    int64_t class_tiles[2] = {0, 0};
    if (group_plan)
        for (int i = 0; i < n; ++i) {
            Wgrad2Params p;
            wgrad2_plan(items[i].M, items[i].N, items[i].K, p);
            class_tiles[p.tn == 256] += p.tiles;
        }
    for (int tn = 128; tn <= 256; tn += 128) {
        int64_t cap = 1 << 30;
        if (group_plan && class_tiles[tn == 256] > 0) {
            const int resident = tn == 256 ? 256 : 512;
            cap = ((int64_t)(3.5 * resident / (double)class_tiles[tn == 256]) + 4) & ~(int64_t)7;
            if (cap < 8) cap = 8;
        }
        Wgrad2Group G;
        G.n = 0; G.first[0] = 0; G.tile0[0] = 0;
        char *w = ws;
        for (int i = 0; i <= n; ++i) {
            bool flush = i == n;
            Wgrad2Params p;
            size_t bytes = 0;
            if (i < n) {
                const VsdeWgradItemC &it = items[i];
                VSDE_CHECK_ARG(it.dy && it.x && it.dW && it.M > 0 && it.N % 8 == 0 && it.K % 8 == 0 && ((uintptr_t)it.dy % 16) == 0 &&
                               ((uintptr_t)it.x % 16) == 0, VSDE_E_BADARG, "bad item %d of linear_wgrad_group", i);
                wgrad2_plan(it.M, it.N, it.K, p);
                bytes = (wgrad2_workspace(p) + 255) & ~(size_t)255;
                if (p.tn == tn) {
                    if (p.nsplit > cap) p.nsplit = (int)cap;
                    p.dy = (const uint16_t *)it.dy; p.x = (const uint16_t *)it.x; p.partial = (float *)w; p.dW = it.dW; p.db = it.db; p.row_map = it.row_map;
                    G.g[G.n] = p;
                    G.first[G.n + 1] = G.first[G.n] + ((p.nsplit + 7) / 8) * 8 * p.tiles;
                    G.tile0[G.n + 1] = G.tile0[G.n] + p.tiles;
                    ++G.n;
                    flush = flush || G.n == W2_MAXG;
                }
                w += bytes;
            }
            if (flush && G.n > 0) {
                const int rc = tn == 256 ? (VSDE_FOUR_WAVES() ? VSDE_FOUR_WAVES_GROUP(G, G.first[G.n], (hipStream_t)stream)
                                                                : wgrad2_launch_group<256>(G, G.first[G.n], (hipStream_t)stream))
                                         : wgrad2_launch_group<128>(G, G.first[G.n], (hipStream_t)stream);
                if (rc != 0) return rc;
                G.n = 0; G.first[0] = 0; G.tile0[0] = 0;
            }
        }
    }
    return 0;
}

extern "C" size_t vsde_linear_wgrad_workspace_bytes(int64_t M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    vsde::Wgrad2Params p;
    vsde::wgrad2_plan(M, N, K, p);
    return vsde::wgrad2_workspace(p);
}

extern "C" int vsde_linear_wgrad_bf16_rows(const void *dy, const void *x, int64_t M, int N, int K, float *dW, float *db,
                                           const int32_t *row_map, void *workspace, size_t workspace_bytes, void *stream);

extern "C" int vsde_linear_wgrad_bf16(const void *dy, const void *x, int64_t M, int N, int K, float *dW, float *db, void *workspace,
                                      size_t workspace_bytes, void *stream) {
    return vsde_linear_wgrad_bf16_rows(dy, x, M, N, K, dW, db, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int vsde_linear_wgrad_bf16_rows(const void *dy, const void *x, int64_t M, int N, int K, float *dW, float *db,
                                           const int32_t *row_map, void *workspace, size_t workspace_bytes, void *stream) {
    using namespace vsde;
    VSDE_CHECK_ARG(dy && x && dW && workspace && M > 0, VSDE_E_BADARG, "bad linear_wgrad arguments");
    VSDE_CHECK_ARG(N % 8 == 0 && K % 8 == 0, VSDE_E_BADARG, "linear_wgrad needs N %% 8 == 0 and K %% 8 == 0 (got %d, %d)", N, K);
    VSDE_CHECK_ARG(((uintptr_t)dy % 16) == 0 && ((uintptr_t)x % 16) == 0, VSDE_E_BADARG, "linear_wgrad operands must be 16-byte aligned");
    Wgrad2Params p;
    wgrad2_plan(M, N, K, p);
    const size_t need = wgrad2_workspace(p);
    VSDE_CHECK_ARG(workspace_bytes >= need, VSDE_E_WORKSPACE, "linear_wgrad workspace too small: %zu < %zu", workspace_bytes, need);
    p.dy = (const uint16_t *)dy; p.x = (const uint16_t *)x; p.partial = (float *)workspace; p.dW = dW; p.db = db; p.row_map = row_map;
#ifdef VSDE_ABLATIONS
    if (p.tn == 256 && wgrad2_four_waves()) return wgrad2_launch<256, 256>(p, (hipStream_t)stream);
#endif
    if (p.tn == 256) return wgrad2_launch<256>(p, (hipStream_t)stream);
    return wgrad2_launch<128>(p, (hipStream_t)stream);
}
