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
//   * cols kernel ("C-stationary", N tile of 128 / 256, any K % 64 == 0): the wave's 32 x N accumulators stay in VGPRs, the
//     loop runs over 64-deep K chunks of W ([N][64]) and of x (fragment loads straight to registers).
// Products are computed swapped (D = W_tile . x_tile^T): the lane that owns activation row r keeps it through the whole
// stripe, and each accumulator register quad is 4 consecutive output columns of that row -- bias, SwiGLU and its derivative
// are then lane-local register math, and the tile leaves through a per-wave LDS staging buffer as full 128-byte row segments.
// Epilogues:
//   EPI_PLAIN        y = acc + bias                                                       (bf16)
//   EPI_SWIGLU       u = acc + bias (kept for the backward), s = silu(a) * b   with [a | b] the two 32-column halves of a
//                    64-column tile: the packed weight interleaves the SwiGLU halves in blocks of 32 rows (primitives/fused.py)
//   EPI_SWIGLU_BWD   acc = ds (gradient of s); reads u, writes du = (da | db) in the same interleaved layout
#include "vsde_common.h"

namespace vsde {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // a 16-byte register quad (native vector: stays in VGPRs)

constexpr int EPI_PLAIN = 0, EPI_SWIGLU = 1, EPI_SWIGLU_BWD = 2;
constexpr int LIN_ROWS = 256;   // rows per workgroup (8 waves x 32)
constexpr int LIN_THREADS = 512;

struct LinParams {
    const uint16_t *A; int64_t lda;    // activations [M][lda] bf16 (row pitch in elements)
    const uint16_t *W;                 // weight [N][K] bf16, contiguous
    const uint16_t *bias;              // [N] bf16 or nullptr
    uint16_t *C; int64_t ldc;          // output [M][ldc]; EPI_SWIGLU: u [M][N] (may be nullptr); EPI_SWIGLU_BWD: du [M][2N]
    uint16_t *S; int64_t lds_;         // EPI_SWIGLU: s [M][N/2]
    const uint16_t *U; int64_t ldu;    // EPI_SWIGLU_BWD: saved u [M][2N]
    int64_t M; int N, K;
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

// 32 rows x 64 columns of bf16 out of the wave's staging buffer (row stride SLD elements) as full 128-byte row segments
template <int SLD>
__device__ __forceinline__ void flush_rows64(const uint16_t *stage, uint16_t *dst, int64_t ld, int64_t row0, int64_t M, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (lane >> 3) + 8 * i, c = lane & 7;
        const uint4 v = *(const uint4 *)(stage + row * SLD + c * 8);
        if (row0 + row < M) *(uint4 *)(dst + (row0 + row) * ld + c * 8) = v;
    }
}

// Epilogue of one 64-column tile held as acc[0], acc[1] (32 columns each) for the wave's 32 rows.
// brow: LDS row with the tile's 64 bias values (bf16; zeros without a bias).  n0: first output column of the tile.
template <int EPI>
__device__ __forceinline__ void tile_epilogue(const LinParams &p, f32x16 (&acc)[2], const uint16_t *brow, uint16_t *stage,
                                              int64_t row0, int n0, int lane) {
    const int r = lane & 31, h = lane >> 5;
    if constexpr (EPI == EPI_PLAIN || EPI == EPI_SWIGLU) {
        constexpr int SLD = 72;   // 64 + 8 elements: 144-byte rows
        uint32_t sp[2][4];        // EPI_SWIGLU: packed s values per register quad
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v[2][4];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const uint2 bb = *(const uint2 *)(brow + nb * 32 + 8 * g + 4 * h);
                v[nb][0] = acc[nb][4 * g + 0] + bf_lo(bb.x); v[nb][1] = acc[nb][4 * g + 1] + bf_hi(bb.x);
                v[nb][2] = acc[nb][4 * g + 2] + bf_lo(bb.y); v[nb][3] = acc[nb][4 * g + 3] + bf_hi(bb.y);
                *(uint2 *)(stage + r * SLD + nb * 32 + 8 * g + 4 * h) = make_uint2(pack_bf16x2(v[nb][0], v[nb][1]), pack_bf16x2(v[nb][2], v[nb][3]));
            }
            if constexpr (EPI == EPI_SWIGLU) {
                float s[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {   // from the bf16-rounded u, as the unfused chain (mlp.py:21-24 under autocast)
                    const float a = rbf(v[0][i]), b = rbf(v[1][i]);
                    s[i] = rbf(a * sigm_f(a)) * b;
                }
                sp[0][g] = pack_bf16x2(s[0], s[1]); sp[1][g] = pack_bf16x2(s[2], s[3]);
            }
        }
        wave_lds_fence();
        if (EPI == EPI_PLAIN || p.C != nullptr) flush_rows64<SLD>(stage, p.C + n0, p.ldc, row0, p.M, lane);
        if constexpr (EPI == EPI_SWIGLU) {
            wave_lds_fence();
#pragma unroll
            for (int g = 0; g < 4; ++g) *(uint2 *)(stage + r * SLD + 8 * g + 4 * h) = make_uint2(sp[0][g], sp[1][g]);
            wave_lds_fence();
#pragma unroll
            for (int i = 0; i < 2; ++i) {   // 32 columns of s = 64 bytes per row: 4 lanes per row, 16 rows per instruction
                const int row = (lane >> 2) + 16 * i, c = lane & 3;
                const uint4 v = *(const uint4 *)(stage + row * SLD + c * 8);
                if (row0 + row < p.M) *(uint4 *)(p.S + (row0 + row) * p.lds_ + (n0 >> 1) + c * 8) = v;
            }
        }
        wave_lds_fence();
    } else {
        // acc[nb] = ds for columns j = n0 + 32 nb + ..; u / du tile: 128 interleaved columns [a(32) | b(32) | a(32) | b(32)]
        constexpr int SLD = 136;  // 128 + 8 elements
#pragma unroll
        for (int i = 0; i < 8; ++i) {   // coalesced load of the wave's 32 x 128 slice of u: 4 rows x 256 bytes per instruction
            const int row = (lane >> 4) + 4 * i, c = lane & 15;
            const uint4 v = row0 + row < p.M ? *(const uint4 *)(p.U + (row0 + row) * p.ldu + 2 * n0 + c * 8) : make_uint4(0, 0, 0, 0);
            *(uint4 *)(stage + row * SLD + c * 8) = v;
        }
        wave_lds_fence();
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint16_t *pa = stage + r * SLD + nb * 64 + 8 * g + 4 * h, *pb = pa + 32;
                const uint2 ua = *(const uint2 *)pa, ub = *(const uint2 *)pb;
                const float a[4] = {bf_lo(ua.x), bf_hi(ua.x), bf_lo(ua.y), bf_hi(ua.y)};
                const float b[4] = {bf_lo(ub.x), bf_hi(ub.x), bf_lo(ub.y), bf_hi(ub.y)};
                float da[4], db[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float gs = acc[nb][4 * g + i], sg = sigm_f(a[i]);
                    da[i] = gs * b[i] * sg * (1.0f + a[i] * (1.0f - sg));
                    db[i] = gs * a[i] * sg;
                }
                *(uint2 *)pa = make_uint2(pack_bf16x2(da[0], da[1]), pack_bf16x2(da[2], da[3]));
                *(uint2 *)pb = make_uint2(pack_bf16x2(db[0], db[1]), pack_bf16x2(db[2], db[3]));
            }
        wave_lds_fence();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = (lane >> 4) + 4 * i, c = lane & 15;
            const uint4 v = *(const uint4 *)(stage + row * SLD + c * 8);
            if (row0 + row < p.M) *(uint4 *)(p.C + (row0 + row) * p.ldc + 2 * n0 + c * 8) = v;
        }
        wave_lds_fence();
    }
}

template <int EPI> constexpr int stage_elems() { return EPI == EPI_SWIGLU_BWD ? 32 * 136 : 32 * 72; }

// weight tile [ROWS][KW] (row pitch ldw in global memory) <-> registers <-> LDS rows of LDB elements
template <int NLD, int KW>
__device__ __forceinline__ void wtile_load(u32x4 (&breg)[NLD], const uint16_t *W, int64_t ldw, int tid) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = tid + LIN_THREADS * i, row = idx / (KW / 8), c = idx % (KW / 8);
        breg[i] = *(const u32x4 *)(W + (int64_t)row * ldw + c * 8);
    }
}
template <int NLD, int KW, int LDB>
__device__ __forceinline__ void wtile_store(const u32x4 (&breg)[NLD], uint16_t *Bs, int tid) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = tid + LIN_THREADS * i, row = idx / (KW / 8), c = idx % (KW / 8);
        *(u32x4 *)(Bs + row * LDB + c * 8) = breg[i];
    }
}

// The 2 x KS products of one 64-column tile for the wave's 32 rows.  Weight fragments are fetched a group (4 k-steps x 2
// column blocks = 8 ds_read_b128) ahead of the MFMAs that use them.  bsrc = tile + r * LDB + 8 h.
template <int KC>
__device__ __forceinline__ void rows_tile_mfma(f32x16 (&acc)[2], const bf16x8 (&afr)[KC / 16], const uint16_t *bsrc) {
    constexpr int KS = KC / 16, LDB = KC + 8, GK = 4, NG = KS / GK;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[nb][e] = 0.f;
    bf16x8 bq[2][GK][2];
#pragma unroll
    for (int k = 0; k < GK; ++k)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) bq[0][k][nb] = *(const bf16x8 *)(bsrc + nb * 32 * LDB + k * 16);
#pragma unroll
    for (int gk = 0; gk < NG; ++gk) {
        if (gk + 1 < NG)
#pragma unroll
            for (int k = 0; k < GK; ++k)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    bq[(gk + 1) & 1][k][nb] = *(const bf16x8 *)(bsrc + nb * 32 * LDB + ((gk + 1) * GK + k) * 16);
        __builtin_amdgcn_sched_barrier(0);   // keep the next group's reads ahead of this group's MFMAs (hipcc sinks them otherwise)
#pragma unroll
        for (int k = 0; k < GK; ++k)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[gk & 1][k][nb], afr[gk * GK + k], acc[nb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ------------------------------------------------------------------------------------------------ rows kernel
template <int KC, int EPI>
__global__ void __launch_bounds__(LIN_THREADS) lin_rows_kernel(LinParams p) {
    constexpr int KS = KC / 16, LDB = KC + 8, TILE = 65 * LDB;   // 64 weight rows + 1 bias row
    constexpr int NLD = 64 * KC / 8 / LIN_THREADS;               // 16-byte loads per thread and tile
    extern __shared__ __attribute__((aligned(16))) uint16_t lsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    uint16_t *stage = lsm + 2 * TILE + wave * stage_elems<EPI>();
    const int64_t row0 = (int64_t)blockIdx.x * LIN_ROWS + wave * 32;

    // the wave's 32 x KC slice of the activations, as MFMA operand fragments (lane: row r, k-half h)
    bf16x8 afr[KS];
    {
        const int64_t m = row0 + r < p.M ? row0 + r : p.M - 1;   // rows past the end repeat the last one; they are never stored
        const uint16_t *src = p.A + m * p.lda + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) afr[ks] = *(const bf16x8 *)(src + ks * 16);
    }
    const int ntiles = p.N / 64;
    // Tiles are visited in a rotated order (first tile = workgroup index mod ntiles): the workgroups of a launch then pull
    // DIFFERENT 32 KB tiles of W out of L2 at any moment instead of all hammering the same lines (same channels).
    const int rot = blockIdx.x % ntiles;
    // Two register sets: tile t travels in breg[t & 1]; its loads are issued two iterations before its LDS store.
    u32x4 breg[2][NLD];
    uint32_t biasreg[2] = {0u, 0u};
#define VSDE_TILE_LOAD(t_, set_)                                                                              \
    do {                                                                                                      \
        const int tile_ = ((t_) + rot) % ntiles;                                                              \
        wtile_load<NLD, KC>(breg[set_], p.W + (int64_t)tile_ * 64 * KC, KC, tid);                             \
        if (tid < 32) biasreg[set_] = p.bias ? *(const uint32_t *)(p.bias + tile_ * 64 + 2 * tid) : 0u;       \
    } while (0)
#define VSDE_TILE_STORE(Bs_, set_)                                                                            \
    do {                                                                                                      \
        wtile_store<NLD, KC, LDB>(breg[set_], (Bs_), tid);                                                    \
        if (tid < 32) *(uint32_t *)((Bs_) + 64 * LDB + 2 * tid) = biasreg[set_];                              \
    } while (0)
// one tile: MFMAs out of LDS buffer PAR_, epilogue, then tile t + 1 (register set 1 - PAR_) goes to the other buffer
#define VSDE_ROWS_BODY(t_, PAR_)                                                                              \
    do {                                                                                                      \
        const uint16_t *Bs = lsm + (PAR_) * TILE;                                                             \
        f32x16 acc[2];                                                                                        \
        rows_tile_mfma<KC>(acc, afr, Bs + r * LDB + 8 * h);                                                   \
        tile_epilogue<EPI>(p, acc, Bs + 64 * LDB, stage, row0, (((t_) + rot) % ntiles) * 64, lane);           \
        if ((t_) + 1 < ntiles) VSDE_TILE_STORE(lsm + (1 - (PAR_)) * TILE, 1 - (PAR_));                        \
        lds_barrier();                                                                                        \
        if ((t_) + 3 < ntiles) VSDE_TILE_LOAD((t_) + 3, 1 - (PAR_));                                          \
    } while (0)
    VSDE_TILE_LOAD(0, 0);
    VSDE_TILE_STORE(lsm, 0);
    lds_barrier();
    if (ntiles > 1) VSDE_TILE_LOAD(1, 1);
    if (ntiles > 2) VSDE_TILE_LOAD(2, 0);
    for (int nt = 0; nt < ntiles; nt += 2) {
        VSDE_ROWS_BODY(nt, 0);
        if (nt + 1 < ntiles) VSDE_ROWS_BODY(nt + 1, 1);
    }
#undef VSDE_ROWS_BODY
#undef VSDE_TILE_LOAD
#undef VSDE_TILE_STORE
}

// The 4 x NB (k-step, column block) products of one 64-deep chunk for the wave's 32 rows; the weight fragments of a group of
// 4 are fetched while the MFMAs of the previous group run.  bsrc = tile + r * 72 + 8 h.
template <int NB>
__device__ __forceinline__ void cols_tile_mfma(f32x16 (&acc)[NB], const bf16x8 (&afr)[4], const uint16_t *bsrc) {
    constexpr int G = 4, NGRP = 4 * NB / G, LDB = 72;
    bf16x8 bq[2][G];
#pragma unroll
    for (int i = 0; i < G; ++i) bq[0][i] = *(const bf16x8 *)(bsrc + (i % NB) * 32 * LDB + (i / NB) * 16);
#pragma unroll
    for (int g = 0; g < NGRP; ++g) {
        if (g + 1 < NGRP)
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const int q = (g + 1) * G + i;
                bq[(g + 1) & 1][i] = *(const bf16x8 *)(bsrc + (q % NB) * 32 * LDB + (q / NB) * 16);
            }
        __builtin_amdgcn_sched_barrier(0);   // keep the next group's reads ahead of this group's MFMAs (hipcc sinks them otherwise)
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int q = g * G + i;
            acc[q % NB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[g & 1][i], afr[q / NB], acc[q % NB], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ------------------------------------------------------------------------------------------------ cols kernel
// NB accumulator blocks of 32 columns (N tile = 32 NB in {128, 256}); blockIdx.y = N tile; K % 64 == 0.
template <int NB>
__global__ void __launch_bounds__(LIN_THREADS) lin_cols_kernel(LinParams p) {
    constexpr int NT = 32 * NB, LDB = 72, TILE = NT * LDB;      // weight tile [NT][64] with 144-byte rows
    constexpr int NLD = NT * 64 / 8 / LIN_THREADS;               // 16-byte loads per thread and tile (2 or 4)
    extern __shared__ __attribute__((aligned(16))) uint16_t lsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    uint16_t *stage = lsm + 2 * TILE + wave * stage_elems<EPI_PLAIN>();
    const int64_t row0 = (int64_t)blockIdx.x * LIN_ROWS + wave * 32;
    const int nbase = blockIdx.y * NT;
    const int64_t m = row0 + r < p.M ? row0 + r : p.M - 1;   // rows past the end repeat the last one; they are never stored
    const uint16_t *asrc = p.A + m * p.lda + 8 * h;
    const int ktiles = p.K / 64;

    // K chunks are visited in a rotated order (see the rows kernel); chunk t travels in breg[t & 1], loaded two iterations
    // before its LDS store; the activation fragments of chunk t + 1 are fetched during the MFMAs of chunk t.
    const int rot = blockIdx.x % ktiles;
    u32x4 breg[2][NLD];
    bf16x8 afr[2][4];
    const uint16_t *wsrc = p.W + (int64_t)nbase * p.K;
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
        cols_tile_mfma<NB>(acc, afr[PAR_], lsm + (PAR_) * TILE + r * LDB + 8 * h);                            \
        if ((t_) + 1 < ktiles) wtile_store<NLD, 64, LDB>(breg[1 - (PAR_)], lsm + (1 - (PAR_)) * TILE, tid);   \
        lds_barrier();                                                                                        \
        if ((t_) + 3 < ktiles) wtile_load<NLD, 64>(breg[1 - (PAR_)], wsrc + VSDE_CHUNK((t_) + 3), p.K, tid);  \
    } while (0)
    wtile_load<NLD, 64>(breg[0], wsrc + VSDE_CHUNK(0), p.K, tid);
    VSDE_A_LOAD(0, 0)
    wtile_store<NLD, 64, LDB>(breg[0], lsm, tid);
    lds_barrier();
    if (ktiles > 1) wtile_load<NLD, 64>(breg[1], wsrc + VSDE_CHUNK(1), p.K, tid);
    if (ktiles > 2) wtile_load<NLD, 64>(breg[0], wsrc + VSDE_CHUNK(2), p.K, tid);
    for (int kt = 0; kt < ktiles; kt += 2) {
        VSDE_COLS_BODY(kt, 0);
        if (kt + 1 < ktiles) VSDE_COLS_BODY(kt + 1, 1);
    }
#undef VSDE_COLS_BODY
#undef VSDE_A_LOAD
#undef VSDE_CHUNK
    // epilogue: 64 columns at a time through the wave's staging buffer; the bias row is staged in the (now free) tile buffer 0
    if (tid < NT / 2) *(uint32_t *)(lsm + 2 * tid) = p.bias ? *(const uint32_t *)(p.bias + nbase + 2 * tid) : 0u;
    lds_barrier();
#pragma unroll
    for (int q = 0; q < NB / 2; ++q) {
        f32x16 pair[2] = {acc[2 * q], acc[2 * q + 1]};
        tile_epilogue<EPI_PLAIN>(p, pair, lsm + 64 * q, stage, row0, nbase + 64 * q, lane);
    }
}

template <int KC, int EPI> static size_t rows_lds_bytes() { return (size_t)(2 * 65 * (KC + 8) + 8 * stage_elems<EPI>()) * sizeof(uint16_t); }
template <int NB> static size_t cols_lds_bytes() { return (size_t)(2 * 32 * NB * 72 + 8 * stage_elems<EPI_PLAIN>()) * sizeof(uint16_t); }

template <int KC, int EPI>
static int launch_rows(const LinParams &p, hipStream_t s) {
    const size_t lds = rows_lds_bytes<KC, EPI>();
    VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)lin_rows_kernel<KC, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((lin_rows_kernel<KC, EPI>), dim3((unsigned)((p.M + LIN_ROWS - 1) / LIN_ROWS)), dim3(LIN_THREADS), lds, s, p);
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}

template <int EPI>
static int launch_rows_k(const LinParams &p, hipStream_t s) {
    return p.K == 128 ? launch_rows<128, EPI>(p, s) : launch_rows<256, EPI>(p, s);
}

template <int NB>
static int launch_cols(const LinParams &p, hipStream_t s) {
    const size_t lds = cols_lds_bytes<NB>();
    VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)lin_cols_kernel<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((lin_cols_kernel<NB>), dim3((unsigned)((p.M + LIN_ROWS - 1) / LIN_ROWS), p.N / (32 * NB)), dim3(LIN_THREADS), lds, s, p);
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}

// 1 = rows kernel, 2 = cols kernel, 0 = shape not covered (the caller keeps its library GEMM)
static int lin_variant(int N, int K, int epilogue) {
    const bool rows_ok = (K == 128 || K == 256) && N % 64 == 0;   // K = 512 would need 172 KB of LDS: cols kernel
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
    return lin_variant(N, K, epilogue);
}

extern "C" int vsde_linear_bf16(const void *x, int64_t ldx, const void *w, const void *bias, void *y, int64_t ldy, int64_t M, int N,
                                int K, int epilogue, void *s_out, int64_t lds, const void *u_in, int64_t ldu, void *stream) {
    VSDE_CHECK_ARG(x && w && M > 0 && N > 0 && K > 0, VSDE_E_BADARG, "bad linear arguments");
    VSDE_CHECK_ARG(epilogue >= 0 && epilogue <= 2, VSDE_E_BADARG, "unknown linear epilogue %d", epilogue);
    const int variant = lin_variant(N, K, epilogue);
    VSDE_CHECK_ARG(variant != 0, VSDE_E_BADARG, "linear shape N=%d K=%d epilogue=%d is not covered by the gfx950 kernels", N, K, epilogue);
    VSDE_CHECK_ARG(ldx >= K && ldx % 8 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0, VSDE_E_BADARG,
                   "linear operands must be 16-byte aligned with row pitches that are multiples of 8 elements");
    LinParams p = {};
    p.A = (const uint16_t *)x; p.lda = ldx; p.W = (const uint16_t *)w; p.bias = (const uint16_t *)bias;
    p.C = (uint16_t *)y; p.ldc = ldy; p.M = M; p.N = N; p.K = K;
    p.S = (uint16_t *)s_out; p.lds_ = lds; p.U = (const uint16_t *)u_in; p.ldu = ldu;
    hipStream_t st = (hipStream_t)stream;
    if (epilogue == EPI_PLAIN) {
        VSDE_CHECK_ARG(y && ldy >= N && ldy % 8 == 0 && ((uintptr_t)y % 16) == 0, VSDE_E_BADARG, "bad linear output");
        if (variant == 1) return launch_rows_k<EPI_PLAIN>(p, st);
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
