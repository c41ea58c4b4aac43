// Weight / bias gradients of the fused head for hidden_dim = 64: the fast path of launch_tn_grouped (vsde_gemm.hip).
//
//   dW[n][k] = sum_m X(m, n) * Y(m, k)      m = (b, t) over all B*T path-steps, X = gate pre-activation gradients [.., 3H = 192]
//
// The generic grouped kernel cuts every problem into 64 x 64 output tiles (28 of them at the Lotka-Volterra shapes, 7 of which
// are padding around the 2 / 3 / 5-wide state, theta and emission operands), re-reads the 192-wide X rows once per tile and
// spends 4 vector-ALU instructions per MFMA on its general operand views.  Here
//   * "wide" problems (NX = 192, NY a multiple of 64) run as 192 x 64 tiles: X rows are staged once per tile and shared by
//     the 4 waves (a wave owns 48 x 64 = 3 x 4 accumulator blocks of v_mfma_f32_16x16x4_f32; exact fp32), operands go to LDS as
//     they sit in memory ([m][cols], 16 rows per step, two buffers, one LDS-only barrier per step; 37 KB and 138 VGPRs: three
//     workgroups per CU), loads are requested two steps ahead and branch-free, every load slot carries its element offset
//     forward by constants (no division or 64-bit multiply in the loop), masked Y rows (h_{t-1} at t = 0) come from a zero buffer;
//   * the narrow operands (the state / theta columns of W_ih0: NY <= 16; the emission rows of out_proj: NX <= 16, computed as
//     out^T with the roles swapped) run in the same launch as 192 x 16 / 64 x 16 tiles with one column block of MFMAs;
//   * the tiles of one row split sit on one XCD, so the X rows the four context tiles share come out of that XCD's L2;
//   * round 6: the 192 x 64 tiles run on the bf16 matrix instruction with operands split exactly into three bf16 pieces on their way
//     into LDS (tn_wide_split_kernel, the default; section "round 6" below), take their workgroups split-major, and all splits add up
//     to whole rounds of resident workgroups.  The fp32 form (tn_wide_kernel) is what the narrow tiles still use and what
//     VSDE_TW_SPLIT=0 runs in the ablation build.
// Splits are summed in a fixed order by the reduce kernels (deterministic).  Reference: kernels/backward.py:108-139,575-590
// (the global fp32 atomics these reductions replace).
#include <stddef.h>
#include <stdlib.h>

#include "vsde_common.h"

namespace vsde {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t twu4 __attribute__((ext_vector_type(4)));
typedef uint32_t twu2 __attribute__((ext_vector_type(2)));

constexpr int TW_N = 192, TW_K = 64, TW_BM = 16;   // rows per step: 16 -> 37 KB of LDS, three workgroups per CU
constexpr int TW_YI = TW_BM * 16 / 256;             // Y chunks per thread and step
constexpr int TW_LDX = TW_N + 16, TW_LDY = TW_K + 16;   // row pitches (floats), 16 past a multiple of 32: the two 16-lane rows a
                                                         // 32-lane ds_read_b32 group touches fall on disjoint banks
constexpr int TW_BUF = TW_BM * (TW_LDX + TW_LDY);        // floats per buffer
constexpr int TW_PART = TW_N * TW_K + TW_N;              // partial tile + bias row
constexpr int TW_MAXTILES = 24;

struct TwTile {
    const float *x;          // X view base
    const void *y;           // Y view base (+ k_blk columns)
    int64_t xbs, xrs, ybs, yrs;   // batch / row strides in elements
    int xshift, yshift, x_split, x_skip, y_bf16;
    int kind;                // 0: 192 x 64,  1: 192 x (NY <= 16),  2: 64 x (NY <= 16) with the roles swapped (out^T is computed)
    int ny;                  // valid Y columns (kinds 1, 2)
    float *out; int64_t ldo; int col_off;   // destination of the reduced tile: out[n * ldo + col_off + k]
    float *bias_out;         // column sums of X (kinds 0, 1) or of Y (kind 2)
    int nsplit, local_begin;  // row splits of this tile (multiple of 8); first `local` workgroup index (id = 8 * local + xcd)
    int64_t part_off;        // offset of its partials in floats
};

struct TwArgs {
    TwTile tile[TW_MAXTILES];
    int ntiles, nlocal, M, T;
    int dbg;                 // ablation build only (VSDE_TW_DBG): 16 | 32 | 64 = only the narrow | 192 x 64 | context tiles run
    int n0, local0;          // the 192 x 64 tiles (tile[0 .. n0), equal splits) and the `local` indices they share split-major: tw_locate
    int64_t chunks;
    float *partial;
};

#ifdef VSDE_ABLATIONS
#define TW_DBG(bit_) ((a.dbg & (bit_)) != 0)
#else
#define TW_DBG(bit_) false
#endif
__device__ float tw_zeros[8];   // never written: the source of masked Y rows

__device__ __forceinline__ void tw_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- round 6: the 192 x 64 tiles on the bf16 matrix instruction with SPLIT operands ------------------------------------------------
// v_mfma_f32_16x16x4_f32 runs at the VALU's rate (32 cycles per instruction) and nothing overlaps with it on its SIMD
// (profiles/r06_mfma_valu_overlap.txt): the fp32 tiles held the matrix pipe 63 % busy at 404 us (profiles/r06_pmc_head_lv.txt).  Here an
// fp32 operand value is cut into three bf16 pieces -- x = h + m + l exactly (8 + 8 + 8 significant bits, tw_split2) -- when its
// rows go to LDS, and a 32 x 32 block over 16 rows is hh + hm + mh + mm + hl + lh: six v_mfma_f32_32x32x16_bf16 (32 cycles each) for
// what took sixteen fp32 instructions of 32 cycles (the dropped products ml, lm, ll are <= 2^-23 of the term: fp32 round-off class);
// with a bf16 context operand (exact in one piece) three.  The pieces sit in LDS as row-major bf16 planes [16 rows][cols]; the MFMA
// operands (8 consecutive ROWS of one column per lane) come out of them through the transposing read ds_read_b64_tr_b16; row pitch =
// row + 64 bytes (64 past a multiple of 256: the 4 rows x 64 bytes a half-wave's read touches fall on disjoint banks).  Two plane
// sets: while the waves multiply block s out of one, they split block s + 1 into the other -- plain VALU work issues in the shadow of
// the bf16 MFMAs (the same profile) -- and one barrier per block separates the two roles.  A wave owns 96 x 32 of the tile.
typedef short twbf8 __attribute__((ext_vector_type(8)));
typedef short twbf4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int TS_SX = 2 * TW_N + 64, TS_XP = TW_BM * TS_SX;   // X planes: row pitch, plane size (bytes)
constexpr int TS_SY = 2 * TW_K + 64, TS_YP = TW_BM * TS_SY;   // Y planes
constexpr int TS_BUF = 3 * TS_XP + 3 * TS_YP;                 // one set of planes: 30,720 bytes
constexpr int TS_LDS = 2 * TS_BUF;                            // 61,440 bytes: two workgroups per CU

// two fp32 values (bit patterns) -> their three bf16 pieces, packed (value 0 in the low half).  Each piece is the value rounded to
// nearest-even bf16 (v_cvt_pk_bf16_f32, two values per instruction), the next one splits what is left: both remainders are exact in
// fp32 and the third (<= 8 significant bits) is exact in bf16: h + m + l == x bit for bit for |x| in [2^-110, 3.38e38]; |m| <= 2^-8 |x|,
// |l| <= 2^-16 |x| with signs of their own, so the dropped products m l + l m + l l are <= 2^-23 |x y| and do not pile up one-sided
// (truncated pieces would all carry the sign of x: 2^-21 and biased).  11 VALU instructions per pair.
typedef float twf2 __attribute__((ext_vector_type(2)));
typedef __bf16 twb2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t tw_pack(float a, float b) {
    const twb2 r = __builtin_convertvector((twf2){a, b}, twb2);
    return *(const uint32_t *)&r;
}
__device__ __forceinline__ void tw_split2(uint32_t v0, uint32_t v1, uint32_t &h, uint32_t &m, uint32_t &l) {
    h = tw_pack(__uint_as_float(v0), __uint_as_float(v1));
    const float r0 = __uint_as_float(v0) - __uint_as_float(h << 16), r1 = __uint_as_float(v1) - __uint_as_float(h & 0xffff0000u);
    m = tw_pack(r0, r1);
    const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
    l = tw_pack(s0, s1);
}
__device__ __forceinline__ twbf8 tw_tr_frag(const char *p0, const char *p1) {   // rows +0..3 and +4..7 of this lane's column
    const twbf4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) twbf4 *)p0);
    const twbf4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) twbf4 *)p1);
    return (twbf8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

// NB: 16-column blocks of X per wave (3: X is 192 wide, 1: 64 wide); NARROW: Y has at most 16 columns (one block, scalar loads)
// SPLIT (NB = 3, wide Y only): the bf16 split form above; YBF: Y is a bf16 operand (one piece); BIAS: the column sums of X are wanted
template <int NB, bool NARROW, bool SPLIT = false, bool YBF = false, bool BIAS = false>
__device__ __forceinline__ void tw_run(const TwArgs &a, const TwTile &P, float *tws, int split) {
    static_assert(!SPLIT || (NB == 3 && !NARROW), "split form: 192 x 64 tiles");
    constexpr int NX = 64 * NB, LDX = NX + 16, JB = NARROW ? 1 : 4, LDY = NARROW ? 48 : TW_LDY;
    constexpr int XI = NX * TW_BM / 4 / 256;          // 16-byte X chunks per thread and step (3 | 1)
    constexpr int XCH = NX / 4;                        // chunks per X row
    constexpr int BUF = TW_BM * (LDX + LDY);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int xr[XI], xc[XI], yr[TW_YI], yc[TW_YI];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int id = tid + 256 * i;
        xr[i] = id / XCH; const int c = (id % XCH) * 4;
        xc[i] = c < P.x_split ? c : c + P.x_skip;   // (dr, du | dc_n) views skip one 64-wide block
    }
#pragma unroll
    for (int i = 0; i < TW_YI; ++i) { const int id = tid + 256 * i; yr[i] = id >> 4; yc[i] = NARROW ? (id & 15) : (id & 15) * 4; }
    // Block k of this split covers rows (k * nsplit + split) * TW_BM ...  Every load slot keeps the time index of its row and the
    // element offset of its 16 bytes; a step adds constants (+ a correction when the time index wraps into the next batch row):
    // no division and no 64-bit multiply in the loop.  M is a multiple of TW_BM and X views have no time shift (launcher), so
    // X needs no masking; a Y row with t + shift < 0 (h_{t-1} at t = 0) or a column past ny is fetched from a zero buffer.
    const int nsteps = (int)((a.chunks - split + P.nsplit - 1) / P.nsplit);
    const int64_t adv_rows = (int64_t)P.nsplit * TW_BM;
    const int adv_b = (int)(adv_rows / a.T), adv_t = (int)(adv_rows % a.T);
    const int64_t x_adv = (int64_t)adv_b * P.xbs + (int64_t)adv_t * P.xrs, x_wrap = P.xbs - (int64_t)a.T * P.xrs;
    const int64_t y_adv = (int64_t)adv_b * P.ybs + (int64_t)adv_t * P.yrs, y_wrap = P.ybs - (int64_t)a.T * P.yrs;
    int xt[XI], yt[TW_YI];
    int64_t xo[XI], yo[TW_YI];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int64_t m = (int64_t)split * TW_BM + xr[i];
        const int bb = (int)(m / a.T); xt[i] = (int)(m % a.T);
        xo[i] = (int64_t)bb * P.xbs + (int64_t)xt[i] * P.xrs + xc[i];
    }
    const bool ybf = P.y_bf16 != 0;
    const int yesz = ybf ? 2 : 4, yhi = ybf ? 0 : 8;
    bool ycol_ok[TW_YI];
#pragma unroll
    for (int i = 0; i < TW_YI; ++i) {
        const int64_t m = (int64_t)split * TW_BM + yr[i];
        const int bb = (int)(m / a.T); yt[i] = (int)(m % a.T);
        yo[i] = ((int64_t)bb * P.ybs + (int64_t)(yt[i] + P.yshift) * P.yrs + yc[i]) * yesz;   // bytes
        ycol_ok[i] = !NARROW || yc[i] < P.ny;
    }
    const int64_t y_adv_b = y_adv * yesz, y_wrap_b = y_wrap * yesz;
    const char *ybase = (const char *)P.y;
    const char *zeros = (const char *)tw_zeros;
    constexpr int NSET = SPLIT ? 4 : 2;   // blocks in flight
    twu4 rx[NSET][XI], ry[NSET][TW_YI];
    int fetched = 0;   // blocks requested so far; the offsets only move on while another block of this split exists, so the
                       // requests past the end re-read the last block instead of running off the operands
#define TW_FETCH(set_)                                                                                                  \
    do {                                                                                                                \
        ++fetched;                                                                                                      \
        const bool go = fetched < nsteps;                                                                               \
        const int at_ = go ? adv_t : 0;                                                                                 \
        const int64_t xa_ = go ? x_adv : 0, ya_ = go ? y_adv_b : 0;                                                     \
        _Pragma("unroll") for (int i = 0; i < XI; ++i) {                                                                \
            rx[set_][i] = *(const twu4 *)(P.x + xo[i]);                                                                 \
            xt[i] += at_; const bool w = xt[i] >= a.T; xt[i] -= w ? a.T : 0; xo[i] += xa_ + (w ? x_wrap : 0);           \
        }                                                                                                               \
        _Pragma("unroll") for (int i = 0; i < TW_YI; ++i) {                                                             \
            const char *src = (ycol_ok[i] && yt[i] + P.yshift >= 0) ? ybase + yo[i] : zeros;                            \
            if constexpr (NARROW) {                                                                                     \
                ry[set_][i].x = *(const uint32_t *)src;                                                                 \
            } else {   /* two 8-byte requests, no branch: the second one repeats the first when the row holds bf16 */   \
                const twu2 lo = *(const twu2 *)src, hi = *(const twu2 *)(src + yhi);                                    \
                ry[set_][i] = (twu4){lo.x, lo.y, hi.x, hi.y};                                                           \
            }                                                                                                           \
            yt[i] += at_; const bool w = yt[i] >= a.T; yt[i] -= w ? a.T : 0; yo[i] += ya_ + (w ? y_wrap_b : 0);         \
        }                                                                                                               \
    } while (0)
#define TW_COMMIT(set_, buf_)                                                                                           \
    do {                                                                                                                \
        float *xs_ = (buf_), *ys_ = (buf_) + TW_BM * LDX;                                                               \
        _Pragma("unroll") for (int i = 0; i < XI; ++i)                                                                  \
            *(twu4 *)(xs_ + xr[i] * LDX + (tid + 256 * i) % XCH * 4) = rx[set_][i];                                     \
        _Pragma("unroll") for (int i = 0; i < TW_YI; ++i) {                                                             \
            if constexpr (NARROW) {                                                                                     \
                *(uint32_t *)(ys_ + yr[i] * LDY + yc[i]) = ry[set_][i].x;                                               \
            } else {                                                                                                    \
                const twu4 r = ry[set_][i];                                                                             \
                *(twu4 *)(ys_ + yr[i] * LDY + yc[i]) =                                                                  \
                    ybf ? (twu4){r.x << 16, r.x & 0xffff0000u, r.y << 16, r.y & 0xffff0000u} : r;                       \
            }                                                                                                           \
        }                                                                                                               \
    } while (0)
    if constexpr (SPLIT) {
        char *lds = (char *)tws;
        constexpr int NYP = YBF ? 1 : 3, NP = YBF ? 3 : 6;
        constexpr int PX[6] = {2, 0, 1, 1, 0, 0}, PY[6] = {0, 2, 1, 0, 1, 0};   // l h, h l, m m, m h, h m, h h: smallest products first
        constexpr int QX[3] = {2, 1, 0};
        const int wn = wave & 1, wk = wave >> 1;   // the wave's 96 rows (n) x 32 columns (k) of the tile
        float bsum[XI][4];
#pragma unroll
        for (int i = 0; i < XI; ++i) bsum[i][0] = bsum[i][1] = bsum[i][2] = bsum[i][3] = 0.f;
        // one 16-row block out of register set set_ into plane set buf_ (+ the column sums of X for the bias gradient: a thread's
        // chunks keep their columns from block to block); MASK_: the block may lie past the end of the split (live_ = 0: it is zeros)
#define TW_COMMIT_S(set_, buf_, MASK_, live_)                                                                                \
        do {                                                                                                            \
            char *b_ = lds + (buf_) * TS_BUF;                                                                           \
            _Pragma("unroll") for (int i = 0; i < XI; ++i) {                                                            \
                char *d_ = b_ + xr[i] * TS_SX + ((tid + 256 * i) % XCH) * 8;                                            \
                const twu4 v_ = (MASK_) ? rx[set_][i] & (live_) : rx[set_][i];                                          \
                if constexpr (BIAS) {                                                                                   \
                    bsum[i][0] += __uint_as_float(v_.x); bsum[i][1] += __uint_as_float(v_.y);                           \
                    bsum[i][2] += __uint_as_float(v_.z); bsum[i][3] += __uint_as_float(v_.w);                           \
                }                                                                                                       \
                uint32_t h0_, m0_, l0_, h1_, m1_, l1_;                                                                  \
                tw_split2(v_.x, v_.y, h0_, m0_, l0_); tw_split2(v_.z, v_.w, h1_, m1_, l1_);                             \
                *(twu2 *)d_ = (twu2){h0_, h1_}; *(twu2 *)(d_ + TS_XP) = (twu2){m0_, m1_}; *(twu2 *)(d_ + 2 * TS_XP) = (twu2){l0_, l1_}; \
            }                                                                                                           \
            {                                                                                                           \
                char *d_ = b_ + 3 * TS_XP + yr[0] * TS_SY + yc[0] * 2;                                                  \
                const twu4 v_ = (MASK_) ? ry[set_][0] & (live_) : ry[set_][0];                                          \
                if constexpr (YBF) {                                                                                    \
                    *(twu2 *)d_ = (twu2){v_.x, v_.y};                                                                   \
                } else {                                                                                                \
                    uint32_t h0_, m0_, l0_, h1_, m1_, l1_;                                                              \
                    tw_split2(v_.x, v_.y, h0_, m0_, l0_); tw_split2(v_.z, v_.w, h1_, m1_, l1_);                         \
                    *(twu2 *)d_ = (twu2){h0_, h1_}; *(twu2 *)(d_ + TS_YP) = (twu2){m0_, m1_}; *(twu2 *)(d_ + 2 * TS_YP) = (twu2){l0_, l1_}; \
                }                                                                                                       \
            }                                                                                                           \
        } while (0)
        f32x16 sacc[3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[i][r] = 0.f;
        // Transposing reads of a 32-column operand block: the 16-lane group (lane >> 4) covers columns 16 ((lane >> 4) & 1) .. of rows
        // 8 (lane >> 5) + {0..3 | 4..7}; lane m of the group supplies row m / 4, columns 4 (m % 4) .. and receives column m.
        const int lrow = 8 * (lane >> 5) + ((lane & 15) >> 2), lcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
        const int xlane = lrow * TS_SX + lcol * 2 + 64 * 3 * wn, ylane = 3 * TS_XP + lrow * TS_SY + lcol * 2 + 64 * wk;
        // Block s_ (plane set k_ & 1): its operands are read first, then its products issue interleaved with the splitting of block
        // s_ + 1 (register set (k_ + 1) % NSET) into the other plane set and with the requests that refill those registers: the reads
        // stand before the writes in program order, so nothing orders the VALU work behind the MFMAs
#define TW_IT_S(k_, s_, MASK_)                                                                                              \
        do {                                                                                                            \
            const char *b_ = lds + ((k_) & 1) * TS_BUF;                                                                 \
            twbf8 yb[NYP], xa[3][3];                                                                                    \
            _Pragma("unroll") for (int pl = 0; pl < NYP; ++pl) {                                                        \
                const char *q_ = b_ + pl * TS_YP + ylane;                                                               \
                yb[pl] = tw_tr_frag(q_, q_ + 4 * TS_SY);                                                                \
            }                                                                                                           \
            _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                                            \
                _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                                         \
                    const char *q_ = b_ + pl * TS_XP + xlane + 64 * i;                                                  \
                    xa[i][pl] = tw_tr_frag(q_, q_ + 4 * TS_SX);                                                         \
                }                                                                                                       \
            _Pragma("unroll") for (int p = 0; p < NP; ++p)                                                              \
                _Pragma("unroll") for (int i = 0; i < 3; ++i)                                                           \
                    sacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[i][YBF ? QX[p % 3] : PX[p]], yb[YBF ? 0 : PY[p]], sacc[i], 0, 0, 0); \
            const uint32_t live_ = (s_) + 1 < nsteps ? 0xffffffffu : 0u;                                                \
            TW_COMMIT_S(((k_) + 1) % NSET, ((k_) + 1) & 1, MASK_, live_);                                               \
            TW_FETCH(((k_) + 1) % NSET);                                                                                \
            _Pragma("unroll") for (int g = 0; g < 3 * NP; ++g) {                                                        \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                  /* one MFMA */                      \
                __builtin_amdgcn_sched_group_barrier(0x002, YBF ? 14 : 8, 0);       /* VALU in its shadow */            \
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                  /* an LDS write */                  \
                if (g % 3 == 2) __builtin_amdgcn_sched_group_barrier(0x020, YBF ? 2 : 1, 0);   /* a memory request */   \
            }                                                                                                           \
            tw_barrier();                                                                                               \
            __builtin_amdgcn_sched_barrier(0);   /* one scheduling region per block */                                  \
        } while (0)
        if (nsteps > 0) {
            static_assert(NSET == 4, "the groups below are written out for four register sets");
            TW_FETCH(0); TW_FETCH(1); TW_FETCH(2); TW_FETCH(3);
            TW_COMMIT_S(0, 0, false, 0u);
            TW_FETCH(0);
            tw_barrier();
            // whole groups of blocks, straight-line (a conditional block inside the body would make hipcc merge the orders in which the
            // register sets can be outstanding and wait with vmcnt(0); leaving the loop from its middle makes it copy the accumulators):
            // in the last group a block past the end of the split is split as zeros
            int s = 0;
            for (; s + NSET < nsteps; s += NSET) {
                TW_IT_S(0, s, false); TW_IT_S(1, s + 1, false); TW_IT_S(2, s + 2, false); TW_IT_S(3, s + 3, false);
            }
            TW_IT_S(0, s, true); TW_IT_S(1, s + 1, true); TW_IT_S(2, s + 2, true); TW_IT_S(3, s + 3, true);
        }
#undef TW_IT_S
#undef TW_COMMIT_S
        // C/D layout of the 32x32 MFMA: column (k) = lane & 31, rows (n) 8 (r / 4) + 4 (lane >> 5) + r % 4
        float *dst = a.partial + P.part_off + (int64_t)split * TW_PART;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                dst[(96 * wn + 32 * i + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3)) * TW_K + 32 * wk + (lane & 31)] = sacc[i][r];
        if constexpr (BIAS) {   // the 16 threads that hold the rows of one column chunk meet in LDS
            float *sc = (float *)lds;
            tw_barrier();
#pragma unroll
            for (int i = 0; i < XI; ++i)
                *(float4 *)(sc + xr[i] * TW_N + ((tid + 256 * i) % XCH) * 4) = make_float4(bsum[i][0], bsum[i][1], bsum[i][2], bsum[i][3]);
            tw_barrier();
            if (tid < TW_N) {
                float t = 0.f;
#pragma unroll
                for (int r = 0; r < TW_BM; ++r) t += sc[r * TW_N + tid];
                dst[TW_N * TW_K + tid] = t;
            }
        }
        return;
    }
    f32x4 acc[NB][JB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < JB; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum[NB], ysum = 0.f;
#pragma unroll
    for (int i = 0; i < NB; ++i) bsum[i] = 0.f;
    const int fr = lane & 15, fq = lane >> 4;
#define TW_BODY(s_, PAR_)                                                                                               \
    do {                                                                                                                \
        const float *xs = tws + (PAR_) * BUF, *ys = xs + TW_BM * LDX;                                                   \
        _Pragma("unroll") for (int ks = 0; ks < TW_BM / 4; ++ks) {                                                      \
            float xa[NB], yb[JB];                                                                                       \
            _Pragma("unroll") for (int i = 0; i < NB; ++i) xa[i] = xs[(4 * ks + fq) * LDX + 16 * NB * wave + 16 * i + fr]; \
            _Pragma("unroll") for (int j = 0; j < JB; ++j) yb[j] = ys[(4 * ks + fq) * LDY + 16 * j + fr];               \
            ysum += yb[0];                                                                                              \
            _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                                            \
                bsum[i] += xa[i];                                                                                       \
                _Pragma("unroll") for (int j = 0; j < JB; ++j)                                                          \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[i], yb[j], acc[i][j], 0, 0, 0);                 \
            }                                                                                                           \
            if (ks & 1) __builtin_amdgcn_sched_barrier(0);                                                              \
        }                                                                                                               \
        if ((s_) + 1 < nsteps) TW_COMMIT(1 - (PAR_), tws + (1 - (PAR_)) * BUF);                                         \
        tw_barrier();                                                                                                   \
        TW_FETCH(1 - (PAR_));                                                                                           \
    } while (0)
    if (nsteps > 0) {
        TW_FETCH(0);
        TW_COMMIT(0, tws);
        tw_barrier();
        TW_FETCH(1);
        TW_FETCH(0);
    }
    int s = 0;
    for (; s + 1 < nsteps; s += 2) {
        TW_BODY(s, 0);
        TW_BODY(s + 1, 1);
    }
    if (s < nsteps) TW_BODY(s, 0);
#undef TW_BODY
#undef TW_COMMIT
#undef TW_FETCH
    // C/D layout of the 16x16 MFMA: column (k) = lane & 15, rows (n) 4 * (lane >> 4) + r.  Partial tile: [n][64] + bias row.
    float *dst = a.partial + P.part_off + (int64_t)split * TW_PART;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < JB; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(16 * NB * wave + 16 * i + 4 * fq + r) * TW_K + 16 * j + fr] = acc[i][j][r];
    // column sums: lanes fr, fr + 16, fr + 32, fr + 48 hold disjoint rows of one column
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        float b = bsum[i];
        b += __shfl_xor(b, 16, 64);
        b += __shfl_xor(b, 32, 64);
        if (fq == 0 && P.kind != 2) dst[TW_N * TW_K + 16 * NB * wave + 16 * i + fr] = b;
    }
    ysum += __shfl_xor(ysum, 16, 64);
    ysum += __shfl_xor(ysum, 32, 64);
    if (P.kind == 2 && wave == 0 && fq == 0) dst[TW_N * TW_K + fr] = ysum;
}

// Workgroup id = 8 * local + xcd -> (tile, split).  Split s of every tile runs on XCD s % 8.  The 192 x 64 tiles take the first
// `local0` locals SPLIT-MAJOR (local = group of 8 splits * n0 + tile): the tiles of one split -- which walk the same rows in the same
// order, four of them (the context tiles) at the same speed -- are dispatched back to back on one XCD and find the X rows one of them
// fetched in that XCD's L2 (tile-major order, round 2-5: the tiles of a split started a whole wave of workgroups apart and every one
// of them read X from memory: 1.33 GB fetched for 0.63 GB of operands).  The narrow tiles follow tile-major, [local_begin, + nsplit / 8).
// The tile descriptor is read with a computed offset straight out of the kernel-argument segment (TwArgs is the only argument: offset 0;
// scalar loads from constant memory) -- indexing the by-value argument itself would make hipcc copy all 3 KB of it to scratch.
__device__ __forceinline__ bool tw_locate(const TwArgs &a, TwTile &P, int &split) {
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const bool shared = local < a.local0;
    const int q = shared ? local / a.n0 : 0;
    int w = shared ? local - q * a.n0 : 0;
    if (!shared) {
#pragma unroll
        for (int i = 1; i < TW_MAXTILES; ++i)
            if (i < a.ntiles && a.tile[i].local_begin >= 0 && local >= a.tile[i].local_begin) w = i;
    }
    typedef __attribute__((address_space(4))) const char kchar;
    kchar *ka = (kchar *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(TwArgs, tile) + (size_t)w * sizeof(TwTile);
    __builtin_memcpy(&P, ka, sizeof(TwTile));
    split = shared ? q * 8 + xcd : (local - P.local_begin) * 8 + xcd;
    return split < P.nsplit;
}

__global__ void __launch_bounds__(256, 3) tn_wide_kernel(TwArgs a) {
    extern __shared__ __attribute__((aligned(16))) float tws[];
    TwTile P; int split;
    if (!tw_locate(a, P, split)) return;
    if (P.kind == 0) tw_run<3, false>(a, P, tws, split);
    else if (P.kind == 1) tw_run<3, true>(a, P, tws, split);
    else tw_run<1, true>(a, P, tws, split);
}

// the same launch with the 192 x 64 tiles in the split bf16 form (58 KB of LDS, two workgroups per CU)
__global__ void __launch_bounds__(256, 2) tn_wide_split_kernel(TwArgs a) {
    extern __shared__ __attribute__((aligned(16))) float tws[];
    TwTile P; int split;
    if (!tw_locate(a, P, split)) return;
    if (TW_DBG(16) && P.kind == 0) return;   // narrow tiles alone
    if (TW_DBG(32) && P.kind != 0) return;   // 192 x 64 tiles alone
    if (TW_DBG(64) && !(P.kind == 0 && P.y_bf16)) return;   // context tiles alone
    if (P.kind == 0) {
        const bool bias = P.bias_out != nullptr;
        if (P.y_bf16) { if (bias) tw_run<3, false, true, true, true>(a, P, tws, split); else tw_run<3, false, true, true, false>(a, P, tws, split); }
        else { if (bias) tw_run<3, false, true, false, true>(a, P, tws, split); else tw_run<3, false, true, false, false>(a, P, tws, split); }
    } else if (P.kind == 1) tw_run<3, true>(a, P, tws, split);
    else tw_run<1, true>(a, P, tws, split);
}

// grid (TW_N * TW_K / 4 / 64 + 1, ntiles), 256 threads = 64 float4 columns x 4 groups of splits (see wgrad_tr_reduce_kernel)
__global__ void __launch_bounds__(256) tn_wide_reduce_kernel(TwArgs a) {
    __shared__ float4 comb[3][64];
    const int tile = blockIdx.y;
    TwTile P = a.tile[0];
#pragma unroll
    for (int i = 1; i < TW_MAXTILES; ++i)
        if (tile == i) P = a.tile[i];
    const float *src = a.partial + P.part_off;
    const int nsplit = P.nsplit;
    const int nrows = P.kind == 2 ? 64 : TW_N, ncols = P.kind == 0 ? TW_K : P.ny;
    if (blockIdx.x == gridDim.x - 1) {
        const int nb = P.kind == 2 ? P.ny : TW_N;
        if (P.bias_out == nullptr || (int)threadIdx.x >= nb) return;
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        int sp = 0;
        for (; sp + 3 < nsplit; sp += 4)
#pragma unroll
            for (int q = 0; q < 4; ++q) t[q] += src[(int64_t)(sp + q) * TW_PART + TW_N * TW_K + threadIdx.x];
        for (; sp < nsplit; ++sp) t[0] += src[(int64_t)sp * TW_PART + TW_N * TW_K + threadIdx.x];
        P.bias_out[threadIdx.x] = (t[0] + t[1]) + (t[2] + t[3]);
        return;
    }
    const int col = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int e = (blockIdx.x * 64 + col) * 4;
    const int n = e / TW_K, k = e % TW_K;
    float4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool live = n < nrows && k < ncols;
    if (live) {
        int sp = g;
        for (; sp + 12 < nsplit; sp += 16) {
            float4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = *(const float4 *)(src + (int64_t)(sp + 4 * q) * TW_PART + e);
#pragma unroll
            for (int q = 0; q < 4; ++q) { acc[q].x += v[q].x; acc[q].y += v[q].y; acc[q].z += v[q].z; acc[q].w += v[q].w; }
        }
        for (; sp < nsplit; sp += 4) {
            const float4 v = *(const float4 *)(src + (int64_t)sp * TW_PART + e);
            acc[0].x += v.x; acc[0].y += v.y; acc[0].z += v.z; acc[0].w += v.w;
        }
    }
    float4 t;
    t.x = (acc[0].x + acc[1].x) + (acc[2].x + acc[3].x); t.y = (acc[0].y + acc[1].y) + (acc[2].y + acc[3].y);
    t.z = (acc[0].z + acc[1].z) + (acc[2].z + acc[3].z); t.w = (acc[0].w + acc[1].w) + (acc[2].w + acc[3].w);
    if (g > 0) comb[g - 1][col] = t;
    __syncthreads();
    if (g == 0 && live) {
#pragma unroll
        for (int q = 0; q < 3; ++q) { const float4 u = comb[q][col]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
        const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (k + c >= ncols) continue;
            // kind 2 holds out^T: partial row = Y' ... = column of the destination
            if (P.kind == 2) P.out[(int64_t)(k + c) * P.ldo + P.col_off + n] = tv[c];
            else P.out[(int64_t)n * P.ldo + P.col_off + k + c] = tv[c];
        }
    }
}

// ---- planning -----------------------------------------------------------------------------------------------------------
// kind of a problem by its shapes only (the same answer for the sizing call with dummy pointers and for the launch):
//   0: NX = 192, NY % 64 == 0 (NY / 64 tiles)   1: NX = 192, NY <= 16   2: NX <= 64, NY = 64 (roles swapped; one tile per 16 columns
//   of the narrow operand: the emission rows of state dimensions 5..9 are 20..54 wide)   -1: not ours
static int tw_kind(const TnProblem &q) {
    if (q.NX == TW_N && q.NY > 0 && q.NY % TW_K == 0) return 0;
    if (q.NX == TW_N && q.NY > 0 && q.NY <= 16) return 1;
    if (q.NX > 0 && q.NX <= 64 && q.NY == 64) return 2;
    return -1;
}
static int tw_ntiles(const TnProblem &q, int kind) { return kind == 0 ? q.NY / TW_K : (kind == 2 ? (q.NX + 15) / 16 : 1); }
static bool tw_plan_shapes(const TnProblem *probs, int nprob, int &tiles) {
    tiles = 0;
    bool any_wide = false;
    for (int i = 0; i < nprob; ++i) {
        const int k = tw_kind(probs[i]);
        if (k < 0) return false;
        tiles += tw_ntiles(probs[i], k);
        any_wide |= k == 0;
    }
    return any_wide && tiles <= TW_MAXTILES;
}
static bool tw_split_form() {
    static int on = -1;
    if (on < 0) on = (int)vsde_knob("VSDE_TW_SPLIT", 1);
    return on != 0;
}
// Row splits per tile, in multiples of 8 (one split per XCD and `local`), adding up to EXACTLY `wgs` workgroups = whole rounds of the
// resident set.  fp32 form: three workgroups per CU, 1536 = two rounds (measured 488 | 447 | 429 us for 768 | 1024 | 1536 at LV).  Split
// form: two per CU, 512 = one round (225 + 8 us with the reduction against 238 + 17 for two rounds: profiles/r06_tn_wide_split.txt; the
// proportional rounding of rounds 2-5 would have given 520 workgroups, i.e. a second round for 8 of them).  The 192 x 64 tiles get
// equal splits (tw_locate relies on it); the narrow tiles share what is left by their time per row relative to a 192 x 64 tile (w1 |
// w2 per cent: a 192 x 16 tile issues a quarter of the MFMAs, the swapped 64 x 16 tile a twelfth, all stage the same rows; in the
// split form they still run on the fp32 instruction).  VSDE_TW_WGS / _W1 / _W2 override (ablation build).
static void tw_splits(const int *kinds, int nt, int64_t chunks, int *nsplit) {
    static int wgs = -1, w1 = -1, w2 = -1;
    if (wgs < 0) {
        wgs = (int)vsde_knob("VSDE_TW_WGS", tw_split_form() ? 512 : 1536);
        w1 = (int)vsde_knob("VSDE_TW_W1", tw_split_form() ? 70 : 35);
        w2 = (int)vsde_knob("VSDE_TW_W2", tw_split_form() ? 40 : 20);
    }
    const float w[3] = {1.0f, 0.01f * w1, 0.01f * w2};
    float tot = 0.f, narrow = 0.f;
    int nwide = 0;
    for (int i = 0; i < nt; ++i) { tot += w[kinds[i]]; if (kinds[i] == 0) ++nwide; else narrow += w[kinds[i]]; }
    const int wide_n = nwide ? ((int)(wgs / tot)) & ~7 : 0;
    int left = wgs - nwide * wide_n;           // for the narrow tiles, handed out in order; the last one takes the remainder
    int last_narrow = -1;
    for (int i = 0; i < nt; ++i) if (kinds[i] != 0) last_narrow = i;
    for (int i = 0; i < nt; ++i) {
        int n;
        if (kinds[i] == 0) n = wide_n;
        else if (i == last_narrow) n = left & ~7;
        else { n = ((int)(left * w[kinds[i]] / narrow + 4.f)) & ~7; left -= n; narrow -= w[kinds[i]]; }
        if (n < 8) n = 8;
        if (n > chunks) n = (int)chunks;
        nsplit[i] = n;
    }
}
static int tw_tile_kinds(const TnProblem *probs, int nprob, int *kinds) {
    int nt = 0;
    for (int i = 0; i < nprob; ++i) {
        const int k = tw_kind(probs[i]);
        for (int kt = 0; kt < tw_ntiles(probs[i], k); ++kt) kinds[nt++] = k;
    }
    return nt;
}

size_t tn_wide_workspace_bytes(const TnProblem *probs, int nprob, int M) {
    int tiles;
    if (!tw_plan_shapes(probs, nprob, tiles)) return 0;
    int kinds[TW_MAXTILES], nsplit[TW_MAXTILES];
    const int nt = tw_tile_kinds(probs, nprob, kinds);
    tw_splits(kinds, nt, ((int64_t)M + TW_BM - 1) / TW_BM, nsplit);
    size_t tot = 0;
    for (int i = 0; i < nt; ++i) tot += (size_t)nsplit[i];
    return tot * TW_PART * sizeof(float);
}

static bool tw_vec_view(const RowView &V, int ncols, int align_bytes) {
    return (uintptr_t)V.base % align_bytes == 0 && V.batch_stride % 4 == 0 && V.row_stride % 4 == 0 && V.col_split >= ncols;
}

// returns 1 when the fast path ran, 0 when the caller has to use the generic kernel, < 0 on error
int launch_tn_wide(const TnProblem *probs, int nprob, int M, void *workspace, size_t workspace_bytes, hipStream_t stream) {
    int tiles;
    if (!tw_plan_shapes(probs, nprob, tiles)) return 0;
    const int T = probs[0].X.rows_per_batch;
    if (T < TW_BM || M % TW_BM != 0) return 0;   // a block must not span more than two batch rows; no partial blocks
    TwArgs a = {};
    int nt = 0;
    for (int i = 0; i < nprob; ++i) {
        const TnProblem &q = probs[i];
        if (q.X.rows_per_batch != T || q.Y.rows_per_batch != T) return 0;
        const int kind = tw_kind(q);
        // kind 2 computes out^T: the 64-wide operand plays X, the narrow one Y
        const RowView &X = kind == 2 ? q.Y : q.X, &Y = kind == 2 ? q.X : q.Y;
        const int nx = kind == 2 ? 64 : TW_N, ny = kind == 2 ? q.NX : q.NY;
        // operand views the branch-free loads can take: X in 16-byte fp32 chunks (a (dr, du | dc_n) remap moves whole chunks)
        if (X.dtype != 0 || (uintptr_t)X.base % 16 || X.batch_stride % 4 || X.row_stride % 4 || X.shift != 0 || Y.shift > 0) return 0;
        if (X.col_split < nx && (X.col_split % 4 || X.col_skip % 4)) return 0;
        if (kind == 0 ? !tw_vec_view(Y, ny, Y.dtype == 0 ? 16 : 8) : (Y.dtype != 0 || Y.col_split < ny)) return 0;
        for (int kt = 0; kt < tw_ntiles(q, kind); ++kt) {
            TwTile &t = a.tile[nt++];
            // kind 0: tile kt = columns 64 kt.. of the wide Y; kind 2: tile kt = columns 16 kt.. of the narrow operand = rows of out
            const int yoff = kind == 0 ? kt * TW_K : (kind == 2 ? kt * 16 : 0);
            t.kind = kind; t.ny = kind == 2 ? (ny - yoff < 16 ? ny - yoff : 16) : ny;
            t.x = (const float *)X.base; t.xbs = X.batch_stride; t.xrs = X.row_stride; t.xshift = X.shift;
            t.x_split = X.col_split < nx ? X.col_split : nx; t.x_skip = X.col_split < nx ? X.col_skip : 0;
            t.y_bf16 = Y.dtype != 0;
            t.y = Y.dtype == 0 ? (const void *)((const float *)Y.base + yoff) : (const void *)((const uint16_t *)Y.base + yoff);
            t.ybs = Y.batch_stride; t.yrs = Y.row_stride; t.yshift = Y.shift;
            if (kind == 2) {
                t.out = q.out + (int64_t)yoff * q.ldo; t.ldo = q.ldo; t.col_off = q.col_off;
                t.bias_out = q.bias_out ? q.bias_out + yoff : nullptr;
            } else {
                t.out = q.out; t.ldo = q.ldo; t.col_off = q.col_off + kt * TW_K; t.bias_out = kt == 0 ? q.bias_out : nullptr;
            }
        }
    }
    a.ntiles = nt; a.M = M; a.T = T; a.chunks = ((int64_t)M + TW_BM - 1) / TW_BM;
    // the 192 x 64 tiles (equal splits: tw_splits) come first and share the first locals split-major, the others follow tile-major
    static int interleave = -1;
    if (interleave < 0) interleave = (int)vsde_knob("VSDE_TW_INTERLEAVE", 1);
    int n0 = 0;
    if (interleave) {
        TwTile sorted[TW_MAXTILES];
        int k = 0;
        for (int i = 0; i < nt; ++i)
            if (a.tile[i].kind == 0) sorted[k++] = a.tile[i];
        n0 = k;
        for (int i = 0; i < nt; ++i)
            if (a.tile[i].kind != 0) sorted[k++] = a.tile[i];
        for (int i = 0; i < nt; ++i) a.tile[i] = sorted[i];
    }
    int kinds[TW_MAXTILES], nsplit[TW_MAXTILES];
    for (int i = 0; i < nt; ++i) kinds[i] = a.tile[i].kind;
    tw_splits(kinds, nt, a.chunks, nsplit);
    int local = n0 > 0 ? n0 * ((nsplit[0] + 7) / 8) : 0;
    int64_t part = 0;
    a.n0 = n0; a.local0 = local;
#ifdef VSDE_ABLATIONS
    a.dbg = (int)vsde_knob("VSDE_TW_DBG", 0);
#endif
    for (int i = 0; i < nt; ++i) {
        a.tile[i].nsplit = nsplit[i]; a.tile[i].part_off = part;
        part += (int64_t)nsplit[i] * TW_PART;
        if (i < n0) { a.tile[i].local_begin = -1; continue; }
        a.tile[i].local_begin = local;
        local += (nsplit[i] + 7) / 8;
    }
    a.nlocal = local;
    const size_t need = (size_t)part * sizeof(float);
    VSDE_CHECK_ARG(workspace_bytes >= need, VSDE_E_WORKSPACE, "TN workspace too small: %zu < %zu", workspace_bytes, need);
    a.partial = (float *)workspace;
    if (tw_split_form()) {
        static_assert(TS_LDS >= 2 * TW_BUF * (int)sizeof(float), "the narrow tiles of the launch use the fp32 buffers");
        VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)tn_wide_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TS_LDS));
        hipLaunchKernelGGL(tn_wide_split_kernel, dim3((unsigned)(a.nlocal * 8)), dim3(256), TS_LDS, stream, a);
    } else {
        const size_t lds = (size_t)2 * TW_BUF * sizeof(float);
        VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)tn_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(tn_wide_kernel, dim3((unsigned)(a.nlocal * 8)), dim3(256), lds, stream, a);
    }
    VSDE_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(tn_wide_reduce_kernel, dim3(TW_N * TW_K / 4 / 64 + 1, nt), dim3(256), 0, stream, a);
    VSDE_CHECK_HIP(hipGetLastError());
    return 1;
}

}  // namespace vsde
