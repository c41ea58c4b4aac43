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
};

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    f32v2 f = {a, b};
    bf16v2 r = __builtin_convertvector(f, bf16v2);  // v_cvt_pk_bf16_f32
    return *(uint32_t *)&r;
}

__global__ void __launch_bounds__(768) attn_fwd_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) uint16_t asmem[];
    uint16_t *Ks = asmem;                      // [npad][AT_KLD]
    uint16_t *Vt = asmem + p.npad * AT_KLD;    // [64][vld]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / p.H, hh = blockIdx.x - b * p.H;
    const int N = p.N, npad = p.npad, vld = p.vld;
    const int64_t ts = (int64_t)p.H * AT_D;    // token stride in elements
    const int64_t base = ((int64_t)b * N * p.H + hh) * AT_D;
    const uint16_t *qb = p.q + base, *kb = p.k + base, *vb = p.v + base;
    // ---- stage K [key][d] and V^T [d][key]: every global load of the workgroup is issued before the first use -----
    constexpr int NT = 768, NW = NT / 64;
    constexpr int KIT = (AT_MAXN * 8 + NT - 1) / NT, VIT = (AT_MAXN + NT - 1) / NT;
    uint4 kreg[KIT], vreg[VIT][8];
#pragma unroll
    for (int it = 0; it < KIT; ++it) {
        const int i = tid + it * NT, n = i >> 3, c = i & 7;
        kreg[it] = (i < npad * 8 && n < N) ? *(const uint4 *)(kb + n * ts + c * 8) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int it = 0; it < VIT; ++it) {  // (npad/8 key blocks) x (8 d-chunks); lanes c = 0..7 read one full 128-byte row
        const int i = tid + it * NT, kblk = i >> 3, c = i & 7;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = kblk * 8 + j;
            vreg[it][j] = (i < npad && n < N) ? *(const uint4 *)(vb + n * ts + c * 8) : make_uint4(0, 0, 0, 0);
        }
    }
    float kss_max = 0.f;  // max_j |k_j|^2 (8 adjacent lanes hold one key row)
#pragma unroll
    for (int it = 0; it < KIT; ++it) {
        const int i = tid + it * NT, n = i >> 3, c = i & 7;
        if (i < npad * 8) *(uint4 *)(Ks + n * AT_KLD + c * 8) = kreg[it];
        const uint32_t w[4] = {kreg[it].x, kreg[it].y, kreg[it].z, kreg[it].w};
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float lo = __uint_as_float(w[e] << 16), hi = __uint_as_float(w[e] & 0xffff0000u);
            ss = fmaf(lo, lo, fmaf(hi, hi, ss));
        }
        ss += __shfl_xor(ss, 1, 64); ss += __shfl_xor(ss, 2, 64); ss += __shfl_xor(ss, 4, 64);
        kss_max = fmaxf(kss_max, ss);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) kss_max = fmaxf(kss_max, __shfl_xor(kss_max, off, 64));
    __shared__ float kred[NW];
    if (lane == 0) kred[wave] = kss_max;
#pragma unroll
    for (int it = 0; it < VIT; ++it) {
        const int i = tid + it * NT, kblk = i >> 3, c = i & 7;
        uint4 ct[8];
        transpose8x8(vreg[it], ct);
        if (i < npad) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {  // row d = c*8+j holds keys kblk*8 .. +7; rows are only 8-byte aligned (vld = 4 mod 8)
                uint2 *dst = (uint2 *)(Vt + (c * 8 + j) * vld + kblk * 8);
                dst[0] = make_uint2(ct[j].x, ct[j].y);
                dst[1] = make_uint2(ct[j].z, ct[j].w);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NW; ++w) kss_max = fmaxf(kss_max, kred[w]);
    const float kmax = sqrtf(kss_max);

    const int fr = lane & 31, h2 = lane >> 5;
    const int nkt = npad >> 5, nqb = npad >> 5;
    const bool ragged = (N & 31) != 0;
    for (int qblk = wave; qblk < nqb; qblk += NW) {
        const int query = qblk * 32 + fr;
        const bool qok = query < N;
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
        qss += __shfl_xor(qss, 32, 64);
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
        if (exact) mx = fmaxf(mx, __shfl_xor(mx, 32, 64));  // the other half-wave holds the other 16 keys of every tile
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
            uint2 vf[8];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const uint16_t *vrow = Vt + (dt * 32 + fr) * vld + kt * 32 + 4 * h2;
#pragma unroll
                for (int x = 0; x < 4; ++x) vf[dt * 4 + x] = *(const uint2 *)(vrow + 8 * x);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (ragged && kt == nkt - 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2 >= N) scur[r] = -INFINITY;
            }
            float pr[16];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                pr[r] = fast_exp2(fmaf(scur[r], c2, -mc)); lsum += pr[r];
                pr[r + 1] = fast_exp2(fmaf(scur[r + 1], c2, -mc)); lsum2 += pr[r + 1];
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
        lsum += __shfl_xor(lsum, 32, 64);
        if (qok) {
            const float inv = 1.0f / lsum;
            uint16_t *orow = p.o + base + query * ts;
#pragma unroll
            for (int g = 0; g < 4; ++g) {  // registers 4g..4g+3 = rows d = 8g + 4h2 + 0..3 (o0) and 32 + ... (o1)
                const int d0 = 8 * g + 4 * h2;
                *(uint2 *)(orow + d0) = make_uint2(pack_bf16(o0[4 * g] * inv, o0[4 * g + 1] * inv), pack_bf16(o0[4 * g + 2] * inv, o0[4 * g + 3] * inv));
                *(uint2 *)(orow + 32 + d0) = make_uint2(pack_bf16(o1[4 * g] * inv, o1[4 * g + 1] * inv), pack_bf16(o1[4 * g + 2] * inv, o1[4 * g + 3] * inv));
            }
            if (h2 == 0) p.lse[((int64_t)b * p.H + hh) * N + query] = mx * p.scale + __logf(lsum);
        }
    }
}

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

struct AttnBwdParams {
    const uint16_t *q, *k, *v, *o, *dout;  // [B][N][H][64] bf16
    const float *lse;                      // [B][H][N]
    float *delta;                          // [B][H][N]  D_i = <dO_i, O_i>   (written by the dq kernel, read by dk/dv)
    uint16_t *dq, *dk, *dv;                // [B][N][H][64] bf16
    int N, H, ntile;                       // ntile = ceil(N / 32)
    float scale, scale_log2e;
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

// ---- staging of whole row-major operands [npad][64] -> LDS [npad][AT_KLD]; all global loads issued before first use ---
constexpr int AT_BT = 768;                                 // threads of a backward workgroup (12 waves: 3 per SIMD)
constexpr int AT_SIT = (AT_MAXN * 8 + AT_BT - 1) / AT_BT;  // 16-byte chunks per thread and operand

__device__ __forceinline__ void stage_two(const uint16_t *a, const uint16_t *b2, int64_t ts, int N, int npad, int tid,
                                          uint16_t *sa, uint16_t *sb) {
    uint4 ra[AT_SIT], rb[AT_SIT];
#pragma unroll
    for (int it = 0; it < AT_SIT; ++it) {
        const int i = tid + it * AT_BT, n = i >> 3, c = i & 7;
        const bool ok = i < npad * 8 && n < N;
        ra[it] = ok ? *(const uint4 *)(a + n * ts + c * 8) : make_uint4(0, 0, 0, 0);
        rb[it] = ok ? *(const uint4 *)(b2 + n * ts + c * 8) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int it = 0; it < AT_SIT; ++it) {
        const int i = tid + it * AT_BT, n = i >> 3, c = i & 7;
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
__global__ void __launch_bounds__(768) attn_bwd_dq_kernel(AttnBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) uint16_t asmem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 31, h2 = lane >> 5;
    const int b = blockIdx.x / p.H, hh = blockIdx.x - b * p.H, N = p.N, npad = p.ntile * 32;
    uint16_t *Ks = asmem, *Vs = asmem + npad * AT_KLD;
    const int64_t ts = (int64_t)p.H * AT_D, base = ((int64_t)b * N * p.H + hh) * AT_D;
    const int64_t srow = ((int64_t)b * p.H + hh) * N;
    bf16x8 qn[4], don[4], on[4];  // fragments of the NEXT round
    float lsen;
    auto request = [&](int qblk) {
        const int query = qblk * 32 + fr;
        const bool ok = qblk < p.ntile && query < N;
        load_bfrag(p.q + base, ts, query, ok, h2, qn);
        load_bfrag(p.dout + base, ts, query, ok, h2, don);
        load_bfrag(p.o + base, ts, query, ok, h2, on);
        lsen = ok ? p.lse[srow + query] : INFINITY;  // padded queries: P = 0
    };
    request(wave);
    stage_two(p.k + base, p.v + base, ts, N, npad, tid, Ks, Vs);
    __syncthreads();
    const float c2 = p.scale_log2e;
    const bool ragged = (N & 31) != 0;
    for (int qblk = wave; qblk < p.ntile; qblk += AT_BT / 64) {
        if (qblk != wave) request(qblk);  // later rounds are rare with 12 waves (N <= 384 needs none): fetch on demand
        const int query = qblk * 32 + fr;
        const bool qok = query < N;
        bf16x8 qf[4], dof[4];
        float dsum = 0.f;  // D_i = <dO_i, O_i>: this lane holds half of the 64 channels of its query
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qf[ks] = qn[ks]; dof[ks] = don[ks];
#pragma unroll
            for (int e = 0; e < 8; ++e)
                dsum = fmaf(__uint_as_float(((uint32_t)(uint16_t)don[ks][e]) << 16), __uint_as_float(((uint32_t)(uint16_t)on[ks][e]) << 16), dsum);
        }
        dsum += __shfl_xor(dsum, 32, 64);
        const float lse2 = lsen * 1.4426950408889634f;
        if (qok && h2 == 0) p.delta[srow + query] = dsum;
        f32x16 acc0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, acc1 = acc0;
#pragma unroll 1
        for (int kt = 0; kt < p.ntile; ++kt) {
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
        if (qok) store_transposed(p.dq + base + query * ts, h2, acc0, acc1, p.scale);
    }
}

// dk, dv: one workgroup per (batch, head); Q, dO, lse and delta resident in LDS, a wavefront owns 32 keys at a time.
__global__ void __launch_bounds__(768) attn_bwd_dkv_kernel(AttnBwdParams p) {
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
    stage_two(p.q + base, p.dout + base, ts, N, npad, tid, Qs, Os);
    for (int i = tid; i < npad; i += AT_BT) {  // padded queries: lse = +inf -> P = 0
        lse2s[i] = i < N ? p.lse[srow + i] * 1.4426950408889634f : INFINITY;
        dels[i] = i < N ? p.delta[srow + i] : 0.f;
    }
    __syncthreads();
    const float c2 = p.scale_log2e;
    for (int kblk = wave; kblk < p.ntile; kblk += AT_BT / 64) {
        if (kblk != wave) request(kblk);
        const int key = kblk * 32 + fr;
        const bool kok = key < N;
        bf16x8 kf[4], vf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { kf[ks] = kn[ks]; vf[ks] = vn[ks]; }
        f32x16 dk0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, dk1 = dk0, dv0 = dk0, dv1 = dk0;
#pragma unroll 1
        for (int qt = 0; qt < p.ntile; ++qt) {
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
        if (kok) {
            store_transposed(p.dk + base + key * ts, h2, dk0, dk1, p.scale);
            store_transposed(p.dv + base + key * ts, h2, dv0, dv1, 1.0f);
        }
    }
}

}  // namespace vsde

using namespace vsde;

// VSDE_ATTN_STREAM=1: take the streamed kernels for every shape (A/B runs)
static bool force_stream() {
    static int f = -1;
    if (f < 0) { const char *e = getenv("VSDE_ATTN_STREAM"); f = e ? atoi(e) : 0; }
    return f != 0;
}

extern "C" int vsde_attention_max_tokens(void) { return AT_MAXN; }

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
    const size_t lds = ((size_t)p.npad * AT_KLD + (size_t)AT_D * p.vld) * sizeof(uint16_t);
    VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(attn_fwd_kernel, dim3((unsigned)(B * H)), dim3(768), lds, (hipStream_t)stream, p);
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
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
    AttnBwdParams p;
    p.q = (const uint16_t *)q; p.k = (const uint16_t *)k; p.v = (const uint16_t *)v; p.o = (const uint16_t *)o;
    p.dout = (const uint16_t *)dout; p.lse = lse; p.delta = delta;
    p.dq = (uint16_t *)dq; p.dk = (uint16_t *)dk; p.dv = (uint16_t *)dv;
    p.N = N; p.H = H; p.ntile = (N + 31) / 32;
    p.scale = (float)scale; p.scale_log2e = (float)(scale * 1.4426950408889634);
    const size_t lds_dq = (size_t)2 * p.ntile * 32 * AT_KLD * sizeof(uint16_t), lds_dkv = lds_dq + (size_t)2 * p.ntile * 32 * sizeof(float);
    VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)attn_bwd_dq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dq));
    VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)attn_bwd_dkv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dkv));
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3((unsigned)(B * H)), dim3(AT_BT), lds_dq, s, p);   // also writes delta, read by the next kernel
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3((unsigned)(B * H)), dim3(AT_BT), lds_dkv, s, p);
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}
