// Weight / bias gradients of the encoder's Linear layers on bf16 MFMA (gfx950).
//
//   dW[n][k] = sum_m dY[m][n] * X[m][k]        db[n] = sum_m dY[m][n]          (M = B * tokens ~ 2e5)
//
// hipBLASLt runs these "reduce over a huge M into a small N x K" GEMMs at 15-270 TF/s (0.45-0.6 ms each
// at the LV shapes whatever their size) and torch adds a separate bf16 column-sum kernel for the bias.
// They are HBM-bound (read dY and X once): this kernel splits M over the grid, every workgroup keeps a
// 128 x 128 fp32 output tile in MFMA accumulators (v_mfma_f32_32x32x16_bf16), stages 64 rows of both
// operands per iteration through LDS with an in-register 8x8 bf16 transpose (the MFMA wants the reduction
// index contiguous per lane, memory has it as the slow index), prefetches the next 64 rows into registers
// during the MFMAs, accumulates the column sums of dY from the staging registers, and a second kernel sums
// the split partials in a fixed order (deterministic; fp32 results, better than the bf16 outputs torch
// produces under autocast).
#include "vsde_common.h"

namespace vsde {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr int WG_BM = 64;          // rows of M per staging step
constexpr int WG_LD = WG_BM + 8;   // LDS row stride in bf16 elements (144 B: conflict-free ds_read_b128)

struct WgradParams {
    const uint16_t *dy;  // [M][N] bf16
    const uint16_t *x;   // [M][K] bf16
    int64_t M;
    int N, K;
    int tile;            // output tile edge: 128 (256 threads) or 256 (512 threads)
    int tiles_k;         // ceil(K / tile)
    int nsplit;
    int64_t rows_per_split;  // multiple of WG_BM
    float *partial;      // [tiles][nsplit][tile*tile + tile]
    float *dW;           // [N][K]
    float *db;           // [N] or nullptr
};

// 8 rows x 8 bf16 (row i in r[i], 4 dwords) -> 8 columns x 8 bf16 (column j in c[j]: rows 0..7)
__device__ __forceinline__ void transpose8x8(const uint4 (&r)[8], uint4 (&c)[8]) {
    const uint32_t *rr = (const uint32_t *)r;
    uint32_t *cc = (uint32_t *)c;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const uint32_t a = rr[(2 * p) * 4 + (j >> 1)], b = rr[(2 * p + 1) * 4 + (j >> 1)];
            cc[j * 4 + p] = (j & 1) ? __builtin_amdgcn_perm(b, a, 0x07060302u) : __builtin_amdgcn_perm(b, a, 0x05040100u);
        }
}

// T x T output tile per workgroup of 2T threads: T=128 -> 4 waves of 64x64, T=256 -> 8 waves of 64(n) x 128(k).
// The larger tile halves the operand re-reads (dY is re-read K/T times, X N/T times).
template <int T>
__global__ void __launch_bounds__(2 * T) wgrad_bf16_kernel(WgradParams p) {
    constexpr int WK = T == 128 ? 64 : 128;  // wave sub-tile width along k
    constexpr int NB = WK / 32;               // MFMA tiles along k per wave
    extern __shared__ __attribute__((aligned(16))) uint16_t wsm[];
    uint16_t *At = wsm;                 // dY^T tile: [n][m]
    uint16_t *Bt = wsm + T * WG_LD;     // X^T  tile: [k][m]
    float *bred = (float *)(wsm + 2 * T * WG_LD);  // [8][T]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int split = blockIdx.x, tile = blockIdx.y;
    const int n_blk = (tile / p.tiles_k) * T, k_blk = (tile % p.tiles_k) * T;
    const bool want_bias = p.db != nullptr && k_blk == 0;
    // staging role: first T threads load dY, the other T load X; each an 8(m) x 8(col) block
    const bool is_a = tid < T;
    // adjacent lanes take adjacent m-blocks of the same column chunk: their 16-byte LDS stores fall into one
    // 128-byte row segment (conflict-free) and every global load instruction covers 8 rows x 128 contiguous bytes
    const int st = is_a ? tid : tid - T, mblk = st & 7, cch = st >> 3;
    const uint16_t *src = is_a ? p.dy : p.x;
    const int ld = is_a ? p.N : p.K;
    const int col0 = (is_a ? n_blk : k_blk) + cch * 8;
    const bool col_ok = col0 < ld;  // N, K are multiples of 8
    uint16_t *dst = (is_a ? At : Bt) + (cch * 8) * WG_LD + mblk * 8;

    const int64_t m_begin = (int64_t)split * p.rows_per_split;
    const int64_t m_end = m_begin + p.rows_per_split < p.M ? m_begin + p.rows_per_split : p.M;
    uint4 rows[8];
    auto fetch = [&](int64_t m0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t m = m0 + mblk * 8 + i;
            rows[i] = (col_ok && m < m_end) ? *(const uint4 *)(src + m * ld + col0) : make_uint4(0, 0, 0, 0);
        }
    };
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 acc[2][NB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    const int wn = (T == 128 ? (wave >> 1) : (wave >> 1)) * 64, wk = (wave & 1) * WK;  // this wave's sub-tile
    const int fr = lane & 31, fh = lane >> 5;

    if (m_begin < m_end) fetch(m_begin);
    for (int64_t m0 = m_begin; m0 < m_end; m0 += WG_BM) {
        uint4 cols[8];
        transpose8x8(rows, cols);
        if (want_bias && is_a) {
            const uint32_t *rr = (const uint32_t *)rows;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint32_t w = rr[i * 4 + (j >> 1)];
                    bsum[j] += __uint_as_float((j & 1) ? (w & 0xffff0000u) : (w << 16));
                }
        }
        __syncthreads();  // previous tile consumed
#pragma unroll
        for (int j = 0; j < 8; ++j) *(uint4 *)(dst + j * WG_LD) = cols[j];
        __syncthreads();
        if (m0 + WG_BM < m_end) fetch(m0 + WG_BM);  // in flight during the MFMAs
#pragma unroll
        for (int ks = 0; ks < WG_BM / 16; ++ks) {
            bf16x8 af[2], bf[NB];
#pragma unroll
            for (int a = 0; a < 2; ++a) af[a] = *(const bf16x8 *)(At + (wn + 32 * a + fr) * WG_LD + ks * 16 + fh * 8);
#pragma unroll
            for (int b = 0; b < NB; ++b) bf[b] = *(const bf16x8 *)(Bt + (wk + 32 * b + fr) * WG_LD + ks * 16 + fh * 8);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
    }
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float *out = p.partial + ((int64_t)tile * p.nsplit + split) * (T * T + T);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = wn + 32 * a + (e & 3) + 8 * (e >> 2) + 4 * fh, col = wk + 32 * b + fr;
                out[row * T + col] = acc[a][b][e];
            }
    if (want_bias) {
        __syncthreads();
        if (is_a)
#pragma unroll
            for (int j = 0; j < 8; ++j) bred[mblk * T + cch * 8 + j] = bsum[j];
        __syncthreads();
        if (tid < T) {
            float s = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) s += bred[g * T + tid];
            out[T * T + tid] = s;
        }
    }
}

// grid (tiles, T*T/256): block (tile, part) reduces 256 outputs (+ the bias row in part 0) over the splits
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(WgradParams p) {
    const int T = p.tile, PART = T * T + T;
    const int tile = blockIdx.x, part = blockIdx.y;
    const int n_blk = (tile / p.tiles_k) * T, k_blk = (tile % p.tiles_k) * T;
    const float *src = p.partial + (int64_t)tile * p.nsplit * PART;
    const int e = part * 256 + threadIdx.x;
    const int n = n_blk + e / T, k = k_blk + e % T;
    if (n < p.N && k < p.K) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // fixed association: deterministic
        int sp = 0;
        for (; sp + 7 < p.nsplit; sp += 8) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = src[(int64_t)(sp + q) * PART + e];  // 8 loads in flight
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] += v[q];
        }
        for (; sp < p.nsplit; ++sp) acc[0] += src[(int64_t)sp * PART + e];
        const float s0 = acc[0] + acc[1], s1 = acc[2] + acc[3], s2 = acc[4] + acc[5], s3 = acc[6] + acc[7];
        p.dW[(int64_t)n * p.K + k] = (s0 + s1) + (s2 + s3);
    }
    if (part == 0 && p.db != nullptr && k_blk == 0 && threadIdx.x < T && n_blk + threadIdx.x < p.N) {
        float s = 0.f;
        for (int sp = 0; sp < p.nsplit; ++sp) s += src[(int64_t)sp * PART + T * T + threadIdx.x];
        p.db[n_blk + threadIdx.x] = s;
    }
}

static void wgrad_plan(int64_t M, int N, int K, WgradParams &p, int &tiles) {
    p.M = M; p.N = N; p.K = K;
    p.tile = (N > 128 && K > 128) ? 256 : 128;
    p.tiles_k = (K + p.tile - 1) / p.tile;
    tiles = ((N + p.tile - 1) / p.tile) * p.tiles_k;
    const int64_t chunks = (M + WG_BM - 1) / WG_BM;
    int64_t nsplit = ((p.tile == 256 ? 512 : 1024) + tiles - 1) / tiles;  // ~2 (4) workgroups per CU
    if (nsplit < 8) nsplit = 8;
    if (nsplit > 256) nsplit = 256;
    if (nsplit > chunks) nsplit = chunks;
    const int64_t cps = (chunks + nsplit - 1) / nsplit;
    p.rows_per_split = cps * WG_BM;
    p.nsplit = (int)((chunks + cps - 1) / cps);
}

static size_t wgrad_lds_bytes(int T) { return (size_t)2 * T * WG_LD * sizeof(uint16_t) + (size_t)8 * T * sizeof(float); }

}  // namespace vsde

using namespace vsde;

extern "C" size_t vsde_linear_wgrad_workspace_bytes(int64_t M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    WgradParams p; int tiles;
    wgrad_plan(M, N, K, p, tiles);
    return (size_t)tiles * p.nsplit * (p.tile * p.tile + p.tile) * sizeof(float);
}

extern "C" int vsde_linear_wgrad_bf16(const void *dy, const void *x, int64_t M, int N, int K, float *dW, float *db,
                                      void *workspace, size_t workspace_bytes, void *stream) {
    VSDE_CHECK_ARG(dy && x && dW && workspace && M > 0, VSDE_E_BADARG, "bad linear_wgrad arguments");
    VSDE_CHECK_ARG(N % 8 == 0 && K % 8 == 0, VSDE_E_BADARG, "linear_wgrad needs N %% 8 == 0 and K %% 8 == 0 (got %d, %d)", N, K);
    VSDE_CHECK_ARG(((uintptr_t)dy % 16) == 0 && ((uintptr_t)x % 16) == 0, VSDE_E_BADARG, "linear_wgrad operands must be 16-byte aligned");
    WgradParams p; int tiles;
    wgrad_plan(M, N, K, p, tiles);
    const size_t need = (size_t)tiles * p.nsplit * (p.tile * p.tile + p.tile) * sizeof(float);
    VSDE_CHECK_ARG(workspace_bytes >= need, VSDE_E_WORKSPACE, "linear_wgrad workspace too small: %zu < %zu", workspace_bytes, need);
    p.dy = (const uint16_t *)dy; p.x = (const uint16_t *)x; p.partial = (float *)workspace; p.dW = dW; p.db = db;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = wgrad_lds_bytes(p.tile);
    if (p.tile == 256) {
        VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)wgrad_bf16_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((wgrad_bf16_kernel<256>), dim3(p.nsplit, tiles), dim3(512), lds, s, p);
    } else {
        hipLaunchKernelGGL((wgrad_bf16_kernel<128>), dim3(p.nsplit, tiles), dim3(256), lds, s, p);
    }
    VSDE_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(tiles, p.tile * p.tile / 256), dim3(256), 0, s, p);
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}
