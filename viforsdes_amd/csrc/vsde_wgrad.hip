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
    int tiles;           // output tiles in total
    int nsplit;
    int64_t rows_per_split;  // multiple of WG_BM
    float *partial;      // [tiles][nsplit][tile*tile + tile]
    float *dW;           // [N][K]
    float *db;           // [N] or nullptr
};

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
    // XCD-aware order: consecutive workgroup ids go round-robin over the 8 XCDs, so id = 8*local + xcd.  The workgroups
    // of one split (same rows of dY / X, different output tiles) are given consecutive `local` on ONE xcd: the operand
    // rows they share are then served by that XCD's L2 instead of being fetched once per tile.
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int tile = local % p.tiles, split = (local / p.tiles) * 8 + xcd;
    if (split >= p.nsplit) return;
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

// Sum of the nsplit partial tiles, in a fixed association (deterministic).  grid (T*T/4/VPB + 1, tiles), 256 threads:
// a block owns VPB = 256/G float4 vectors of one tile; thread (v, g) adds the splits g, g+G, ... (8 loads in flight),
// the G group sums are combined through LDS in order.  The extra block (last blockIdx.x) sums the tile's bias row.
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(WgradParams p, int G) {
    __shared__ float4 red[256];
    const int T = p.tile, PART = T * T + T;
    const int tile = blockIdx.y;
    const int n_blk = (tile / p.tiles_k) * T, k_blk = (tile % p.tiles_k) * T;
    const float *src = p.partial + (int64_t)tile * p.nsplit * PART;
    if (blockIdx.x == gridDim.x - 1) {  // bias row: column c, split group g
        if (p.db == nullptr || k_blk != 0) return;
        const int GB = 256 / T, c = threadIdx.x % T, g = threadIdx.x / T;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int sp = g;
        for (; sp + 7 * GB < p.nsplit; sp += 8 * GB) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = src[(int64_t)(sp + q * GB) * PART + T * T + c];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] += v[q];
        }
        for (; sp < p.nsplit; sp += GB) acc[0] += src[(int64_t)sp * PART + T * T + c];
        float *r = (float *)red;
        r[threadIdx.x] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
        __syncthreads();
        if (g == 0 && n_blk + c < p.N) {
            float t = r[c];
            for (int gg = 1; gg < GB; ++gg) t += r[gg * T + c];
            p.db[n_blk + c] = t;
        }
        return;
    }
    const int VPB = 256 / G, vl = threadIdx.x % VPB, g = threadIdx.x / VPB;
    const int e = (blockIdx.x * VPB + vl) * 4;  // first of 4 consecutive elements (same row of the tile)
    float4 acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    int sp = g;
    for (; sp + 7 * G < p.nsplit; sp += 8 * G) {
        float4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = *(const float4 *)(src + (int64_t)(sp + q * G) * PART + e);
#pragma unroll
        for (int q = 0; q < 8; ++q) { acc[q].x += v[q].x; acc[q].y += v[q].y; acc[q].z += v[q].z; acc[q].w += v[q].w; }
    }
    for (; sp < p.nsplit; sp += G) {
        const float4 v = *(const float4 *)(src + (int64_t)sp * PART + e);
        acc[0].x += v.x; acc[0].y += v.y; acc[0].z += v.z; acc[0].w += v.w;
    }
    float4 t;
    t.x = ((acc[0].x + acc[1].x) + (acc[2].x + acc[3].x)) + ((acc[4].x + acc[5].x) + (acc[6].x + acc[7].x));
    t.y = ((acc[0].y + acc[1].y) + (acc[2].y + acc[3].y)) + ((acc[4].y + acc[5].y) + (acc[6].y + acc[7].y));
    t.z = ((acc[0].z + acc[1].z) + (acc[2].z + acc[3].z)) + ((acc[4].z + acc[5].z) + (acc[6].z + acc[7].z));
    t.w = ((acc[0].w + acc[1].w) + (acc[2].w + acc[3].w)) + ((acc[4].w + acc[5].w) + (acc[6].w + acc[7].w));
    red[threadIdx.x] = t;
    __syncthreads();
    if (g == 0) {
        for (int gg = 1; gg < G; ++gg) {
            const float4 u = red[gg * VPB + vl];
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        const int n = n_blk + e / T, k = k_blk + e % T;
        if (n < p.N && k < p.K) *(float4 *)(p.dW + (int64_t)n * p.K + k) = t;  // K % 8 == 0: all four or none
    }
}

static void wgrad_plan(int64_t M, int N, int K, WgradParams &p, int &tiles) {
    p.M = M; p.N = N; p.K = K;
    p.tile = (N > 128 && K > 128) ? 256 : 128;
    p.tiles_k = (K + p.tile - 1) / p.tile;
    tiles = ((N + p.tile - 1) / p.tile) * p.tiles_k;
    const int64_t chunks = (M + WG_BM - 1) / WG_BM;
    // one round of workgroups: the 256 tile runs one workgroup per CU (register-limited: 8 waves x ~200 VGPRs), the 128
    // tile two (measured optimum on MI355X: 256 / 512 workgroups); a multiple of 8 so that every XCD gets whole splits
    int64_t nsplit = ((p.tile == 256 ? 256 : 512) / tiles) & ~7;
    if (nsplit < 8) nsplit = 8;
    if (nsplit > 256) nsplit = 256;
    if (nsplit > chunks) nsplit = chunks;
    const int64_t cps = (chunks + nsplit - 1) / nsplit;
    p.rows_per_split = cps * WG_BM;
    p.nsplit = (int)((chunks + cps - 1) / cps);
    p.tiles = tiles;
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
    const dim3 grid((unsigned)(((p.nsplit + 7) / 8) * 8 * tiles));
    if (p.tile == 256) {
        VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)wgrad_bf16_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((wgrad_bf16_kernel<256>), grid, dim3(512), lds, s, p);
    } else {
        hipLaunchKernelGGL((wgrad_bf16_kernel<128>), grid, dim3(256), lds, s, p);
    }
    VSDE_CHECK_HIP(hipGetLastError());
    const int64_t vecs = (int64_t)tiles * p.tile * p.tile / 4;
    int G = 1;
    while (G < 16 && vecs * G < 524288 && 2 * G <= p.nsplit) G *= 2;  // enough threads to cover the load latency
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(p.tile * p.tile / 4 / (256 / G) + 1, tiles), dim3(256), 0, s, p, G);
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}
