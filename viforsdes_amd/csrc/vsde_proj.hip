// The two non-recurrent GEMMs of the fused head on bf16 MFMA without giving up fp32 accuracy (gfx950):
//
//   forward   G[m][n]  = sum_c ctx(m, c) W_c[n][c] + b_ih0[n]        ctx bf16 [M, C], W_c fp32 [3H, C]  -> fp32 [M, 3H]
//   backward  gC[m][c] = sum_n dpre0(m, n) W_c[n][c]                 dpre0 fp32 [M, 3H]                 -> bf16 [M, C]
//
// (the hoisted context projection of kernels/helpers.py:42-72 and grad_context of kernels/backward.py:550-564).  The generic
// kernel (vsde_gemm.hip: gemm_nt_kernel) runs them on v_mfma_f32_16x16x4_f32 at 1/16 of the bf16 rate: 0.26 + 0.28 ms at the
// Lotka-Volterra shapes, MFMA-bound.  Here
//   * forward: the context already IS bf16, so only the weight needs care: W_c = hi + mid + lo with three bf16 planes (8 + 8 + 8
//     mantissa bits, exact) and G = ctx hi^T + ctx mid^T + ctx lo^T -- every product is exact in fp32 and the accumulation is
//     fp32 as before: the same result as the fp32 GEMM up to summation order, at 3/16 of its MFMA time;
//   * backward: the output is bf16 (2^-9), so two planes per operand and the products hi hi + hi lo + lo hi (relative error
//     2^-16 before the final rounding) are more than it can show.
// Both are then HBM streaming kernels shaped like lin_rows_kernel (vsde_linear.hip): a wave keeps its 32 rows' operand slices in
// VGPRs, the weight planes stream through LDS in tiles of 32 output columns, products are computed swapped (D = W_tile x^T, the
// lane that owns a row keeps it), outputs leave as full 128-byte row segments through a per-wave staging buffer.
#include "vsde_common.h"

namespace vsde {

typedef float pf32x16 __attribute__((ext_vector_type(16)));
typedef short pbf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t pu32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 phwbf16x2 __attribute__((ext_vector_type(2)));
typedef float pf32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t p_pack(float lo, float hi) {
    const pf32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, phwbf16x2));
}
__device__ __forceinline__ void p_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// planes[n][p * K + k], p < NP: W = sum_p plane_p.  Truncating splits: every remainder is exact in fp32 and has 8 fewer
// significant bits, so three planes reproduce a 24-bit mantissa exactly.
template <int NP>
__global__ void __launch_bounds__(256) split_planes_kernel(const float *W, int ldw, uint16_t *planes, int N, int K) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= N * K) return;
    const int n = e / K, k = e % K;
    float r = W[(int64_t)n * ldw + k];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const uint32_t hi = __float_as_uint(r) & 0xffff0000u;
        planes[(int64_t)n * NP * K + p * K + k] = (uint16_t)(hi >> 16);
        r -= __uint_as_float(hi);
    }
}

struct ProjParams {
    const void *A; int64_t abs_, ars; int T;   // row m = (b, t): A + b * abs_ + t * ars   (elements)
    const uint16_t *planes;                    // [N][NP * K] bf16
    const float *bias;                         // forward: [N] fp32 or nullptr
    void *C; int64_t ldc; int out_rpb; int64_t out_bstride;   // row m lands at b * out_bstride + t * ldc when out_rpb > 0
    int64_t M; int N;
};

constexpr int PJ_THREADS = 256;

// ---------------------------------------------------------------------------------------------------- forward projection
// KC = context width (A bf16), three weight planes; tile = 32 output columns x 3 KC; fp32 output.
// NKH > 1: context width NKH * KC; a 32-column tile is multiplied in NKH k-slices of KC through the same LDS tile buffer.
template <int KC, int NKH = 1>
__global__ void __launch_bounds__(PJ_THREADS, 2) proj_fwd_kernel(ProjParams p) {
    constexpr int KS = KC / 16, KW = 3 * KC, LDB = KW + 8, TILE = 32 * LDB, NLD = 32 * KW / 8 / PJ_THREADS, SLD = 36;
    constexpr int KT = KC * NKH, KWT = 3 * KT;   // full context width, row pitch of the planes
    extern __shared__ __attribute__((aligned(16))) uint16_t psm[];
    float *stage_all = (float *)(psm + TILE);
    __shared__ float sbias[512];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    float *stage = stage_all + wave * 32 * SLD;
    const int64_t row0 = (int64_t)blockIdx.x * 128 + wave * 32;
    for (int i = tid; i < p.N; i += PJ_THREADS) sbias[i] = p.bias ? p.bias[i] : 0.f;
    pbf16x8 afr[KS * NKH];
    {
        const int64_t m = row0 + r < p.M ? row0 + r : p.M - 1;   // rows past the end repeat the last one (never stored)
        const int64_t b = m / p.T, t = m - b * p.T;
        const uint16_t *src = (const uint16_t *)p.A + b * p.abs_ + t * p.ars + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS * NKH; ++ks) afr[ks] = *(const pbf16x8 *)(src + ks * 16);
    }
    const int ntiles = p.N / 32;
    const int rot = blockIdx.x % ntiles;   // workgroups pull different tiles out of L2 at any moment
    pu32x4 breg[NLD];
#define PJ_LOAD(t_)                                                                                           \
    do {                                                                                                      \
        const uint16_t *w_ = p.planes + (int64_t)(((t_) / NKH + rot) % ntiles) * 32 * KWT + ((t_) % NKH) * KC; \
        _Pragma("unroll") for (int i = 0; i < NLD; ++i) {                                                     \
            const int idx = tid + PJ_THREADS * i, row = idx / (KW / 8), c = idx % (KW / 8);                   \
            breg[i] = *(const pu32x4 *)(w_ + (int64_t)row * KWT + (c * 8 / KC) * KT + (c * 8) % KC);          \
        }                                                                                                     \
    } while (0)
#define PJ_STORE()                                                                                            \
    do {                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < NLD; ++i) {                                                     \
            const int idx = tid + PJ_THREADS * i, row = idx / (KW / 8), c = idx % (KW / 8);                   \
            *(pu32x4 *)(psm + row * LDB + c * 8) = breg[i];                                                   \
        }                                                                                                     \
    } while (0)
    PJ_LOAD(0);
    PJ_STORE();
    __syncthreads();
    const int nsteps = ntiles * NKH;
    if (nsteps > 1) PJ_LOAD(1);
    pf32x16 acc;
    for (int t = 0; t < nsteps; ++t) {
        const int kh = t % NKH;
        if (kh == 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        }
        const uint16_t *bsrc = psm + r * LDB + 8 * h;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const pbf16x8 *)(bsrc + pl * KC + ks * 16), NKH == 1 ? afr[ks] : (kh == 0 ? afr[ks] : afr[(NKH - 1) * KS + ks]), acc, 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        p_barrier();   // every wave is done with the tile
        if (t + 1 < nsteps) PJ_STORE();
        if (t + 2 < nsteps) PJ_LOAD(t + 2);
        if (kh != NKH - 1) { p_barrier(); continue; }   // more k-slices of this tile to come
        // epilogue: this lane holds row r, columns 8 g + 4 h + i of the tile
        const int n0 = ((t / NKH + rot) % ntiles) * 32;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 v = make_float4(acc[4 * g + 0] + sbias[n0 + 8 * g + 4 * h + 0], acc[4 * g + 1] + sbias[n0 + 8 * g + 4 * h + 1],
                                         acc[4 * g + 2] + sbias[n0 + 8 * g + 4 * h + 2], acc[4 * g + 3] + sbias[n0 + 8 * g + 4 * h + 3]);
            *(float4 *)(stage + r * SLD + 8 * g + 4 * h) = v;
        }
        wave_lds_fence();
#pragma unroll
        for (int i = 0; i < 4; ++i) {   // 32 rows x 128 bytes: 8 lanes per row
            const int row = (lane >> 3) + 8 * i, c = lane & 7;
            const float4 v = *(const float4 *)(stage + row * SLD + c * 4);
            if (row0 + row < p.M) *(float4 *)((float *)p.C + (row0 + row) * p.ldc + n0 + c * 4) = v;
        }
        wave_lds_fence();
        p_barrier();   // next tile visible
    }
#undef PJ_LOAD
#undef PJ_STORE
}

// ---------------------------------------------------------------------------------------------------- grad_context
// KC = 3H (A fp32, split into hi / lo on the way to registers), two weight planes; tile = 32 output columns x 2 KC; bf16 output
// in pairs of tiles (64 columns = 128-byte row segments).
template <int KC>
__global__ void __launch_bounds__(PJ_THREADS, 2) proj_bwd_kernel(ProjParams p) {
    constexpr int KS = KC / 16, KW = 2 * KC, LDB = KW + 8, TILE = 32 * LDB, NLD = (32 * KW / 8 + PJ_THREADS - 1) / PJ_THREADS, SLD = 72;
    constexpr int NCH = 32 * KW / 8;
    extern __shared__ __attribute__((aligned(16))) uint16_t psm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    uint16_t *stage = psm + 2 * TILE + wave * 32 * SLD;
    const int64_t row0 = (int64_t)blockIdx.x * 128 + wave * 32;
    pbf16x8 ahi[KS], alo[KS];
    {
        const int64_t m = row0 + r < p.M ? row0 + r : p.M - 1;
        const int64_t b = m / p.T, t = m - b * p.T;
        const float *src = (const float *)p.A + b * p.abs_ + t * p.ars + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float4 v0 = *(const float4 *)(src + ks * 16), v1 = *(const float4 *)(src + ks * 16 + 4);
            const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            uint32_t hw[4], lw[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                hw[q] = p_pack(x[2 * q], x[2 * q + 1]);   // round to nearest even
                const float r0 = x[2 * q] - __uint_as_float(hw[q] << 16), r1 = x[2 * q + 1] - __uint_as_float(hw[q] & 0xffff0000u);
                lw[q] = p_pack(r0, r1);
            }
            const pu32x4 hv = {hw[0], hw[1], hw[2], hw[3]}, lv = {lw[0], lw[1], lw[2], lw[3]};
            ahi[ks] = __builtin_bit_cast(pbf16x8, hv); alo[ks] = __builtin_bit_cast(pbf16x8, lv);
        }
    }
    const int ntiles = p.N / 32;                 // even (launcher)
    const int rot = 2 * (blockIdx.x % (ntiles / 2));
    pu32x4 breg[NLD];
#define PB_LOAD(t_)                                                                                           \
    do {                                                                                                      \
        const uint16_t *w_ = p.planes + (int64_t)(((t_) + rot) % ntiles) * 32 * KW;                           \
        _Pragma("unroll") for (int i = 0; i < NLD; ++i) {                                                     \
            int idx = tid + PJ_THREADS * i; idx = idx < NCH ? idx : NCH - 1;                                  \
            breg[i] = *(const pu32x4 *)(w_ + (int64_t)(idx / (KW / 8)) * KW + (idx % (KW / 8)) * 8);          \
        }                                                                                                     \
    } while (0)
#define PB_STORE(Bs_)                                                                                         \
    do {                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < NLD; ++i) {                                                     \
            const int idx = tid + PJ_THREADS * i;                                                             \
            if (idx < NCH) *(pu32x4 *)((Bs_) + (idx / (KW / 8)) * LDB + (idx % (KW / 8)) * 8) = breg[i];      \
        }                                                                                                     \
    } while (0)
    // output row placement of this lane's flush rows (8 lanes per row, 4 passes of 8 rows)
    int64_t orow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t m = row0 + (lane >> 3) + 8 * i;
        if (p.out_rpb > 0) { const int64_t b = m / p.out_rpb; orow[i] = b * p.out_bstride + (m - b * p.out_rpb) * p.ldc; }
        else orow[i] = m * p.ldc;
    }
#define PB_BODY(t_, PAR_)                                                                                     \
    do {                                                                                                      \
        const uint16_t *bsrc = psm + (PAR_) * TILE + r * LDB + 8 * h;                                         \
        pf32x16 acc;                                                                                          \
        _Pragma("unroll") for (int e = 0; e < 16; ++e) acc[e] = 0.f;                                          \
        __builtin_amdgcn_s_setprio(1);                                                                        \
        _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                                   \
            const pbf16x8 wh = *(const pbf16x8 *)(bsrc + ks * 16), wl = *(const pbf16x8 *)(bsrc + KC + ks * 16); \
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, ahi[ks], acc, 0, 0, 0);                         \
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, alo[ks], acc, 0, 0, 0);                         \
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, ahi[ks], acc, 0, 0, 0);                         \
        }                                                                                                     \
        __builtin_amdgcn_s_setprio(0);                                                                        \
        _Pragma("unroll") for (int g = 0; g < 4; ++g)                                                         \
            *(uint2 *)(stage + r * SLD + (PAR_) * 32 + 8 * g + 4 * h) =                                       \
                make_uint2(p_pack(acc[4 * g + 0], acc[4 * g + 1]), p_pack(acc[4 * g + 2], acc[4 * g + 3]));   \
        if ((PAR_) == 1) {                                                                                    \
            wave_lds_fence();                                                                                 \
            const int n0 = (((t_) + rot) % ntiles) * 32 - 32;                                                 \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                   \
                const int row = (lane >> 3) + 8 * i, c = lane & 7;                                            \
                const pu32x4 v = *(const pu32x4 *)(stage + row * SLD + c * 8);                                \
                if (row0 + row < p.M) __builtin_nontemporal_store(v, (pu32x4 *)((uint16_t *)p.C + orow[i] + n0 + c * 8)); \
            }                                                                                                 \
            wave_lds_fence();                                                                                 \
        }                                                                                                     \
        if ((t_) + 1 < ntiles) PB_STORE(psm + (1 - (PAR_)) * TILE);                                           \
        p_barrier();                                                                                          \
        if ((t_) + 2 < ntiles) PB_LOAD((t_) + 2);                                                             \
    } while (0)
    PB_LOAD(0);
    PB_STORE(psm);
    p_barrier();
    PB_LOAD(1);
    for (int t = 0; t < ntiles; t += 2) {
        PB_BODY(t, 0);
        PB_BODY(t + 1, 1);
    }
#undef PB_BODY
#undef PB_LOAD
#undef PB_STORE
}

// returns 1 when the fast path ran, 0 when the caller has to use the generic kernel, < 0 on error.  `scratch` holds the planes.
size_t proj_planes_bytes(int N, int K) { return (size_t)N * 3 * K * sizeof(uint16_t); }

int launch_proj_fwd_bf16(const RowView &A, int64_t M, int K, const float *W, int ldw, int N, const float *bias, float *G, int64_t ldc,
                         void *scratch, size_t scratch_bytes, hipStream_t s) {
    if (A.dtype != 1 || (K != 256 && K != 512) || N % 32 || N > 512 || M <= 0) return 0;
    if ((uintptr_t)A.base % 16 || A.batch_stride % 8 || A.row_stride % 8 || A.col_split < K || A.shift != 0) return 0;
    if ((uintptr_t)G % 16 || ldc % 4 || scratch == nullptr || scratch_bytes < proj_planes_bytes(N, K) || (uintptr_t)scratch % 16) return 0;
    hipLaunchKernelGGL(split_planes_kernel<3>, dim3((N * K + 255) / 256), dim3(256), 0, s, W, ldw, (uint16_t *)scratch, N, K);
    VSDE_CHECK_HIP(hipGetLastError());
    ProjParams p = {};
    p.A = A.base; p.abs_ = A.batch_stride; p.ars = A.row_stride; p.T = A.rows_per_batch; p.planes = (const uint16_t *)scratch;
    p.bias = bias; p.C = G; p.ldc = ldc; p.M = M; p.N = N;
    constexpr int KC = 256, LDB = 3 * KC + 8;
    const size_t lds = (size_t)32 * LDB * sizeof(uint16_t) + (size_t)4 * 32 * 36 * sizeof(float);
    if (K == 512) {   // two k-slices per tile (the synthetic stress configuration's context width)
        VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)proj_fwd_kernel<KC, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((proj_fwd_kernel<KC, 2>), dim3((unsigned)((M + 127) / 128)), dim3(PJ_THREADS), lds, s, p);
    } else {
        VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)proj_fwd_kernel<KC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(proj_fwd_kernel<KC>, dim3((unsigned)((M + 127) / 128)), dim3(PJ_THREADS), lds, s, p);
    }
    VSDE_CHECK_HIP(hipGetLastError());
    return 1;
}

int launch_proj_bwd_bf16(const RowView &A, int64_t M, int K, const float *Wt, int ldw, int N, void *C, int64_t ldc, int out_rpb,
                         int64_t out_bstride, void *scratch, size_t scratch_bytes, hipStream_t s) {
    // A = dpre0 fp32 [M][K = 192], Wt = W_c^T [N = C][K], C bf16
    if (A.dtype != 0 || K != 192 || N % 64 || M <= 0) return 0;
    if ((uintptr_t)A.base % 16 || A.batch_stride % 4 || A.row_stride % 4 || A.col_split < K || A.shift != 0) return 0;
    if ((uintptr_t)C % 16 || ldc % 8 || out_bstride % 8 || scratch == nullptr || scratch_bytes < (size_t)N * 2 * K * 2 || (uintptr_t)scratch % 16) return 0;
    hipLaunchKernelGGL(split_planes_kernel<2>, dim3((N * K + 255) / 256), dim3(256), 0, s, Wt, ldw, (uint16_t *)scratch, N, K);
    VSDE_CHECK_HIP(hipGetLastError());
    ProjParams p = {};
    p.A = A.base; p.abs_ = A.batch_stride; p.ars = A.row_stride; p.T = A.rows_per_batch; p.planes = (const uint16_t *)scratch;
    p.C = C; p.ldc = ldc; p.out_rpb = out_rpb; p.out_bstride = out_bstride; p.M = M; p.N = N;
    constexpr int KC = 192, LDB = 2 * KC + 8;
    const size_t lds = (size_t)2 * 32 * LDB * sizeof(uint16_t) + (size_t)4 * 32 * 72 * sizeof(uint16_t);
    VSDE_CHECK_HIP(hipFuncSetAttribute((const void *)proj_bwd_kernel<KC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(proj_bwd_kernel<KC>, dim3((unsigned)((M + 127) / 128)), dim3(PJ_THREADS), lds, s, p);
    VSDE_CHECK_HIP(hipGetLastError());
    return 1;
}

}  // namespace vsde
