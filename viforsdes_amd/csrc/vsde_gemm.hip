// fp32 MFMA GEMMs for the non-recurrent part of the fused head (gfx950).
//
// The reference keeps the context projection inside the per-path time loop
// (project_scalar_rzn, kernels/helpers.py:42-72: C serial scalar loads per step) and builds
// all weight gradients with global atomics (kernels/backward.py:108-139,575-590).  Neither
// is recurrent, so here they are dense contractions over all B*T path-steps:
//   gemm_nt      G[m][n]  = sum_k ctx(m,k) W_c[n][k] + b_ih0[n]     (forward, hoisted projection)
//                gC[m][c] = sum_n dpre0(m,n) W_c^T[c][n]            (grad_context)
//   tn_grouped   dW[n][k] = sum_m dpre(m,n) * input(m,k)            (all weight/bias gradients)
// Arithmetic is exact fp32 (v_mfma_f32_16x16x4_f32 == fmaf chain), no reduced precision.
#include <stdarg.h>

#include <stdlib.h>

#include "vsde_common.h"

namespace vsde {

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char *last_error() { return g_err; }

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------- NT
constexpr int NT_BM = 64, NT_BN = 64, NT_BK = 32, NT_LD = 36;  // LD 36: conflict-free ds_read_b64

struct NtParams {
    RowView A;
    int M, K, N;
    const float *Bt;
    int ldb;
    const float *bias;
    float *C;
    int64_t ldc;
    int a_vec;  // A rows may be read with 16-byte (f32) / 8-byte (bf16) vector loads
    int b_vec;
    // output placement: row m = b*out_rpb + t lands at C + b*out_bstride + t*ldc when out_rpb > 0 (lets the result be the
    // leading T rows of every [T+1, N] batch slab); out_dtype 0 = f32, 1 = bf16 (round to nearest even)
    int out_rpb;
    int64_t out_bstride;
    int out_dtype;
};

__device__ __forceinline__ float4 load_a4(const RowView &A, int64_t off, bool valid, int k, int K, int vec) {
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!valid) return r;
    if (vec && k + 3 < K) {
        if (A.dtype == 0) return *(const float4 *)((const float *)A.base + off + k);
        uint2 raw = *(const uint2 *)((const uint16_t *)A.base + off + k);
        r.x = __uint_as_float(raw.x << 16); r.y = __uint_as_float(raw.x & 0xffff0000u);
        r.z = __uint_as_float(raw.y << 16); r.w = __uint_as_float(raw.y & 0xffff0000u);
        return r;
    }
    if (k + 0 < K) r.x = rowview_load(A, off, k + 0);
    if (k + 1 < K) r.y = rowview_load(A, off, k + 1);
    if (k + 2 < K) r.z = rowview_load(A, off, k + 2);
    if (k + 3 < K) r.w = rowview_load(A, off, k + 3);
    return r;
}

__global__ void __launch_bounds__(256) gemm_nt_kernel(NtParams p) {
    __shared__ __attribute__((aligned(16))) float As[NT_BM * NT_LD];
    __shared__ __attribute__((aligned(16))) float Bs[NT_BN * NT_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m_blk = blockIdx.x * NT_BM, n_blk = blockIdx.y * NT_BN;
    const int srow = tid >> 3, sk4 = (tid & 7) * 4;  // staging coordinates (two passes of 32 rows)

    int64_t a_off[2]; bool a_ok[2];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        int m = m_blk + srow + 32 * ps;
        bool v = false; int64_t off = 0;
        if (m < p.M) off = rowview_offset(p.A, m, v);
        a_off[ps] = off; a_ok[ps] = v && (m < p.M);
    }
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    for (int k0 = 0; k0 < p.K; k0 += NT_BK) {
        float4 av[2], bv[2];
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            av[ps] = load_a4(p.A, a_off[ps], a_ok[ps], k0 + sk4, p.K, p.a_vec);
            int n = n_blk + srow + 32 * ps;
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n < p.N) {
                const float *bp = p.Bt + (int64_t)n * p.ldb + k0 + sk4;
                if (p.b_vec && k0 + sk4 + 3 < p.K) r = *(const float4 *)bp;
                else {
                    if (k0 + sk4 + 0 < p.K) r.x = bp[0];
                    if (k0 + sk4 + 1 < p.K) r.y = bp[1];
                    if (k0 + sk4 + 2 < p.K) r.z = bp[2];
                    if (k0 + sk4 + 3 < p.K) r.w = bp[3];
                }
            }
            bv[ps] = r;
        }
        __syncthreads();  // previous tile fully consumed
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            *(float4 *)&As[(srow + 32 * ps) * NT_LD + sk4] = av[ps];
            *(float4 *)&Bs[(srow + 32 * ps) * NT_LD + sk4] = bv[ps];
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < NT_BK / 8; ++c) {
            // lane group q supplies k = 8c + 2q + s to MFMA s (same permutation for A and B)
            float2 a2 = *(const float2 *)&As[(16 * wave + fr) * NT_LD + 8 * c + 2 * fq];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                float2 b2 = *(const float2 *)&Bs[(16 * nt + fr) * NT_LD + 8 * c + 2 * fq];
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.x, b2.x, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.y, b2.y, acc[nt], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        int n = n_blk + 16 * nt + fr;
        if (n >= p.N) continue;
        float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int m = m_blk + 16 * wave + 4 * fq + r;
            if (m < p.M) {
                int64_t off = (int64_t)m * p.ldc;
                if (p.out_rpb > 0) { const int bb = m / p.out_rpb; off = (int64_t)bb * p.out_bstride + (int64_t)(m - bb * p.out_rpb) * p.ldc; }
                const float val = acc[nt][r] + bias;
                if (p.out_dtype == 0) p.C[off + n] = val;
                else {
                    uint32_t u = __float_as_uint(val);
                    u += 0x7fffu + ((u >> 16) & 1u);
                    ((uint16_t *)p.C)[off + n] = (uint16_t)(u >> 16);
                }
            }
        }
    }
}

static bool rowview_vec_ok(const RowView &v, int K) {
    if (v.col_split < K) return false;  // column remap in use
    if ((uintptr_t)v.base % 16) return false;
    return (v.batch_stride % 4 == 0) && (v.row_stride % 4 == 0);
}

int launch_gemm_nt(const RowView &A, int M, int K, const float *Bt, int ldb, int N, const float *bias,
                   float *C, int64_t ldc, hipStream_t stream, int out_rpb, int64_t out_bstride, int out_dtype) {
    if (M <= 0 || N <= 0) return 0;
    NtParams p;
    p.A = A; p.M = M; p.K = K; p.N = N; p.Bt = Bt; p.ldb = ldb; p.bias = bias; p.C = C; p.ldc = ldc;
    p.out_rpb = out_rpb; p.out_bstride = out_bstride; p.out_dtype = out_dtype;
    p.a_vec = rowview_vec_ok(A, K) ? 1 : 0;
    p.b_vec = ((uintptr_t)Bt % 16 == 0 && ldb % 4 == 0) ? 1 : 0;
    dim3 grid((M + NT_BM - 1) / NT_BM, (N + NT_BN - 1) / NT_BN);
    hipLaunchKernelGGL(gemm_nt_kernel, grid, dim3(256), 0, stream, p);
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}

// -------------------------------------------------------------------------- grouped TN
constexpr int TN_BM = 64, TN_T = 64, TN_LD = 72;  // LD 72: conflict-free ds_read_b32 fragments

struct TnKernelArgs {
    TnProblem prob[kMaxTnProblems];
    int tile_begin[kMaxTnProblems + 1];  // prefix sum of 64x64 output tiles per problem
    int nprob;
    int M;
    int rows_per_split;  // multiple of TN_BM
    int nsplit;
    int ntiles;
    float *partial;      // [ntiles][nsplit][64*64 + 64]  (tile + column sums of X for the bias gradient)
};
constexpr int TN_PART = TN_T * TN_T + TN_T;

__device__ __forceinline__ float4 load_row4(const RowView &V, int64_t off, bool valid, int c, int NC, int vec) {
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!valid) return r;
    if (vec && c + 3 < NC) {
        if (V.dtype == 0) return *(const float4 *)((const float *)V.base + off + c);
        uint2 raw = *(const uint2 *)((const uint16_t *)V.base + off + c);
        r.x = __uint_as_float(raw.x << 16); r.y = __uint_as_float(raw.x & 0xffff0000u);
        r.z = __uint_as_float(raw.y << 16); r.w = __uint_as_float(raw.y & 0xffff0000u);
        return r;
    }
    if (c + 0 < NC) r.x = rowview_load(V, off, c + 0);
    if (c + 1 < NC) r.y = rowview_load(V, off, c + 1);
    if (c + 2 < NC) r.z = rowview_load(V, off, c + 2);
    if (c + 3 < NC) r.w = rowview_load(V, off, c + 3);
    return r;
}

// out tile (n_blk.., k_blk..) += X^T Y over this block's row range; rows are staged 64 at a time through
// LDS with the NEXT 64 rows prefetched into registers while the MFMAs of the current ones run.
__global__ void __launch_bounds__(256) tn_grouped_kernel(TnKernelArgs a) {
    __shared__ __attribute__((aligned(16))) float Xs[TN_BM * TN_LD];
    __shared__ __attribute__((aligned(16))) float Ys[TN_BM * TN_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // consecutive workgroup ids go round-robin over the 8 XCDs (id = 8 * local + xcd): the tiles of one split -- same rows of the
    // operands, different column blocks -- get consecutive `local` on ONE XCD, so the rows they share come out of that XCD's L2
    // instead of HBM once per tile (2.15 GB fetched per launch at LV before, against ~0.65 GB of distinct operand bytes)
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int tile = local % a.ntiles, split = (local / a.ntiles) * 8 + xcd;
    if (split >= a.nsplit) return;
    int pi = 0;
    while (pi + 1 < a.nprob && tile >= a.tile_begin[pi + 1]) ++pi;
    // copy the descriptor with static indices (a runtime-indexed kernarg array would live in scratch)
    TnProblem P = a.prob[0];
#pragma unroll
    for (int i = 1; i < kMaxTnProblems; ++i)
        if (pi == i) P = a.prob[i];
    const int tiles_k = (P.NY + TN_T - 1) / TN_T;
    const int lt = tile - a.tile_begin[pi];
    const int n_blk = (lt / tiles_k) * TN_T, k_blk = (lt % tiles_k) * TN_T;
    const bool want_bias = P.bias_out != nullptr && k_blk == 0;
    const int x_vec = (P.X.col_split >= P.NX && ((uintptr_t)P.X.base % 16 == 0) && P.X.batch_stride % 4 == 0 &&
                       P.X.row_stride % 4 == 0) ? 1 : 0;
    const int y_vec = (P.Y.col_split >= P.NY && ((uintptr_t)P.Y.base % 16 == 0) && P.Y.batch_stride % 4 == 0 &&
                       P.Y.row_stride % 4 == 0) ? 1 : 0;

    const int srow = tid >> 4, sc4 = (tid & 15) * 4;  // staging: 16 rows x 64 cols per pass, 4 passes
    const int fr = lane & 15, fq = lane >> 4;
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;

    const int m_begin = split * a.rows_per_split;
    const int m_end = min(a.M, m_begin + a.rows_per_split);
    float4 xv[4], yv[4];
    // (batch, step) of this thread's four staging rows, advanced by 64 rows per chunk: no division in the loop.  Both views of a
    // problem share rows_per_batch (every operand is a [b][t] record); the shifts differ.
    int rb[4], rt[4];
    const int rpb = P.X.rows_per_batch;
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) { const int m = m_begin + srow + 16 * ps; rb[ps] = m / rpb; rt[ps] = m - rb[ps] * rpb; }
    const int adv_b = TN_BM / rpb, adv_t = TN_BM - adv_b * rpb;
    auto fetch = [&](int m0) {
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const bool in = m0 + srow + 16 * ps < m_end;
            const int tx = rt[ps] + P.X.shift, ty = rt[ps] + P.Y.shift;
            const int64_t ox = (int64_t)rb[ps] * P.X.batch_stride + (int64_t)tx * P.X.row_stride;
            const int64_t oy = (int64_t)rb[ps] * P.Y.batch_stride + (int64_t)ty * P.Y.row_stride;
            xv[ps] = load_row4(P.X, ox, in && tx >= 0, n_blk + sc4, P.NX, x_vec);
            yv[ps] = load_row4(P.Y, oy, in && ty >= 0, k_blk + sc4, P.NY, y_vec);
            rb[ps] += adv_b; rt[ps] += adv_t;
            if (rt[ps] >= rpb) { rt[ps] -= rpb; ++rb[ps]; }
        }
    };
    if (m_begin < m_end) fetch(m_begin);
    for (int m0 = m_begin; m0 < m_end; m0 += TN_BM) {
        __syncthreads();  // previous rows fully consumed
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            *(float4 *)&Xs[(srow + 16 * ps) * TN_LD + sc4] = xv[ps];
            *(float4 *)&Ys[(srow + 16 * ps) * TN_LD + sc4] = yv[ps];
        }
        __syncthreads();
        if (m0 + TN_BM < m_end) fetch(m0 + TN_BM);  // in flight during the MFMAs below
#pragma unroll
        for (int c = 0; c < TN_BM / 8; ++c) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int row = 8 * c + 2 * fq + s;
                const float xa = Xs[row * TN_LD + 16 * wave + fr];
                bsum += xa;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    const float yb = Ys[row * TN_LD + 16 * kt + fr];
                    acc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, yb, acc[kt], 0, 0, 0);
                }
            }
        }
    }
    float *dst = a.partial + ((int64_t)tile * a.nsplit + split) * TN_PART;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(16 * wave + 4 * fq + r) * TN_T + 16 * kt + fr] = acc[kt][r];
    if (want_bias) {  // column sums of X: lanes fr, fr+16, fr+32, fr+48 hold disjoint rows of column 16*wave+fr
        bsum += __shfl_xor(bsum, 16, 64);
        bsum += __shfl_xor(bsum, 32, 64);
        if (fq == 0) dst[TN_T * TN_T + 16 * wave + fr] = bsum;
    }
}

// 16 blocks per tile: block (tile, part) reduces 256 outputs (+ the bias row in part 0) over the splits
__global__ void __launch_bounds__(256) tn_reduce_kernel(TnKernelArgs a) {
    const int tile = blockIdx.x, part = blockIdx.y;
    int pi = 0;
    while (pi + 1 < a.nprob && tile >= a.tile_begin[pi + 1]) ++pi;
    TnProblem P = a.prob[0];
#pragma unroll
    for (int i = 1; i < kMaxTnProblems; ++i)
        if (pi == i) P = a.prob[i];
    const int tiles_k = (P.NY + TN_T - 1) / TN_T;
    const int lt = tile - a.tile_begin[pi];
    const int n_blk = (lt / tiles_k) * TN_T, k_blk = (lt % tiles_k) * TN_T;
    const float *src = a.partial + (int64_t)tile * a.nsplit * TN_PART;
    {
        const int e = part * 256 + threadIdx.x;
        const int n = n_blk + e / TN_T, k = k_blk + e % TN_T;
        if (n < P.NX && k < P.NY) {
            float s = 0.f;
            for (int sp = 0; sp < a.nsplit; ++sp) s += src[(int64_t)sp * TN_PART + e];  // fixed order
            P.out[(int64_t)n * P.ldo + P.col_off + k] = s;
        }
    }
    if (part == 0 && P.bias_out != nullptr && k_blk == 0 && threadIdx.x < TN_T) {
        const int n = n_blk + threadIdx.x;
        if (n < P.NX) {
            float s = 0.f;
            for (int sp = 0; sp < a.nsplit; ++sp) s += src[(int64_t)sp * TN_PART + TN_T * TN_T + threadIdx.x];
            P.bias_out[n] = s;
        }
    }
}

static int tn_plan(const TnProblem *probs, int nprob, int M, TnKernelArgs &a) {
    a.nprob = nprob; a.M = M;
    int tiles = 0;
    for (int i = 0; i < kMaxTnProblems; ++i) {
        a.prob[i] = probs[i < nprob ? i : 0];
        if (i < nprob) {
            a.tile_begin[i] = tiles;
            tiles += ((probs[i].NX + TN_T - 1) / TN_T) * ((probs[i].NY + TN_T - 1) / TN_T);
        }
    }
    for (int i = nprob; i <= kMaxTnProblems; ++i) a.tile_begin[i] = tiles;
    int chunks = (M + TN_BM - 1) / TN_BM;
    // Whole rounds of resident workgroups (4 per CU: 36 KB of LDS each): all workgroups take the same time, so 2072 of them on
    // 1024 slots cost three rounds, 2044 cost two.
    static int target = -1;
    if (target < 0) target = (int)vsde_knob("VSDE_TN_WGS", 2048);
    int want = tiles > 0 ? target / tiles : 1;
    int nsplit = want < 1 ? 1 : want;
    if (nsplit > chunks) nsplit = chunks;
    if (nsplit > 256) nsplit = 256;
    int cps = (chunks + nsplit - 1) / nsplit;
    a.rows_per_split = cps * TN_BM;
    a.nsplit = (chunks + cps - 1) / cps;
    return tiles;
}

size_t tn_workspace_bytes(const TnProblem *probs, int nprob, int M) {
    TnKernelArgs a;
    int tiles = tn_plan(probs, nprob, M, a);
    const size_t generic = (size_t)tiles * a.nsplit * TN_PART * sizeof(float), wide = tn_wide_workspace_bytes(probs, nprob, M);
    return generic > wide ? generic : wide;   // the fast path can still decline at launch (operand alignment)
}

int launch_tn_grouped(const TnProblem *probs, int nprob, int M, void *workspace, size_t workspace_bytes,
                      hipStream_t stream) {
    if (nprob == 0 || M <= 0) return 0;
    VSDE_CHECK_ARG(nprob <= kMaxTnProblems, VSDE_E_BADARG, "too many grouped TN problems (%d)", nprob);
    for (int i = 0; i < nprob; ++i)
        VSDE_CHECK_ARG(probs[i].X.rows_per_batch == probs[i].Y.rows_per_batch && probs[i].X.rows_per_batch > 0, VSDE_E_BADARG,
                       "grouped TN problem %d: both operands must share rows_per_batch", i);
    {
        static int generic_only = -1;   // VSDE_TN_GENERIC=1: A/B against the generic kernel
        if (generic_only < 0) generic_only = (int)vsde_knob("VSDE_TN_GENERIC", 0);
        const int rc = generic_only ? 0 : launch_tn_wide(probs, nprob, M, workspace, workspace_bytes, stream);
        if (rc != 0) return rc < 0 ? rc : 0;
    }
    TnKernelArgs a;
    int tiles = tn_plan(probs, nprob, M, a);
    size_t need = (size_t)tiles * a.nsplit * TN_PART * sizeof(float);
    VSDE_CHECK_ARG(workspace_bytes >= need, VSDE_E_WORKSPACE, "TN workspace too small: %zu < %zu", workspace_bytes, need);
    a.partial = (float *)workspace;
    a.ntiles = tiles;
    hipLaunchKernelGGL(tn_grouped_kernel, dim3((unsigned)(((a.nsplit + 7) / 8) * 8 * tiles)), dim3(256), 0, stream, a);
    VSDE_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(tn_reduce_kernel, dim3(tiles, TN_T * TN_T / 256), dim3(256), 0, stream, a);
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace vsde

extern "C" const char *vsde_last_error(void) { return vsde::last_error(); }
extern "C" int vsde_abi_version(void) { return VSDE_ABI_VERSION; }
extern "C" int vsde_build_ablations(void) {
#ifdef VSDE_ABLATIONS
    return 1;
#else
    return 0;
#endif
}
