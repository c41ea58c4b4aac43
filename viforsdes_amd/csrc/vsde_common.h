// Shared device/host helpers for libvsde_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/vsde_hip.h"

namespace vsde {

constexpr int kWave = 64;      // CDNA wavefront
constexpr int kHP = 64;        // hidden units padded to one wavefront
constexpr int kChunks = 16;    // kHP / 4 (float4 chunks along the reduction index)
constexpr int kMatF4 = kChunks * 3 * kHP;  // float4 elements of one packed 64x192 matrix

void set_error(const char *fmt, ...);

// A/B and ablation switches.  The SHIPPED library reads no environment variable: `vsde_knob(name, default)` is its default and the
// losing kernel variants behind such switches are not compiled.  `python -m viforsdes_amd.build --ablations` builds
// libvsde_hip_abl.so with -DVSDE_ABLATIONS (the tools load it through VSDE_HIP_LIB): there the knobs are read from the environment
// (once per call site) and the variants exist.  Timing-only ablation switches (a kernel skips part of its work: RESULTS ARE WRONG) go
// through `ablation_env`, which announces the first non-zero value on stderr.
#ifdef VSDE_ABLATIONS
static inline long long vsde_knob(const char *name, long long dflt) {
    const char *e = getenv(name);
    return (e && *e) ? atoll(e) : dflt;
}
static inline int ablation_env(const char *name) {
    const char *e = getenv(name);
    const int v = e ? atoi(e) : 0;
    if (v != 0) fprintf(stderr, "libvsde_hip: %s=%d is a TIMING-ONLY ablation switch -- results of the affected kernels are WRONG\n", name, v);
    return v;
}
#else
#define vsde_knob(name, dflt) ((long long)(dflt))
#define ablation_env(name) (0)
#endif

#define VSDE_CHECK_ARG(cond, code, ...)          \
    do {                                         \
        if (!(cond)) {                           \
            vsde::set_error(__VA_ARGS__);        \
            return (code);                       \
        }                                        \
    } while (0)

#define VSDE_CHECK_HIP(expr)                                                         \
    do {                                                                             \
        hipError_t e_ = (expr);                                                      \
        if (e_ != hipSuccess) {                                                      \
            vsde::set_error("%s failed: %s", #expr, hipGetErrorString(e_));          \
            return (int)e_;                                                          \
        }                                                                            \
    } while (0)

// ---------------------------------------------------------------------------------------
// Row-mapped 2-D operand view used by the GEMM kernels.  Logical row m in [0, M) maps to
//   b = m / rows_per_batch, t = m % rows_per_batch + shift   (row is all-zero when t < 0)
//   element(m, k) = base[b*batch_stride + t*row_stride + colmap(k)]
// with colmap(k) = k < col_split ? k : k + col_skip (lets one view read (dr,du | dcn) out
// of a [dr,du,dn,dcn] record).  dtype: 0 = f32, 1 = bf16.
struct RowView {
    const void *base;
    int64_t batch_stride;
    int64_t row_stride;
    int rows_per_batch;
    int shift;
    int col_split;
    int col_skip;
    int dtype;
};

__device__ __forceinline__ float bf16_to_f32(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

__device__ __forceinline__ int64_t rowview_offset(const RowView &v, int m, bool &valid) {
    int b = m / v.rows_per_batch;
    int t = m - b * v.rows_per_batch + v.shift;
    valid = t >= 0;
    return (int64_t)b * v.batch_stride + (int64_t)t * v.row_stride;
}

__device__ __forceinline__ float rowview_load(const RowView &v, int64_t off, int k) {
    int kk = k < v.col_split ? k : k + v.col_skip;
    if (v.dtype == 0) return ((const float *)v.base)[off + kk];
    return bf16_to_f32(((const uint16_t *)v.base)[off + kk]);
}

// fast transcendental forms (v_exp_f32 / v_rcp_f32); abs error ~1e-7 on outputs in [-1, 1]
// NOTE: __frcp_rn expands to a 12-instruction correctly-rounded division; the raw v_rcp_f32 (1 ulp) is what we want.
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_sigmoid(float x) { return fast_rcp(1.0f + fast_exp2(-1.4426950408889634f * x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.0f - 2.0f * fast_rcp(1.0f + fast_exp2(2.8853900817779268f * x)); }

__device__ __forceinline__ void wave_lds_fence() {
    // LDS operations of one wavefront retire in order; this only stops the compiler from
    // moving LDS reads above the preceding LDS writes of other lanes.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// sum over the 64 lanes; result valid in every lane
// v(lane) + v(lane ^ 32) without the lane index (__shfl_xor builds its ds_bpermute address from v_mbcnt; hoisted out of a loop that
// uses every VGPR that index is spilled and reloaded per trip): v_permlane32_swap of two copies gives [lo, lo] and [hi, hi]
__device__ __forceinline__ float sum_xor32(float v) {
    const uint32_t u = __float_as_uint(v);
    const auto s = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(s[0]) + __uint_as_float(s[1]);
}

// v(lane ^ 32) the same way, and v(lane ^ K) for K < 32 as a ds_swizzle bit-mode pattern (an immediate): neither needs a lane index
__device__ __forceinline__ float xor32(float v) {
    const uint32_t u = __float_as_uint(v);
    const auto s = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float((threadIdx.x & 32) ? s[0] : s[1]);
}
template <int K>
__device__ __forceinline__ float xor_lane(float v) {
    static_assert(K > 0 && K < 32, "ds_swizzle bit mode reaches the 32 lanes of a half wave");
    return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), (K << 10) | 0x1f));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

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

// DPP cross-lane add: v + v[permuted lane] in one VALU instruction (no LDS crossbar round trip)
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

// Sum over the 16 quads of a wavefront of a value that is identical in the 4 lanes of each quad; result wave-uniform.
// row_half_mirror and row_mirror fold the 4 quads of each 16-lane row, v_readlane picks the 4 row sums.
__device__ __forceinline__ float wave_sum_of_quads(float v) {
    v = dpp_add<0x141>(v);  // row_half_mirror: lane i += lane 7-i (within 8)
    v = dpp_add<0x140>(v);  // row_mirror:      lane i += lane 15-i (within 16)
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}

// ---- GEMM launchers (vsde_gemm.hip) ----------------------------------------------------
// C[m][n] = sum_k A(m,k) * Bt[n][k] (+ bias[n]);  Bt row-major [N][K] with leading dim ldb.
int launch_gemm_nt(const RowView &A, int M, int K, const float *Bt, int ldb, int N, const float *bias,
                   float *C, int64_t ldc, hipStream_t stream, int out_rpb = 0, int64_t out_bstride = 0, int out_dtype = 0);

// Grouped "X^T Y" reductions over the row index:  out[n][col_off + k] = sum_m X(m,n) * Y(m,k)
// and, when bias_out != nullptr, bias_out[n] = sum_m X(m,n).
struct TnProblem {
    RowView X;
    RowView Y;
    int NX, NY;
    float *out;       // [NX][ldo]
    int64_t ldo;
    int col_off;
    float *bias_out;  // [NX] or nullptr
};
constexpr int kMaxTnProblems = 12;
size_t tn_workspace_bytes(const TnProblem *probs, int nprob, int M);
int launch_tn_grouped(const TnProblem *probs, int nprob, int M, void *workspace, size_t workspace_bytes,
                      hipStream_t stream);

// fast path for hidden_dim 64 (vsde_tn_wide.hip): 0 = not applicable (use the generic kernel), 1 = done, < 0 = error
size_t tn_wide_workspace_bytes(const TnProblem *probs, int nprob, int M);
int launch_tn_wide(const TnProblem *probs, int nprob, int M, void *workspace, size_t workspace_bytes, hipStream_t stream);

// ---- bf16-plane fast paths of the two head GEMMs (vsde_proj.hip): 1 = done, 0 = not applicable (use launch_gemm_nt), < 0 = error
size_t proj_planes_bytes(int N, int K);
int launch_proj_fwd_bf16(const RowView &A, int64_t M, int K, const float *W, int ldw, int N, const float *bias, float *G, int64_t ldc,
                         void *scratch, size_t scratch_bytes, hipStream_t s);
int launch_proj_bwd_bf16(const RowView &A, int64_t M, int K, const float *Wt, int ldw, int N, void *C, int64_t ldc, int out_rpb,
                         int64_t out_bstride, void *scratch, size_t scratch_bytes, hipStream_t s);

// ---- multi-path MFMA forward of the GRU head (vsde_head_mp.hip): hidden_dim 64, L <= 2, state_dim <= 2 ----------------------
struct MpLaunch {
    int B, T, S, P, C, L, save;
    int np;                      // paths per workgroup: 4, 8 or 16; anything else = chosen by batch size
    const float *x0, *theta, *eps, *G;
    const float *W_ih0, *W_hh0, *W_ih_st, *W_hh_st, *out_W;
    const float *b_hh0, *b_ih_st, *b_hh_st, *out_b;
    void *frags;                 // mp_frag_bytes(L, S) of workspace
    float dt, sqdt, diag_min;
    float *paths, *means, *chol, *chol_raw, *acts;
};
bool mp_applicable(int H, int L, int S);
bool mp_weights_overflowed();   // sticky: a weight left the f16 range of the multi-path kernels (no device synchronisation)
void mp_clear_overflow();        // tests: forget it
size_t mp_frag_bytes(int L, int S);
// mark: the profile-event hook of vsde_head.hip (slot 0 = the time-stepping kernel), may be nullptr
int launch_head_fwd_mp(const MpLaunch &a, hipStream_t s, void (*mark)(int, int, hipStream_t));

// multi-path MFMA reverse-time sweep (two layers, state_dim <= 2): writes D4, DO, g_x0, g_theta like head_bwd_v2_kernel
struct MpBwdLaunch {
    int B, T, S, P, C;
    int np;                      // paths per workgroup: 4 or 8 (16 -> 8); anything else = chosen by batch size
    const float *g_paths, *g_means, *g_chol, *eps, *chol_raw, *acts;
    const float *W_ih0, *W_hh0, *W_ih_st, *W_hh_st, *out_W;
    void *frags;                 // mp_bwd_frag_bytes() of workspace
    float dt, sqdt, diag_min;
    float *D4, *DO, *g_x0, *g_theta;
};
bool mp_bwd_applicable(int H, int L, int S);
size_t mp_bwd_frag_bytes(void);
int launch_head_bwd_mp(const MpBwdLaunch &a, hipStream_t s, void (*mark)(int, int, hipStream_t));

// ---- streamed attention kernels (vsde_attn_stream.hip): any N, head_dim 64 or 128 ------------
int launch_attention_stream_fwd(const void *q, const void *k, const void *v, void *o, float *lse, int64_t B, int N, int H, int D,
                                double scale, hipStream_t s);
int launch_attention_stream_bwd(const void *dout, const void *q, const void *k, const void *v, const void *o, const float *lse, void *dq,
                                void *dk, void *dv, float *delta, int64_t B, int N, int H, int D, double scale, hipStream_t s);

}  // namespace vsde
