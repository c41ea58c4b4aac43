// Do the MFMAs of one wave and the VALU instructions of ANOTHER wave on the same SIMD overlap?  (gfx950)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_valu_overlap.hip -o gpurun_out/overlap && gpurun_out/overlap
// One workgroup per CU of 8 waves: waves 0..3 (one per SIMD) run role A, waves 4..7 role B.  Roles: 0 idle, 1 chain of DEPENDENT
// 32x32x16 bf16 MFMAs (one accumulator), 2 MFMAs over 4 independent accumulators, 3 VALU (v_fma chains, 8 independent), 4 VALU with
// a quarter of transcendentals (v_exp / v_rcp), 5 ds_read_b128 stream.  Prints kernel time per (A, B) pair and the time of each alone.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int ROLE>
__device__ __forceinline__ float run_role(int iters, float seed, char *lds) {
    float out = 0.f;
    if constexpr (ROLE == 1) {
        f32x16 acc = {0};
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + i); b[i] = (short)(0x3c00 + i); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
        out = acc[0] + acc[7];
    } else if constexpr (ROLE == 2) {
        f32x16 acc[4] = {{0}, {0}, {0}, {0}};
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + i); b[i] = (short)(0x3c00 + i); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k & 3], 0, 0, 0);
        }
        out = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    } else if constexpr (ROLE == 3) {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = seed + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 12; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], 0.999f, 0.001f);
        }
        for (int i = 0; i < 8; ++i) out += v[i];
    } else if constexpr (ROLE == 4) {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = seed + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_exp2f(-0.5f * v[i]);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_rcpf(1.0f + v[i]);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], 0.999f, 0.001f);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = v[i] * 1.0001f;
            }
        }
        for (int i = 0; i < 8; ++i) out += v[i];
    } else if constexpr (ROLE == 6 || ROLE == 9) {   // ONE wave: dependent MFMAs with 6 (ROLE 6) / 8 (ROLE 9: 4 of them exp/rcp) VALU in every gap
        f32x16 acc = {0};
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + i); b[i] = (short)(0x3c00 + i); }
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = seed + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
                if constexpr (ROLE == 6) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) v[i] = __builtin_fmaf(v[i], 0.999f, 0.001f);
                } else {
#pragma unroll
                    for (int i = 0; i < 2; ++i) v[i] = __builtin_amdgcn_exp2f(-0.5f * v[i]);
#pragma unroll
                    for (int i = 2; i < 4; ++i) v[i] = __builtin_amdgcn_rcpf(1.0f + v[i]);
#pragma unroll
                    for (int i = 4; i < 8; ++i) v[i] = __builtin_fmaf(v[i], 0.999f, 0.001f);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        out = acc[0] + acc[7];
        for (int i = 0; i < 8; ++i) out += v[i];
    } else if constexpr (ROLE == 7) {
        __builtin_amdgcn_s_setprio(3);
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = seed + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 12; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], 0.999f, 0.001f);
        }
        for (int i = 0; i < 8; ++i) out += v[i];
    } else if constexpr (ROLE == 8) {
        __builtin_amdgcn_s_setprio(3);
        f32x16 acc = {0};
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + i); b[i] = (short)(0x3c00 + i); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
        out = acc[0] + acc[7];
    } else if constexpr (ROLE == 5) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 s = {0, 0, 0, 0};
        const char *src = lds + (threadIdx.x & 63) * 16;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 16; ++k) { const u32x4 v = *(const volatile u32x4 *)(src + k * 1024); s[0] += v[0]; s[1] ^= v[3]; }
        }
        out = (float)(s[0] + s[1]);
    }
    return out;
}

template <int RA, int RB>
__global__ void __launch_bounds__(512, 2) probe(float *out, int iters, float seed) {
    __shared__ __attribute__((aligned(16))) char lds[16384];
    for (int i = threadIdx.x; i < 4096; i += 512) ((float *)lds)[i] = seed;
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float r;
    if (wave < 4) r = run_role<RA>(iters, seed, lds);
    else r = run_role<RB>(iters, seed, lds);
    if (r == 123.456f) out[threadIdx.x] = r;
}

template <int RA, int RB>
static float timeit(float *out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<RA, RB>), dim3(256), dim3(512), 0, 0, out, iters, 1.0f);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((probe<RA, RB>), dim3(256), dim3(512), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5 * 1e3f;
}

int main() {
    float *out;
    hipMalloc(&out, 4096);
    const int it = 2000;
    printf("role alone (other half of the waves idle), us for %d iterations:\n", it);
    printf("  1 dependent MFMA chain (16/iter): %8.1f   -> %.1f cycles per MFMA at 2.1 GHz\n", timeit<1, 0>(out, it), timeit<1, 0>(out, it) * 2100.0 / (16.0 * it));
    printf("  2 independent MFMAs (16/iter):    %8.1f   -> %.1f cycles per MFMA\n", timeit<2, 0>(out, it), timeit<2, 0>(out, it) * 2100.0 / (16.0 * it));
    printf("  3 VALU fma (96/iter):             %8.1f   -> %.1f cycles per instruction\n", timeit<3, 0>(out, it), timeit<3, 0>(out, it) * 2100.0 / (96.0 * it));
    printf("  4 VALU with exp/rcp (96/iter):    %8.1f   -> %.1f cycles per instruction\n", timeit<4, 0>(out, it), timeit<4, 0>(out, it) * 2100.0 / (96.0 * it));
    printf("  5 ds_read_b128 (16/iter):         %8.1f   -> %.1f cycles per read\n", timeit<5, 0>(out, it), timeit<5, 0>(out, it) * 2100.0 / (16.0 * it));
    printf("pairs on the same SIMD (A = waves 0..3, B = waves 4..7):\n");
    printf("  dep MFMA + VALU fma:       %8.1f\n", timeit<1, 3>(out, it));
    printf("  dep MFMA + VALU exp/rcp:   %8.1f\n", timeit<1, 4>(out, it));
    printf("  indep MFMA + VALU fma:     %8.1f\n", timeit<2, 3>(out, it));
    printf("  indep MFMA + VALU exp/rcp: %8.1f\n", timeit<2, 4>(out, it));
    printf("  dep MFMA + VALU fma @prio3:%8.1f\n", timeit<1, 7>(out, it));
    printf("  dep MFMA @prio3 + VALU fma:%8.1f\n", timeit<8, 3>(out, it));
    printf("  ONE wave: MFMA + 6 fma per gap (16 MFMA + 96 VALU / iter), other idle: %8.1f\n", timeit<6, 0>(out, it));
    printf("  ONE wave: MFMA + 4 fma + 2 exp + 2 rcp per gap, other idle:           %8.1f\n", timeit<9, 0>(out, it));
    printf("  both waves: MFMA + 6 fma per gap each:                                %8.1f\n", timeit<6, 6>(out, it));
    printf("  dep MFMA + dep MFMA:       %8.1f\n", timeit<1, 1>(out, it));
    printf("  indep MFMA + indep MFMA:   %8.1f\n", timeit<2, 2>(out, it));
    printf("  VALU fma + VALU fma:       %8.1f\n", timeit<3, 3>(out, it));
    printf("  dep MFMA + ds_read:        %8.1f\n", timeit<1, 5>(out, it));
    printf("  VALU fma + ds_read:        %8.1f\n", timeit<3, 5>(out, it));
    return 0;
}
