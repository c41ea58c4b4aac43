// Do the MFMAs of one wave and the VALU instructions of ANOTHER wave on the same SIMD overlap?  (gfx950; round 5, hardened in round 6)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_valu_overlap.hip -o tools/probes/overlap_probe.bin
//   tools/probes/overlap_probe.bin [iterations, default 200000: >= 50 ms per timing, the clock has settled]
//   rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES --kernel-trace -d DIR -- tools/probes/overlap_probe.bin 20000
//   (every (role A, role B) pair is its own kernel instantiation: the counters come out per pair; tools/pmc_summary.py DB probe)
// One workgroup per CU of 8 waves: waves 0..3 (one per SIMD) run role A, waves 4..7 role B.  Roles:
//   0 idle
//   1 chain of DEPENDENT v_mfma_f32_32x32x16_bf16 (one accumulator; 8 passes = 32 cycles, 16 K per instruction: the double-rate form)
//   2 the same over 4 independent accumulators
//   3 VALU: v_fma_f32, 8 independent chains                          4 VALU with a quarter of transcendentals (v_exp / v_rcp)
//   5 ds_read_b128 stream
//   6 / 9 ONE wave: dependent MFMAs with 6 v_fma (6) / 4 v_fma + 2 v_exp + 2 v_rcp (9) in every gap
//   7 / 8 roles 3 / 1 at s_setprio 3
//   10 chain of dependent v_mfma_f32_32x32x8_bf16_1k (the half-rate form: 8 K per instruction, same 16 passes? -- measured below)
//   11 chain of dependent v_mfma_f32_32x32x2_f32 (fp32 inputs, 64 cycles per instruction: an eighth of the operand bytes per cycle)
//   12 chain of dependent v_mfma_f32_16x16x32_bf16 (4 accumulator registers)
//   13 VALU: v_mov_b32 between 8 registers (no arithmetic, one source operand)
//   14 / 15 / 16 ONE wave: INDEPENDENT 32x32x16 bf16 MFMAs (4 accumulators, round robin) with 2 / 4 / 6 v_fma in every gap
//   17 ONE wave: the same with 4 v_mov_b32 in every gap;  18: 4 v_cvt_pk_bf16_f32;  19: 2 v_exp_f32
//   (the fillers of the one-wave roles are asm volatile statements between sched_barriers: they stay in their gap)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

// fillers that stay where they are written: the compiler clumps plain fmaf() calls of several gaps together (and SLP-packs them into
// v_pk_fma_f32); an asm volatile statement keeps its place between the sched_barriers
__device__ __forceinline__ void fill_fma(float &v, float c) { asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v) : "v"(c)); }
__device__ __forceinline__ void fill_exp(float &v) { asm volatile("v_exp_f32 %0, %0" : "+v"(v)); }
__device__ __forceinline__ void fill_rcp(float &v) { asm volatile("v_rcp_f32 %0, %0" : "+v"(v)); }
__device__ __forceinline__ void fill_cvt(uint32_t &d, float a, float b) { asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); }

template <int ROLE>
__device__ __forceinline__ float run_role(int iters, float seed, char *lds) {
    float out = 0.f;
    if constexpr (ROLE == 1 || ROLE == 8) {
        if constexpr (ROLE == 8) __builtin_amdgcn_s_setprio(3);
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
    } else if constexpr (ROLE == 10) {
        f32x16 acc = {0};
        bf16x4 a, b;
        for (int i = 0; i < 4; ++i) { a[i] = (short)(0x3f80 + i); b[i] = (short)(0x3c00 + i); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a, b, acc, 0, 0, 0);
        }
        out = acc[0] + acc[7];
    } else if constexpr (ROLE == 11) {
        f32x16 acc = {0};
        float a = seed, b = 0.5f * seed;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        out = acc[0] + acc[7];
    } else if constexpr (ROLE == 12) {
        f32x4 acc = {0};
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + i); b[i] = (short)(0x3c00 + i); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
        }
        out = acc[0] + acc[3];
    } else if constexpr (ROLE == 3 || ROLE == 7) {
        if constexpr (ROLE == 7) __builtin_amdgcn_s_setprio(3);
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = seed + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 12; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], 0.999f, 0.001f);
        }
        for (int i = 0; i < 8; ++i) out += v[i];
    } else if constexpr (ROLE == 13) {
        uint32_t v[8];
        for (int i = 0; i < 8; ++i) v[i] = __float_as_uint(seed) + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                // a rotation of 8 registers: 8 v_mov_b32 per round through inline asm (the compiler would fold plain moves away)
                uint32_t t;
                asm volatile("v_mov_b32 %0, %1" : "=v"(t) : "v"(v[0]));
#pragma unroll
                for (int i = 0; i < 7; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(v[i]) : "v"(v[i + 1]));
                v[7] = t;
            }
        }
        for (int i = 0; i < 8; ++i) out += __uint_as_float(v[i]);
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
    } else if constexpr (ROLE == 6 || ROLE == 9) {
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
                    for (int i = 0; i < 6; ++i) fill_fma(v[i], 0.999f);
                } else {
#pragma unroll
                    for (int i = 0; i < 2; ++i) fill_exp(v[i]);
#pragma unroll
                    for (int i = 2; i < 4; ++i) fill_rcp(v[i]);
#pragma unroll
                    for (int i = 4; i < 8; ++i) fill_fma(v[i], 0.999f);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        out = acc[0] + acc[7];
        for (int i = 0; i < 8; ++i) out += v[i];
    } else if constexpr (ROLE >= 14 && ROLE <= 19) {
        f32x16 acc[4] = {{0}, {0}, {0}, {0}};
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + i); b[i] = (short)(0x3c00 + i); }
        float v[8];
        uint32_t u[8];
        for (int i = 0; i < 8; ++i) { v[i] = seed + i; u[i] = __float_as_uint(seed) + i; }
        constexpr int NF = ROLE == 14 ? 2 : (ROLE == 15 ? 4 : (ROLE == 16 ? 6 : 0));
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k & 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < NF; ++i) fill_fma(v[i], 0.999f);
                if constexpr (ROLE == 18) {   // 4 v_cvt_pk_bf16_f32
#pragma unroll
                    for (int i = 0; i < 4; ++i) fill_cvt(u[i], v[i], v[i + 4]);
                }
                if constexpr (ROLE == 19) {   // 2 v_exp_f32
                    fill_exp(v[0]); fill_exp(v[1]);
                }
                if constexpr (ROLE == 17) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(u[i + 4]));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        out = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
        for (int i = 0; i < 8; ++i) out += v[i] + __uint_as_float(u[i]);
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

static int g_iters = 200000;
template <int RA, int RB>
static float timeit(float *out) {   // us per launch, median of 3 after one warm-up launch of the same length (clock ramp)
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<RA, RB>), dim3(256), dim3(512), 0, 0, out, g_iters, 1.0f);
    float t[3];
    for (int i = 0; i < 3; ++i) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<RA, RB>), dim3(256), dim3(512), 0, 0, out, g_iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&t[i], e0, e1);
    }
    const float lo = t[0] < t[1] ? t[0] : t[1], hi = t[0] < t[1] ? t[1] : t[0];
    const float med = t[2] < lo ? lo : (t[2] > hi ? hi : t[2]);
    return med * 1e3f;
}

template <int RA, int RB>
static void line(const char *name, float *out, double per_iter_a, double alone_a, double alone_b) {
    const float t = timeit<RA, RB>(out);
    printf("  %-52s %10.1f us", name, t);
    if (per_iter_a > 0) printf("   %6.2f us-cycles@2.1GHz per instruction of A", t * 2100.0 / (per_iter_a * g_iters));
    if (alone_a > 0 && alone_b > 0) printf("   alone %9.1f + %9.1f = %9.1f;  exposed share of B: %4.0f %%", alone_a, alone_b, alone_a + alone_b, 100.0 * (t - alone_a) / alone_b);
    printf("\n");
}

int main(int argc, char **argv) {
    if (argc > 1) g_iters = atoi(argv[1]);
    float *out;
    hipMalloc(&out, 4096);
    printf("iterations per launch: %d (one warm-up launch + median of three timed launches per line)\n", g_iters);
    printf("roles alone (the other wave of the SIMD idle):\n");
    const float m16 = timeit<1, 0>(out), m16i = timeit<2, 0>(out), m8 = timeit<10, 0>(out), mf = timeit<11, 0>(out), m1616 = timeit<12, 0>(out);
    const float fma = timeit<3, 0>(out), tr = timeit<4, 0>(out), mov = timeit<13, 0>(out), dsr = timeit<5, 0>(out);
    auto cyc = [&](float us, double n) { return us * 2100.0 / (n * g_iters); };
    printf("  1  dependent v_mfma_f32_32x32x16_bf16 (16/iter)  %10.1f us   %6.2f cycles@2.1GHz each\n", m16, cyc(m16, 16));
    printf("  2  independent 32x32x16_bf16, 4 accumulators     %10.1f us   %6.2f\n", m16i, cyc(m16i, 16));
    printf("  10 dependent v_mfma_f32_32x32x8_bf16_1k          %10.1f us   %6.2f\n", m8, cyc(m8, 16));
    printf("  11 dependent v_mfma_f32_32x32x2_f32              %10.1f us   %6.2f\n", mf, cyc(mf, 16));
    printf("  12 dependent v_mfma_f32_16x16x32_bf16            %10.1f us   %6.2f\n", m1616, cyc(m1616, 16));
    printf("  3  v_fma_f32, 8 chains (96/iter)                 %10.1f us   %6.2f\n", fma, cyc(fma, 96));
    printf("  4  VALU with exp/rcp (96/iter)                   %10.1f us   %6.2f\n", tr, cyc(tr, 96));
    printf("  13 v_mov_b32 (96/iter)                           %10.1f us   %6.2f\n", mov, cyc(mov, 96));
    printf("  5  ds_read_b128 (16/iter)                        %10.1f us   %6.2f\n", dsr, cyc(dsr, 16));
    printf("pairs on one SIMD (A = waves 0..3, B = waves 4..7); 'exposed share of B' = (pair - A alone) / B alone: 0 %% = B hidden, 100 %% = times add\n");
    line<1, 3>("32x32x16 bf16 (dep) + v_fma", out, 0, m16, fma);
    line<2, 3>("32x32x16 bf16 (indep) + v_fma", out, 0, m16i, fma);
    line<10, 3>("32x32x8 bf16_1k (dep) + v_fma", out, 0, m8, fma);
    line<11, 3>("32x32x2 f32 (dep) + v_fma", out, 0, mf, fma);
    line<12, 3>("16x16x32 bf16 (dep) + v_fma", out, 0, m1616, fma);
    line<1, 13>("32x32x16 bf16 (dep) + v_mov", out, 0, m16, mov);
    line<11, 13>("32x32x2 f32 (dep) + v_mov", out, 0, mf, mov);
    line<1, 4>("32x32x16 bf16 (dep) + VALU exp/rcp", out, 0, m16, tr);
    line<11, 4>("32x32x2 f32 (dep) + VALU exp/rcp", out, 0, mf, tr);
    line<1, 7>("32x32x16 bf16 (dep) + v_fma @ s_setprio 3", out, 0, m16, fma);
    line<8, 3>("32x32x16 bf16 (dep) @ s_setprio 3 + v_fma", out, 0, m16, fma);
    line<1, 5>("32x32x16 bf16 (dep) + ds_read_b128", out, 0, m16, dsr);
    line<11, 5>("32x32x2 f32 (dep) + ds_read_b128", out, 0, mf, dsr);
    line<3, 5>("v_fma + ds_read_b128", out, 0, fma, dsr);
    line<1, 1>("32x32x16 bf16 (dep) + the same", out, 0, m16, m16);
    line<3, 3>("v_fma + v_fma", out, 0, fma, fma);
    printf("one wave (B idle):\n");
    line<6, 0>("32x32x16 bf16 (dep) with 6 v_fma in every gap", out, 16, 0, 0);
    line<9, 0>("32x32x16 bf16 (dep) with 4 v_fma + 2 exp + 2 rcp per gap", out, 16, 0, 0);
    line<6, 6>("both waves: MFMA + 6 v_fma per gap each", out, 16, 0, 0);
    line<14, 0>("32x32x16 bf16 (4 independent acc) with 2 v_fma per gap", out, 16, 0, 0);
    line<15, 0>("32x32x16 bf16 (4 independent acc) with 4 v_fma per gap", out, 16, 0, 0);
    line<16, 0>("32x32x16 bf16 (4 independent acc) with 6 v_fma per gap", out, 16, 0, 0);
    line<17, 0>("32x32x16 bf16 (4 independent acc) with 4 v_mov per gap", out, 16, 0, 0);
    line<18, 0>("32x32x16 bf16 (4 independent acc) with 4 v_cvt_pk per gap", out, 16, 0, 0);
    line<19, 0>("32x32x16 bf16 (4 independent acc) with 2 v_exp per gap", out, 16, 0, 0);
    line<15, 15>("both waves: 4 independent acc + 4 v_fma per gap each", out, 16, 0, 0);
    return 0;
}
