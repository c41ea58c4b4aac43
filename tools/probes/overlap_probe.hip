// Do the matrix pipe and the VALU overlap on a gfx950 SIMD, (a) between two waves, (b) inside one wave?  And what is the shader
// clock under each load?  One workgroup per CU; every wave runs R rounds of a "tile": 16 MFMA 32x32x16 bf16 (512 pipe cycles)
// and / or 112 VALU issue slots shaped like an online-softmax tile (16 fma, 16 exp, 16 add, 8 cvt_pk, 8 max3).
//   mode 0  MFMA only                     mode 1  VALU only
//   mode 2  MFMA block then VALU block    mode 3  same wave, interleaved 1 MFMA : 7 VALU instructions (sched_group_barrier)
//   mode 4  waves 0-3 MFMA only, waves 4-7 VALU only (needs 8 waves: one of each per SIMD)
//   mode 6  as 2 plus a workgroup barrier per round (the streamed kernels' lock step)
//   mode 7  as 6, waves 4-7 run the VALU block first (half a round out of phase with waves 0-3)
//   mode 5  as 2 with independent data (the VALU block does not consume the MFMA results: the compiler may hoist / overlap)
// Prints the time per round and wave in ns and in shader cycles (s_memtime) and the shader clock = cycles / wall time.
// build: hipcc --offload-arch=gfx950 -O3 overlap_probe.hip -o overlap_probe.bin ; run: ./overlap_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 hbf16x2 __attribute__((ext_vector_type(2)));
typedef float f32v2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pack(float a, float b) {
    const f32v2 f = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, hbf16x2));
}

__device__ __forceinline__ void valu_tile(const f32x16 &s, float c2, float &m, float &lsum, float &mx, uint32_t (&pk)[8]) {
    float pr[16];
#pragma unroll
    for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(s[r], s[r + 1]), mx);
#pragma unroll
    for (int r = 0; r < 16; ++r) { pr[r] = __builtin_amdgcn_exp2f(fmaf(s[r], c2, -m)); lsum += pr[r]; }
#pragma unroll
    for (int r = 0; r < 8; ++r) pk[r] ^= pack(pr[2 * r], pr[2 * r + 1]);
    m += 1e-9f * pr[15];   // the next round's exponentials depend on this round (nothing is loop-invariant)
}

template <int MODE>
__global__ void __launch_bounds__(512) probe(float *out, long long *clk, int R, float c2) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3c00 + lane + i); b[i] = (short)(0x3b00 + lane * 3 + i); }
    f32x16 acc[4], s;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = 0.01f * (lane + e);
    float lsum = 0.f, mx = 0.f, m = 1.0f;
    uint32_t pk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool do_mfma = MODE == 0 || MODE == 2 || MODE == 3 || MODE == 5 || MODE == 6 || MODE == 7 || (MODE == 4 && wave < 4);
    const bool do_valu = MODE == 1 || MODE == 2 || MODE == 3 || MODE == 5 || MODE == 6 || MODE == 7 || (MODE == 4 && wave >= 4);
    __syncthreads();
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int r = 0; r < R; ++r) {
        if (MODE == 3) {
            // one tile's MFMAs and the VALU work of the previous tile's scores, interleaved in the instruction stream
            f32x16 sv = s;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j & 3], 0, 0, 0);
            valu_tile(sv, c2, m, lsum, mx, pk);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);   // 5 VALU
            }
            s[r & 15] += acc[0][0] * 1e-30f;
        } else if (MODE == 7 && wave >= 4) {
            valu_tile(s, c2, m, lsum, mx, pk);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            s[r & 15] += acc[0][0] * 1e-30f;
            __builtin_amdgcn_s_barrier();
        } else {
            if (do_mfma) {
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j & 3], 0, 0, 0);
            }
            if (MODE == 2 || MODE == 6 || MODE == 7) { __builtin_amdgcn_sched_barrier(0); s[r & 15] += acc[0][0] * 1e-30f; }   // the VALU block waits for the MFMA results
            if (do_valu) valu_tile(s, c2, m, lsum, mx, pk);
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 6 || MODE == 7) __builtin_amdgcn_s_barrier();
        }
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    float r0 = lsum + mx;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) r0 += acc[j][e];
#pragma unroll
    for (int i = 0; i < 8; ++i) r0 += (float)pk[i];
    if (r0 == 123.456f) out[threadIdx.x] = r0;
    if (blockIdx.x == 0 && lane == 0) { clk[2 * wave] = t1 - t0; clk[2 * wave + 1] = w1 - w0; }
}

template <int MODE> void run(int waves, int R, float *out, long long *clk) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(64 * waves), 0, 0, out, clk, R, 0.125f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(64 * waves), 0, 0, out, clk, R, 0.125f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[16]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    // wall_clock64 ticks at 100 MHz
    const double cyc = (double)h[0] / R, wall_ns = (double)h[1] * 10.0 / R;
    const double cyc_l = (double)h[2 * (waves - 1)] / R;
    printf("mode %d waves %d: %8.1f ns/round (event) | wave0 %7.0f cyc/round, last wave %7.0f | shader clock %.2f GHz\n", MODE, waves, ms * 1e6 / R, cyc, cyc_l,
           cyc / wall_ns);
}

int main() {
    float *out; long long *clk;
    hipMalloc(&out, 4096); hipMalloc(&clk, 256);
    const int R = 20000;
    for (int waves : {4, 8}) {
        run<0>(waves, R, out, clk); run<1>(waves, R, out, clk); run<2>(waves, R, out, clk); run<5>(waves, R, out, clk); run<3>(waves, R, out, clk);
        if (waves == 8) { run<4>(waves, R, out, clk); run<6>(waves, R, out, clk); run<7>(waves, R, out, clk); }
    }
    return 0;
}
