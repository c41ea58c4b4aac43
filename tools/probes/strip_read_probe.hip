// How fast can 256 CUs stream a row-major [M][K] bf16 matrix when every workgroup walks ITS 256 rows in column strips?  (round 6)
// The deep-reduction GEMMs (csrc/vsde_mlp.hip::deep256p_kernel) read their activation operand exactly like that: per tile of 64
// reduction indices a wave requests 32 rows x 128 bytes (4 x global_load_dwordx4, 8 lanes per row), two tiles ahead.  This probe issues
// the same requests without any MFMA / LDS work, for several lengths of the contiguous run per row and request group:
//     RUN = 128 bytes (the GEMM's pattern) | 256 | 512 | the whole row (rows-kernel pattern: a wave streams its rows end to end)
// Build + run on the GPU box:   hipcc --offload-arch=gfx950 -O3 -o /tmp/strip tools/probes/strip_read_probe.hip && /tmp/strip [M K]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// one workgroup = 8 waves x 32 rows; a wave walks its 32 rows in steps of RUN bytes per row; per step it covers 32 rows x RUN bytes with
// (32 * RUN / 1024) loads of 1 KB (64 lanes x 16 B, RUN / 16 lanes per row); DEPTH steps in flight (registers)
template <int RUN, int DEPTH>
__global__ void __launch_bounds__(512) strip_kernel(const char *X, int64_t M, int64_t pitch, int rotate, uint32_t *out) {
    constexpr int NL = 32 * RUN / 1024;        // loads per step
    constexpr int LPR = RUN / 16;              // lanes per row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row0 = (int64_t)blockIdx.x * 256 + wave * 32;
    if (row0 >= M) return;
    const int steps = (int)(pitch / RUN);
    const int rot = rotate ? (int)((blockIdx.x * 5u) % (unsigned)steps) : 0;
    u32x4 acc = {0, 0, 0, 0};
    u32x4 buf[DEPTH][NL];
    auto req = [&](int s, u32x4 (&b)[NL]) {
        const int ss = ((s < steps ? s : steps - 1) + rot) % steps;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            int64_t r = row0 + (64 / LPR) * i + lane / LPR;
            r = r < M ? r : M - 1;
            b[i] = __builtin_nontemporal_load((const u32x4 *)(X + r * pitch + (int64_t)ss * RUN + (lane % LPR) * 16));
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) req(d, buf[d]);
    for (int s = 0; s < steps; s += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int i = 0; i < NL; ++i) acc ^= buf[d][i];
            req(s + d + DEPTH, buf[d]);
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;   // keeps the loads alive
}

template <int RUN, int DEPTH>
static void run(const char *name, const char *X, int64_t M, int64_t pitch, int rotate, uint32_t *out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid((unsigned)((M + 255) / 256));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((strip_kernel<RUN, DEPTH>), grid, dim3(512), 0, 0, X, M, pitch, rotate, out);
    hipEventRecord(e0);
    const int n = 20;
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL((strip_kernel<RUN, DEPTH>), grid, dim3(512), 0, 0, X, M, pitch, rotate, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / n;
    printf("%-44s rotate %d  %8.1f us  %6.0f GB/s\n", name, rotate, us, (double)M * pitch / us / 1e3);
}

int main(int argc, char **argv) {
    const int64_t M = argc > 1 ? atoll(argv[1]) : 205312, K = argc > 2 ? atoll(argv[2]) : 1408;
    const int64_t pitch = K * 2;
    char *X; uint32_t *out;
    hipMalloc(&X, M * pitch); hipMalloc(&out, 64);
    hipMemset(X, 1, M * pitch);
    printf("M = %lld, K = %lld (row pitch %lld bytes, %.0f MB), grid %lld workgroups of 256 rows\n", (long long)M, (long long)K,
           (long long)pitch, M * pitch / 1e6, (long long)((M + 255) / 256));
    for (int rot = 0; rot < 2; ++rot) {
        run<128, 2>("run 128 B, 2 steps (8 KB/wave) in flight", X, M, pitch, rot, out);
        run<128, 4>("run 128 B, 4 steps (16 KB/wave) in flight", X, M, pitch, rot, out);
        if (pitch % 256 == 0) run<256, 2>("run 256 B, 2 steps (16 KB/wave) in flight", X, M, pitch, rot, out);
        if (pitch % 512 == 0) run<512, 1>("run 512 B, 1 step (16 KB/wave) in flight", X, M, pitch, rot, out);
        if (pitch % 256 == 0) run<256, 1>("run 256 B, 1 step (8 KB/wave) in flight", X, M, pitch, rot, out);
    }
    return 0;
}
