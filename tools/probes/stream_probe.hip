// Read-stream rates of a [M][K] bf16 matrix (M = 205,312): what access SHAPE does HBM want?
//   mode 0: contiguous -- workgroup g reads its slab front to back, 16 B per lane, 8 loads in flight per lane
//   mode 1: K-chunked  -- workgroup g owns 256-row tiles; per 64-element K chunk it reads 256 rows x 128 B (8 lanes per row),
//           chunk after chunk (the access order of a C-stationary GEMM)
//   mode 2: as 1 with 128-element chunks (256 B per row and chunk)
// build: hipcc --offload-arch=gfx950 -O3 stream_probe.hip -o stream_probe ; run: ./stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(512) probe(const uint16_t *A, uint32_t *out, int64_t M, int K, int mode, int rows_per_wg) {
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg, r1 = r0 + rows_per_wg < M ? r0 + rows_per_wg : M;
    u4 acc = {0, 0, 0, 0};
    if (mode == 0) {
        const u4 *p = (const u4 *)(A + r0 * K);
        const int64_t n = (r1 - r0) * K / 8;
        for (int64_t i = tid; i < n; i += 512 * 8) {
            u4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = i + j * 512 < n ? p[i + j * 512] : acc;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc ^= v[j];
        }
    } else {
        const int ck = mode == 1 ? 64 : 128, lpr = ck / 8;   // lanes per row
        const int rpi = 512 / lpr;                             // rows per pass of the workgroup
        for (int64_t t = r0; t < r1; t += 256)
            for (int kt = 0; kt < K / ck; ++kt) {
                u4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int row = j * rpi + tid / lpr;
                    int64_t m = t + row; m = m < M ? m : M - 1;
                    v[j] = row < 256 ? *(const u4 *)(A + m * K + kt * ck + (tid % lpr) * 8) : acc;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) acc ^= v[j];
            }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[blockIdx.x] = acc.x;
}

int main() {
    const int64_t M = 205312;
    for (int K : {704, 1408}) {
        uint16_t *A; uint32_t *out;
        hipMalloc(&A, M * K * 2); hipMalloc(&out, 4096 * 4);
        hipMemset(A, 1, M * K * 2);
        for (int wgs : {256, 512, 1024, 2048})
            for (int mode = 0; mode < 3; ++mode) {
                int rows = (int)((M + wgs - 1) / wgs); rows = (rows + 255) / 256 * 256;
                const int grid = (int)((M + rows - 1) / rows);
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(probe, dim3(grid), dim3(512), 0, 0, A, out, M, K, mode, rows);
                hipEventRecord(e0);
                for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(probe, dim3(grid), dim3(512), 0, 0, A, out, M, K, mode, rows);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                printf("K=%4d wgs=%4d (grid %4d, %5d rows) mode %d: %7.1f us  %.2f TB/s\n", K, wgs, grid, rows, mode, ms * 100, M * K * 2 / (ms / 10 * 1e-3) / 1e12);
            }
        hipFree(A); hipFree(out);
    }
    return 0;
}
