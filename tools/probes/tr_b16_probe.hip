// Probe of ds_read_b64_tr_b16 lane semantics on gfx950: LDS holds lds[i] = i (as u16); lane l supplies the address of 4
// contiguous elements; prints what every lane receives.  Hypothesis: within a 16-lane group, lane m supplies block row m/4,
// columns (m%4)*4..+3 of a [4][16] block, and lane i receives column i (rows 0..3).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void probe(unsigned short *out, int R) {
    __shared__ unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x, m = l & 15, g = l >> 4;
    const int addr = g * 16 + (m >> 2) * R + (m & 3) * 4;
    s4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4 *)(lds + addr));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)r[j];
}
int main() {
    unsigned short *d, h[256];
    hipMalloc(&d, sizeof(h));
    const int R = 72;
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, R);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int m = l & 15, g = l >> 4;
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) {
            const int expect = g * 16 + j * R + m;
            printf(" %5d%s", h[l * 4 + j], h[l * 4 + j] == expect ? "" : "*");
            bad += h[l * 4 + j] != expect;
        }
        printf("\n");
    }
    printf("mismatches vs hypothesis: %d\n", bad);
    return 0;
}
