// Probe of global_load_lds_dwordx4 on gfx950: does lane l land at lds_base + 16*l, with a wave-uniform base?  Also a
// partially masked instruction (lanes >= 40 off) and a second instruction at base + 1024.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(const float4 *g, float *out) {
    __shared__ float4 lds[256];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = make_float4(-1.f, -1.f, -1.f, -1.f);
    __syncthreads();
    const int lane = threadIdx.x;
    __builtin_amdgcn_global_load_lds((const void *)(g + lane), (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
    if (lane < 40)
        __builtin_amdgcn_global_load_lds((const void *)(g + 64 + lane), (__attribute__((address_space(3))) void *)(lds + 64), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i].x;
}
int main() {
    float4 h[256]; float o[256];
    for (int i = 0; i < 256; ++i) h[i] = make_float4((float)i, 0, 0, 0);
    float4 *d; float *dout;
    hipMalloc(&d, sizeof(h)); hipMalloc(&dout, sizeof(o));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, dout);
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 128; ++i) {
        const float expect = (i < 64 || i < 104) ? (float)i : -1.f;
        if (o[i] != expect) { ++bad; printf("slot %d: got %g expected %g\n", i, o[i], expect); }
    }
    printf("first 8: %g %g %g %g %g %g %g %g | slot 103 %g slot 104 %g | mismatches %d\n", o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], o[103], o[104], bad);
    return 0;
}
