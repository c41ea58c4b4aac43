// VALU issue-rate probe for gfx950: cycles per wave-instruction (s_memtime) and chip throughput for
// independent / dependent v_fma_f32, v_pk_fma_f32, transcendentals, DPP adds and LDS-broadcast operands at 1, 2 and 4
// waves per SIMD.  Decides how the serial GRU kernels should split a path over waves.
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue_probe.bin valu_issue_probe.hip && ./valu_issue_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

template <int KIND>
__global__ void probe(float *out, long long *cyc, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b0 = seed * 0.5f, b1 = seed * 0.25f, w0 = 1.0001f, w1 = 0.9999f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, q = {b0, b1}, w = {w0, w1};
    __shared__ float lds[256];
    lds[threadIdx.x & 255] = seed;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {  // 8 independent chains of v_fma_f32, 64 instructions per iteration
            asm volatile(REP4(REP4("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"))
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(w0));
        } else if (KIND == 1) {  // 4 independent chains of v_pk_fma_f32, 64 instructions
            asm volatile(REP16("v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q), "v"(w));
        } else if (KIND == 2) {  // one dependent chain of v_fma_f32
            asm volatile(REP16(REP4("v_fmac_f32 %0, %1, %2\n")) : "+v"(a0) : "v"(b0), "v"(w0));
        } else if (KIND == 3) {  // v_exp_f32, 4 independent
            asm volatile(REP16("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        } else if (KIND == 4) {  // quad_perm DPP adds, 4 independent
            asm volatile(REP16("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        } else if (KIND == 5) {  // v_rcp_f32
            asm volatile(REP16("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        } else if (KIND == 6) {  // pk_fma with a broadcast (op_sel) operand: acc.lo += q.lo*w.lo, acc.hi += q.hi*w.lo
            asm volatile(REP16("v_pk_fma_f32 %0, %4, %5, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %4, %5, %1 op_sel_hi:[1,0,1]\n"
                               "v_pk_fma_f32 %2, %4, %5, %2 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, %4, %5, %3 op_sel_hi:[1,0,1]\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q), "v"(w));
        } else if (KIND == 7) {  // v_fma_f32 with 3 distinct VGPR sources (non-fmac encoding), 8 chains
            asm volatile(REP4(REP4("v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n"))
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(w0));
        } else if (KIND == 8) {  // v_mul_f32 / v_add_f32 mix, independent
            asm volatile(REP16("v_add_f32 %0, %4, %0\n v_mul_f32 %1, %4, %1\n v_add_f32 %2, %5, %2\n v_mul_f32 %3, %5, %3\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(w0));
        } else if (KIND == 9) {  // v_dot2c_f32_bf16 (2 bf16 MACs per lane), 4 chains
            asm volatile(REP16("v_dot2c_f32_bf16 %0, %4, %5\n v_dot2c_f32_bf16 %1, %4, %5\n v_dot2c_f32_bf16 %2, %4, %5\n v_dot2c_f32_bf16 %3, %4, %5\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(w0));
        } else if (KIND == 10) {  // ds_read_b128 broadcast reads (all lanes of a quad share an address), 16 per iteration
            float4 r;
            asm volatile(REP16("ds_read_b128 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : "=v"(r) : "v"((threadIdx.x & 3) * 64));
            a0 += r.x;
        }
    }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int KIND>
static void run(const char *name, int per_iter, int macs_per_lane_instr) {
    float *out; long long *cyc;
    hipMalloc(&out, 256 * 1024 * sizeof(float) * 4); hipMalloc(&cyc, 64);
    const int iters = 4000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int threads = 256 * wps > 1024 ? 1024 : 256 * wps, blocks = 256 * (256 * wps / threads);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.0f);
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double n = (double)iters * per_iter;
        const double waves = (double)blocks * threads / 64;
        printf("%-34s waves/SIMD %d: %6.2f cycles/instr/wave (s_memtime), SIMD issue interval %5.2f cyc at %.2f GHz-equiv, "
               "%.1f G wave-instr/s", name, wps, c / n, c / n / wps, (double)c / (ms * 1e6), n * waves / (ms * 1e6));
        if (macs_per_lane_instr) printf(", %.1f TFLOP/s", 2.0 * macs_per_lane_instr * 64 * n * waves / (ms * 1e9));
        printf("\n");
    }
}

int main() {
    run<0>("v_fmac_f32 x8 independent", 64, 1);
    run<7>("v_fma_f32 (VOP3) x8 independent", 64, 1);
    run<2>("v_fmac_f32 dependent chain", 64, 1);
    run<1>("v_pk_fma_f32 x4 independent", 64, 2);
    run<6>("v_pk_fma_f32 op_sel broadcast", 64, 2);
    run<8>("v_add/v_mul mix", 64, 0);
    run<3>("v_exp_f32", 64, 0);
    run<5>("v_rcp_f32", 64, 0);
    run<4>("v_add_f32_dpp quad_perm", 64, 0);
    run<9>("v_dot2c_f32_bf16", 64, 2);
    run<10>("ds_read_b128 quad-broadcast", 16, 0);
    return 0;
}
