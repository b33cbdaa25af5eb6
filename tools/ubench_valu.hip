// ubench_valu.hip — VALU issue-rate microbenchmark for gfx950: plain vs packed fp32 ops,
// VGPR vs SGPR operands, 1..8 waves per SIMD.  Prints lane-ops/clk/CU for each case.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_valu.bin tools/ubench_valu.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float float2v __attribute__((ext_vector_type(2)));

#define REP8(X) X X X X X X X X

template <int MODE>
__global__ void __launch_bounds__(256) k_valu(float* out, float k, int iters) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float2v p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x0}, p5 = {x3, x2}, p6 = {x5, x4}, p7 = {x7, x6};
    float2v k2 = {k, k};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {   // v_mul_f32, SGPR operand
            REP8(asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                              "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(k));)
        } else if (MODE == 1) {   // v_pk_mul_f32, SGPR pair operand
            REP8(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                              "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "s"(k2));)
        } else if (MODE == 2) {   // v_add_f32 VGPR operands
            REP8(asm volatile("v_add_f32 %0, %1, %0\n v_add_f32 %1, %2, %1\n v_add_f32 %2, %3, %2\n v_add_f32 %3, %4, %3\n"
                              "v_add_f32 %4, %5, %4\n v_add_f32 %5, %6, %5\n v_add_f32 %6, %7, %6\n v_add_f32 %7, %0, %7\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));)
        } else if (MODE == 3) {   // v_fma_f32
            REP8(asm volatile("v_fma_f32 %0, %8, %0, %0\n v_fma_f32 %1, %8, %1, %1\n v_fma_f32 %2, %8, %2, %2\n v_fma_f32 %3, %8, %3, %3\n"
                              "v_fma_f32 %4, %8, %4, %4\n v_fma_f32 %5, %8, %5, %5\n v_fma_f32 %6, %8, %6, %6\n v_fma_f32 %7, %8, %7, %7\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(k));)
        } else if (MODE == 4) {   // v_pk_add_f32 VGPR operands
            REP8(asm volatile("v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %2, %2, %3\n v_pk_add_f32 %3, %3, %4\n"
                              "v_pk_add_f32 %4, %4, %5\n v_pk_add_f32 %5, %5, %6\n v_pk_add_f32 %6, %6, %7\n v_pk_add_f32 %7, %7, %0\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7));)
        } else if (MODE == 5) {   // v_pk_fma_f32
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %1, %1, %8, %1\n v_pk_fma_f32 %2, %2, %8, %2\n v_pk_fma_f32 %3, %3, %8, %3\n"
                              "v_pk_fma_f32 %4, %4, %8, %4\n v_pk_fma_f32 %5, %5, %8, %5\n v_pk_fma_f32 %6, %6, %8, %6\n v_pk_fma_f32 %7, %7, %8, %7\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "s"(k2));)
        } else if (MODE == 6) {   // v_cmp_lt_f32 to SGPR pair (VOP3)
            unsigned long long m;
            REP8(asm volatile("v_cmp_lt_f32 %0, %1, %2\n v_cmp_lt_f32 %0, %2, %3\n v_cmp_lt_f32 %0, %3, %4\n v_cmp_lt_f32 %0, %4, %1\n"
                              "v_cmp_lt_f32 %0, %1, %3\n v_cmp_lt_f32 %0, %2, %4\n v_cmp_lt_f32 %0, %3, %1\n v_cmp_lt_f32 %0, %4, %2\n"
                              : "=s"(m) : "v"(x0), "v"(x1), "v"(x2), "v"(x3));)
            if (m == 0x1234567ull) x0 += 1.0f;
        } else if (MODE == 7) {   // v_rcp_f32
            REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                              "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));)
        }
    }
    float r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y +
              p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
    if (r == 12345.678f) out[0] = r;
}

template <int MODE>
double run(int waves_per_simd, int iters, float* d_out, int cus) {
    dim3 block(256), grid(cus * waves_per_simd);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(k_valu<MODE>, grid, block, 0, 0, d_out, 1.0000001f, 16);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k_valu<MODE>, grid, block, 0, 0, d_out, 1.0000001f, iters);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a);
    hipEventDestroy(b);
    double insts_per_wave = 64.0 * iters;                    // 8 x 8 instructions per iteration
    double wave_insts = insts_per_wave * 4.0 * grid.x;       // 4 waves per block
    return wave_insts / (ms * 1e-3);                         // wave-instructions per second, whole chip
}

int main() {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { printf("no device\n"); return 1; }
    int cus = prop.multiProcessorCount;
    double clk = prop.clockRate * 1e3;
    printf("device %s, %d CUs, clock %.0f MHz\n", prop.name, cus, clk / 1e6);
    float* d_out;
    hipMalloc(&d_out, 4);
    const char* names[] = {"v_mul_f32 (sgpr)", "v_pk_mul_f32 (sgpr pair)", "v_add_f32", "v_fma_f32 (sgpr)",
                           "v_pk_add_f32", "v_pk_fma_f32 (sgpr pair)", "v_cmp_lt_f32 ->sgpr", "v_rcp_f32"};
    int iters = 20000;
    for (int w = 1; w <= 8; w *= 2) {
        double r[8];
        r[0] = run<0>(w, iters, d_out, cus);
        r[1] = run<1>(w, iters, d_out, cus);
        r[2] = run<2>(w, iters, d_out, cus);
        r[3] = run<3>(w, iters, d_out, cus);
        r[4] = run<4>(w, iters, d_out, cus);
        r[5] = run<5>(w, iters, d_out, cus);
        r[6] = run<6>(w, iters, d_out, cus);
        r[7] = run<7>(w, iters, d_out, cus);
        for (int m = 0; m < 8; ++m)
            printf("waves/SIMD %d  %-26s  %8.3f Gwave-inst/s  = %6.3f wave-inst/clk/SIMD (cycles per wave-inst per SIMD %.2f at nominal clk)\n",
                   w, names[m], r[m] / 1e9, r[m] / (cus * 4.0) / clk, (cus * 4.0) * clk / r[m]);
    }
    return 0;
}
