// ubench_fp64.hip — issue cost of the fp64 / conversion / transcendental instructions the
// shading kernel's exact-division path uses, gfx950.  Cycles per wave64 instruction per SIMD.
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench_fp64.hip -o tools/ubench_fp64.bin
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP8(X) X X X X X X X X
#define D8(OP) REP8(asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) \
        : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(yd));)
#define F8(OP) REP8(asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) \
        : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(yf));)
#define DF8(OP) REP8(asm volatile(OP(0, 8) OP(1, 9) OP(2, 10) OP(3, 11) OP(4, 12) OP(5, 13) OP(6, 14) OP(7, 15) \
        : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), \
          "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7));)

#define OP_FMA64(i) "v_fma_f64 %" #i ", %8, %" #i ", %" #i "\n"
#define OP_MUL64(i) "v_mul_f64 %" #i ", %8, %" #i "\n"
#define OP_ADD64(i) "v_add_f64 %" #i ", %8, %" #i "\n"
#define OP_RCP64(i) "v_rcp_f64 %" #i ", %" #i "\n"
#define OP_CVT_D_F(i, j) "v_cvt_f64_f32 %" #i ", %" #j "\n"
#define OP_CVT_F_D(i, j) "v_cvt_f32_f64 %" #j ", %" #i "\n"
#define OP_FMA32(i) "v_fma_f32 %" #i ", %8, %" #i ", %" #i "\n"
#define OP_RCP32(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define OP_SQRT32(i) "v_sqrt_f32 %" #i ", %" #i "\n"
#define OP_CVT_F_U(i) "v_cvt_f32_u32 %" #i ", %" #i "\n"

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    double d0 = threadIdx.x + 1.0, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7;
    float f0 = threadIdx.x + 1.0f, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
    double yd = 1.0000001 + threadIdx.x * 1e-9;
    float yf = 1.0000001f;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { D8(OP_FMA64) }
        if (MODE == 1) { D8(OP_MUL64) }
        if (MODE == 2) { D8(OP_ADD64) }
        if (MODE == 3) { D8(OP_RCP64) }
        if (MODE == 4) { DF8(OP_CVT_D_F) }
        if (MODE == 5) { DF8(OP_CVT_F_D) }
        if (MODE == 6) { F8(OP_FMA32) }
        if (MODE == 7) { F8(OP_RCP32) }
        if (MODE == 8) { F8(OP_SQRT32) }
        if (MODE == 9) { F8(OP_CVT_F_U) }
    }
    double r = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
    if (r == 12345.678) out[0] = (float)r;
}

template <int MODE>
void run(const char* name, float* d_out) {
    const int iters = 2000, blocks = 256 * 8;   // 8 blocks of 4 waves per CU: 8 waves per SIMD
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double insts_per_simd = (double)iters * 64 * (blocks * 4 / 1024.0);   // wave-insts per SIMD
    printf("%-16s %.3f ms  %.2f cycles per wave64 inst per SIMD (at 2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / insts_per_simd);
}

int main() {
    float* d_out;
    hipMalloc(&d_out, 4);
    run<0>("v_fma_f64", d_out); run<1>("v_mul_f64", d_out); run<2>("v_add_f64", d_out); run<3>("v_rcp_f64", d_out);
    run<4>("v_cvt_f64_f32", d_out); run<5>("v_cvt_f32_f64", d_out); run<6>("v_fma_f32", d_out); run<7>("v_rcp_f32", d_out);
    run<8>("v_sqrt_f32", d_out); run<9>("v_cvt_f32_u32", d_out);
    return 0;
}
