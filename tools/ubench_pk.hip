// ubench_pk.hip — does gfx950 issue packed FP32 (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32: two IEEE fp32 operations per lane)
// at the rate of the scalar forms?  Cycles per wave64 instruction PER SIMD with 1, 2 and 4 wavefronts per SIMD, operands in VGPRs.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_pk.hip -o tools/ubench_pk.bin && tools/ubench_pk.bin
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP8(X) X X X X X X X X
#define OPS(OP) REP8(asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y));)
#define MUL(i)   "v_mul_f32 %" #i ", %8, %" #i "\n"
#define ADD(i)   "v_add_f32 %" #i ", %8, %" #i "\n"
#define FMA(i)   "v_fma_f32 %" #i ", %8, %" #i ", %" #i "\n"
#define PKMUL(i) "v_pk_mul_f32 %" #i ", %9, %" #i "\n"
#define PKADD(i) "v_pk_add_f32 %" #i ", %9, %" #i "\n"
#define PKFMA(i) "v_pk_fma_f32 %" #i ", %9, %" #i ", %" #i "\n"
#define OPS2(OP) REP8(asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) \
    : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : "v"(y), "v"(y2));)

typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(1024) k(float* out, long long* cyc, int iters) {
    float a0 = threadIdx.x + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float2v b0 = {a0, a1}, b1 = {a1, a2}, b2 = {a2, a3}, b3 = {a3, a4}, b4 = {a4, a5}, b5 = {a5, a6}, b6 = {a6, a7}, b7 = {a7, a0};
    float y = 1.0000001f;
    float2v y2 = {1.0000001f, 0.9999999f};
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { OPS(MUL) }
        if (MODE == 1) { OPS(ADD) }
        if (MODE == 2) { OPS(FMA) }
        if (MODE == 3) { OPS2(PKMUL) }
        if (MODE == 4) { OPS2(PKADD) }
        if (MODE == 5) { OPS2(PKFMA) }
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0.x + b1.y + b2.x + b3.y + b4.x + b5.y + b6.x + b7.y;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
static void run(const char* name, float* d_out, long long* d_cyc) {
    const int iters = 20000;
    for (int threads : {256, 512, 1024, 2048}) {   // 1, 2, 4 wavefronts per SIMD (one block per CU); 8: two blocks of 1024 per CU
        const int blocks = threads > 1024 ? 512 : 256, tpb = threads > 1024 ? 1024 : threads;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(tpb), 0, 0, d_out, d_cyc, iters);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(tpb), 0, 0, d_out, d_cyc, iters);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0.0f;
        hipEventElapsedTime(&ms, e0, e1);
        long long c = 0;
        hipMemcpy(&c, d_cyc, sizeof c, hipMemcpyDeviceToHost);
        const double per_wave = (double)c / (iters * 64.0), waves = threads / 256.0;
        // wall clock: every SIMD issues iters * 64 * waves instructions in ms
        printf("%-14s %d wave(s)/SIMD: %.2f s_memtime ticks per instruction per wave; wall clock %.3f ms = %.2f ns per instruction per SIMD\n", name, (int)waves,
               per_wave, ms, ms * 1e6 / (iters * 64.0 * waves));
    }
}

int main() {
    float* d_out; long long* d_cyc;
    hipMalloc(&d_out, 512 * 1024 * sizeof(float)); hipMalloc(&d_cyc, sizeof(long long));
    run<0>("v_mul_f32", d_out, d_cyc); run<1>("v_add_f32", d_out, d_cyc); run<2>("v_fma_f32", d_out, d_cyc);
    run<3>("v_pk_mul_f32", d_out, d_cyc); run<4>("v_pk_add_f32", d_out, d_cyc); run<5>("v_pk_fma_f32", d_out, d_cyc);
    return 0;
}
