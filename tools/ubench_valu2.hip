// ubench_valu2.hip — second VALU issue-rate probe for gfx950: which operand kinds /
// encodings run at the SIMD-32 rate (2 cycles per wave64 instruction) and which at 4.
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP8(X) X X X X X X X X
#define BODY8(OP) \
    REP8(asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) \
        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(k), "v"(y), "s"(m64) : "vcc");)

#define OP_MUL_V(i)   "v_mul_f32 %" #i ", %9, %" #i "\n"
#define OP_ADD_S(i)   "v_add_f32 %" #i ", %8, %" #i "\n"
#define OP_FMA_V(i)   "v_fma_f32 %" #i ", %9, %" #i ", %" #i "\n"
#define OP_ADD3_ABS(i) "v_add_f32_e64 %" #i ", |%9|, %" #i "\n"
#define OP_SUB_NEG(i)  "v_sub_f32_e64 %" #i ", -%9, %" #i "\n"
#define OP_CMP_VCC(i) "v_cmp_lt_f32 vcc, %9, %" #i "\n"
#define OP_CMP_VCC_S(i) "v_cmp_lt_f32 vcc, %8, %" #i "\n"
#define OP_CNDMASK(i) "v_cndmask_b32 %" #i ", %9, %" #i ", vcc\n"
#define OP_CNDMASK_S(i) "v_cndmask_b32_e64 %" #i ", %9, %" #i ", %10\n"
#define OP_MOV_S(i)   "v_mov_b32 %" #i ", %8\n"
#define OP_MOV_V(i)   "v_mov_b32 %" #i ", %9\n"
#define OP_MUL_LIT(i) "v_mul_f32 %" #i ", 0x3f800001, %" #i "\n"
#define OP_MUL_INL(i) "v_mul_f32 %" #i ", 1.0, %" #i "\n"
#define OP_XOR_V(i)   "v_xor_b32 %" #i ", %9, %" #i "\n"
#define OP_MAX_V(i)   "v_max_f32 %" #i ", %9, %" #i "\n"
#define OP_SQRT(i)    "v_sqrt_f32 %" #i ", %" #i "\n"
#define OP_MULLEG(i)  "v_mul_legacy_f32 %" #i ", %9, %" #i "\n"
#define OP_RDLANE(i)  "v_readfirstlane_b32 s20, %" #i "\n"
#define OP_DIVFIX(i)  "v_div_fixup_f32 %" #i ", %9, %" #i ", %" #i "\n"
#define OP_MED3(i)    "v_med3_f32 %" #i ", %9, %" #i ", %" #i "\n"

template <int MODE>
__global__ void __launch_bounds__(256) k_valu(float* out, float k, int iters) {
    float x0 = threadIdx.x + 1.0f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float y = 1.0000001f + threadIdx.x * 1e-9f;
    unsigned long long m64 = 0x5555555555555555ull + (unsigned long long)iters;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { BODY8(OP_MUL_V) }
        if (MODE == 1) { BODY8(OP_ADD_S) }
        if (MODE == 2) { BODY8(OP_FMA_V) }
        if (MODE == 3) { BODY8(OP_ADD3_ABS) }
        if (MODE == 4) { BODY8(OP_SUB_NEG) }
        if (MODE == 5) { BODY8(OP_CMP_VCC) }
        if (MODE == 6) { BODY8(OP_CMP_VCC_S) }
        if (MODE == 7) { BODY8(OP_CNDMASK) }
        if (MODE == 8) { BODY8(OP_CNDMASK_S) }
        if (MODE == 9) { BODY8(OP_MOV_S) }
        if (MODE == 10) { BODY8(OP_MOV_V) }
        if (MODE == 11) { BODY8(OP_MUL_LIT) }
        if (MODE == 12) { BODY8(OP_MUL_INL) }
        if (MODE == 13) { BODY8(OP_XOR_V) }
        if (MODE == 14) { BODY8(OP_MAX_V) }
        if (MODE == 15) { BODY8(OP_SQRT) }
        if (MODE == 16) { BODY8(OP_DIVFIX) }
        if (MODE == 17) { BODY8(OP_MED3) }
    }
    float r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if (r == 12345.678f) out[0] = r;
}

// LDS broadcast read rate: every lane reads the same 16 B (ds_read_b128) / 4 B (ds_read_b32).
template <int MODE>
__global__ void __launch_bounds__(256) k_lds(float* out, int iters) {
    __shared__ float4 tile[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) tile[i] = make_float4(i, i + 1, i + 2, i + 3);
    __syncthreads();
    float4 acc = make_float4(0, 0, 0, 0);
    float a1 = 0;
    unsigned base = (unsigned)(size_t)tile;   // LDS byte address (low 32 bits of the generic pointer are the LDS offset)
    for (int i = 0; i < iters; ++i) {
        unsigned addr = ((i * 64) & 1023) * 16u;
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                float4 v;
                asm volatile("ds_read_b128 %0, %1 offset:%2\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr), "i"(j * 64));
                acc.x += v.x;
            }
        } else if (MODE == 1) {
            float4 v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[j]) : "v"(addr), "i"(j * 64));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 16; ++j) acc.x += v[j].x;
        } else {
            float v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[j]) : "v"(addr), "i"(j * 64));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 16; ++j) a1 += v[j];
        }
    }
    (void)base;
    if (acc.x + a1 == 12345.678f) out[0] = acc.x;
}

template <int MODE>
double run(int w, int iters, float* d_out, int cus) {
    dim3 block(256), grid(cus * w);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k_valu<MODE>, grid, block, 0, 0, d_out, 1.0000001f, 16);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL(k_valu<MODE>, grid, block, 0, 0, d_out, 1.0000001f, iters);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    return 64.0 * iters * 4.0 * grid.x / (ms * 1e-3);
}
template <int MODE>
double run_lds(int w, int iters, float* d_out, int cus) {
    dim3 block(256), grid(cus * w);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k_lds<MODE>, grid, block, 0, 0, d_out, 16);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL(k_lds<MODE>, grid, block, 0, 0, d_out, iters);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    return 16.0 * iters * 4.0 * grid.x / (ms * 1e-3);   // ds wave-instructions per second
}

int main() {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { printf("no device\n"); return 1; }
    int cus = prop.multiProcessorCount;
    double clk = prop.clockRate * 1e3;
    float* d_out;
    (void)hipMalloc(&d_out, 4);
    const char* names[] = {"v_mul_f32 vgpr,vgpr", "v_add_f32 sgpr,vgpr", "v_fma_f32 vgpr x3 (VOP3)", "v_add_f32_e64 |v|,v (VOP3 mod)",
                           "v_sub_f32_e64 -v,v", "v_cmp_lt_f32 vcc,v,v (VOPC)", "v_cmp_lt_f32 vcc,s,v", "v_cndmask_b32 v,v,vcc",
                           "v_cndmask_b32_e64 v,v,s[]", "v_mov_b32 v,s", "v_mov_b32 v,v", "v_mul_f32 literal,v", "v_mul_f32 inline 1.0,v",
                           "v_xor_b32 v,v", "v_max_f32 v,v", "v_sqrt_f32", "v_div_fixup_f32", "v_med3_f32"};
    int iters = 10000;
    for (int w = 2; w <= 8; w *= 4) {
        double r[18];
        r[0] = run<0>(w, iters, d_out, cus);   r[1] = run<1>(w, iters, d_out, cus);   r[2] = run<2>(w, iters, d_out, cus);
        r[3] = run<3>(w, iters, d_out, cus);   r[4] = run<4>(w, iters, d_out, cus);   r[5] = run<5>(w, iters, d_out, cus);
        r[6] = run<6>(w, iters, d_out, cus);   r[7] = run<7>(w, iters, d_out, cus);   r[8] = run<8>(w, iters, d_out, cus);
        r[9] = run<9>(w, iters, d_out, cus);   r[10] = run<10>(w, iters, d_out, cus); r[11] = run<11>(w, iters, d_out, cus);
        r[12] = run<12>(w, iters, d_out, cus); r[13] = run<13>(w, iters, d_out, cus); r[14] = run<14>(w, iters, d_out, cus);
        r[15] = run<15>(w, iters, d_out, cus); r[16] = run<16>(w, iters, d_out, cus); r[17] = run<17>(w, iters, d_out, cus);
        for (int m = 0; m < 18; ++m)
            printf("waves/SIMD %d  %-32s  cycles per wave-inst per SIMD %.2f\n", w, names[m], (cus * 4.0) * clk / r[m]);
    }
    for (int w = 1; w <= 8; w *= 2) {
        double a = run_lds<0>(w, 4000, d_out, cus), b = run_lds<1>(w, 4000, d_out, cus), c = run_lds<2>(w, 4000, d_out, cus);
        printf("waves/SIMD %d  ds_read_b128 broadcast: dependent %.1f cyc/inst/CU, pipelined %.1f cyc/inst/CU; ds_read_b32 pipelined %.1f cyc/inst/CU\n",
               w, cus * clk / a, cus * clk / b, cus * clk / c);
    }
    return 0;
}
