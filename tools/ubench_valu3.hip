// ubench_valu3.hip — third VALU issue-rate probe for gfx950: the packed fp32 forms
// (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32, IEEE-exact per component) against their scalar
// forms, and v_cndmask_b32 with vcc under different producers of vcc (round 1's probe measured
// 22.9 cycles for the VOP2 form with a never-written vcc and left it unexplained).
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(X) X X X X X X X X
#define BODY8P(OP) \
    REP8(asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) \
        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(q), "s"(sq) : "vcc");)
#define BODY8(OP) \
    REP8(asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) \
        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y), "s"(m64) : "vcc");)
#define BODY8PRE(PRE, OP) \
    REP8(asm volatile(PRE OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) \
        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y), "s"(m64) : "vcc");)

#define OP_PK_MUL(i)  "v_pk_mul_f32 %" #i ", %8, %" #i "\n"
#define OP_PK_ADD(i)  "v_pk_add_f32 %" #i ", %8, %" #i "\n"
#define OP_PK_FMA(i)  "v_pk_fma_f32 %" #i ", %8, %" #i ", %" #i "\n"
#define OP_PK_MUL_S(i) "v_pk_mul_f32 %" #i ", %9, %" #i "\n"
#define OP_PK_MUL_SEL(i) "v_pk_mul_f32 %" #i ", %8, %" #i " op_sel:[1,0] op_sel_hi:[0,1]\n"
#define OP_PK_MOV(i)  "v_pk_mov_b32 %" #i ", %8, %" #i " op_sel:[1,0]\n"
#define OP_MUL_V(i)   "v_mul_f32 %" #i ", %8, %" #i "\n"
#define OP_CNDMASK(i) "v_cndmask_b32 %" #i ", %8, %" #i ", vcc\n"
#define OP_CMP_CND(i) "v_cmp_lt_f32 vcc, %8, %" #i "\nv_cndmask_b32 %" #i ", %8, %" #i ", vcc\n"
#define OP_CMP_S_CND(i) "v_cmp_lt_f32_e64 s[20:21], %8, %" #i "\nv_cndmask_b32_e64 %" #i ", %8, %" #i ", s[20:21]\n"
#define OP_CMP_MUL_CND(i) "v_cmp_lt_f32 vcc, %8, %" #i "\nv_mul_f32 %" #i ", %8, %" #i "\nv_cndmask_b32 %" #i ", %8, %" #i ", vcc\n"

template <int MODE>
__global__ void __launch_bounds__(256) k_valu(float* out, float k, int iters) {
    float x0 = threadIdx.x + 1.0f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float y = 1.0000001f + threadIdx.x * 1e-9f;
    f2 p0 = {x0, x1}, p1 = {x1, x2}, p2 = {x2, x3}, p3 = {x3, x4}, p4 = {x4, x5}, p5 = {x5, x6}, p6 = {x6, x7}, p7 = {x7, x0};
    f2 q = {y, y * 1.0000002f};
    f2 sq = {k, k};
    unsigned long long m64 = 0x5555555555555555ull + (unsigned long long)iters;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { BODY8P(OP_PK_MUL) }
        if (MODE == 1) { BODY8P(OP_PK_ADD) }
        if (MODE == 2) { BODY8P(OP_PK_FMA) }
        if (MODE == 3) { BODY8P(OP_PK_MUL_S) }
        if (MODE == 4) { BODY8P(OP_PK_MUL_SEL) }
        if (MODE == 5) { BODY8P(OP_PK_MOV) }
        if (MODE == 6) { BODY8(OP_MUL_V) }
        if (MODE == 7) { BODY8(OP_CNDMASK) }                                     // vcc never written inside the loop
        if (MODE == 8) { BODY8PRE("s_mov_b64 vcc, %9\n", OP_CNDMASK) }           // vcc written by the SALU once per 8
        if (MODE == 9) { BODY8PRE("v_cmp_lt_f32 vcc, %8, %0\n", OP_CNDMASK) }     // vcc written by the VALU once per 8
        if (MODE == 10) { BODY8(OP_CMP_CND) }                                    // the compiler's usual pair (counts 2 per op)
        if (MODE == 11) { BODY8(OP_CMP_S_CND) }                                  // the same pair through an SGPR pair
        if (MODE == 12) { BODY8(OP_CMP_MUL_CND) }                                // pair with one independent instruction between (counts 3)
    }
    float r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p1.x + p2.x + p3.x + p4.x + p5.x + p6.x + p7.x + p0.y + p1.y + p2.y + p3.y + p4.y +
              p5.y + p6.y + p7.y;
    if (r == 12345.678f) out[0] = r;
}

template <int MODE>
double run(int w, int iters, float* d_out, int cus, int per_op) {
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
    return 64.0 * per_op * iters * 4.0 * grid.x / (ms * 1e-3);   // wave-instructions per second
}

int main() {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { printf("no device\n"); return 1; }
    int cus = prop.multiProcessorCount;
    double clk = prop.clockRate * 1e3;
    float* d_out;
    (void)hipMalloc(&d_out, 4);
    const char* names[] = {"v_pk_mul_f32 v,v (2 products)", "v_pk_add_f32 v,v", "v_pk_fma_f32 v,v,v", "v_pk_mul_f32 s[],v", "v_pk_mul_f32 v,v op_sel swap",
                           "v_pk_mov_b32 v,v", "v_mul_f32 v,v (for scale)", "v_cndmask vcc: vcc never written", "v_cndmask vcc: s_mov vcc per 8",
                           "v_cndmask vcc: v_cmp vcc per 8", "v_cmp vcc + v_cndmask vcc (per inst)", "v_cmp s[] + v_cndmask s[] (per inst)",
                           "v_cmp vcc, v_mul, v_cndmask (per inst)"};
    int iters = 10000;
    for (int w = 2; w <= 8; w *= 4) {
        double r[13];
        r[0] = run<0>(w, iters, d_out, cus, 1);   r[1] = run<1>(w, iters, d_out, cus, 1);   r[2] = run<2>(w, iters, d_out, cus, 1);
        r[3] = run<3>(w, iters, d_out, cus, 1);   r[4] = run<4>(w, iters, d_out, cus, 1);   r[5] = run<5>(w, iters, d_out, cus, 1);
        r[6] = run<6>(w, iters, d_out, cus, 1);   r[7] = run<7>(w, iters, d_out, cus, 1);   r[8] = run<8>(w, iters, d_out, cus, 1);
        r[9] = run<9>(w, iters, d_out, cus, 1);   r[10] = run<10>(w, iters, d_out, cus, 2); r[11] = run<11>(w, iters, d_out, cus, 2);
        r[12] = run<12>(w, iters, d_out, cus, 3);
        for (int m = 0; m < 13; ++m)
            printf("waves/SIMD %d  %-40s  cycles per wave-inst per SIMD %.2f\n", w, names[m], (cus * 4.0) * clk / r[m]);
    }
    return 0;
}
