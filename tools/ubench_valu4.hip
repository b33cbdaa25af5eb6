// ubench_valu4.hip — fourth VALU issue probe for gfx950: do the "slow" forms (v_cmp, v_cndmask,
// SGPR operands, conversions, transcendentals; ~4.1 / 8.2 cycles per wave64 instruction when
// issued back to back) overlap with a VGPR-only instruction that follows them?  ubench_valu3
// measured v_cmp + v_mul + v_cndmask at 7.1 cycles for the three against 8.0 for v_cmp +
// v_cndmask alone, which says part of the cost is a dependency/port bubble that independent
// work can fill.  Every pattern below is reported per instruction AND per group.
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP8(X) X X X X X X X X
#define BODY8(OP) \
    REP8(asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) \
        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+v"(z0), "+v"(z1), "+v"(z2), "+v"(z3) \
        : "v"(y), "s"(m64), "s"(k), "v"(ui) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35");)
// operands: %0..%7 x (rw), %8..%11 z (rw), %12 y (v), %13 m64 (s pair), %14 k (s), %15 ui (v, integer)

#define MUL(i)       "v_mul_f32 %" #i ", %12, %" #i "\n"
#define P_CMP_VCC(i) "v_cmp_lt_f32 vcc, %12, %" #i "\n"
#define P_CMP_S0(i)  "v_cmp_lt_f32_e64 s[20:21], %12, %" #i "\n"
#define P_CMP_SD(i)  "v_cmp_lt_f32_e64 s[" SD(i) "], %12, %" #i "\n"
#define SD(i) SD_##i
#define SD_0 "20:21"
#define SD_1 "22:23"
#define SD_2 "24:25"
#define SD_3 "26:27"
#define SD_4 "28:29"
#define SD_5 "30:31"
#define SD_6 "32:33"
#define SD_7 "34:35"
#define P_CND_S(i)   "v_cndmask_b32_e64 %" #i ", %12, %" #i ", %13\n"
#define P_CND_VCC(i) "v_cndmask_b32 %" #i ", %12, %" #i ", vcc\n"
#define P_ADD_S(i)   "v_add_f32 %" #i ", %14, %" #i "\n"
#define P_MOV_S(i)   "v_mov_b32 %" #i ", %14\n"
#define P_CVT(i)     "v_cvt_f32_u32 %" #i ", %15\n"
#define P_RCP(i)     "v_rcp_f32 %" #i ", %" #i "\n"
#define P_SQRT(i)    "v_sqrt_f32 %" #i ", %" #i "\n"
#define P_MAX(i)     "v_max_f32 %" #i ", %12, %" #i "\n"
#define P_XOR(i)     "v_xor_b32 %" #i ", %15, %" #i "\n"
#define P_LSHL(i)    "v_lshlrev_b32 %" #i ", 3, %" #i "\n"
#define P_LSHLXOR(i) "v_lshl_add_u32 %" #i ", %" #i ", 3, %15\n"   /* VOP3 integer */
#define P_FMA(i)     "v_fma_f32 %" #i ", %12, %" #i ", %" #i "\n"
#define P_FMAC(i)    "v_fmac_f32 %" #i ", %12, %12\n"

// groups
#define G0(i)  P_CMP_SD(i)
#define G1(i)  P_CMP_VCC(i) MUL(i)
#define G2(i)  P_CMP_SD(i) MUL(i)
#define G3(i)  P_CND_S(i) MUL(i)
#define G4(i)  P_CMP_VCC(i) MUL(i) MUL(i) P_CND_VCC(i)
#define G5(i)  P_CMP_VCC(i) P_CND_VCC(i) MUL(i) MUL(i)
#define G6(i)  P_ADD_S(i) MUL(i)
#define G7(i)  P_MOV_S(i) MUL(i)
#define G8(i)  P_CVT(i) MUL(i)
#define G9(i)  P_RCP(i) MUL(i)
#define G10(i) P_RCP(i) MUL(i) MUL(i) MUL(i)
#define G11(i) P_SQRT(i) MUL(i) MUL(i) MUL(i)
#define G12(i) P_MAX(i) MUL(i)
#define G13(i) P_XOR(i)
#define G14(i) P_LSHL(i)
#define G15(i) P_LSHLXOR(i)
#define G16(i) P_FMA(i) MUL(i)
#define G17(i) P_FMAC(i)
#define G18(i) P_CMP_S0(i) "v_cndmask_b32_e64 %" #i ", %12, %" #i ", s[20:21]\n"
#define G19(i) P_CMP_SD(i) MUL(i) "v_cndmask_b32_e64 %" #i ", %12, %" #i ", s[" SD(i) "]\n"
#define G20(i) P_CVT(i)
#define G21(i) P_MAX(i) P_MAX(i)

template <int MODE>
__global__ void __launch_bounds__(256) k_valu(float* out, float k, int iters) {
    float x0 = threadIdx.x + 1.0f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float z0 = x0 * 0.5f, z1 = x1 * 0.5f, z2 = x2 * 0.5f, z3 = x3 * 0.5f;
    float y = 1.0000001f + threadIdx.x * 1e-9f;
    unsigned ui = threadIdx.x * 2654435761u + (unsigned)iters;
    unsigned long long m64 = 0x5555555555555555ull + (unsigned long long)iters;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { BODY8(G0) }
        if (MODE == 1) { BODY8(G1) }
        if (MODE == 2) { BODY8(G2) }
        if (MODE == 3) { BODY8(G3) }
        if (MODE == 4) { BODY8(G4) }
        if (MODE == 5) { BODY8(G5) }
        if (MODE == 6) { BODY8(G6) }
        if (MODE == 7) { BODY8(G7) }
        if (MODE == 8) { BODY8(G8) }
        if (MODE == 9) { BODY8(G9) }
        if (MODE == 10) { BODY8(G10) }
        if (MODE == 11) { BODY8(G11) }
        if (MODE == 12) { BODY8(G12) }
        if (MODE == 13) { BODY8(G13) }
        if (MODE == 14) { BODY8(G14) }
        if (MODE == 15) { BODY8(G15) }
        if (MODE == 16) { BODY8(G16) }
        if (MODE == 17) { BODY8(G17) }
        if (MODE == 18) { BODY8(G18) }
        if (MODE == 19) { BODY8(G19) }
        if (MODE == 20) { BODY8(G20) }
        if (MODE == 21) { BODY8(G21) }
    }
    float r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + z0 + z1 + z2 + z3;
    if (r == 12345.678f) out[0] = r;
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
    return 64.0 * iters * 4.0 * grid.x / (ms * 1e-3);   // groups per second
}

int main() {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { printf("no device\n"); return 1; }
    int cus = prop.multiProcessorCount;
    double clk = prop.clockRate * 1e3;
    float* d_out;
    (void)hipMalloc(&d_out, 4);
    struct { const char* name; int n; } pat[] = {
        {"v_cmp_e64 -> 8 different SGPR pairs", 1}, {"v_cmp vcc | v_mul", 2}, {"v_cmp_e64 s[i] | v_mul", 2}, {"v_cndmask_e64 s[] | v_mul", 2},
        {"v_cmp vcc | v_mul | v_mul | v_cndmask vcc", 4}, {"v_cmp vcc | v_cndmask vcc | v_mul | v_mul", 4}, {"v_add s,v | v_mul", 2},
        {"v_mov v,s | v_mul", 2}, {"v_cvt_f32_u32 | v_mul", 2}, {"v_rcp | v_mul", 2}, {"v_rcp | 3 v_mul", 4}, {"v_sqrt | 3 v_mul", 4},
        {"v_max | v_mul", 2}, {"v_xor v,v", 1}, {"v_lshlrev imm", 1}, {"v_lshl_add_u32 (VOP3 int)", 1}, {"v_fma | v_mul", 2}, {"v_fmac v,v (VOP2)", 1},
        {"v_cmp_e64 s[20:21] | v_cndmask_e64 s[20:21]", 2}, {"v_cmp_e64 s[i] | v_mul | v_cndmask_e64 s[i]", 3}, {"v_cvt_f32_u32", 1},
        {"v_max | v_max", 2}};
    int iters = 10000;
    for (int w = 2; w <= 8; w *= 4) {
        double r[22];
        r[0] = run<0>(w, iters, d_out, cus);   r[1] = run<1>(w, iters, d_out, cus);   r[2] = run<2>(w, iters, d_out, cus);
        r[3] = run<3>(w, iters, d_out, cus);   r[4] = run<4>(w, iters, d_out, cus);   r[5] = run<5>(w, iters, d_out, cus);
        r[6] = run<6>(w, iters, d_out, cus);   r[7] = run<7>(w, iters, d_out, cus);   r[8] = run<8>(w, iters, d_out, cus);
        r[9] = run<9>(w, iters, d_out, cus);   r[10] = run<10>(w, iters, d_out, cus); r[11] = run<11>(w, iters, d_out, cus);
        r[12] = run<12>(w, iters, d_out, cus); r[13] = run<13>(w, iters, d_out, cus); r[14] = run<14>(w, iters, d_out, cus);
        r[15] = run<15>(w, iters, d_out, cus); r[16] = run<16>(w, iters, d_out, cus); r[17] = run<17>(w, iters, d_out, cus);
        r[18] = run<18>(w, iters, d_out, cus); r[19] = run<19>(w, iters, d_out, cus); r[20] = run<20>(w, iters, d_out, cus);
        r[21] = run<21>(w, iters, d_out, cus);
        for (int m = 0; m < 22; ++m) {
            const double per_group = (cus * 4.0) * clk / r[m];
            printf("waves/SIMD %d  %-46s  cycles per group %.2f  per instruction %.2f\n", w, pat[m].name, per_group, per_group / pat[m].n);
        }
    }
    return 0;
}
