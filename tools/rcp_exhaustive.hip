// Exhaustive check: candidate reciprocal sequences vs the compiler's IEEE 1.0f/a on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ __forceinline__ float rcpA(float a) {   // rcp + 2 Newton steps
    float r = __builtin_amdgcn_rcpf(a);
    float e = __builtin_fmaf(-a, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    e = __builtin_fmaf(-a, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float rcpB(float a) {   // compiler's chain without scaling/fixup
    float r0 = __builtin_amdgcn_rcpf(a);
    float e0 = __builtin_fmaf(-a, r0, 1.0f);
    float r1 = __builtin_fmaf(e0, r0, r0);
    float e1 = __builtin_fmaf(-a, r1, 1.0f);
    float q1 = __builtin_fmaf(e1, r1, r1);
    float e2 = __builtin_fmaf(-a, q1, 1.0f);
    return __builtin_fmaf(e2, r1, q1);
}
__device__ __forceinline__ float rcpC(float a) {   // rcp + 1 Newton step
    float r = __builtin_amdgcn_rcpf(a);
    float e = __builtin_fmaf(-a, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__global__ void check(uint32_t exp_lo, uint32_t exp_hi, unsigned long long* bad) {
    // all mantissas for exponents [exp_lo, exp_hi], both signs
    unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long total = (unsigned long long)(exp_hi - exp_lo + 1) << 24;
    for (; i < total; i += (unsigned long long)gridDim.x * blockDim.x) {
        uint32_t sign = (uint32_t)(i & 1), man = (uint32_t)((i >> 1) & 0x7fffff), ex = exp_lo + (uint32_t)(i >> 24);
        uint32_t bits = (sign << 31) | (ex << 23) | man;
        float a = __uint_as_float(bits);
        float ref = 1.0f / a;
        if (__float_as_uint(rcpA(a)) != __float_as_uint(ref)) atomicAdd(&bad[0], 1ull);
        if (__float_as_uint(rcpB(a)) != __float_as_uint(ref)) atomicAdd(&bad[1], 1ull);
        if (__float_as_uint(rcpC(a)) != __float_as_uint(ref)) atomicAdd(&bad[2], 1ull);
    }
}
int main() {
    unsigned long long* d; hipMalloc(&d, 24); 
    for (int range = 0; range < 3; ++range) {
        uint32_t lo = range == 0 ? 127 - 100 : (range == 1 ? 1 : 127 - 20), hi = range == 0 ? 127 + 100 : (range == 1 ? 254 : 127 + 84);
        hipMemset(d, 0, 24);
        hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, lo, hi, d);
        unsigned long long h[3]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("exponents [%u,%u] (2^%d..2^%d): mismatches  2-step %llu  3-step %llu  1-step %llu of %llu\n", lo, hi, (int)lo - 127, (int)hi - 127, h[0], h[1], h[2], (unsigned long long)(hi - lo + 1) << 24);
    }
    return 0;
}
