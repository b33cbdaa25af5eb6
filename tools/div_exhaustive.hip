// div_exhaustive.hip — proof by exhaustion for the shading kernel's fp32 quotient
//     y = RN(1/b) (v_rcp_f32 + one Newton step, itself checked exhaustively by the GPU tests),
//     q0 = RN(x*y);  r = RN(x - q0*b) (fma, exact);  q = RN(q0 + r*y)
// against the IEEE quotient x / b for EVERY pair of significands: x = 1.mx, b = 1.mb
// (2^23 x 2^23 = 7.0e13 pairs; quotients cover both result binades (0.5, 2)).  Scaling x or b
// by a power of two changes nothing as long as no intermediate under/overflows, which the
// kernel's range guard ensures.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt tools/div_exhaustive.hip -o tools/div_exhaustive.bin
// run:   tools/div_exhaustive.bin [first_chunk] [n_chunks]     (64 chunks of 2^17 divisors; ~1 s each)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void __launch_bounds__(256) k(unsigned mb0, unsigned long long* bad, unsigned* first_bad) {
    const unsigned mb = mb0 + blockIdx.x;                  // one divisor per block
    const float b = __uint_as_float(0x3f800000u | mb);
    float y = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, y, 1.0f);
    y = __builtin_fmaf(e, y, y);
    unsigned long long mine = (__float_as_uint(y) != __float_as_uint(1.0f / b)) ? 1ull : 0ull;
    for (unsigned mx = threadIdx.x; mx < (1u << 23); mx += 256) {
        const float x = __uint_as_float(0x3f800000u | mx);
        const float q0 = x * y;
        const float r = __builtin_fmaf(-q0, b, x);
        const float q = __builtin_fmaf(r, y, q0);
        if (__float_as_uint(q) != __float_as_uint(x / b)) {
            ++mine;
            first_bad[0] = mb; first_bad[1] = mx;
        }
    }
    if (mine) atomicAdd(bad, mine);
}

int main(int argc, char** argv) {
    const int first = argc > 1 ? atoi(argv[1]) : 0, n = argc > 2 ? atoi(argv[2]) : 64;
    unsigned long long* d_bad; unsigned* d_first;
    hipMalloc(&d_bad, 8); hipMalloc(&d_first, 8);
    hipMemset(d_bad, 0, 8); hipMemset(d_first, 0, 8);
    unsigned long long total = 0;
    for (int c = first; c < first + n && c < 64; ++c) {
        hipLaunchKernelGGL(k, dim3(1u << 17), dim3(256), 0, 0, (unsigned)c << 17, d_bad, d_first);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
        unsigned long long bad = 0;
        hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost);
        total += 1ull << 40;
        printf("chunk %2d: divisors 1.%06x..: pairs so far %.3e, mismatches so far %llu\n", c, (unsigned)c << 17, (double)total, bad);
        fflush(stdout);
    }
    unsigned long long bad = 0; unsigned fb[2];
    hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost);
    hipMemcpy(fb, d_first, 8, hipMemcpyDeviceToHost);
    printf("TOTAL pairs %.6e mismatches %llu (last offender mb=0x%06x mx=0x%06x)\n", (double)total, bad, fb[0], fb[1]);
    return bad ? 1 : 0;
}
