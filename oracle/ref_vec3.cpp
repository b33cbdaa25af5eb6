// oracle/ref_vec3.cpp — TEST INFRASTRUCTURE, not part of the product.
//
// Builds the reference's OWN vec3 arithmetic: core/vec3.h is entirely `__host__ __device__` and includes only
// <math.h>, <stdlib.h>, <iostream>, so it compiles for the HOST with the image's own HIP front end
//     clang++ -x hip --cuda-host-only -no-hip-rt -O2 -ffp-contract=off
// (the qualifiers come from the toolchain's HIP headers, nothing is stood in for; the file is included from where it
// lies under /root/reference — path passed as -DREF_VEC3_HEADER=... — nothing is copied; output only into oracle/_ref/).
// tests/test_vec3_ref.py holds the oracle's restatement (crt_oracle.c:26-84), the product's host helpers
// (csrc/crt_internal.hpp: hsub/hmul/hcross/hunit/hrotate) and — through the committed fixture
// tests/golden/vec3_ref_vectors.npz (tests/make_vec3_fixtures.py) — the device arithmetic of the kernels to it, bit
// for bit.  This pins SURVEY §8 row a6 (vec3 helpers); the `__device__`-only headers (hitable.h, triangle.h,
// transform.h, aabb.h, render.h's kernels) cannot be called on the host without redefining their qualifier and stay
// unpinned.
#include REF_VEC3_HEADER

// op codes shared with o_vec3_op (oracle/crt_oracle.c) and crt_debug_vec3 (include/crt.h)
enum {
    V3_ADD = 0, V3_SUB, V3_MULV, V3_MULS, V3_DIVV, V3_DIVS, V3_DOT, V3_CROSS, V3_LENGTH, V3_SQLEN, V3_UNIT, V3_LERP,
    V3_CLIP, V3_DIV_ASSIGN, V3_ROTATE, V3_MAKE_UNIT, V3_NEG, V3_SUM, V3_N_OPS
};

// out[i] = op(a[i], b[i], s[i]); a, b, out: n x 3 floats, s: n floats.  Scalar results go to out[3i], out[3i+1..2] = 0.
extern "C" int ref_vec3_op(int op, long long n, const float* a, const float* b, const float* s, float* out) {
    if (op < 0 || op >= V3_N_OPS) return -1;
    for (long long i = 0; i < n; ++i) {
        vec3 A(a[3 * i], a[3 * i + 1], a[3 * i + 2]), B(b[3 * i], b[3 * i + 1], b[3 * i + 2]), R(0.0f, 0.0f, 0.0f);
        const float t = s[i];
        switch (op) {
            case V3_ADD: R = A + B; break;                                 // vec3.h:73-75
            case V3_SUB: R = A - B; break;                                 // :77-79
            case V3_MULV: R = A * B; break;                                // :89-91
            case V3_MULS: R = t * A; break;                                // :93-95
            case V3_DIVV: R = A / B; break;                                // :101-103
            case V3_DIVS: R = A / t; break;                                // :105-107
            case V3_DOT: R = vec3(dot(A, B), 0.0f, 0.0f); break;           // :109-111
            case V3_CROSS: R = cross(A, B); break;                         // :113-117
            case V3_LENGTH: R = vec3(A.length(), 0.0f, 0.0f); break;       // :38-40
            case V3_SQLEN: R = vec3(A.squared_length(), 0.0f, 0.0f); break;   // :42-44
            case V3_UNIT: R = unit_vector(A); break;                       // :191-193
            case V3_LERP: R = lerp(t, A, B); break;                        // :196-198
            case V3_CLIP: R = clip(A); break;                              // :119-131 (defaults min 0, max 1)
            case V3_DIV_ASSIGN: R = A; R /= t; break;                      // :182-189
            case V3_ROTATE: R = rotate(A, B); break;                       // :200-217
            case V3_MAKE_UNIT: R = A; R.make_unit_vector(); break;         // :63-66
            case V3_NEG: R = -A; break;                                    // :24
            case V3_SUM: R = vec3(A.sum(), 0.0f, 0.0f); break;             // :68-71
        }
        out[3 * i] = R.x(); out[3 * i + 1] = R.y(); out[3 * i + 2] = R.z();
    }
    return 0;
}
