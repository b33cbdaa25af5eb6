// crt_kernels.hip — gfx950 (MI355X, CDNA4) kernels of the list-render path.
//
// What the reference does per pixel (core/render.h:105-129 and callees, paths relative
// to CudaTest/src of the reference) is reproduced bit for bit; how it is done is not:
//
//   * The reference chases Hitable** -> Triangle (80 B) -> Transform (36 B) per ray and
//     triangle and redoes two normalisations and 18 sin/cos there (hitable.h:29-38,
//     transform.h:11-54).  All triangles carry the identity Transform(0,0,1)
//     (createScene.h:184), so the direction every Triangle::collision_detection sees is
//     d2 = unit(unit(d0)) — per ray, not per triangle — and rotate(.,0) is the identity.
//   * The camera is a pinhole on this path (aperture 0 at every call site,
//     createScene.h:157-160), so s = o - v0, q = cross(s,e1) and dot(e2,q) are per
//     triangle, not per ray: a prepass computes them once per frame with the reference's
//     operation order and the render kernel reads one 64-B record per triangle.
//   * One lane = one pixel (samples of a pixel are chained through its RNG stream, so the
//     sample loop stays serial per lane); one wavefront = one 8x8 pixel tile.  The
//     triangle index is wave-uniform, so the record is fetched with scalar loads
//     (s_load_dwordx16 through the scalar cache) and feeds the VALU as SGPR operands:
//     no VGPRs, no LDS bandwidth and no VALU slots are spent on broadcasting it.
//   * After each of the reference's rejection tests the wavefront votes
//     (__builtin_amdgcn_ballot_w64) and leaves the triangle as soon as no lane is alive.
//
// No fused multiply-add may be formed anywhere in this file (-ffp-contract=off).
#include "crt_device.hpp"

#include <float.h>

#include <stdlib.h>

#include <algorithm>
#include <type_traits>

namespace crt {

// ------------------------------------------------------------------ XORWOW (cuRAND)
// curand(curandStateXORWOW_t*) of CUDA 12.2's curand_kernel.h, restated.
struct Xorwow {
    uint32_t d, v0, v1, v2, v3, v4;
};

__device__ __forceinline__ uint32_t xw_next(Xorwow& s) {
    uint32_t t = (s.v0 ^ (s.v0 >> 2));
    s.v0 = s.v1;
    s.v1 = s.v2;
    s.v2 = s.v3;
    s.v3 = s.v4;
    s.v4 = (s.v4 ^ (s.v4 << 4)) ^ (t ^ (t << 1));
    s.d += 362437u;
    return s.v4 + s.d;
}

// _curand_uniform: x * 2^-32 + 2^-33, two roundings, result in (0, 1].
__device__ __forceinline__ float xw_uniform(Xorwow& s) {
    uint32_t x = xw_next(s);
    const float k = 2.3283064e-10f;
    // (float)x * 2^-32 is exact (a power-of-two scaling of 0 or a float >= 1: no rounding, no underflow), so the fused form rounds
    // once where the reference's product-then-sum rounds once: the same bits for every x, one instruction less per draw
    return __builtin_fmaf((float)x, k, k / 2.0f);
}

// 2.0f * curand_uniform(state) - 1.0f (material.h:14, camera.h:9): doubling is exact and commutes with the rounding of
// the sum (no underflow: the terms are >= 2^-33), so 2u = x * 2^-31 + 2^-32 with the same single rounding — one
// instruction less per draw in the rejection loops.
__device__ __forceinline__ float xw_uniform_2m1(Xorwow& s) {
    uint32_t x = xw_next(s);
    const float k = 4.6566129e-10f;   // 2^-31
    return __builtin_fmaf((float)x, k, k / 2.0f) - 1.0f;   // (the product is exact: see xw_uniform)
}

// random_init (core/deviceManage.h:120-128): curand_init(seed, pixel_index, 0).
// The 2^67-step subsequence skip is a product of precomputed GF(2) matrices
// J_k = T^(2^(67+k)), one per set bit of pixel_index.  Matrix rows are wave-uniform
// scalar loads; each lane masks them with its own state bits.
// One matrix-vector product: r = v * M (row-vector convention), M wave-uniform.
__device__ __forceinline__ void xw_matvec(const uint32_t* __restrict__ M, const uint32_t v[kXwWords], uint32_t r[kXwWords]) {
    uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0;
#pragma unroll
    for (int w = 0; w < kXwWords; ++w) {
        uint32_t word = v[w];
        for (int b = 0; b < 32; ++b) {
            uint32_t mask = 0u - ((word >> b) & 1u);
            const uint32_t* row = M + (size_t)(w * 32 + b) * kXwWords;   // uniform address
            r0 ^= row[0] & mask;
            r1 ^= row[1] & mask;
            r2 ^= row[2] & mask;
            r3 ^= row[3] & mask;
            r4 ^= row[4] & mask;
        }
    }
    r[0] = r0; r[1] = r1; r[2] = r2; r[3] = r3; r[4] = r4;
}

// Element i of the state array belongs to pixel x = i % nx of local row i / nx.  A thread derives the
// state of its FIRST element from the seed (one product per set bit of the pixel index: ~10 for a
// 1080p frame), then walks kRngRun - 1 further elements 64 apart with ONE product each: elements 64
// apart inside a row are 64 pixels apart, i.e. one application of J_6 = T^(2^73) (the table's entry
// for bit 6).  A wavefront thus covers kRngRun * 64 consecutive elements with coalesced stores, at
// roughly (10 + kRngRun - 1) / kRngRun products per pixel instead of 10.  A lane whose step leaves the
// row (or crosses into another strip) starts over from the seed.
constexpr int kRngRun = 16;
__global__ void __launch_bounds__(256)
rng_init_kernel(uint32_t* __restrict__ state, long long first_pixel, int nx, int strip_rows, int strip_step,
                long long count, uint32_t seed_lo, uint32_t seed_hi, const uint32_t* __restrict__ jump,
                int nbits) {
    const long long wave_id = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const long long i0 = wave_id * (64 * kRngRun) + lane;

    // _curand_init_scratch seed scrambling
    const uint32_t s0 = seed_lo ^ 0xaad26b49u;
    const uint32_t s1 = seed_hi ^ 0xf7dcefddu;
    const uint32_t t0 = 1099087573u * s0;
    const uint32_t t1 = 2591861531u * s1;
    const uint32_t d = 6615241u + t1 + t0;
    uint32_t v[kXwWords] = {0u, 0u, 0u, 0u, 0u};
    unsigned long long prev_pi = 0ull;
    bool have_prev = false;
    for (int j = 0; j < kRngRun; ++j) {
        const long long i = i0 + 64ll * j;
        const bool live = i < count;
        if (__builtin_amdgcn_ballot_w64(live) == 0ull) break;
        const long long il = live ? i : 0;
        const int ly = (int)(il / nx), x = (int)(il % nx);
        // subsequence = GLOBAL pixel index y*nx + x (deviceManage.h:126-127)
        const unsigned long long pi = (unsigned long long)first_pixel +
                                      (unsigned long long)global_row(0, strip_rows, strip_step, ly) * (unsigned long long)nx +
                                      (unsigned long long)x;
        const bool step = have_prev && live && pi == prev_pi + 64ull;   // the same stream family, 64 subsequences on
        uint32_t stepped[kXwWords] = {0u, 0u, 0u, 0u, 0u};
        if (__builtin_amdgcn_ballot_w64(step) != 0ull) xw_matvec(jump + (size_t)6 * kXwBits * kXwWords, v, stepped);
        if (__builtin_amdgcn_ballot_w64(live && !step) != 0ull) {       // some lane starts from the seed (first element, row change)
            uint32_t f[kXwWords] = {123456789u + t0, 362436069u ^ t0, 521288629u + t1, 88675123u ^ t1, 5783321u + t0};
            for (int k = 0; k < nbits; ++k) {
                const bool take = live && !step && ((pi >> k) & 1ull);
                if (__builtin_amdgcn_ballot_w64(take) == 0ull) continue;   // wave-uniform skip
                uint32_t r[kXwWords];
                xw_matvec(jump + (size_t)k * kXwBits * kXwWords, f, r);
                if (take) { f[0] = r[0]; f[1] = r[1]; f[2] = r[2]; f[3] = r[3]; f[4] = r[4]; }
            }
            if (!step) { stepped[0] = f[0]; stepped[1] = f[1]; stepped[2] = f[2]; stepped[3] = f[3]; stepped[4] = f[4]; }
        }
        v[0] = stepped[0]; v[1] = stepped[1]; v[2] = stepped[2]; v[3] = stepped[3]; v[4] = stepped[4];
        prev_pi = pi;
        have_prev = live;
        if (live) {   // SoA: coalesced
            state[0 * count + i] = d;
            state[1 * count + i] = v[0];
            state[2 * count + i] = v[1];
            state[3 * count + i] = v[2];
            state[4 * count + i] = v[3];
            state[5 * count + i] = v[4];
        }
    }
}

hipError_t launch_rng_init(uint32_t* d_state, long long first_pixel, long long count, int nx, int strip_rows,
                           int strip_step, uint64_t seed, const uint32_t* d_jump, hipStream_t stream) {
    if (count <= 0 || nx <= 0) return hipSuccess;
    const int last_ly = (int)((count - 1) / nx);
    unsigned long long last = (unsigned long long)first_pixel +
                              (unsigned long long)(global_row(0, strip_rows, strip_step, last_ly) + 1) * (unsigned long long)nx;
    int nbits = 0;
    while (nbits < kXwJumps && (last >> nbits) != 0ull) ++nbits;
    const long long waves = (count + 64 * kRngRun - 1) / (64 * kRngRun);
    dim3 block(256), grid((unsigned)((waves + 3) / 4));
    hipLaunchKernelGGL(rng_init_kernel, grid, block, 0, stream, d_state, first_pixel, nx, strip_rows, strip_step,
                       count, (uint32_t)seed, (uint32_t)(seed >> 32), d_jump, nbits);
    return hipGetLastError();
}

// ------------------------------------------------------------------ triangle prepass
// Per triangle, once per camera origin: the ray-independent part of
// Triangle::collision_detection (shapes/triangle.h:69-70,81,87,92) in its own order.
__device__ __forceinline__ void derive_record(const float* __restrict__ b, float ox, float oy, float oz, float4 o[4]) {
    float v0x = b[0], v0y = b[1], v0z = b[2];
    float v1x = b[3], v1y = b[4], v1z = b[5];
    float v2x = b[6], v2y = b[7], v2z = b[8];
    float e1x = v1x - v0x, e1y = v1y - v0y, e1z = v1z - v0z;       // edge1 = vertex1 - vertex0
    float e2x = v2x - v0x, e2y = v2y - v0y, e2z = v2z - v0z;       // edge2 = vertex2 - vertex0
    float sx = ox - v0x, sy = oy - v0y, sz = oz - v0z;             // s = r.origin() - vertex0
    float qx = (sy * e1z - sz * e1y);                              // q = cross(s, edge1), vec3.h:113-117
    float qy = (-(sx * e1z - sz * e1x));
    float qz = (sx * e1y - sy * e1x);
    float tq = e2x * qx + e2y * qy + e2z * qz;                     // dot(edge2, q)
    o[0] = make_float4(b[9], b[10], b[11], e2x);
    o[1] = make_float4(e2y, e2z, e1x, e1y);
    o[2] = make_float4(e1z, sx, sy, sz);
    o[3] = make_float4(qx, qy, qz, tq);
}

__global__ void __launch_bounds__(256)
tri_prepass_kernel(const float* __restrict__ base, int n, float ox, float oy, float oz,
                   float* __restrict__ rec) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 r4[4];
    derive_record(base + (size_t)i * kBaseFloats, ox, oy, oz, r4);
    float4* o = reinterpret_cast<float4*>(rec + (size_t)i * kRecFloats);
    o[0] = r4[0]; o[1] = r4[1]; o[2] = r4[2]; o[3] = r4[3];
}

// update_pose (createScene.h:99-110; a single device thread in the reference): lane = triangle,
// the three vertices are re-read from the new point positions through the triangle's (already
// 0-based, validated) indices in the pack's reversed winding; the stored normal stays.
__global__ void __launch_bounds__(256)
update_pose_kernel(const float* __restrict__ points, const int* __restrict__ idx, int n, float* __restrict__ base) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float* b = base + (size_t)i * kBaseFloats;
#pragma unroll
    for (int k = 0; k < 3; ++k) {   // v = {points[idx[2]], points[idx[1]], points[idx[0]]}
        const float* q = points + 3 * (size_t)idx[3 * (size_t)i + (2 - k)];
        b[3 * k + 0] = q[0];
        b[3 * k + 1] = q[1];
        b[3 * k + 2] = q[2];
    }
}

hipError_t launch_update_pose(const float* d_points, const int* d_idx, int n_tris, float* d_base, hipStream_t stream) {
    if (n_tris <= 0) return hipSuccess;
    hipLaunchKernelGGL(update_pose_kernel, dim3((unsigned)((n_tris + 255) / 256)), dim3(256), 0, stream, d_points, d_idx,
                       n_tris, d_base);
    return hipGetLastError();
}

hipError_t launch_tri_prepass(const float* d_base, int n_tris, const float origin[3],
                              float* d_rec, hipStream_t stream) {
    if (n_tris <= 0) return hipSuccess;
    dim3 block(256), grid((unsigned)((n_tris + 255) / 256));
    hipLaunchKernelGGL(tri_prepass_kernel, grid, block, 0, stream, d_base, n_tris, origin[0],
                       origin[1], origin[2], d_rec);
    return hipGetLastError();
}

// ------------------------------------------------------------------ render
struct V3 {
    float x, y, z;
};

// ---- correctly rounded fp32 quotients without the IEEE division sequence (used only in the
// tiled shading kernel).  With y = RN(1/b):
//     q0 = RN(x*y);   r = RN(x - q0*b)  (fma: exact);   q = RN(q0 + r*y)  ==  RN(x / b)
// for every pair of floats whose quotient, remainder and reciprocal stay in the normal range.
// Proof by exhaustion: tools/div_exhaustive.hip compares q with the IEEE quotient for all
// 2^23 x 2^23 pairs of significands (7.04e13 pairs, 0 mismatches: profiles/r1_div_exhaustive.txt);
// a power-of-two scaling of x or b changes no rounding while nothing under/overflows, which
// the callers' range guards ensure.  y itself is reciprocal<true> (v_rcp_f32 + one Newton
// step), equal to the IEEE 1.0f/b for every float in 2^-100..2^100 (GPU test, exhaustive).
template <bool FAST>
__device__ __forceinline__ float reciprocal(float a) {
    if (FAST) {
        float r = __builtin_amdgcn_rcpf(a);
        float e = __builtin_fmaf(-a, r, 1.0f);
        return __builtin_fmaf(e, r, r);
    }
    return 1.0f / a;
}

__device__ __forceinline__ float div_by_rcp(float x, float b, float y) {
    const float q0 = x * y;
    const float r = __builtin_fmaf(-q0, b, x);
    return __builtin_fmaf(r, y, q0);
}

// unit_vector through sqrt_normal and div_by_rcp; identical bits to unit_vector below.
__device__ __forceinline__ V3 unit_vector_fast(V3 v);

// unit_vector(v) = v / v.length()  (core/vec3.h:191-193,38-40,105-107): three IEEE divides.
__device__ __forceinline__ V3 unit_vector(V3 v) {
    float len = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    return V3{v.x / len, v.y / len, v.z / len};
}

// Camera::get_ray's direction with offset = 0 (core/camera.h:43):
// ((lower_left_corner + s*horizontal) + t*vertical) - origin, component-wise.
__device__ __forceinline__ V3 camera_direction(const CameraPOD& c, float u, float v) {
    V3 d;
    d.x = ((c.llc[0] + u * c.horizontal[0]) + v * c.vertical[0]) - c.origin[0];
    d.y = ((c.llc[1] + u * c.horizontal[1]) + v * c.vertical[1]) - c.origin[1];
    d.z = ((c.llc[2] + u * c.horizontal[2]) + v * c.vertical[2]) - c.origin[2];
    return d;
}

// Correctly rounded sqrt of a NORMAL float (2^-100 <= x <= 2^100): v_sqrt_f32 is within one ulp,
// the exact residuals of its two neighbours decide — the compiler's own sqrtf expansion without
// the denormal pre-scaling and the class fix-up.  Equal to sqrtf for every float of that range
// (crt_debug_check_division mode 3, exhaustive).
__device__ __forceinline__ float sqrt_normal(float x) {
    float s = __builtin_amdgcn_sqrtf(x);
    const float lo = __int_as_float(__float_as_int(s) - 1), hi = __int_as_float(__float_as_int(s) + 1);
    const float r_lo = __builtin_fmaf(-lo, s, x), r_hi = __builtin_fmaf(-hi, s, x);
    s = (r_lo <= 0.0f) ? lo : s;
    s = (r_hi > 0.0f) ? hi : s;
    return s;
}

// Guard of the fast path: every component >= 2^-40 in magnitude (so an exact zero component
// takes the IEEE path too) and len <= 2^40.  Then the sum of squares is a normal float >= 2^-80,
// |q| >= 2^-80, the exact remainder is a multiple of 2^-126 and 1/len is in 2^-40..2^40.
// Anything else — including inf/NaN, for which both tests are false — recomputes with the
// IEEE sqrtf and divisions.
__device__ __forceinline__ V3 unit_vector_fast(V3 v) {
    const float s2 = v.x * v.x + v.y * v.y + v.z * v.z;
    float len = sqrt_normal(s2);
    const float amin = fminf(fminf(fabsf(v.x), fabsf(v.y)), fabsf(v.z));
    const bool ordinary = amin >= 9.094947017729282e-13f && len <= 1099511627776.0f;   // 2^-40, 2^40
    if (__builtin_amdgcn_ballot_w64(!ordinary) != 0ull) {
        len = sqrtf(s2);
        return V3{v.x / len, v.y / len, v.z / len};
    }
    const float y = reciprocal<true>(len);
    return V3{div_by_rcp(v.x, len, y), div_by_rcp(v.y, len, y), div_by_rcp(v.z, len, y)};
}

// The same WITHOUT the range guard, for callers that check the range themselves afterwards (the BVH mode's hot direction
// chain: one test on the last element covers the whole chain, see hot_chain).  On inputs outside the guarded range the
// result is merely not the IEEE quotient — no trap, no loop — and the caller must not use it.
__device__ __forceinline__ V3 unit_vector_fast_ng(V3 v) {
    const float s2 = v.x * v.x + v.y * v.y + v.z * v.z;
    const float len = sqrt_normal(s2);
    const float y = reciprocal<true>(len);
    return V3{div_by_rcp(v.x, len, y), div_by_rcp(v.y, len, y), div_by_rcp(v.z, len, y)};
}

// the y component of unit_vector(v) alone (sky-only tiles)
__device__ __forceinline__ float unit_vector_fast_y(V3 v) {
    const float s2 = v.x * v.x + v.y * v.y + v.z * v.z;
    float len = sqrt_normal(s2);
    // s2 >= v.y^2 >= 2^-80 up to rounding: normal; x, z only add to it
    const bool ordinary = fabsf(v.y) >= 9.094947017729282e-13f && len <= 1099511627776.0f;
    if (__builtin_amdgcn_ballot_w64(!ordinary) != 0ull) return v.y / sqrtf(s2);
    return div_by_rcp(v.y, len, reciprocal<true>(len));
}

// backgroundSky (core/render.h:41-46) given v = unit_vector(d0) already.
__device__ __forceinline__ V3 sky_from_unit(V3 v) {
    float t = 0.5f * (v.y + 1.0f);
    // lerp(t, vec3(1), vec3(0.5,0.7,1.0)) = from + t*(to - from)   (vec3.h:196-198)
    return V3{1.0f + t * (0.5f - 1.0f), 1.0f + t * (0.7f - 1.0f), 1.0f + t * (1.0f - 1.0f)};
}

// Triangle::collision_detection (shapes/triangle.h:57-100) on one derived record for this
// lane's twice-normalised direction d, followed by the list's accept rule
// `hit && tmp_rec.t < closest_so_far` (hitable/hitable_list.h:84).  `r` and `i` are
// wave-uniform: the record arrives through scalar loads and feeds the VALU as SGPR operands;
// after each of the reference's rejection tests the wavefront votes and leaves as soon as no
// lane is alive.
#ifndef CRT_WALK_BATCH
#define CRT_WALK_BATCH 2
#endif
constexpr int kWalkBatch = CRT_WALK_BATCH;

struct Rec16 {
    float f[kRecFloats];
};

// Wave-uniform address -> one s_load_dwordx16 into SGPRs.
__device__ __forceinline__ Rec16 load_rec(const float* __restrict__ rec, int idx_raw) {
    const float* __restrict__ r = rec + (size_t)(idx_raw & 0x7fffffff) * kRecFloats;
    Rec16 o;
#pragma unroll
    for (int k = 0; k < kRecFloats; ++k) o.f[k] = r[k];
    return o;
}

// f = 1.0 / a (shapes/triangle.h:80; the double division rounded to float equals the fp32
// IEEE quotient).  v_rcp_f32 followed by ONE Newton step is bit-identical to the IEEE
// quotient 1.0f/a for every float with 2^-100 <= |a| <= 2^100 on gfx950 — verified
// exhaustively (tools/rcp_exhaustive.hip, tests/test_gpu_parity.py::test_fast_reciprocal_
// is_exact).  On this path |a| >= EPSILON = 2^-20 for every lane whose result is used and
// |a| <= |e1||e2| <= 2^84 because the host only selects FAST for scenes whose coordinates are
// bounded by 2^38; otherwise the compiler's IEEE sequence is used.

// Votes are kept as 64-bit lane masks in SGPRs (v_cmp -> s_and_b64 -> s_cbranch_scc), the
// per-lane predicate is only rebuilt for the final conditional update.
template <bool FAST>
__device__ __forceinline__ void exact_test(const Rec16& rc, int i_raw, V3 d, float& closest, int& hit_idx) {
    const int i = i_raw & 0x7fffffff;                 // bit 31 (wave-uniform): the binning proved dn >= 0 for the whole tile
    const float nx = rc.f[0], ny = rc.f[1], nz = rc.f[2];
    const float e2x = rc.f[3], e2y = rc.f[4], e2z = rc.f[5];
    const float e1x = rc.f[6], e1y = rc.f[7], e1z = rc.f[8];
    const float sx = rc.f[9], sy = rc.f[10], sz = rc.f[11];
    const float qx = rc.f[12], qy = rc.f[13], qz = rc.f[14];
    const float tq = rc.f[15];

    // if (dot(r.direction(), normal) < 0) return false;            triangle.h:61
    unsigned long long alive = __builtin_amdgcn_ballot_w64(true);
    if (i_raw >= 0) {
        float dn = d.x * nx + d.y * ny + d.z * nz;
        alive = __builtin_amdgcn_ballot_w64(!(dn < 0.0f));
        if (alive == 0ull) return;
    }

    // h = cross(r.direction(), edge2); a = dot(edge1, h);            triangle.h:71-72
    float hx = (d.y * e2z - d.z * e2y);
    float hy = (-(d.x * e2z - d.z * e2x));
    float hz = (d.x * e2y - d.y * e2x);
    float a = e1x * hx + e1y * hy + e1z * hz;
    // if (a > -EPSILON && a < EPSILON) return false;                triangle.h:77 (backCulling is false)
    alive &= __builtin_amdgcn_ballot_w64(!(fabsf(a) < kTriEpsilon));
    if (alive == 0ull) return;

    float f = reciprocal<FAST>(a);                                    // triangle.h:80
    float u = f * (sx * hx + sy * hy + sz * hz);                      // triangle.h:82
    alive &= __builtin_amdgcn_ballot_w64(!(u < 0.0f)) & __builtin_amdgcn_ballot_w64(!(u > 1.0f));   // triangle.h:84
    if (alive == 0ull) return;

    float v = f * (d.x * qx + d.y * qy + d.z * qz);                   // triangle.h:88
    alive &= __builtin_amdgcn_ballot_w64(!(v < 0.0f)) & __builtin_amdgcn_ballot_w64(!(u + v > 1.0f));   // triangle.h:89
    if (alive == 0ull) return;

    float t = f * tq;                                                 // triangle.h:92
    // hit(...) && tmp_rec.t < closest_so_far                         hitable_list.h:84
    // The reference walks in index order with a strict <, i.e. it keeps the smallest t and, among
    // equal t, the lowest index.  Written order-independently (tile lists are unordered):
    if (__builtin_amdgcn_inverse_ballot_w64(alive) && (t < closest || (t == closest && i < hit_idx))) {
        closest = t;
        hit_idx = i;
    }
}

// exact_test over list positions [k0, k1); index_at(k) is the (wave-uniform) triangle index
// of position k.  (A manual one-ahead prefetch of the next record was measured 1.75x SLOWER:
// SMEM returns out of order, so hipcc must drain lgkmcnt to 0 before the first use of any
// scalar-loaded register and then shuffles the record through s_mov copies.)
template <bool FAST, class IndexAt>
__device__ __forceinline__ void walk(const float* __restrict__ rec, int k0, int k1, IndexAt index_at,
                                     V3 d, float& closest, int& hit_idx) {
    // kWalkBatch records are requested together and waited for once: the scalar-load latency
    // is paid per batch, not per triangle (it bounds the time of a heavy tile).
    int k = k0;
    for (; k + kWalkBatch <= k1; k += kWalkBatch) {
        int idx[kWalkBatch];
        Rec16 r[kWalkBatch];
#pragma unroll
        for (int j = 0; j < kWalkBatch; ++j) {
            idx[j] = index_at(k + j);
            r[j] = load_rec(rec, idx[j]);
        }
        // Both records are consumed here, so hipcc issues the scalar loads back to back and
        // waits once; left alone it sinks every piece of a record below the vote that first
        // needs it (up to three dependent scalar round trips per triangle).
        if (kWalkBatch == 2)
            asm volatile("" ::"s"(r[0].f[0]), "s"(r[0].f[2]), "s"(r[0].f[3]), "s"(r[0].f[10]), "s"(r[0].f[11]), "s"(r[0].f[14]),
                         "s"(r[0].f[15]), "s"(r[1].f[0]), "s"(r[1].f[2]), "s"(r[1].f[3]), "s"(r[1].f[10]), "s"(r[1].f[11]),
                         "s"(r[1].f[14]), "s"(r[1].f[15]));
#pragma unroll
        for (int j = 0; j < kWalkBatch; ++j) exact_test<FAST>(r[j], idx[j], d, closest, hit_idx);
    }
    for (; k < k1; ++k) {
        const int idx = index_at(k);
        const Rec16 r1 = load_rec(rec, idx);
        asm volatile("" ::"s"(r1.f[0]), "s"(r1.f[2]), "s"(r1.f[3]), "s"(r1.f[10]), "s"(r1.f[11]), "s"(r1.f[14]), "s"(r1.f[15]));
        exact_test<FAST>(r1, idx, d, closest, hit_idx);
    }
}

// Same over a survivor list in global memory: 64 indices at a time are fetched with one
// coalesced vector load (lane j holds position base+j) and handed out with v_readlane.
template <bool FAST>
__device__ __forceinline__ void walk_list(const float* __restrict__ rec, const uint32_t* __restrict__ lst,
                                          int k0, int k1, int lane, V3 d, float& closest, int& hit_idx) {
    for (int base = k0; base < k1; base += 64) {
        const int n = min(64, k1 - base);
        const int mine = (lane < n) ? (int)lst[base + lane] : 0;
        walk<FAST>(rec, 0, n, [&](int k) { return __builtin_amdgcn_readlane(mine, k); }, d, closest, hit_idx);
    }
}
// the same for the entries first + j * step, j in [j0, j1)
template <bool FAST>
__device__ __forceinline__ void walk_list_strided(const float* __restrict__ rec, const uint32_t* __restrict__ lst, int first, int step,
                                                  int j0, int j1, int lane, V3 d, float& closest, int& hit_idx) {
    for (int base = j0; base < j1; base += 64) {
        const int n = min(64, j1 - base);
        const int mine = (lane < n) ? (int)lst[first + (base + lane) * step] : 0;
        walk<FAST>(rec, 0, n, [&](int k) { return __builtin_amdgcn_readlane(mine, k); }, d, closest, hit_idx);
    }
}

// BVH mode: the entries of a slice beyond its staged ones.  The record of slot s is tested like above, but what orders ties
// and comes back in hit_idx is the TREE's order — key = (0x3fffff - partner[s]) << 9 | tag (smaller wins = later leaf, first
// in its leaf; tag marks the entry as unstaged: the master looks the slot up through the partner table) — with the pool
// word's cull-sure bit on top.  Slots and keys are wave-uniform (readlane of one coalesced load / gather per 64 entries).
template <bool FAST>
__device__ __forceinline__ void walk_list_strided_keyed(const float* __restrict__ rec, const uint32_t* __restrict__ lst,
                                                        const int* __restrict__ partner, unsigned tag, int first, int step, int j0, int j1,
                                                        int lane, V3 d, float& closest, int& hit_idx) {
    for (int base = j0; base < j1; base += 64) {
        const int n = min(64, j1 - base);
        const int mine = (lane < n) ? (int)lst[first + (base + lane) * step] : 0;
        const int key = (lane < n) ? (int)(((unsigned)mine & 0x80000000u) | ((unsigned)(0x3fffff - partner[mine & 0x7fffffff]) << 9) | tag) : 0;
        for (int k = 0; k < n; ++k) {
            const int slot = __builtin_amdgcn_readlane(mine, k), kk = __builtin_amdgcn_readlane(key, k);
            const Rec16 r1 = load_rec(rec, slot);
            asm volatile("" ::"s"(r1.f[0]), "s"(r1.f[2]), "s"(r1.f[3]), "s"(r1.f[10]), "s"(r1.f[11]), "s"(r1.f[14]), "s"(r1.f[15]));
            exact_test<FAST>(r1, kk, d, closest, hit_idx);
        }
    }
}

// ---- the same test with the record in VGPRs.  On gfx950 a VALU instruction with an SGPR
// operand issues at half the rate of a VGPR-only one (profiles/r1_ubench_valu*.txt).  For
// light tiles (<= 64 survivors) the wave copies the records of its list into LDS once per tile
// and every sample re-reads them with broadcast ds_read_b128 (uniform address): VGPR operands,
// LDS latency instead of two dependent scalar-memory round trips.
struct RecV {
    float4 a, b, c, d;   // n.xyz e2.x | e2.yz e1.xy | e1.z s.xyz | q.xyz tq
};

template <bool FAST>
__device__ __forceinline__ void exact_test_vec(const RecV& rc, int i_raw, V3 d, float& closest, int& hit_idx) {
    const int i = i_raw & 0x7fffffff;
    const float nx = rc.a.x, ny = rc.a.y, nz = rc.a.z;
    const float e2x = rc.a.w, e2y = rc.b.x, e2z = rc.b.y;
    const float e1x = rc.b.z, e1y = rc.b.w, e1z = rc.c.x;
    const float sx = rc.c.y, sy = rc.c.z, sz = rc.c.w;
    const float qx = rc.d.x, qy = rc.d.y, qz = rc.d.z;
    const float tq = rc.d.w;

    unsigned long long alive = __builtin_amdgcn_ballot_w64(true);
    if (i_raw >= 0) {                                                 // not "cull-sure"
        float dn = d.x * nx + d.y * ny + d.z * nz;                     // triangle.h:61
        alive = __builtin_amdgcn_ballot_w64(!(dn < 0.0f));
        if (alive == 0ull) return;
    }
    float hx = (d.y * e2z - d.z * e2y);                               // triangle.h:71-72
    float hy = (-(d.x * e2z - d.z * e2x));
    float hz = (d.x * e2y - d.y * e2x);
    float a = e1x * hx + e1y * hy + e1z * hz;
    alive &= __builtin_amdgcn_ballot_w64(!(fabsf(a) < kTriEpsilon));  // triangle.h:77
    if (alive == 0ull) return;
    float f = reciprocal<FAST>(a);                                    // triangle.h:80
    float u = f * (sx * hx + sy * hy + sz * hz);                      // triangle.h:82
    alive &= __builtin_amdgcn_ballot_w64(!(u < 0.0f)) & __builtin_amdgcn_ballot_w64(!(u > 1.0f));   // triangle.h:84
    if (alive == 0ull) return;
    float v = f * (d.x * qx + d.y * qy + d.z * qz);                   // triangle.h:88
    alive &= __builtin_amdgcn_ballot_w64(!(v < 0.0f)) & __builtin_amdgcn_ballot_w64(!(u + v > 1.0f));   // triangle.h:89
    if (alive == 0ull) return;
    float t = f * tq;                                                 // triangle.h:92
    if (__builtin_amdgcn_inverse_ballot_w64(alive) && (t < closest || (t == closest && i < hit_idx))) {  // hitable_list.h:84, order-independent
        closest = t;
        hit_idx = i;
    }
}

// A wave-uniform float in a VGPR.  On gfx950 a VALU instruction with an SGPR source issues at half the
// rate of a VGPR-only one (profiles/r1_ubench_valu2.txt: 4.1 vs 2.3 cycles per wave64 instruction);
// the empty asm hides the value's origin, so hipcc keeps using the register copy inside the loop.
// (float)n of a wave-uniform int, converted INSIDE the opaque instruction: written as in_vgpr((float)n) the conversion is a loop
// invariant of its own that hipcc hoists to the kernel's entry and — at six waves per SIMD — spills (two of the list kernel's six
// scratch slots were (float)nx and (float)ny).
__device__ __forceinline__ float in_vgpr_from_int(int n) {
    float v;
    asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(v) : "s"(n));
    return v;
}
__device__ __forceinline__ float in_vgpr(float x) {
    float v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(x));
    return v;
}

// The body of the `render` kernel for one pixel (core/render.h:105-129).  `traverse(d2,
// closest, hit_idx)` must return the result of HitableList::collision_detection
// (hitable/hitable_list.h:76-91): minimum t over all triangles that pass
// Triangle::collision_detection (any sign of t, no range test), lowest index on ties, NaN
// never accepted.
// `traverse(d1, d2, closest, hit_idx)`: d1 = unit_vector(d0) (what the world-level ScaleRay produces),
// d2 = unit_vector(d1) (what the triangle-level one produces; computed only if WANT_D2 — the BVH
// world derives its own per-level directions from d1).
// SPEC: `traverse` takes a fifth argument, a functor that runs the hit branch's rejection draws on a COPY of the stream
// (the master wave of a shared tile calls it while the other waves trace); a lane that turns out to have hit adopts
// the copy, a lane that missed keeps its stream as it was — the draws a pixel consumes are the same either way.
// MAT: LambertShade with a Material* per triangle (crt_scene_set_materials; render.h:75-83 with material.h:57-161): what the hit
// triangle's material draws and returns as attenuation / emitted; otherwise the scene's one Lambertian.
template <bool FASTDIV = false, bool SKYONLY = false, bool WANT_D2 = true, bool SPEC = false, bool MAT = false, class Traverse>
__device__ __forceinline__ void render_pixel(const RenderParams& p, int x, int ly,
                                             const float* __restrict__ shade_n,
                                             uint32_t* __restrict__ state, float* __restrict__ rgb,
                                             bool store, Traverse traverse, MatView mv = MatView{}) {
    static_assert(!(SPEC && MAT), "the speculative hit-branch draws assume one Lambertian: what is drawn depends on the hit material");
    const int y = global_row(p.row_begin, p.strip_rows, p.strip_step, ly);
    const long long pixel_index = (long long)y * p.nx + x;             // render.h:111 (global)
    const long long si = (long long)ly * p.nx + x;                     // element of this launch's states
    const long long stride = p.rng_count;

    Xorwow rs;
    rs.d = state[0 * stride + si];
    rs.v0 = state[1 * stride + si];
    rs.v1 = state[2 * stride + si];
    rs.v2 = state[3 * stride + si];
    rs.v3 = state[4 * stride + si];
    rs.v4 = state[5 * stride + si];

    const CameraPOD& c = p.cam;
    float colx = 0.0f, coly = 0.0f, colz = 0.0f;
#ifndef CRT_VGPR_CAM
#define CRT_VGPR_CAM 1   // measured on the bench frame: shade 0.816 -> 0.769 ms (profiles/r2_tiled/sweep_occ.txt)
#endif
    const bool kVgprCam = FASTDIV && CRT_VGPR_CAM && WANT_D2;   // ray set-up constants in VGPRs (see in_vgpr); not in the BVH mode (WANT_D2 false), whose kernel is short of registers as it is
    const float fnx = kVgprCam ? in_vgpr_from_int(p.nx) : (float)p.nx, fny = kVgprCam ? in_vgpr_from_int(p.ny) : (float)p.ny;
    const float rnx = kVgprCam ? in_vgpr(p.rnx) : p.rnx, rny = kVgprCam ? in_vgpr(p.rny) : p.rny;
    CameraPOD cv = c;
    if (kVgprCam) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            cv.llc[k] = in_vgpr(c.llc[k]); cv.horizontal[k] = in_vgpr(c.horizontal[k]);
            cv.vertical[k] = in_vgpr(c.vertical[k]); cv.origin[k] = in_vgpr(c.origin[k]);
        }
    }

    const bool wave_has_pixel0 = __builtin_amdgcn_ballot_w64(pixel_index == 0) != 0ull;
    for (int smp = 0; smp < p.samples; ++smp) {
        float u, v;
        if (FASTDIV) {   // same bits as the IEEE divides (see div_by_rcp; numerators >= 2^-33, divisors 1..2^24)
            u = div_by_rcp((float)x + xw_uniform(rs), fnx, rnx);       // render.h:116
            v = div_by_rcp((float)y + xw_uniform(rs), fny, rny);       // render.h:117
        } else {
            u = ((float)x + xw_uniform(rs)) / fnx;                     // render.h:116
            v = ((float)y + xw_uniform(rs)) / fny;                     // render.h:117
        }
        // get_ray(u, v, state): the lens sample comes from state[0] (render.h:118,
        // camera.h:6-13,41).  With lens_radius == 0 only its side effect on stream 0
        // exists; under serial-order semantics only pixel 0 observes it.
        if (wave_has_pixel0 && pixel_index == 0) {                      // (wave-uniform test first: one scalar branch for every other tile)
            float px, py;
            do {
                px = xw_uniform_2m1(rs);
                py = xw_uniform_2m1(rs);
            } while (px * px + py * py + 0.0f * 0.0f >= 1.0f);          // dot(p,p) with p.z = 2*0-0
        }
        // direction = lower_left_corner + s*horizontal + t*vertical - origin - offset, offset = 0
        V3 d0 = camera_direction(cv, u, v);

        if (SKYONLY) {   // a tile without survivors: every sample is background, which needs unit_vector(d0).y only
            const V3 sky = sky_from_unit(V3{0.0f, unit_vector_fast_y(d0), 0.0f});
            colx += sky.x;                                             // render.h:85
            coly += sky.y;
            colz += sky.z;
            continue;
        }
        // world list Transform then triangle Transform: ScaleRay normalises twice
        // (transform.h:50-54 via hitable.h:34 at both levels).
        V3 d1 = FASTDIV ? unit_vector_fast(d0) : unit_vector(d0);
        V3 d2 = d1;
        if (WANT_D2) d2 = FASTDIV ? unit_vector_fast(d1) : unit_vector(d1);

        float closest = FLT_MAX;   // double closest_so_far = t_max (FLT_MAX), hitable_list.h:82
        int hit_idx = -1;
        Xorwow rs_hit = rs;        // SPEC: the stream after the hit branch's draws
        if constexpr (SPEC) {
            traverse(d1, d2, closest, hit_idx, [&]() {
                float sxr, syr, szr;
                do {
                    sxr = xw_uniform_2m1(rs_hit);
                    syr = xw_uniform_2m1(rs_hit);
                    szr = xw_uniform_2m1(rs_hit);
                } while (sxr * sxr + syr * syr + szr * szr >= 1.0f);
            });
        } else {
            traverse(d1, d2, closest, hit_idx);
        }

        V3 sky = sky_from_unit(d1);                                    // backgroundSky(r.direction())
        if (hit_idx >= 0) {                                            // LambertShade, render.h:75-83
            // Lambertian::scatter -> random_in_unit_sphere: only the draws matter (material.h:11-19,66)
            const float* sn = shade_n + (size_t)hit_idx * 3;           // normal after both rotate() calls
            const float snx = sn[0], sny = sn[1], snz = sn[2];         // requested before the draws: the latency overlaps them
            float ax = p.albedo[0], ay = p.albedo[1], az = p.albedo[2], ex = 0.0f, ey = 0.0f, ez = 0.0f;
            if constexpr (MAT) {
                if (mv.normal_shader != 0) {                           // shade_normal, render.h:98: return rec.normal
                    colx += snx;
                    coly += sny;
                    colz += snz;
                    continue;
                }
                // mat_ptr->scatter(...) for its draws and `attenuation`, mat_ptr->emitted (render.h:78-79):
                //   Lambertian / Metal: random_in_unit_sphere (material.h:66,87), attenuation = albedo
                //   Dielectric: ONE uniform (material.h:134), attenuation = (1, 1, 1)
                //   DiffuseLight: no draw, attenuation stays unwritten (defined as 0, as the oracle), emitted = the light's colour
                const MaterialPOD m = mv.materials[mv.material_of ? mv.material_of[hit_idx] : 0];
                if (m.kind <= 1) {
                    float sxr, syr, szr;
                    do {
                        sxr = xw_uniform_2m1(rs);
                        syr = xw_uniform_2m1(rs);
                        szr = xw_uniform_2m1(rs);
                    } while (sxr * sxr + syr * syr + szr * szr >= 1.0f);
                    ax = m.albedo[0]; ay = m.albedo[1]; az = m.albedo[2];
                } else if (m.kind == 2) {
                    (void)xw_uniform(rs);
                    ax = 1.0f; ay = 1.0f; az = 1.0f;
                } else {
                    ax = 0.0f; ay = 0.0f; az = 0.0f;
                    ex = m.albedo[0]; ey = m.albedo[1]; ez = m.albedo[2];
                }
            } else if constexpr (SPEC) {
                rs = rs_hit;
            } else {
                float sxr, syr, szr;
                do {
                    sxr = xw_uniform_2m1(rs);
                    syr = xw_uniform_2m1(rs);
                    szr = xw_uniform_2m1(rs);
                } while (sxr * sxr + syr * syr + szr * szr >= 1.0f);
            }
            float t = d0.x * snx + d0.y * sny + d0.z * snz;            // dot(r.direction(), rec.normal)
            if (t < 0.0f) t = 0.0f;
            // attenuation * t * backgroundSky(...) * 0.2 + emitted
            colx += 0.2f * ((t * ax) * sky.x) + ex;
            coly += 0.2f * ((t * ay) * sky.y) + ey;
            colz += 0.2f * ((t * az) * sky.z) + ez;
        } else {
            colx += sky.x;                                             // render.h:85
            coly += sky.y;
            colz += sky.z;
        }
    }

    // col /= float(ns): k = float(1.0 / double(ns)); col *= k                vec3.h:182-189
    float k = (float)(1.0 / (double)(float)p.samples);
    colx *= k;
    coly *= k;
    colz *= k;
    colx = sqrtf(colx);                                                // render.h:124-126
    coly = sqrtf(coly);
    colz = sqrtf(colz);
    // clip(col): if (f > 1) 1 else if (f < 0) 0 else f (NaN passes through)   vec3.h:119-131
    colx = (colx > 1.0f) ? 1.0f : ((colx < 0.0f) ? 0.0f : colx);
    coly = (coly > 1.0f) ? 1.0f : ((coly < 0.0f) ? 0.0f : coly);
    colz = (colz > 1.0f) ? 1.0f : ((colz < 0.0f) ? 0.0f : colz);

    if (!store) return;
    // The addresses of the stores are formed HERE, from copies of x and ly the compiler cannot trace back (an empty asm):
    // left alone it keeps the 64-bit addresses it formed for the loads at the top alive across the whole sample loop —
    // three VGPR pairs the kernel does not have at six waves per SIMD and spills, per tile (46 MB of scratch writes per frame).
    int x_st = x, ly_st = ly;
    asm volatile("" : "+v"(x_st), "+v"(ly_st));
    const long long si_st = (long long)ly_st * p.nx + x_st;
    const size_t o = (size_t)si_st * 3;
    rgb[o + 0] = colx;
    rgb[o + 1] = coly;
    rgb[o + 2] = colz;

    state[0 * stride + si_st] = rs.d;
    state[1 * stride + si_st] = rs.v0;
    state[2 * stride + si_st] = rs.v1;
    state[3 * stride + si_st] = rs.v2;
    state[4 * stride + si_st] = rs.v3;
    state[5 * stride + si_st] = rs.v4;
}

// The `render` kernel for a pixel of a tile WITHOUT survivors: every sample is background
// (render.h:85), which needs unit_vector(d0).y only.  Same arithmetic as render_pixel<true, true>;
// the camera and the reciprocals live in VGPRs for the loop (VGPR-only operands), and the range
// guard of the fast quotient is decided once per tile when the tile's d0.y cannot come near zero:
// d0.y is affine in (u, v), so its range over the tile follows from the four corners, and
// |d0| <= 2^40 holds for every bounded camera.
template <bool HAS_PIXEL0>   // does the wavefront hold pixel 0 of the frame (the only one that draws lens samples)?
__device__ __forceinline__ void render_sky_pixel_impl(const RenderParams& p, int x, int ly, uint32_t* __restrict__ state,
                                                      float* __restrict__ rgb) {
    const int y = global_row(p.row_begin, p.strip_rows, p.strip_step, ly);
    const long long pixel_index = (long long)y * p.nx + x;
    const long long si = (long long)ly * p.nx + x, stride = p.rng_count;
    Xorwow rs;
    rs.d = state[0 * stride + si]; rs.v0 = state[1 * stride + si]; rs.v1 = state[2 * stride + si];
    rs.v2 = state[3 * stride + si]; rs.v3 = state[4 * stride + si]; rs.v4 = state[5 * stride + si];
    const CameraPOD& c = p.cam;
    const float llx = in_vgpr(c.llc[0]), lly = in_vgpr(c.llc[1]), llz = in_vgpr(c.llc[2]);
    const float hx = in_vgpr(c.horizontal[0]), hy = in_vgpr(c.horizontal[1]), hz = in_vgpr(c.horizontal[2]);
    const float vx = in_vgpr(c.vertical[0]), vy = in_vgpr(c.vertical[1]), vz = in_vgpr(c.vertical[2]);
    const float ox = in_vgpr(c.origin[0]), oy = in_vgpr(c.origin[1]), oz = in_vgpr(c.origin[2]);
    const float fnx = in_vgpr_from_int(p.nx), fny = in_vgpr_from_int(p.ny), rnx = in_vgpr(p.rnx), rny = in_vgpr(p.rny);
    const float fx = (float)x, fy = (float)y;
    // can some ray of this wavefront's tile have |d0.y| < 2^-30 or a huge length?  (u, v) of the tile lie in
    // [x0/nx, (x0+8)/nx] x [y0/ny, (y0+8)/ny] up to rounding; a margin of one pixel each way covers that.
    bool guard_needed;
    {
        const float u0 = ((float)(x & ~7) - 1.0f) * p.rnx, u1 = ((float)(x & ~7) + 9.0f) * p.rnx;
        const float v0 = ((float)(y & ~7) - 1.0f) * p.rny, v1 = ((float)(y & ~7) + 9.0f) * p.rny;
        float lo = 3.0e38f, hi = -3.0e38f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float uu = (k & 1) ? u1 : u0, vv = (k & 2) ? v1 : v0;
            const float dy = ((c.llc[1] + uu * c.horizontal[1]) + vv * c.vertical[1]) - c.origin[1];
            lo = fminf(lo, dy);
            hi = fmaxf(hi, dy);
        }
        const float mag = p.cam_mag;   // sum of |component| over llc, horizontal, vertical, origin (host)
        // safe: over the tile d0.y keeps one sign with |d0.y| >= 2^-20 mag at the corners; the float evaluation here and
        // per sample is within 2^-22 mag of the affine model each, so every sample has |d0.y| >= 2^-21 mag >= 2^-39,
        // and |d0| <= sqrt(3) mag < 2^40: the conditions unit_vector_fast_y votes on hold for all of them
        const float thr = p.cam_thr;   // 2^-20 mag (host)
        const bool safe = mag < 274877906944.0f && thr >= 3.637978807091713e-12f && ((lo > 0.0f && lo >= thr) || (hi < 0.0f && -hi >= thr));
        guard_needed = __builtin_amdgcn_ballot_w64(!safe) != 0ull;
    }
    float colx = 0.0f, coly = 0.0f, colz = 0.0f;
    auto one_sample = [&](auto guarded) {   // guarded: std::true_type = keep the per-sample range vote of unit_vector_fast_y
        const float u = div_by_rcp(fx + xw_uniform(rs), fnx, rnx);     // render.h:116
        const float v = div_by_rcp(fy + xw_uniform(rs), fny, rny);     // render.h:117
        if (HAS_PIXEL0 && pixel_index == 0) {                          // lens sample from state[0] (render.h:118): DESIGN.md §1
            float px, py;
            do {
                px = xw_uniform_2m1(rs);
                py = xw_uniform_2m1(rs);
            } while (px * px + py * py + 0.0f * 0.0f >= 1.0f);
        }
        const float dx = ((llx + u * hx) + v * vx) - ox;               // camera.h:43
        const float dy = ((lly + u * hy) + v * vy) - oy;
        const float dz = ((llz + u * hz) + v * vz) - oz;
        if (decltype(guarded)::value) {
            const float uy = unit_vector_fast_y(V3{dx, dy, dz});
            const float t = 0.5f * (uy + 1.0f);                        // backgroundSky, render.h:41-46
            colx += 1.0f + t * (0.5f - 1.0f);
            coly += 1.0f + t * (0.7f - 1.0f);
            colz += 1.0f + t * (1.0f - 1.0f);
        } else {   // unit_vector_fast_y without its per-sample range vote
            const float s2 = dx * dx + dy * dy + dz * dz;
            const float len = sqrt_normal(s2);
            const float uy = div_by_rcp(dy, len, reciprocal<true>(len));
            const float t = 0.5f * (uy + 1.0f);
            colx += 1.0f + t * (0.5f - 1.0f);
            coly += 1.0f + t * (0.7f - 1.0f);
            colz += 1.0f;                                              // t is finite here: 1 + t * (1 - 1) is 1 exactly
        }
    };
    // Five samples = ten draws bring the five XORWOW words back to the registers they started in: in blocks of five
    // the state rotation needs no copies (a tile holding pixel 0 draws a varying number and keeps the plain loop).
    if (guard_needed) {
        for (int smp = 0; smp < p.samples; ++smp) one_sample(std::true_type{});
    } else {
        const std::false_type g{};
        int smp = 0;
        if (!HAS_PIXEL0)
            for (; smp + 5 <= p.samples; smp += 5) { one_sample(g); one_sample(g); one_sample(g); one_sample(g); one_sample(g); }
        for (; smp < p.samples; ++smp) one_sample(g);
    }
    const float k = (float)(1.0 / (double)(float)p.samples);           // vec3.h:182-189
    colx *= k; coly *= k; colz *= k;
    colx = sqrtf(colx); coly = sqrtf(coly); colz = sqrtf(colz);         // render.h:124-126
    colx = (colx > 1.0f) ? 1.0f : ((colx < 0.0f) ? 0.0f : colx);        // clip, vec3.h:119-131
    coly = (coly > 1.0f) ? 1.0f : ((coly < 0.0f) ? 0.0f : coly);
    colz = (colz > 1.0f) ? 1.0f : ((colz < 0.0f) ? 0.0f : colz);
    int x_st = x, ly_st = ly;                                          // store addresses formed here (see render_pixel)
    asm volatile("" : "+v"(x_st), "+v"(ly_st));
    const long long si_st = (long long)ly_st * p.nx + x_st;
    const size_t o = (size_t)si_st * 3;
    rgb[o + 0] = colx; rgb[o + 1] = coly; rgb[o + 2] = colz;
    state[0 * stride + si_st] = rs.d; state[1 * stride + si_st] = rs.v0; state[2 * stride + si_st] = rs.v1;
    state[3 * stride + si_st] = rs.v2; state[4 * stride + si_st] = rs.v3; state[5 * stride + si_st] = rs.v4;
}

__device__ __forceinline__ void render_sky_pixel(const RenderParams& p, int x, int ly, uint32_t* __restrict__ state,
                                                 float* __restrict__ rgb) {
    const bool wave_has_pixel0 =
        __builtin_amdgcn_ballot_w64(x == 0 && global_row(p.row_begin, p.strip_rows, p.strip_step, ly) == 0) != 0ull;
    if (wave_has_pixel0) render_sky_pixel_impl<true>(p, x, ly, state, rgb);
    else render_sky_pixel_impl<false>(p, x, ly, state, rgb);
}

// ---------------------------------------------------------------- variant 0: brute force
// One lane = one pixel; one wavefront = an 8x8 pixel tile; one 256-thread block = 16x16.
// Every ray walks the whole list.
template <bool FAST>
__global__ void __launch_bounds__(256)
render_list_kernel(RenderParams p, const float* __restrict__ rec,
                   const float* __restrict__ shade_n, uint32_t* __restrict__ state,
                   float* __restrict__ rgb) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int x = blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7);
    const int ly = blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3);   // row inside the band
    if (x >= p.nx || ly >= p.row_count) return;                        // render.h:109
    const int n_tris = p.n_tris;
    render_pixel(p, x, ly, shade_n, state, rgb, true, [&](V3, V3 d2, float& closest, int& hit_idx) {
        walk<FAST>(rec, 0, n_tris, [](int k) { return k; }, d2, closest, hit_idx);
    });
}

#include "crt_bvh_kernels.hpp"   // BVH world: traversal kernel + the device functions the tiled kernel's BVH mode shares

// ---------------------------------------------------------------- variant 1: tiled
// Same pixels, far fewer tests.  A triangle can only be hit by a ray of a screen rectangle
// if the reference's own float predicates can pass for some direction of that rectangle.
// The four quantities those predicates are built from — dn, a, su = dot(s,h),
// dq = dot(d,q) — are linear forms of the direction, hence affine in the screen
// coordinates (u,v) up to the positive factor 1/|d0| and up to rounding.  The prepass
// stores the exact affine coefficients per triangle (computed in fp64); a rectangle is
// classified with interval bounds of those forms plus a slack that covers every rounding
// the reference's float evaluation can commit (DESIGN.md §4.4 has the derivation).
//   binning (lane = triangle) -> per-tile survivor lists; shading (lane = pixel) -> the reference
//   arithmetic (exact_test) on the tile's survivors only.
constexpr int kHeavyMinSmall = 16, kHeavyMinMedium = 24;   // same for small launches (< 24 / < 96 tiles per CU)
constexpr int kHeavyMin = 32;  // tiles with more survivors than this are traced by all waves of a block together (measured optimum, 12..64 swept)
constexpr unsigned kNoList = 0xffffffffu;   // tile_desc.y: the pool was full, walk every triangle
constexpr unsigned kCullSureBit = 0x80000000u;   // pool entry flag: the cull test cannot reject in this tile

// ---- binning, triangle-major (lane = triangle).  Two regular grids over this launch's rows:
// regions (128 x 32 pixels = 16 x 4 tiles; wide, so that multi-GPU strips stay 32 rows) and
// 8 x 8 tiles.  A cell of pixel columns [x_lo,x_hi) has u = fl((x + U)/nx) with fl(x + U) in
// [x_lo, x_hi], i.e. u within 6e-8 of [x_lo/nx, x_hi/nx]; the binning uses the interval
//      centre = fl((x_lo + w/2) * fl(1/nx)),  half = (w/2) fl(1/nx) * 1.00001 + 4e-7 + pos_du
// which contains it (centre is within 1.2e-7 of the true midpoint) widened by the camera's
// position slack (RenderParams::pos_du); likewise v from global rows.  Cells are
// pure arithmetic on wave-uniform integers: no tables, no loads in the loops.
//   bin_tri_kernel<0>   : ONE coalesced pass over the classification records per screen slice;
//                         regions are selected by the triangle's (u,v) bounding box alone,
//                         tiles by box and by the five affine forms; wave votes keep the
//                         control flow uniform; counts per tile
//   bin_alloc_kernel    : pool space per tile, descriptors, work queues
//   bin_tri_kernel<1>   : same walk, writes the triangle index (+ cull-sure bit) into the
//                         tile's list.  Lists are unordered; the exact test breaks ties by index.
// The centre value of a form on a tile is F(uc[col], vc[row]) = (A + vc[row]*C) + uc[col]*B:
// one FMA per form and tile once the row part is known.
constexpr int kRegionW = 128, kRegionH = 32;
constexpr int kRegionTilesX = kRegionW / 8, kRegionTilesY = kRegionH / 8;
constexpr int kBinThreads = 64;
constexpr int kBinWavesMin = 16384;   // the screen is cut into slices of region rows (blockIdx.y) / columns (blockIdx.z)
                                     // until the launch has this many wavefronts: few large triangles are not one wave's work
constexpr int kBigRegions = 16;     // a triangle whose box spans at least this many regions is worked on its own (bin_big)
constexpr int kDirectTiles = 12;    // a triangle whose box touches at most this many tiles (and rows, and columns) tests them itself (bin_direct)
constexpr int kFewTriangles = 8;     // a region with at most this many candidates is classified lane = tile (measured: 4..16 equal, 0 and 32+ slower)

struct FormSet {          // the five affine forms of one triangle (dn, a, su, dq, w = a - su - dq)
    float A[5], B[5], C[5], M[5];
};

__device__ __forceinline__ FormSet load_forms(const float4* __restrict__ cls, int n_pad, int idx) {
    const float4 p0 = cls[(size_t)0 * n_pad + idx], p1 = cls[(size_t)1 * n_pad + idx];
    const float4 p2 = cls[(size_t)2 * n_pad + idx], p3 = cls[(size_t)3 * n_pad + idx];
    FormSet f;
    f.A[0] = p0.x; f.B[0] = p0.y; f.C[0] = p0.z; f.M[0] = p0.w;
    f.A[1] = p1.x; f.B[1] = p1.y; f.C[1] = p1.z; f.M[1] = p1.w;
    f.A[2] = p2.x; f.B[2] = p2.y; f.C[2] = p2.z; f.M[2] = p2.w;
    f.A[3] = p3.x; f.B[3] = p3.y; f.C[3] = p3.z; f.M[3] = p3.w;
    // third barycentric numerator w = a - su - dq, as its own affine form (tight radius)
    f.A[4] = p1.x - p2.x - p3.x; f.B[4] = p1.y - p2.y - p3.y; f.C[4] = p1.z - p2.z - p3.z;
    f.M[4] = p2.w + p3.w + 1.25f * p1.w;
    return f;
}

// some ray of the cell may pass Triangle::collision_detection (NaN anywhere -> true).
// c[] = centre values, q[] = radius*1.0001 + K*M of the five forms.
__device__ __forceinline__ bool cell_may_hit(const float c[5], const float q[5]) {
    const bool cull_ok = !(c[0] < -q[0]);                                   // some ray may have dn >= 0
    const bool pos = !(c[1] < -q[1]) && !(c[2] < -q[2]) && !(c[3] < -q[3]) && !(c[4] < -q[4]);
    const bool neg = !(c[1] > q[1]) && !(c[2] > q[2]) && !(c[3] > q[3]) && !(c[4] > q[4]);
    return cull_ok && (pos || neg);
}

// ONE walk (round 2; round 1 walked twice, count then fill, and the fill pass waited on one returning atomic per
// wavefront and tile): the walk counts the survivors of every tile (fire-and-forget atomics) and logs what it
// found — records (tile, first triangle of the wavefront, 64-bit survivor mask, 64-bit cull-sure mask) collected
// in LDS and appended to a global log kBinLogCap at a time.  After bin_alloc_kernel has placed the tiles' lists,
// bin_fill_kernel turns every record into list entries, one record per lane: the returning atomics that order
// the writers of a tile are then spread over thousands of lanes instead of serialised inside the busiest walk.
// Control flow is wave-uniform (ballots).
constexpr int kBinLogCap = 128;   // records buffered per wavefront (3 KB of LDS)

__device__ __forceinline__ void bin_log_flush(BinRecord* slog, unsigned& n_log, const TiledWork& w, int lane) {
    if (n_log == 0u) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const unsigned seg = (blockIdx.x + blockIdx.y * 7u + blockIdx.z * 13u) % (unsigned)kLogSegments, seg_cap = w.log_cap / (unsigned)kLogSegments;
    unsigned base = 0;
    if (lane == 0) base = atomicAdd(&w.counters[kCtrLogCursor + seg * kCtrLine], n_log);
    base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
    if ((unsigned long long)base + n_log > (unsigned long long)seg_cap) {
        if (lane == 0) w.counters[kCtrLogOverflow] = 1u;   // bin_alloc_kernel then gives every tile the whole list
    } else {
        for (unsigned k = (unsigned)lane; k < n_log; k += 64u) w.log[(size_t)seg * seg_cap + base + k] = slog[k];
    }
    n_log = 0u;
    __builtin_amdgcn_wave_barrier();
}

// The local rows [ly_a, ly_b] of this launch whose GLOBAL rows lie in [g_a, g_b] (none: ly_b < ly_a).  The launch's rows are
// strips of strip_rows rows every strip_rows * strip_step (global_row), so the map is monotone over the rows it owns.
__device__ __forceinline__ void local_rows(const RenderParams& p, int g_a, int g_b, int& ly_a, int& ly_b) {
    const int period = p.strip_rows * p.strip_step;
    const int da = max(g_a - p.row_begin, 0), db = g_b - p.row_begin;
    int sa = da / period, ra = da - sa * period;
    if (ra >= p.strip_rows) { sa += 1; ra = 0; }
    ly_a = sa * p.strip_rows + ra;
    ly_b = -1;
    if (db >= 0) {
        const int sb = db / period, rb = min(db - sb * period, p.strip_rows - 1);
        ly_b = min(sb * p.strip_rows + rb, p.row_count - 1);
    }
}

__global__ void __launch_bounds__(kBinThreads)
bin_tri_kernel(RenderParams p, const float4* __restrict__ cls, int n_pad, TiledWork w,
               unsigned long long* __restrict__ stats_in, int few_triangles) {
    __shared__ BinRecord slog[kBinLogCap];
    unsigned n_log = 0u;   // wave-uniform
    unsigned n_region_hit = 0, n_tile_box = 0, n_tile_hit = 0;   // diagnostics (wave-uniform)
    // diagnostics: the timeline mode (crt_debug_tiled_stats(2)) keeps this kernel's timing intact and records, on the 100 MHz
    // device clock, the last end of a wavefront (slot 30) and the longest wavefront: ticks << 40 | ticks after slot 29's
    // reference (the first wavefront that got there) << 24 | block (slot 31)
    const bool timeline = stats_in != nullptr && stats_in[kStatsTimeline] != 0ull;
    unsigned long long* __restrict__ stats = timeline ? nullptr : stats_in;
    const unsigned long long wall0 = timeline ? (unsigned long long)wall_clock64() : 0ull;
    const long long clk0 = stats ? (long long)clock64() : 0;
    const int tri = blockIdx.x * kBinThreads + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const unsigned long long lanemask_lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const bool have = tri < p.n_tris;
    const int region_nx = (w.tiles_x + kRegionTilesX - 1) / kRegionTilesX;
    const int region_ny = (w.tiles_y + kRegionTilesY - 1) / kRegionTilesY;
    const int rows_per_slice = (region_ny + (int)gridDim.y - 1) / (int)gridDim.y;
    const int ry_begin = (int)blockIdx.y * rows_per_slice, ry_end = min(ry_begin + rows_per_slice, region_ny);
    const int cols_per_slice = (region_nx + (int)gridDim.z - 1) / (int)gridDim.z;
    const int rx_begin = (int)blockIdx.z * cols_per_slice, rx_end = min(rx_begin + cols_per_slice, region_nx);

    const float rnx = 1.0f / (float)p.nx, rny = 1.0f / (float)p.ny;
    const float hu_t = (4.0f * rnx) * 1.00001f + 4.0e-7f + p.pos_du, hv_t = (4.0f * rny) * 1.00001f + 4.0e-7f + p.pos_dv;
    const float hu_r = (0.5f * kRegionW * rnx) * 1.00001f + 4.0e-7f + p.pos_du;
    const float hv_r = (0.5f * kRegionH * rny) * 1.00001f + 4.0e-7f + p.pos_dv;
    // bounding box (u_lo, u_hi, v_lo, v_hi) of the triangle; box tests as "cell centre within
    // [lo - half, hi + half]", written so that NaN keeps the triangle
    const float4 bb = cls[(size_t)4 * n_pad + (have ? tri : 0)];
    const float rv_lo = bb.z - hv_r, rv_hi = bb.w + hv_r;
    // slice test first: most wavefronts of a slice have nothing to do in it
    {
        const float v_first = (float)global_row(p.row_begin, p.strip_rows, p.strip_step, ry_begin * kRegionH) * rny;
        const int ly_last = min(ry_end * kRegionH, p.row_count) - 1;
        const float v_last = (float)(global_row(p.row_begin, p.strip_rows, p.strip_step, max(ly_last, 0)) + 1) * rny;
        const float u_first = (float)(rx_begin * kRegionW) * rnx, u_last = (float)(rx_end * kRegionW) * rnx;
        const bool in_slice = have && ry_begin < ry_end && rx_begin < rx_end && !(bb.w < v_first - hv_t) &&
                              !(bb.z > v_last + hv_t) && !(bb.y < u_first - hu_t) && !(bb.x > u_last + hu_t);
        if (__builtin_amdgcn_ballot_w64(in_slice) == 0ull) return;
    }
    // Small triangles (round 4): a lane whose box touches at most kDirectTiles tiles of this launch tests THOSE tiles itself,
    // one per step (bin_direct below) — the walk over the union of the wavefront's boxes tested 174 tiles per wavefront on the
    // C3 frame (191 on C5) for the 15 that hold a survivor.  The tile range comes from the same thresholds the walk compares the
    // tile centres with (tu_lo .. tv_hi), one tile of margin each way; every tile in it then passes through the walk's own
    // float tests, so both routes select the same (tile, triangle) pairs.
    const float tu_lo = bb.x - hu_t, tu_hi = bb.y + hu_t, tv_lo = bb.z - hv_t, tv_hi = bb.w + hv_t;
    int dtx0 = 0, dnx = 0, dty0 = 0, dny = 0;
    bool small = false;
    if (have && fabsf(bb.x) < 1.0e3f && fabsf(bb.y) < 1.0e3f && fabsf(bb.z) < 1.0e3f && fabsf(bb.w) < 1.0e3f &&   // (NaN fails; the conversions
        p.nx < (1 << 20) && p.ny < (1 << 20)) {                                                                      //  below stay in int range)
        // columns: centre (tx * 8 + 4) / nx within [tu_lo, tu_hi]; 0.01 tile covers the rounding of the float comparison (1e-4)
        const int tx_a = max((int)ceilf((tu_lo * (float)p.nx - 4.0f) * 0.125f - 0.01f), rx_begin * kRegionTilesX);
        const int tx_b = min((int)floorf((tu_hi * (float)p.nx - 4.0f) * 0.125f + 0.01f), min(rx_end * kRegionTilesX, w.tiles_x) - 1);
        // rows: tiles whose first GLOBAL pixel row g has g + 4 within [tv_lo, tv_hi] * ny; the rows of this launch are strips of
        // strip_rows rows every strip_rows * strip_step (global_row): ly_a = the first local row at or above g_a, ly_b = the last
        // at or below g_b; tiles start at local rows that are multiples of 8
        const int g_a = (int)ceilf(tv_lo * (float)p.ny - 4.08f), g_b = (int)floorf(tv_hi * (float)p.ny - 3.92f);
        int ly_a, ly_b;
        local_rows(p, g_a, g_b, ly_a, ly_b);
        const int ty_a = max((ly_a + 7) >> 3, ry_begin * kRegionTilesY);
        const int ty_b = min(ly_b >> 3, min(ry_end * kRegionTilesY, w.tiles_y) - 1);
        dtx0 = tx_a; dty0 = ty_a;
        dnx = max(tx_b - tx_a + 1, 0); dny = (ly_b < 0) ? 0 : max(ty_b - ty_a + 1, 0);
        small = dnx * dny <= kDirectTiles && dnx <= kDirectTiles && dny <= kDirectTiles;
    }
    // Big lanes (round 4): a triangle whose box spans many regions — an edge-on sliver on the silhouette: its relaxed hit set is a
    // long thin strip across the screen — made the walk below scan the whole rectangle of regions around it, one region per
    // step with 63 lanes idle: ONE such wavefront lasted as long as the whole kernel (340 of 400 us on the C5 frame, 45 of 46 us
    // on the bench frame; tools/bin_timeline.py).  Such a lane is now taken out of the walk and worked on its own (bin_big
    // below): 64 regions of ITS rectangle per step, lane = region, then the 64 tiles of every region that passes, lane = tile —
    // the float tests of the walk's two levels with the same operands, so the same (tile, triangle) pairs.
    int brx0 = 0, bnx = 0, bry0 = 0, bny = 0;
    bool big = false;
    if (have && !small && fabsf(bb.x) < 1.0e3f && fabsf(bb.y) < 1.0e3f && fabsf(bb.z) < 1.0e3f && fabsf(bb.w) < 1.0e3f &&
        p.nx < (1 << 20) && p.ny < (1 << 20)) {
        // regions whose centre ((rx * 128 + 64) / nx, (global row of the region's first row + 16) / ny) lies within the
        // box + region slack: the walk's first test (s1_box), as integer ranges with the float test's rounding covered
        const float ru_lo_ = bb.x - hu_r, ru_hi_ = bb.y + hu_r, rv_lo_ = bb.z - hv_r, rv_hi_ = bb.w + hv_r;
        const int rx_a = max((int)ceilf((ru_lo_ * (float)p.nx - 64.0f) * (1.0f / kRegionW) - 0.01f), rx_begin);
        const int rx_b = min((int)floorf((ru_hi_ * (float)p.nx - 64.0f) * (1.0f / kRegionW) + 0.01f), rx_end - 1);
        int ly_a, ly_b;
        local_rows(p, (int)ceilf(rv_lo_ * (float)p.ny - 16.3f), (int)floorf(rv_hi_ * (float)p.ny - 15.7f), ly_a, ly_b);
        const int ry_a = max((ly_a + kRegionH - 1) / kRegionH, ry_begin), ry_b = min(ly_b / kRegionH, ry_end - 1);
        brx0 = rx_a; bry0 = ry_a;
        bnx = max(rx_b - rx_a + 1, 0); bny = (ly_b < 0) ? 0 : max(ry_b - ry_a + 1, 0);
        big = bnx * bny >= kBigRegions;
    }
    const bool have_w = have && !small && !big;   // lanes that take the walk
    // Region range of the walking lanes' union box (a superset is enough: every region is still
    // tested per lane): with a coherent triangle order it is a handful of the screen's regions.
    int rx_lo = rx_begin, rx_hi = rx_end, ry_lo = ry_begin, ry_hi = ry_end;
#ifdef CRT_DBG_BIN_NOWALK   // timing experiment only (lists incomplete)
    const bool any_walk = false;
#else
    const bool any_walk = __builtin_amdgcn_ballot_w64(have_w) != 0ull;
#endif
    if (any_walk) {
        const bool odd = have_w && !(bb.x == bb.x && bb.y == bb.y && bb.z == bb.z && bb.w == bb.w);   // NaN box: keep everything
        if (__builtin_amdgcn_ballot_w64(odd) == 0ull) {
            float u0 = have_w ? bb.x : 1.0e30f, u1 = have_w ? bb.y : -1.0e30f, v0 = have_w ? bb.z : 1.0e30f, v1 = have_w ? bb.w : -1.0e30f;
#pragma unroll
            for (int ofs = 32; ofs > 0; ofs >>= 1) {
                u0 = fminf(u0, __shfl_xor(u0, ofs, 64)); u1 = fmaxf(u1, __shfl_xor(u1, ofs, 64));
                v0 = fminf(v0, __shfl_xor(v0, ofs, 64)); v1 = fmaxf(v1, __shfl_xor(v1, ofs, 64));
            }
            const float cap = 1.0e6f;   // keeps the conversions to int in range; one region of margin each way
            const float fx0 = fminf(fmaxf((u0 - hu_r) * (float)p.nx * (1.0f / kRegionW), -cap), cap);
            const float fx1 = fminf(fmaxf((u1 + hu_r) * (float)p.nx * (1.0f / kRegionW), -cap), cap);
            rx_lo = max(rx_begin, (int)floorf(fx0) - 1);
            rx_hi = min(rx_end, (int)floorf(fx1) + 2);
            if (p.strip_step == 1) {   // contiguous rows: local row = global row - row_begin
                const float fy0 = fminf(fmaxf(((v0 - hv_r) * (float)p.ny - (float)p.row_begin) * (1.0f / kRegionH), -cap), cap);
                const float fy1 = fminf(fmaxf(((v1 + hv_r) * (float)p.ny - (float)p.row_begin) * (1.0f / kRegionH), -cap), cap);
                ry_lo = max(ry_begin, (int)floorf(fy0) - 1);
                ry_hi = min(ry_end, (int)floorf(fy1) + 2);
            }
        }
    }
    const float ru_lo = bb.x - hu_r, ru_hi = bb.y + hu_r;
    const FormSet f = load_forms(cls, n_pad, have ? tri : 0);   // coalesced: 1 KiB per plane and wavefront
    float qt[5], qr[5];   // radius * 1.0001 + K * M on a tile / on a region
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        qt[k] = __builtin_fmaf(p.K_slack, f.M[k], __builtin_fmaf(hv_t, fabsf(f.C[k]), hu_t * fabsf(f.B[k])) * 1.0001f);
        qr[k] = __builtin_fmaf(p.K_slack, f.M[k], __builtin_fmaf(hv_r, fabsf(f.C[k]), hu_r * fabsf(f.B[k])) * 1.0001f);
    }
    // the wavefront's forms as (A, B, C, q on a tile) + the tile-level box thresholds, for the lane = tile mode
    __shared__ float4 sform[6][kBinThreads];
#pragma unroll
    for (int k = 0; k < 5; ++k) sform[k][lane] = make_float4(f.A[k], f.B[k], f.C[k], qt[k]);
    sform[5][lane] = make_float4(tu_lo, tu_hi, tv_lo, tv_hi);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float c[5];
    {   // bin_direct: step k tests tile k of every small lane's own range; lanes that found the same tile share one record
#ifdef CRT_DBG_BIN_NODIRECT   // timing experiment only (lists incomplete)
        const int n_mine = 0;
#else
        const int n_mine = small ? dnx * dny : 0;
#endif
        int n_steps = n_mine;
#pragma unroll
        for (int ofs = 32; ofs > 0; ofs >>= 1) n_steps = max(n_steps, __shfl_xor(n_steps, ofs, 64));
        int ix = 0, ty = dty0;
        float vct = (float)(global_row(p.row_begin, p.strip_rows, p.strip_step, dty0 * 8) + 4) * rny;
        for (int k = 0; k < n_steps; ++k) {
            if (ix == dnx && k < n_mine) {   // next tile row of this lane's range
                ix = 0;
                ++ty;
                vct = (float)(global_row(p.row_begin, p.strip_rows, p.strip_step, ty * 8) + 4) * rny;
            }
            const int tx = dtx0 + ix;
            ++ix;
            const int tile = ty * w.tiles_x + tx;
            const float uct = (float)(tx * 8 + 4) * rnx;
#pragma unroll
            for (int j = 0; j < 5; ++j) c[j] = __builtin_fmaf(uct, f.B[j], __builtin_fmaf(vct, f.C[j], f.A[j]));
            const bool s2 = k < n_mine && !(vct < tv_lo) && !(vct > tv_hi) && !(uct < tu_lo) && !(uct > tu_hi) && cell_may_hit(c, qt);
            const bool sure = s2 && c[0] > qt[0];
            n_tile_box += 1u;
            unsigned long long todo = __builtin_amdgcn_ballot_w64(s2);
            while (todo != 0ull) {
                const int t_first = __builtin_amdgcn_readlane(tile, (int)__builtin_ctzll(todo));
                const unsigned long long m = __builtin_amdgcn_ballot_w64(s2 && tile == t_first);
                const unsigned long long msure = __builtin_amdgcn_ballot_w64(sure && tile == t_first);
                ++n_tile_hit;
                if (n_log + 1u > (unsigned)kBinLogCap) bin_log_flush(slog, n_log, w, lane);
                if (lane == 0) {
                    atomicAdd(&w.tile_count[t_first], (unsigned)__builtin_popcountll(m));
                    BinRecord r;
                    r.tile = (uint32_t)t_first;
                    r.base = (uint32_t)(blockIdx.x * kBinThreads);
                    r.mask = m;
                    r.sure = msure;
                    slog[n_log] = r;
                }
                n_log += 1u;
                todo &= ~m;
            }
        }
    }
    {   // bin_big: one big lane at a time, its values broadcast (wave-uniform: scalar operands — this is not the hot path)
        auto bc = [&](float x, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), src)); };
        for (unsigned long long mbig = __builtin_amdgcn_ballot_w64(big); mbig != 0ull; mbig &= mbig - 1ull) {
            const int src = (int)__builtin_ctzll(mbig);
            const int rx0 = __builtin_amdgcn_readlane(brx0, src), nxr = __builtin_amdgcn_readlane(bnx, src);
            const int ry0 = __builtin_amdgcn_readlane(bry0, src), n_reg = nxr * __builtin_amdgcn_readlane(bny, src);
            float fA[5], fB[5], fC[5], sqt[5], sqr[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) { fA[k] = bc(f.A[k], src); fB[k] = bc(f.B[k], src); fC[k] = bc(f.C[k], src); sqt[k] = bc(qt[k], src); sqr[k] = bc(qr[k], src); }
            const float s_ru_lo = bc(ru_lo, src), s_ru_hi = bc(ru_hi, src), s_rv_lo = bc(rv_lo, src), s_rv_hi = bc(rv_hi, src);
            const float s_tu_lo = bc(tu_lo, src), s_tu_hi = bc(tu_hi, src), s_tv_lo = bc(tv_lo, src), s_tv_hi = bc(tv_hi, src);
            for (int base = 0; base < n_reg; base += 64) {   // lane = region of the rectangle
                const int l = base + lane;
                const int jy = l / nxr;
                const int rx = rx0 + (l - jy * nxr), ry = ry0 + jy;
                const int gy = global_row(p.row_begin, p.strip_rows, p.strip_step, ry * kRegionH);
                const float vcr = (float)(gy + kRegionH / 2) * rny, ucr = (float)(rx * kRegionW + kRegionW / 2) * rnx;
#pragma unroll
                for (int k = 0; k < 5; ++k) c[k] = __builtin_fmaf(ucr, fB[k], __builtin_fmaf(vcr, fC[k], fA[k]));
                const bool s1 = l < n_reg && !(vcr < s_rv_lo) && !(vcr > s_rv_hi) && !(ucr < s_ru_lo) && !(ucr > s_ru_hi) && cell_may_hit(c, sqr);
                for (unsigned long long m1 = __builtin_amdgcn_ballot_w64(s1); m1 != 0ull; m1 &= m1 - 1ull) {   // lane = tile of a passing region
                    const int lr = (int)__builtin_ctzll(m1);
                    const int prx = __builtin_amdgcn_readlane(rx, lr), pry = __builtin_amdgcn_readlane(ry, lr), pgy = __builtin_amdgcn_readlane(gy, lr);
                    ++n_region_hit;
                    const int tx = prx * kRegionTilesX + (lane & (kRegionTilesX - 1)), lty = lane / kRegionTilesX;
                    const int ty = pry * kRegionTilesY + lty;
                    const bool tile_ok = tx < w.tiles_x && ty < w.tiles_y;
                    const int tile = ty * w.tiles_x + tx;
                    const float uct = (float)(tx * 8 + 4) * rnx, vct = (float)(pgy + lty * 8 + 4) * rny;
                    float ct[5];
#pragma unroll
                    for (int k = 0; k < 5; ++k) ct[k] = __builtin_fmaf(uct, fB[k], __builtin_fmaf(vct, fC[k], fA[k]));
                    const bool s2 = tile_ok && !(vct < s_tv_lo) && !(vct > s_tv_hi) && !(uct < s_tu_lo) && !(uct > s_tu_hi) && cell_may_hit(ct, sqt);
                    n_tile_box += 64u;
                    const unsigned long long ms = __builtin_amdgcn_ballot_w64(s2);
                    if (ms == 0ull) continue;
                    const unsigned n_rec = (unsigned)__builtin_popcountll(ms);
                    if (n_log + n_rec > (unsigned)kBinLogCap) bin_log_flush(slog, n_log, w, lane);
                    if (s2) {   // one record per surviving tile: this triangle alone
                        atomicAdd(&w.tile_count[tile], 1u);
                        BinRecord r;
                        r.tile = (uint32_t)tile;
                        r.base = (uint32_t)(blockIdx.x * kBinThreads);
                        r.mask = 1ull << src;
                        r.sure = (ct[0] > sqt[0]) ? (1ull << src) : 0ull;
                        slog[n_log + (unsigned)__builtin_popcountll(ms & lanemask_lt)] = r;
                    }
                    n_log += n_rec;
                }
            }
        }
    }
    for (int ry = ry_lo; any_walk && ry < ry_hi; ++ry) {
        // 32 local rows: inside one strip, so the region's global rows are contiguous
        const int gy = global_row(p.row_begin, p.strip_rows, p.strip_step, ry * kRegionH);
        const float vcr = (float)(gy + kRegionH / 2) * rny;
        const bool in_row = have_w && !(vcr < rv_lo) && !(vcr > rv_hi);
        if (__builtin_amdgcn_ballot_w64(in_row) == 0ull) continue;
        float row1[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) row1[k] = __builtin_fmaf(vcr, f.C[k], f.A[k]);
        for (int rx = rx_lo; rx < rx_hi; ++rx) {
            const float ucr = (float)(rx * kRegionW + kRegionW / 2) * rnx;
            const bool s1_box = in_row && !(ucr < ru_lo) && !(ucr > ru_hi);
            if (__builtin_amdgcn_ballot_w64(s1_box) == 0ull) continue;
            // the forms on the whole region: drops the regions a long thin box only passes by
#pragma unroll
            for (int k = 0; k < 5; ++k) c[k] = __builtin_fmaf(ucr, f.B[k], row1[k]);
            const bool s1 = s1_box && cell_may_hit(c, qr);
            const unsigned long long m1 = __builtin_amdgcn_ballot_w64(s1);
            if (m1 == 0ull) continue;
            ++n_region_hit;
            if (__builtin_popcountll(m1) <= few_triangles) {
                // few candidates: lane = tile of the region (16 x 4), one triangle at a time,
                // broadcast from its lane.  Same arithmetic as below, so both passes and both
                // modes select the same (tile, triangle) pairs.
                const int tx = rx * kRegionTilesX + (lane & (kRegionTilesX - 1)), lty = lane / kRegionTilesX;
                const int ty = ry * kRegionTilesY + lty;
                const bool tile_ok = tx < w.tiles_x && ty < w.tiles_y;
                const int tile = ty * w.tiles_x + tx;
                const float uct = (float)(tx * 8 + 4) * rnx, vct = (float)(gy + lty * 8 + 4) * rny;
                for (unsigned long long mm = m1; mm != 0ull; mm &= mm - 1ull) {
                    const int src = (int)__builtin_ctzll(mm);
                    float q5[5];
#pragma unroll
                    for (int k = 0; k < 5; ++k) {   // broadcast LDS reads: VGPR operands (readlane would feed the VALU through SGPRs)
                        const float4 fk = sform[k][src];
                        c[k] = __builtin_fmaf(uct, fk.y, __builtin_fmaf(vct, fk.z, fk.x));
                        q5[k] = fk.w;
                    }
                    const float4 bx = sform[5][src];   // tu_lo, tu_hi, tv_lo, tv_hi
                    const bool s2 = tile_ok && !(vct < bx.z) && !(vct > bx.w) && !(uct < bx.x) && !(uct > bx.y) && cell_may_hit(c, q5);
                    n_tile_box += 64u;
                    const unsigned long long ms = __builtin_amdgcn_ballot_w64(s2);
                    if (ms == 0ull) continue;
                    const unsigned n_rec = (unsigned)__builtin_popcountll(ms);
                    if (n_log + n_rec > (unsigned)kBinLogCap) bin_log_flush(slog, n_log, w, lane);
                    if (s2) {   // one record per surviving tile: this triangle alone
                        atomicAdd(&w.tile_count[tile], 1u);
                        BinRecord r;
                        r.tile = (uint32_t)tile;
                        r.base = (uint32_t)(blockIdx.x * kBinThreads);
                        r.mask = 1ull << src;
                        r.sure = (c[0] > q5[0]) ? (1ull << src) : 0ull;
                        slog[n_log + (unsigned)__builtin_popcountll(ms & lanemask_lt)] = r;
                    }
                    n_log += n_rec;
                }
                continue;
            }
            const int ty1 = min((ry + 1) * kRegionTilesY, w.tiles_y), tx1 = min((rx + 1) * kRegionTilesX, w.tiles_x);
            for (int ty = ry * kRegionTilesY; ty < ty1; ++ty) {
                const float vct = (float)(gy + (ty - ry * kRegionTilesY) * 8 + 4) * rny;
                const bool s_row = s1 && !(vct < tv_lo) && !(vct > tv_hi);
                if (__builtin_amdgcn_ballot_w64(s_row) == 0ull) continue;
                float row2[5];
#pragma unroll
                for (int k = 0; k < 5; ++k) row2[k] = __builtin_fmaf(vct, f.C[k], f.A[k]);
                for (int tx = rx * kRegionTilesX; tx < tx1; ++tx) {
                    const float uct = (float)(tx * 8 + 4) * rnx;
                    const bool s_box = s_row && !(uct < tu_lo) && !(uct > tu_hi);
                    if (__builtin_amdgcn_ballot_w64(s_box) == 0ull) continue;
                    ++n_tile_box;
#pragma unroll
                    for (int k = 0; k < 5; ++k) c[k] = __builtin_fmaf(uct, f.B[k], row2[k]);
                    const bool s2 = s_box && cell_may_hit(c, qt);
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(s2);
                    if (m == 0ull) continue;
                    ++n_tile_hit;
                    const int tile = ty * w.tiles_x + tx;
                    const unsigned n = (unsigned)__builtin_popcountll(m);
                    // cull-sure: c_dn - r_dn > S_dn with the same r, S  <=>  c_dn > q_dn
                    const unsigned long long msure = __builtin_amdgcn_ballot_w64(s2 && c[0] > qt[0]);
                    if (n_log + 1u > (unsigned)kBinLogCap) bin_log_flush(slog, n_log, w, lane);
                    if (lane == 0) {
                        atomicAdd(&w.tile_count[tile], n);
                        BinRecord r;
                        r.tile = (uint32_t)tile;
                        r.base = (uint32_t)(blockIdx.x * kBinThreads);
                        r.mask = m;
                        r.sure = msure;
                        slog[n_log] = r;
                    }
                    n_log += 1u;
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    bin_log_flush(slog, n_log, w, lane);
    if (timeline && threadIdx.x == 0) {
        const unsigned long long wall1 = (unsigned long long)wall_clock64();
        unsigned long long ref = atomicCAS(&stats_in[29], 0ull, wall0);   // the first to arrive sets the reference
        if (ref == 0ull) ref = wall0;
        atomicMax(&stats_in[30], wall1 - ref + (1ull << 32));             // (+ 2^32: a wavefront that started before the reference stays positive)
        atomicMax(&stats_in[31], ((wall1 - wall0) << 40) | (((wall0 - ref + (1ull << 15)) & 0xffffull) << 24) | (unsigned long long)(blockIdx.x & 0xffffffu));
    }
    if (stats && lane == 0) {   // cells a wavefront walks, per level
        atomicAdd(&stats[0], 1ull);
        atomicAdd(&stats[1], (unsigned long long)n_region_hit);
        atomicAdd(&stats[2], (unsigned long long)n_tile_box);
        atomicAdd(&stats[3], (unsigned long long)n_tile_hit);
        atomicMax(&stats[29], (unsigned long long)n_region_hit);     // the busiest wavefront
        atomicMax(&stats[30], (unsigned long long)n_tile_box);
        atomicMax(&stats[31], ((unsigned long long)(clock64() - clk0) << 24) | (unsigned long long)(blockIdx.x & 0xffffffu));   // cycles << 24 | wavefront
    }
}

// lane = tile: reserve the tile's pool space, descriptor, work queue.  Blocks of 1024 tiles
// aggregate in LDS first: one global atomic per block for the pool and one per weight class.
constexpr int kAllocThreads = 1024;
__global__ void __launch_bounds__(kAllocThreads)
bin_alloc_kernel(TiledWork w, int heavy_min, unsigned shade_blocks, unsigned long long* __restrict__ stats) {
    __shared__ unsigned wave_total[kAllocThreads / 64], class_count[kClasses], class_base[kClasses];
    __shared__ unsigned long long block_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x * kAllocThreads + tid;
    const bool valid = tile < w.tiles_total;
    if (tid < kClasses) class_count[tid] = 0u;
    // block b of the shading kernel starts with heavy item b; later items are pulled from here
    if (blockIdx.x == 0 && tid == 0) w.counters[kCtrHeavyCursor] = shade_blocks;
    const unsigned cnt = valid ? w.tile_count[tile] : 0u;
    unsigned incl = cnt;                                   // inclusive scan inside the wavefront
#pragma unroll
    for (int ofs = 1; ofs < 64; ofs <<= 1) {
        const unsigned v = __shfl_up(incl, ofs, 64);
        if (lane >= ofs) incl += v;
    }
    if (lane == 63) wave_total[wave] = incl;
    __syncthreads();
    unsigned before = 0, total = 0;
#pragma unroll
    for (int k = 0; k < kAllocThreads / 64; ++k) {
        const unsigned v = wave_total[k];
        before += (k < wave) ? v : 0u;
        total += v;
    }
    if (tid == 0) block_base = total ? atomicAdd(&w.counters64[0], (unsigned long long)total) : 0ull;
    __syncthreads();
    const unsigned long long base = block_base;
    const bool fits = base + total <= (unsigned long long)w.pool_cap && w.counters[kCtrLogOverflow] == 0u;   // per block of tiles
    if (valid) {
        w.tile_desc[tile] = fits ? make_uint2((unsigned)base + before + (incl - cnt), cnt) : make_uint2(0u, kNoList);
        if (!fits) w.tile_count[tile] = 0u;                // no fill pass for this tile
    }
    // weight class (heaviest first); the last heavy class starts at heavy_min, below is light
    const unsigned weight = fits ? cnt : 0xffffffffu;
    const unsigned mins[kHeavyClasses] = {768u, 512u, 384u, 256u, 192u, 128u, 96u, 64u, (unsigned)heavy_min};
    const unsigned light_mins[kLightClasses - 1] = {48u, 32u, 16u, 8u, 0u};
    int c = kClasses - 1;
#pragma unroll
    for (int k = kLightClasses - 2; k >= 0; --k)
        if (weight > light_mins[k]) c = kHeavyClasses + k;
#pragma unroll
    for (int k = kHeavyClasses - 1; k >= 0; --k)
        if (weight > mins[k] && weight > (unsigned)heavy_min) c = k;
    if (!valid) c = -1;
    // rank inside the block's share of the class queue: one LDS atomic per wavefront and class
    const unsigned long long lanemask_lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    unsigned rank = 0;
    unsigned long long todo = __builtin_amdgcn_ballot_w64(valid);
    while (todo != 0ull) {
        const int leader = (int)__builtin_ctzll(todo);
        const int cc = __builtin_amdgcn_readlane(c, leader);
        const unsigned long long mc = __builtin_amdgcn_ballot_w64(c == cc);
        unsigned qbase = 0;
        if (lane == leader) qbase = atomicAdd(&class_count[cc], (unsigned)__builtin_popcountll(mc));
        qbase = (unsigned)__builtin_amdgcn_readlane((int)qbase, leader);
        if (c == cc) rank = qbase + (unsigned)__builtin_popcountll(mc & lanemask_lt);
        todo &= ~mc;
    }
    __syncthreads();
    if (tid < kClasses && class_count[tid] != 0u) class_base[tid] = atomicAdd(&w.counters[kCtrQueueLen + tid], class_count[tid]);
    __syncthreads();
    if (valid) w.queue[(size_t)c * w.tiles_total + class_base[c] + rank] = (uint32_t)tile;
    if (stats && valid) {                                   // diagnostics only (crt_debug_tiled_stats)
        atomicAdd(&stats[4], 1ull);
        atomicAdd(&stats[5], (unsigned long long)cnt);
        atomicMax(&stats[6], (unsigned long long)cnt);
        atomicAdd(&stats[7], fits ? 0ull : 1ull);
        int bucket = 0;
        while ((cnt >> bucket) != 0 && bucket < 15) ++bucket;   // bucket b: cnt in [2^(b-1), 2^b)
        atomicAdd(&stats[8 + bucket], 1ull);
    }
}

// One log record per lane: its survivors become entries of the tile's list.  The count doubles as the fill cursor
// (it is back at 0 afterwards); lists are unordered, the exact test breaks ties by index.
__global__ void __launch_bounds__(256)
bin_fill_kernel(TiledWork w) {
    if (w.counters[kCtrLogOverflow] != 0u) return;   // every tile walks the whole list (bin_alloc_kernel)
    // block b works on segment b % kLogSegments of the log, together with the other gridDim.x / kLogSegments blocks of that segment
    const unsigned seg_cap = w.log_cap / (unsigned)kLogSegments, seg = blockIdx.x % (unsigned)kLogSegments;
    const unsigned n_rec = min(w.counters[kCtrLogCursor + seg * kCtrLine], seg_cap);
    const unsigned per_seg = gridDim.x / (unsigned)kLogSegments;   // (the launch uses a multiple of kLogSegments blocks)
    for (unsigned r = (blockIdx.x / (unsigned)kLogSegments) * blockDim.x + threadIdx.x; r < n_rec; r += per_seg * blockDim.x) {
        const BinRecord rec = w.log[(size_t)seg * seg_cap + r];
        const uint2 desc = w.tile_desc[rec.tile];
        if (desc.y == kNoList) continue;
        const unsigned n = (unsigned)__builtin_popcountll(rec.mask);
        unsigned at = atomicSub(&w.tile_count[rec.tile], n) - n;
        for (unsigned long long m = rec.mask; m != 0ull; m &= m - 1ull) {
            const int j = (int)__builtin_ctzll(m);
            w.pool[(size_t)desc.x + at++] = (rec.base + (unsigned)j) | (((rec.sure >> j) & 1ull) ? kCullSureBit : 0u);
        }
    }
}

// binning of one launch: walk (count + log), allocate, fill (all on `stream`)
static hipError_t launch_binning(const RenderParams& p, const float4* d_cls, int n_pad, const TiledWork& w,
                                 int heavy_min, unsigned shade_blocks, unsigned long long* d_stats, bool prezeroed, hipStream_t stream) {
    hipError_t e = hipSuccess;
    if (!prezeroed) {   // launch_frame_prepass zeroes the scratch of its frame itself
        e = hipMemsetAsync(w.counters, 0, kCtrWords * sizeof(uint32_t), stream);
        if (e != hipSuccess) return e;
        e = hipMemsetAsync(w.tile_count, 0, (size_t)w.tiles_total * sizeof(uint32_t), stream);
        if (e != hipSuccess) return e;
    }
    const int region_nx = (w.tiles_x + kRegionTilesX - 1) / kRegionTilesX, region_ny = (w.tiles_y + kRegionTilesY - 1) / kRegionTilesY;
    const int n_waves = (p.n_tris + kBinThreads - 1) / kBinThreads;
    int waves_min = kBinWavesMin, few = kFewTriangles;
#ifdef CRT_TUNING   // tuning builds only (make tuning): the shipped library reads no environment
    if (const char* ev = getenv("CRT_BIN_FEW")) few = std::max(0, atoi(ev));
    if (const char* ev = getenv("CRT_BIN_WAVES")) waves_min = std::max(1, atoi(ev));
#endif
    const int slices_y = std::max(1, std::min(region_ny, waves_min / std::max(n_waves, 1)));
    const int slices_x = std::max(1, std::min(region_nx, waves_min / std::max(n_waves * slices_y, 1)));
    const dim3 grid_tri((unsigned)n_waves, (unsigned)slices_y, (unsigned)slices_x);
    if (p.n_tris > 0) hipLaunchKernelGGL(bin_tri_kernel, grid_tri, dim3(kBinThreads), 0, stream, p, d_cls, n_pad, w, d_stats, few);
    hipLaunchKernelGGL(bin_alloc_kernel, dim3((unsigned)((w.tiles_total + kAllocThreads - 1) / kAllocThreads)), dim3(kAllocThreads), 0, stream, w,
                       heavy_min, shade_blocks, d_stats);
    if (p.n_tris > 0) hipLaunchKernelGGL(bin_fill_kernel, dim3(1024), dim3(256), 0, stream, w);
    return hipGetLastError();
}

// ---- per-pixel candidate masks (shading kernel).  A tile's list holds every triangle some ray
// of the 8x8 tile may hit; one pixel is hit by few of them, and a wavefront pays for a list
// entry as soon as ONE lane needs it.  So before the sample loop each lane classifies the staged
// entries (at most 64) against ITS pixel — the binning's rectangle test on the
// cell [x, x+1) x [y, y+1): same forms, same slack, same arithmetic — and keeps a bit mask of
// candidates.  Every sample then walks the lane's own candidates: the number of steps is the
// longest per-pixel list of the wavefront instead of the tile's list.
struct Cand {
    unsigned lo, hi;             // bit j: staged entry j (resp. 32 + j) may be hit from this lane's pixel
    unsigned sure_lo, sure_hi;   // ... and the cull test `dot(d, n) < 0` cannot reject it there
};

typedef float4 StageArea[4][64];   // one wavefront's LDS staging area: 4 planes x 64 entries x 16 B (plane-major:
                                   // per-lane reads of different entries fall into different banks)
typedef float4 StageArea32[4][32]; // the BVH mode stages 32 entries per wavefront (its LDS also holds slots and leaf boxes; 3 blocks per CU)

// q = radius * 1.0001 + K * M of one form on a 1-pixel cell (hu, hv: pixel half extents incl. position slack)
__device__ __forceinline__ float pixel_q(float K, float hu, float hv, float B, float C, float M) {
    return __builtin_fmaf(K, M, __builtin_fmaf(hv, fabsf(C), hu * fabsf(B)) * 1.0001f);
}
__device__ __forceinline__ void pixel_half_extents(const RenderParams& p, float& hu, float& hv) {
    hu = p.pix_hu;   // (0.5f * p.rnx) * 1.00001f + 4.0e-7f + p.pos_du, formed on the host (crt_api.cpp): in the kernel it was a loop
    hv = p.pix_hv;   // invariant hoisted to the entry and kept in (or spilled from) a VGPR
}
// The rectangle test on the pixel (x, y_global) from forms stored as (A, B, C, q); q4 = q of w.
__device__ __forceinline__ bool pixel_may_hit(const RenderParams& p, float4 f0, float4 f1, float4 f2, float4 f3, float q4,
                                              int x, int y_global, bool& sure) {
    const float uc = ((float)x + 0.5f) * p.rnx, vc = ((float)y_global + 0.5f) * p.rny;
    const float Aw = f1.x - f2.x - f3.x, Bw = f1.y - f2.y - f3.y, Cw = f1.z - f2.z - f3.z;
    float c[5], q[5] = {f0.w, f1.w, f2.w, f3.w, q4};
    c[0] = __builtin_fmaf(uc, f0.y, __builtin_fmaf(vc, f0.z, f0.x));
    c[1] = __builtin_fmaf(uc, f1.y, __builtin_fmaf(vc, f1.z, f1.x));
    c[2] = __builtin_fmaf(uc, f2.y, __builtin_fmaf(vc, f2.z, f2.x));
    c[3] = __builtin_fmaf(uc, f3.y, __builtin_fmaf(vc, f3.z, f3.x));
    c[4] = __builtin_fmaf(uc, Bw, __builtin_fmaf(vc, Cw, Aw));
    const bool may = cell_may_hit(c, q);
    sure = may && c[0] > q[0];   // c_dn - r_dn > S_dn
    return may;
}
// (A, B, C, M) records of the four forms -> (A, B, C, q) on a pixel; returns q of the fifth form
__device__ __forceinline__ float pixel_forms(const RenderParams& p, float4& f0, float4& f1, float4& f2, float4& f3) {
    float hu, hv;
    pixel_half_extents(p, hu, hv);
    const float Bw = f1.y - f2.y - f3.y, Cw = f1.z - f2.z - f3.z, Mw = f2.w + f3.w + 1.25f * f1.w;
    const float q4 = pixel_q(p.K_slack, hu, hv, Bw, Cw, Mw);
    f0.w = pixel_q(p.K_slack, hu, hv, f0.y, f0.z, f0.w);
    f1.w = pixel_q(p.K_slack, hu, hv, f1.y, f1.z, f1.w);
    f2.w = pixel_q(p.K_slack, hu, hv, f2.y, f2.z, f2.w);
    f3.w = pixel_q(p.K_slack, hu, hv, f3.y, f3.z, f3.w);
    return q4;
}

// lane j < n holds list entry j (`entry`, raw pool word).  Leaves the exact records of the n
// entries in `area` and the lane's candidate masks in `cand`.
template <int CAP = 64, class Area>   // CAP: entries the area holds (n <= CAP); CAP <= 32: only the low mask words are used
__device__ __forceinline__ void stage_candidates(Area& area, float* qw, int* ent, const RenderParams& p,
                                                 const float4* __restrict__ cls, int n_pad,
                                                 const float* __restrict__ rec, int entry, int n, int lane,
                                                 int x, int y_global, Cand& cand) {
    const int t = entry & 0x7fffffff;
    if (lane < n) {   // forms of entry `lane` as (A, B, C, q), q = radius * 1.0001 + K * M on a pixel
        float4 f0 = cls[(size_t)0 * n_pad + t], f1 = cls[(size_t)1 * n_pad + t];
        float4 f2 = cls[(size_t)2 * n_pad + t], f3 = cls[(size_t)3 * n_pad + t];
        qw[lane] = pixel_forms(p, f0, f1, f2, f3);
        area[0][lane] = f0; area[1][lane] = f1; area[2][lane] = f2; area[3][lane] = f3;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    cand.lo = cand.hi = cand.sure_lo = cand.sure_hi = 0u;
    for (int k = 0; k < n; ++k) {
        bool sure_b;
        const unsigned may = pixel_may_hit(p, area[0][k], area[1][k], area[2][k], area[3][k], qw[k], x, y_global, sure_b) ? 1u : 0u;   // broadcast reads
        const unsigned sure = sure_b ? 1u : 0u;
        if (CAP <= 32 || k < 32) { cand.lo |= may << (k & 31); cand.sure_lo |= sure << (k & 31); }
        else                     { cand.hi |= may << (k - 32); cand.sure_hi |= sure << (k - 32); }
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < n) {   // now the exact records, same slots, and the pool words (lanes may be switched off later)
        ent[lane] = entry;
        const float4* r = reinterpret_cast<const float4*>(rec + (size_t)t * kRecFloats);
        area[0][lane] = r[0]; area[1][lane] = r[1]; area[2][lane] = r[2]; area[3][lane] = r[3];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// exact_test_vec with a per-lane record: lanes test different triangles (or none: !active).
template <bool FAST>
__device__ __forceinline__ void exact_test_lane(const RecV& rc, int i, bool sure, bool active, V3 d,
                                                float& closest, int& hit_idx) {
    const float nx = rc.a.x, ny = rc.a.y, nz = rc.a.z;
    const float e2x = rc.a.w, e2y = rc.b.x, e2z = rc.b.y;
    const float e1x = rc.b.z, e1y = rc.b.w, e1z = rc.c.x;
    const float sx = rc.c.y, sy = rc.c.z, sz = rc.c.w;
    const float qx = rc.d.x, qy = rc.d.y, qz = rc.d.z;
    const float tq = rc.d.w;

    unsigned long long alive = __builtin_amdgcn_ballot_w64(active);
    if (__builtin_amdgcn_ballot_w64(active && !sure) != 0ull) {
        float dn = d.x * nx + d.y * ny + d.z * nz;                     // triangle.h:61
        alive &= __builtin_amdgcn_ballot_w64(sure || !(dn < 0.0f));
        if (alive == 0ull) return;
    }
    float hx = (d.y * e2z - d.z * e2y);                               // triangle.h:71-72
    float hy = (-(d.x * e2z - d.z * e2x));
    float hz = (d.x * e2y - d.y * e2x);
    float a = e1x * hx + e1y * hy + e1z * hz;
    alive &= __builtin_amdgcn_ballot_w64(!(fabsf(a) < kTriEpsilon));  // triangle.h:77
    if (alive == 0ull) return;
    float f = reciprocal<FAST>(a);                                    // triangle.h:80
    float u = f * (sx * hx + sy * hy + sz * hz);                      // triangle.h:82
    alive &= __builtin_amdgcn_ballot_w64(!(u < 0.0f)) & __builtin_amdgcn_ballot_w64(!(u > 1.0f));   // triangle.h:84
    if (alive == 0ull) return;
    float v = f * (d.x * qx + d.y * qy + d.z * qz);                   // triangle.h:88
    alive &= __builtin_amdgcn_ballot_w64(!(v < 0.0f)) & __builtin_amdgcn_ballot_w64(!(u + v > 1.0f));   // triangle.h:89
    if (alive == 0ull) return;
    float t = f * tq;                                                 // triangle.h:92
    if (__builtin_amdgcn_inverse_ballot_w64(alive) && (t < closest || (t == closest && i < hit_idx))) {  // hitable_list.h:84, order-independent
        closest = t;
        hit_idx = i;
    }
}

// one sample of one lane against its candidates
template <int CAP = 64, class Area>
__device__ __forceinline__ void trace_candidates(Area& area, const int* ent, const Cand& cand, V3 d2,
                                                 float& closest, int& hit_idx) {
#pragma unroll
    for (int half = 0; half < (CAP > 32 ? 2 : 1); ++half) {
        unsigned m = half ? cand.hi : cand.lo;
        const unsigned sm = half ? cand.sure_hi : cand.sure_lo;
        while (__builtin_amdgcn_ballot_w64(m != 0u) != 0ull) {
            const bool active = m != 0u;
            const int j = active ? __builtin_ctz(m) : 0;
            const bool sure = ((sm >> j) & 1u) != 0u;
            m &= m - 1u;
            const int k = half * 32 + j;
            const int e = ent[k];
            RecV rc;
            rc.a = area[0][k];
            rc.b = area[1][k];
            rc.c = area[2][k];
            rc.d = area[3][k];
            exact_test_lane<true>(rc, e & 0x7fffffff, sure || e < 0, active, d2, closest, hit_idx);
        }
    }
}

// (t, index) as one 64-bit key whose unsigned order is the reference's accept order: smaller t
// first (IEEE order of non-NaN floats; -0 is folded into +0 because the reference's `<` does not
// separate them), lower triangle index on equal t (hitable_list.h:84 walks in index order with
// a strict <).  Accepted hits never carry a NaN t.
__device__ __forceinline__ unsigned long long hit_key(float t, int idx) {
    const unsigned b = __float_as_uint(t + 0.0f);
    const unsigned ordered = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    return ((unsigned long long)ordered << 32) | (unsigned long long)(unsigned)idx;
}

// The t of a key (inverse of hit_key's order-preserving map; -0 comes back as +0, which no comparison tells apart).
__device__ __forceinline__ float key_t(unsigned long long key) {
    const unsigned ordered = (unsigned)(key >> 32);
    return __uint_as_float((ordered & 0x80000000u) ? (ordered & 0x7fffffffu) : ~ordered);
}

// ---- balanced candidate walk.  The step count of trace_candidates is the LONGEST per-pixel list
// of the wavefront while the mean is much shorter.  When the (pixel, entry) pairs of the staged
// entries fit the wave's pair buffer they are laid out once per tile, evenly over the lanes:
// every sample then takes ceil(pairs / 64) steps, each lane testing the ray of its pair's pixel
// against its pair's entry; the results meet per pixel as (ordered t, index) keys (atomicMin =
// the reference's accept rule).  Pair word: pixel | entry << 6 | cull-sure << 12.
// Returns the number of pairs, or 0xffffffff when they do not fit (nothing written).
template <int CAP = 64>
__device__ __forceinline__ unsigned build_pairs(const Cand& cand, int lane, unsigned short* __restrict__ pairs, unsigned cap) {
    const unsigned pc = (unsigned)(__builtin_popcount(cand.lo) + (CAP > 32 ? __builtin_popcount(cand.hi) : 0));
    unsigned incl = pc;
#pragma unroll
    for (int ofs = 1; ofs < 64; ofs <<= 1) {
        const unsigned v = __shfl_up(incl, ofs, 64);
        if (lane >= ofs) incl += v;
    }
    const unsigned total = __shfl(incl, 63, 64);
    if (total > cap) return 0xffffffffu;
    unsigned at = incl - pc;
#pragma unroll
    for (int half = 0; half < (CAP > 32 ? 2 : 1); ++half) {
        unsigned m = half ? cand.hi : cand.lo;
        const unsigned sm = half ? cand.sure_hi : cand.sure_lo;
        while (m != 0u) {
            const int j = __builtin_ctz(m);
            m &= m - 1u;
            pairs[at++] = (unsigned short)((unsigned)lane | (unsigned)(half * 32 + j) << 6 | ((sm >> j) & 1u) << 12);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return total;
}

// One sample: all pairs against the rays ray_of(pixel); minima into keys[pixel].  All 64 lanes active.
template <class Area, class RayOf>
__device__ __forceinline__ void trace_pairs(Area& area, const int* ent, const unsigned short* __restrict__ pairs,
                                            unsigned n_pairs, int lane, RayOf ray_of, unsigned long long* keys) {
    for (unsigned base = 0; base < n_pairs; base += 64u) {
        const unsigned j = base + (unsigned)lane;
        const bool active = j < n_pairs;
        const unsigned pr = active ? (unsigned)pairs[j] : 0u;
        const int pix = (int)(pr & 63u), k = (int)((pr >> 6) & 63u);
        const V3 dp = ray_of(pix);
        const int e = ent[k];
        RecV rc;
        rc.a = area[0][k];
        rc.b = area[1][k];
        rc.c = area[2][k];
        rc.d = area[3][k];
        float t_pair = FLT_MAX;
        int i_pair = -1;
        exact_test_lane<true>(rc, e & 0x7fffffff, ((pr >> 12) & 1u) != 0u || e < 0, active, dp, t_pair, i_pair);
        if (i_pair >= 0) atomicMin(&keys[pix], hit_key(t_pair, i_pair));
    }
}

// A wave-uniform 32-bit value in an SGPR of its own (see shade_tiles_kernel).
__device__ __forceinline__ float sgpr_copy(float x) {
    float y;
    asm volatile("s_mov_b32 %0, %1" : "=s"(y) : "s"(x));
    return y;
}
__device__ __forceinline__ int sgpr_copy(int x) {
    int y;
    asm volatile("s_mov_b32 %0, %1" : "=s"(y) : "s"(x));
    return y;
}

// ---- far pairs: the balanced walk for a slice of a shared tile that is LONGER than its staging area.  Rounds 1-3 staged the
// first 64 entries and walked the rest with wave-uniform records — every lane tests every remaining entry whatever its pixel,
// one scalar-memory round trip per entry: 46 entries cost a 768-entry tile 10 us per sample, 219 entries cost the ~1 980-entry
// pole tiles of the C5 frame 45 us, and those 64-sample chains ended the launch (and every rank's share of it).  Instead the
// wave classifies ALL entries of its slice against the pixels once per tile (same test as stage_candidates, forms read with
// wave-uniform loads) and lays the surviving (pixel, entry) pairs out in its staging area as 16-bit words
// pixel | position in the slice << 6 (the area holds 2 048 of them; a slice up to 1 024 entries); every sample then takes
// ceil(pairs / 64) steps, each lane fetching ITS pair's pool word and record from global memory (L2 after the first sample) —
// the pool word two steps ahead, the record one step ahead of the test.
// Returns the number of pairs, or 0xffffffff when they do not fit `cap` words (the caller then stages as before).
constexpr int kFarMaxSlice = 1024;
#ifndef CRT_NO_FAR          // A/B switches of tuning builds
#define CRT_NO_FAR 0
#endif
#ifndef CRT_TL_LIGHT        // diagnostic builds (make EXTRA=-DCRT_TL_LIGHT=1): the timeline also logs every wavefront's last light tile with
#define CRT_TL_LIGHT 0      // geometry (tools/tiled_stats.py); off in the shipped library — its eight SGPRs of state across the light loop cost
#endif                      // the bench frame 1-2 % through spills
__device__ __forceinline__ unsigned build_far_pairs(unsigned short* __restrict__ out, unsigned cap, const RenderParams& p,
                                                    const float4* __restrict__ cls, int n_pad, const uint32_t* __restrict__ lst,
                                                    int first, int step, int n, int lane, bool valid, int x, int y_global,
                                                    bool count_all, unsigned& total_out) {
    const unsigned long long lanemask_lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    unsigned total = 0;
    for (int base = 0; base < n && (total <= cap || count_all); base += 64) {
        const int m = min(64, n - base);
        const unsigned mine = (lane < m) ? lst[first + (base + lane) * step] : 0u;
#pragma unroll 2
        for (int k = 0; k < m; ++k) {
            const int t = __builtin_amdgcn_readlane((int)mine, k) & 0x7fffffff;
            float4 f0 = cls[(size_t)0 * n_pad + t], f1 = cls[(size_t)1 * n_pad + t];
            float4 f2 = cls[(size_t)2 * n_pad + t], f3 = cls[(size_t)3 * n_pad + t];
            const float q4 = pixel_forms(p, f0, f1, f2, f3);
            bool sure;
            const bool may = pixel_may_hit(p, f0, f1, f2, f3, q4, x, y_global, sure) && valid;
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(may);
            const unsigned cnt = (unsigned)__builtin_popcountll(mask);
            if (may && total + cnt <= cap)
                out[total + (unsigned)__builtin_popcountll(mask & lanemask_lt)] = (unsigned short)((unsigned)lane | (unsigned)(base + k) << 6);
            total += cnt;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    total_out = total;   // diagnostics (count_all: the whole slice is counted even when it does not fit)
    return total <= cap ? total : 0xffffffffu;
}

__device__ __forceinline__ RecV load_rec_lane(const float* __restrict__ rec, unsigned t) {
    const float4* r = reinterpret_cast<const float4*>(rec + (size_t)t * kRecFloats);
    RecV rc;
    rc.a = r[0]; rc.b = r[1]; rc.c = r[2]; rc.d = r[3];
    return rc;
}

// One sample over the far pairs (n >= 1) of the slice lst[first + pos * step]: minima into keys[pixel].
// KEYED (BVH mode: the list holds slots): what orders ties and comes back in the key is the tree's order,
// (0x3fffff - partner[slot]) << 9 | tag (walk_list_strided_keyed's keys: the master looks the slot up through the partner table).
template <bool KEYED, class RayOf>
__device__ __forceinline__ void trace_far_pairs(const unsigned short* far, unsigned n, const float* __restrict__ rec,
                                                const uint32_t* __restrict__ lst, int first, int step, int lane, RayOf ray_of,
                                                unsigned long long* keys, const int* __restrict__ partner = nullptr, unsigned tag = 0u) {
    const unsigned ln = (unsigned)lane;
    unsigned w0 = far[ln < n ? ln : 0u], w1 = far[ln + 64u < n ? ln + 64u : 0u];
    unsigned e0 = lst[first + (int)(w0 >> 6) * step], e1 = lst[first + (int)(w1 >> 6) * step];
    RecV cur = load_rec_lane(rec, e0 & 0x7fffffffu);
    int k0 = KEYED ? partner[e0 & 0x7fffffffu] : 0;
    for (unsigned base = 0; base < n; base += 64u) {
        RecV nxt = cur;
        int k1 = k0;
        unsigned w2 = w1, e2 = e1;
        if (base + 64u < n) {   // in flight while this step's record is tested
            nxt = load_rec_lane(rec, e1 & 0x7fffffffu);
            if (KEYED) k1 = partner[e1 & 0x7fffffffu];
        }
        if (base + 128u < n) {
            w2 = far[base + 128u + ln < n ? base + 128u + ln : 0u];
            e2 = lst[first + (int)(w2 >> 6) * step];
        }
        const int pix = (int)(w0 & 63u);
        const V3 dp = ray_of(pix);
        float t_pair = FLT_MAX;
        int i_pair = -1;
        const int idx = KEYED ? (int)(((unsigned)(0x3fffff - k0) << 9) | tag) : (int)(e0 & 0x7fffffffu);
        exact_test_lane<true>(cur, idx, (e0 & kCullSureBit) != 0u, base + ln < n, dp, t_pair, i_pair);
        if (i_pair >= 0) atomicMin(&keys[pix], hit_key(t_pair, i_pair));
        w0 = w1; e0 = e1; cur = nxt; k0 = k1;
        w1 = w2; e1 = e2;
    }
}

// ---- shading kernel of the tiled variant.  Persistent blocks of WAVES wavefronts (exactly as
// many blocks as are resident) take tiles from the work queues.  Heavy tiles (heaviest first;
// block b starts with item b) are traced by all waves of a block together: wave 0 owns the
// pixels' RNG streams, generates the 64 rays of a sample and publishes them in LDS; every wave
// traces them against its own slice of the tile's list; the partial results meet in LDS as
// 64-bit (ordered t, index) keys (atomicMin = the reference's accept rule); wave 0 shades.
// Light tiles are then pulled one per wave from 64 interleaved chunks of the light queue
// (one cursor per cache line).  VREC: the slice / list is staged in LDS, every lane classifies
// the entries against ITS pixel (stage_candidates) and the surviving (pixel, entry) pairs are
// walked evenly spread over the lanes (build_pairs / trace_pairs; trace_candidates when they
// do not fit);  !VREC (variant 4, A/B): wave-uniform walks with scalar-operand records.
// BVH (the tree as the world, primary rays): the scene arrays are in leaf ("slot") order, so the binning
// yields slot lists.  A triangle is hit through the tree iff it passes Triangle::collision_detection
// (with the direction of its depth: c3 on the hot chain) AND the boxes of its leaf and of all the
// leaf's ancestors pass AABB::hit; the tile's list is a superset of the first condition, so per
// sample the staged walk finds every pixel's winner under the tree's tie order (BvhTiles::partner) and the
// winner's chain is checked — once per hit pixel, at full lane utilisation: on the hot chain that is the
// leaf's box (staged in LDS), which implies every ancestor from depth 2 on, plus the root and one of its
// children where the tile's cone does not prove them (crt_bvh_kernels.hpp, "the chain of boxes of a winner");
// a winner whose chain fails is excluded for that pixel and the walk repeated (rare); entries whose leaf's
// box no ray can pass are dropped when the tile is staged.  Tiles with more than 32 survivors are shared by
// the block (the master wave publishes c3, merges the slices' keys and checks the winner's chain); tiles
// without a list and wave-samples off the hot direction chain take the tree itself (bvh_trace) or the
// literal chain (general_winner, chain_ok_general).
// FULL: the launch policy of a full frame as compile-time constants — light tiles dealt in groups of four, no row slabs except
// for the heaviest class (four; eight in the BVH mode) — instead of the decode of split_packed that small launches (one rank's strips),
// test hooks and the tuning build need (round 3 decoded it in every launch: +2 % on the bench frame's shading kernel, same-box
// A/B in profiles/r4/ab_r2_vs_head.txt).
template <bool VREC, int OCC, int WAVES, bool BVH = false, bool MAT = false, bool FULL = false>
__global__ void __launch_bounds__(WAVES * 64, OCC)
shade_tiles_kernel(RenderParams p_arg, const float* __restrict__ rec, const float4* __restrict__ cls, int n_pad,
                   const float* __restrict__ shade_n, uint32_t* __restrict__ state, float* __restrict__ rgb,
                   TiledWork w, unsigned long long* __restrict__ stats_in, int dbg_repeat_in, BvhTiles bvh,
                   unsigned split_packed, MatView mv) {
    // The scalars of `p` that the per-sample code reads, as registers of their own: the kernel's arguments arrive in 16-dword
    // tuples (s_load_dwordx16), a tuple is spilled and reloaded as a whole, and with a few more values live across the
    // shared-tile phase (the far pairs) hipcc spilled the tuple that holds the albedo — 20 v_readlane per hit sample of
    // every light tile, +3.5 % on the bench frame.
    // (a real s_mov each: an empty asm with a tied operand is coalesced back into the tuple's sub-register)
    RenderParams p = p_arg;
    p.albedo[0] = sgpr_copy(p_arg.albedo[0]);
    p.albedo[1] = sgpr_copy(p_arg.albedo[1]);
    p.albedo[2] = sgpr_copy(p_arg.albedo[2]);
    p.samples = sgpr_copy(p_arg.samples);
    p.nx = sgpr_copy(p_arg.nx);
    // bvh by value: its pointers come with the kernel's arguments and are global memory to the compiler (loaded from a struct in
    // device memory they were generic: 119 FLAT loads in the BVH mode — which cost nothing measurable, 1.16 against 1.17 ms)
    // diagnostics: stats_in[kStatsTimeline] != 0 selects the light-weight timeline (wave start/end times only), which
    // leaves the kernel's timing intact; otherwise the full counters (which roughly double the kernel time)
    const bool timeline = stats_in != nullptr && stats_in[kStatsTimeline] != 0ull;
    unsigned long long* __restrict__ stats = timeline ? nullptr : stats_in;
    const unsigned long long wall0 = timeline ? (unsigned long long)wall_clock64() : 0ull;
    unsigned long long tl_heavy = 0ull, tl_light = 0ull, tl_light_t = 0ull, tl_counts = 0ull, tl_light_end = 0ull;
    __shared__ unsigned long long best_key[64];   // per pixel: min over the waves of (ordered t, index)
    __shared__ float ray_d[3][64];
    __shared__ unsigned s_item;
    // kStage: list entries one wavefront stages (forms, then exact records).  The BVH mode stages 32: with its slots and leaf
    // boxes next to them a 64-entry area made the block 76 KB = two blocks per CU = 4 wavefronts per SIMD; at 32 the block is
    // below 53 KB and three fit.  (Light tiles never hold more than 32 entries; a heavy tile's slice beyond its staged part
    // takes the keyed scalar walk.)
    constexpr int kStage = BVH ? 32 : 64;
    using Area = typename std::conditional<BVH, StageArea32, StageArea>::type;
    __shared__ Area lrec[VREC ? WAVES : 1];                // this wave's staged list entries: forms, then exact records (4 KB per wave; BVH: 2 KB)
    __shared__ float lqw[VREC ? WAVES : 1][kStage];        // slack + radius of the fifth form of the staged entries
    __shared__ int lent[VREC ? WAVES : 1][kStage];         // pool words of the staged entries
    // balanced candidate walk of light tiles (8-wave blocks only: the LDS of 16-wave blocks is spoken for)
    constexpr int kPairCap = (VREC && WAVES <= 8) ? 512 : 1;
    __shared__ unsigned short lpairs[(VREC && WAVES <= 8) ? WAVES : 1][kPairCap];   // (pixel | entry << 6 | cull-sure << 12)
    __shared__ unsigned long long lkey[(VREC && WAVES <= 8) ? WAVES : 1][64];       // per pixel: min (ordered t, index)
    // BVH mode, staged tiles: per list entry its slot, the box of its LEAF (min.xyz, max.x | max.yz) and a flag (stage_leaf): on
    // the hot chain the winner's chain of boxes is its leaf's box + the root + one child of the root (crt_bvh_kernels.hpp,
    // "the chain of boxes of a winner"), so the common sample needs no node loads at all
    __shared__ int lslot[BVH ? WAVES : 1][BVH ? kStage : 1];
    __shared__ int lflag[BVH ? WAVES : 1][BVH ? kStage : 1];
    __shared__ float4 lbox0[BVH ? WAVES : 1][BVH ? kStage : 1];
    __shared__ float2 lbox1[BVH ? WAVES : 1][BVH ? kStage : 1];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_tris = p.n_tris;
    const int dbg_pairs = (dbg_repeat_in < 0) ? 0 : 1;   // tuning hook: a negative repeat count switches the balanced walk off
    const int dbg_repeat = (dbg_repeat_in < 0) ? -dbg_repeat_in : dbg_repeat_in;
    const long long clk0 = stats ? (long long)clock64() : 0;

    // ---- heavy tiles: the classes form one list, heaviest first.  Block b starts with item b
    // (no atomic: all blocks are resident), later items come from one shared cursor.
    // Sub-items: a heavy tile of weight class c is dealt out as 1 << sl(c) items (sl = 3 bits of split_packed per class),
    // each the same list for a slab of 8 >> sl rows of the tile's pixels.  The 64 samples of a pixel are a serial
    // chain (the RNG position of sample k + 1 depends on the rejection loop of sample k), so the time of a shared tile
    // is 64 x (time of one sample); the (pixel, entry) pairs of one sample split evenly over the slabs, and the blocks
    // that take the slabs run side by side — the longest chain of a launch gets shorter by the split factor, down to
    // the master wave's own per-sample work.
    auto split_of = [&](int c) -> unsigned {   // log2 of the row slabs a heavy tile of class c is dealt out as
        if (FULL) return c == 0 ? (BVH ? 3u : 2u) : 0u;
        return (split_packed >> (3 * c)) & 7u;
    };
    unsigned total_heavy = 0;
#pragma unroll
    for (int c = 0; c < kHeavyClasses; ++c) total_heavy += w.counters[kCtrQueueLen + c] << split_of(c);
#ifndef CRT_PRIO
#define CRT_PRIO 1
#endif
#ifndef CRT_PRIO_HEAVY
#define CRT_PRIO_HEAVY 3
#define CRT_PRIO_GEOM 1
#define CRT_PRIO_SKY 0
#endif
#if CRT_PRIO
    // Issue priority by item length: the shared (heavy) tiles are the longest serial chains of the launch and decide
    // when it ends; light tiles with geometry come next; sky tiles are the filler.
    __builtin_amdgcn_s_setprio(CRT_PRIO_HEAVY);
#endif
    {
        unsigned item = blockIdx.x;
        const int lane_h = lane;
        while (item < total_heavy) {
            int lane = lane_h;   // opaque per-item copy (see the light loop): LDS addresses are formed per item, not hoisted and spilled
            asm volatile("" : "+v"(lane));
            int c = 0;
            unsigned idx = item;
            for (; c < kHeavyClasses - 1; ++c) {
                const unsigned len = w.counters[kCtrQueueLen + c] << split_of(c);
                if (idx < len) break;
                idx -= len;
            }
            const unsigned sl = split_of(c);
            const int sub = (int)(idx & ((1u << sl) - 1u));
            const int tile_id = (int)w.queue[(size_t)c * w.tiles_total + (idx >> sl)];
            const uint2 desc = w.tile_desc[tile_id];
            const bool walk_all = desc.y == kNoList;
            // this item: rows [sub * sub_rows, + sub_rows) of the tile.  A tile without a list (pool overflow) walks every
            // triangle whatever its pixels: one item for the whole tile, its other items are empty.
            const int sub_rows = walk_all ? 8 : (8 >> sl);
            const bool skip_item = (walk_all && sub != 0) || ((tile_id / w.tiles_x) * 8 + sub * sub_rows >= p.row_count);
            const int n_src = skip_item ? 0 : (walk_all ? n_tris : (int)desc.y);
            const uint32_t* __restrict__ lst = w.pool + desc.x;
            if (timeline) tl_heavy = ((unsigned long long)(unsigned)n_src << 24) | (unsigned)tile_id;
            // The master wave (wave 0) has no slice — it runs the rejection draws of the hit branch on a copy of the streams while
            // the other WAVES - 1 waves trace (render_pixel<.., SPEC>) — unless (BVH world) seven slices would not fit the staging
            // areas (7 x 32 entries): then eight slices, the master tracing one itself.  An unstaged winner costs the master three
            // dependent global loads per sample (slot, path, nodes): the two 247-entry pole tiles of the C3 frame took 440 or 860 us
            // with seven slices of 36, depending on where the binning's atomics had left the winners in the list.
            const int slice_waves = (BVH && n_src > (WAVES - 1) * (BVH ? 32 : 64)) ? WAVES : WAVES - 1;
            const int swave = (slice_waves == WAVES) ? wave : wave - 1;
            const int per = (n_src + slice_waves - 1) / slice_waves;
            const int k0 = (swave < 0) ? 0 : min(n_src, swave * per), k1 = (swave < 0) ? 0 : min(n_src, k0 + per);
            // Tiles with a list: the slices are INTERLEAVED (wave s takes entries s, s + 7, s + 14, ...).  Neighbours in a
            // tile's list are neighbours in the mesh and cover the same pixels; dealt round-robin, every slice gets its share
            // of the (pixel, entry) pairs, and none overflows its pair buffer while the others idle at the barrier (measured:
            // with contiguous slices the 247-entry pole tiles took 245 or 640 us depending on the order the binning's
            // atomics happened to leave the list in — and ended the launch in the second case).
            const bool interleaved = !walk_all;
            const int cnt_w = (!interleaved || swave < 0 || swave >= n_src) ? 0 : (n_src - swave + slice_waves - 1) / slice_waves;
            // local rows of the tile (tiles are 8 local rows; they never straddle a strip)
            const int hx0 = (tile_id % w.tiles_x) * 8, tly0 = (tile_id / w.tiles_x) * 8;   // tly0: the tile's first local row
            const int hly0 = tly0 + sub * sub_rows;                                         // hly0: this item's
            const int hx1 = min(hx0 + 8, p.nx), hly1 = min(hly0 + sub_rows, p.row_count);
            const int xr = hx0 + (lane & 7), lyr = hly0 + (lane >> 3);
            const bool lane_valid = xr < hx1 && lyr < hly1;
            // lanes off the item's slab (or off the image) replay a valid pixel of the tile, unstored and without candidates
            const int x = min(xr, hx1 - 1), ly = max(min(lyr, hly1 - 1), tly0);
            // Master/worker: wave 0 owns the pixels' RNG streams, generates the 64 rays of each
            // sample and shades; the other waves only trace.  Two barriers per sample:
            // B1 "rays are in LDS", B2 "partial minima are in LDS".
            // this wave's slice: its first kStage entries are staged in LDS like a light tile's list (per-pixel candidate masks,
            // balanced pair walk); the rest of a longer slice is walked with scalar-operand records — in the BVH mode in the
            // tree's tie order (walk_list_strided_keyed; rounds 2-3 passed up to four chunks of 64 through the staging area per
            // sample instead, whose candidate masks were 24 VGPRs live across the sample loop: most of that kernel's scratch).
            // Tiles without a list walk the tree, wave 0 alone.
            const bool tree_tile = BVH && (walk_all || !VREC);
            // list world, a slice longer than the staging area: all of it as far pairs (build_far_pairs), nothing staged
            unsigned n_far = 0xffffffffu, n_far_all = 0u;
            if (!CRT_NO_FAR && VREC && WAVES <= 8 && __builtin_expect(!walk_all && cnt_w > kStage && cnt_w <= kFarMaxSlice && dbg_pairs != 0, 0))
                n_far = build_far_pairs(reinterpret_cast<unsigned short*>(&lrec[wave]), (unsigned)(sizeof(Area) / sizeof(unsigned short)), p, cls, n_pad,
                                        lst, swave, slice_waves, cnt_w, lane, lane_valid, x,
                                        global_row(p.row_begin, p.strip_rows, p.strip_step, ly), stats != nullptr, n_far_all);
            const bool far = n_far != 0xffffffffu;
            if (stats && lane == 0 && VREC && !walk_all && cnt_w > kStage) {   // slices as far pairs, their pairs, slices that did not fit, the most pairs of a slice
                atomicAdd(&stats[43], far ? 1ull : 0ull);
                atomicAdd(&stats[44], far ? (unsigned long long)n_far : 0ull);
                atomicAdd(&stats[45], far ? 0ull : 1ull);
                atomicMax(&stats[46], (unsigned long long)n_far_all);
            }
            const int n_lds = (VREC && !walk_all && !tree_tile && !far) ? min(kStage, cnt_w) : 0;
            int slice_mine = 0;
            Cand cand = {0u, 0u, 0u, 0u};
            if (n_lds > 0) {   // per-pixel candidates among the staged entries of this wave's slice
                slice_mine = (lane < n_lds) ? (int)lst[swave + lane * slice_waves] : 0;
                stage_candidates<kStage>(lrec[wave], lqw[wave], lent[wave], p, cls, n_pad, rec, slice_mine, n_lds, lane, x,
                                         global_row(p.row_begin, p.strip_rows, p.strip_step, ly), cand);
            }
            // BVH mode: which of the root's and its children's boxes every ray of the tile passes (tile_top_proven), and for a staged
            // slice — like a light tile's list — per entry: the tree's tie order with the entry's place (wave, position) in the
            // low bits of the key, its slot, its leaf's box.  The master then finds the winner's staged data one LDS read away
            // and tests its chain without node loads (rounds 1-2 went through slot_leaf, the ancestor table and the nodes: three
            // dependent round trips per sample, 3.3 us of the 7.5 us a shared tile's sample took; round 3 one round of loads).
            unsigned top_h = 0u;
            if (BVH && !tree_tile && !skip_item) {
                const BvhTiles& bt = bvh;
                const V3 o = V3{p.cam.origin[0], p.cam.origin[1], p.cam.origin[2]};
                if (wave == 0) top_h = tile_top_proven(bt, o, tile_cone(p, hx0, global_row(p.row_begin, p.strip_rows, p.strip_step, tly0)));
                if (n_lds > 0) {
                    bool dead = false;
                    if (lane < n_lds) {
                        const int slot = slice_mine & 0x7fffffff;
                        // key: (tie order, 22 bits: smaller wins = later leaf, first in its leaf) | wave | position (host: at most 2^22 triangles)
                        lent[wave][lane] = (int)(((unsigned)slice_mine & kCullSureBit) | ((unsigned)(0x3fffff - bt.partner[slot]) << 9) |
                                                 ((unsigned)wave << 6) | (unsigned)lane);
                        lslot[wave][lane] = slot;
                        stage_leaf(bt, slot, lflag[wave][lane], lbox0[wave][lane], lbox1[wave][lane]);
                        dead = box_never_hit(lbox0[wave][lane], lbox1[wave][lane], o);
                    }
                    cand.lo &= ~(unsigned)__builtin_amdgcn_ballot_w64(dead);   // (see the light tiles)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
            }
            unsigned slice_pairs = 0xffffffffu;   // balanced walk of the staged part of the slice (rays and keys are in LDS anyway)
            if (VREC && WAVES <= 8 && n_lds > 0 && dbg_pairs != 0) {
                if (!lane_valid) cand.lo = cand.hi = 0u;
                slice_pairs = build_pairs<kStage>(cand, lane, lpairs[wave], (unsigned)kPairCap);
            }
            auto trace_slice = [&](V3 d2, float& closest, int& hit_idx) {
              for (int rep = 0; rep < dbg_repeat; ++rep) {
                if (!CRT_NO_FAR && __builtin_expect(far, 0)) {
                    if (n_far != 0u)
                        trace_far_pairs<BVH>(reinterpret_cast<const unsigned short*>(&lrec[wave]), n_far, rec, lst, swave, slice_waves, lane,
                                             [&](int pix) { return V3{ray_d[0][pix], ray_d[1][pix], ray_d[2][pix]}; }, best_key,
                                             BVH ? bvh.partner : nullptr, ((unsigned)wave << 6) | 32u);
                    continue;
                }
                if (slice_pairs != 0xffffffffu)
                    trace_pairs(lrec[wave], lent[wave], lpairs[wave], slice_pairs, lane,
                                [&](int pix) { return V3{ray_d[0][pix], ray_d[1][pix], ray_d[2][pix]}; }, best_key);
                else if (n_lds > 0) trace_candidates<kStage>(lrec[wave], lent[wave], cand, d2, closest, hit_idx);
                if (BVH) {   // (a tree tile never comes here) the unstaged rest of the slice, keys in the tree's order, tagged: position bit 5
                    if (interleaved) walk_list_strided_keyed<true>(rec, lst, bvh.partner, ((unsigned)wave << 6) | 32u, swave, slice_waves, n_lds,
                                                                   cnt_w, lane, d2, closest, hit_idx);
                } else if (interleaved) walk_list_strided<true>(rec, lst, swave, slice_waves, n_lds, cnt_w, lane, d2, closest, hit_idx);
                else walk<true>(rec, k0, k1, [](int k) { return k; }, d2, closest, hit_idx);   // no list: a range of all triangles
              }
            };
            if (skip_item) {
                // nothing to render: every wave of the block goes on to the next item
            } else if (BVH && tree_tile) {
                if (wave == 0) {
                    BvhView view;
                    view.nodes = bvh.nodes;
                    view.n_nodes = bvh.n_nodes;
#pragma unroll
                    for (int k = 0; k < 9; ++k) view.rot[k] = bvh.rot[k];
                    const V3 o = V3{p.cam.origin[0], p.cam.origin[1], p.cam.origin[2]};
                    render_pixel<true, false, false, false, MAT>(p, x, ly, shade_n, state, rgb, lane_valid, [&](V3 d1, V3, float& closest, int& hit_idx) {
                        V3 used;
                        bvh_trace<true>(view, rec, o, d1, 0, closest, hit_idx, used);
                    }, mv);
                }
            } else if (BVH && wave == 0) {
                // Master wave, BVH mode: publishes c3 (what the triangles see on the hot chain), merges the slices' keys,
                // checks the winner's chain of boxes; a lane off the hot chain or with an unreachable winner resolves its
                // pixel alone from the tile's whole list (general_winner + chain_ok_general, excluding what fails).
                const BvhTiles& bt = bvh;
                const V3 o = V3{p.cam.origin[0], p.cam.origin[1], p.cam.origin[2]};
                const unsigned top = top_h;
                int tl_sample_bvh = 0;   // timeline mode: (slices done, chains checked) per sample of the first blocks' first tile
                render_pixel<true, false, false, !MAT, MAT>(p, x, ly, shade_n, state, rgb, lane_valid,
                             [&](V3 d1, V3, float& closest, int& hit_idx, auto&&... hit_branch_draws) {
                    V3 c3;   // (hot_chain: the chain's other elements are recomputed in the rare branches that need them)
                    const bool hot = hot_chain(d1, c3) && dbg_repeat_in != -77;
                    ray_d[0][lane] = c3.x;
                    ray_d[1][lane] = c3.y;
                    ray_d[2][lane] = c3.z;
                    best_key[lane] = ~0ull;
                    __syncthreads();   // B1
                    if (swave >= 0) {   // eight slices: the master traces one too
                        int q = -1;
                        float tb = FLT_MAX;
                        trace_slice(c3, tb, q);
                        if (q >= 0) atomicMin(&best_key[lane], hit_key(tb, q));
                    }
                    (hit_branch_draws(), ...);   // one Lambertian: the hit branch's draws on a copy of the stream, while the other waves trace
                    __syncthreads();   // B2
                    if (timeline && blockIdx.x < (unsigned)kStatsSampleBlocks && lane == 0 && tl_sample_bvh < kStatsSampleSlots)
                        stats_in[kStatsSampleLog + (blockIdx.x * kStatsSampleSlots + tl_sample_bvh) * 2 + 0] = (unsigned long long)wall_clock64();   // slices done
                    const unsigned long long kbest = best_key[lane];
                    int slot = -1;
                    bool alone = !hot;
#ifdef CRT_DBG_NOCHAIN_HEAVY   // timing experiment only (wrong pixels where a chain fails)
                    if (hot && kbest != ~0ull) {
                        const int qk0 = (int)(unsigned)(kbest & 0xffffffffull);
                        slot = (qk0 & 32) ? bt.partner[0x3fffff - ((qk0 >> 9) & 0x3fffff)] : lslot[(qk0 >> 6) & 7][qk0 & 31];
                        alone = false;
                    }
                    if (false) {
#else
                    if (hot && kbest != ~0ull) {
#endif
                        const int qk = (int)(unsigned)(kbest & 0xffffffffull);
                        if (qk & 32) {   // an unstaged entry of a long slice: the slot through the tie order, leaf and flag through the tables
                            slot = bt.partner[0x3fffff - ((qk >> 9) & 0x3fffff)];
                            alone = !chain_ok_hot(bt, slot, top, o, d1, c3);
                        } else {         // staged: the winner's wave and position are in the key
                            const int sw = (qk >> 6) & 7, kw = qk & 31;
                            slot = lslot[sw][kw];
                            const int flag = lflag[sw][kw];
                            bool ok = (flag & 2) != 0 || leaf_box_hit(lbox0[sw][kw], lbox1[sw][kw], o, c3);
                            const bool rest = ok && chain_needs_top(flag, top);
                            if (__builtin_amdgcn_ballot_w64(rest) != 0ull && rest) ok = chain_top_ok(bt, slot, flag, o, d1, c3);
                            alone = !ok;
                        }
                    }
                    if (alone) {   // per lane, over the tile's whole list, with the literal chain
                        const V3 c1 = rot0_exact(bt.rot, d1);
                        const V3 c2 = chain_next<true>(bt.rot, c1);
                        const V3 c3l = chain_next<true>(bt.rot, c2);
                        const V3 c4 = chain_next<true>(bt.rot, c3l);
                        const V3 c5 = rot0(bt.rot, unit_vector(c4));
                        const int mode = same_bits(c4, c3l) ? 0 : (same_bits(c5, c4) ? 0 : (same_bits(c5, c3l) ? 1 : 2));
                        int gex[8];
                        int n_ex = 0;
                        for (;;) {
                            const int qg = general_winner_all(bt, rec, lst, n_src, gex, n_ex, o, c1, c2, c3l, c4, mode, slot);
                            if (qg < 0 || chain_ok_general(bt, slot, o, c1, c2, c3l, c4, mode)) break;
                            if (n_ex == 8) {   // more unreachable winners than this lane keeps track of: the tree itself decides
                                BvhView view;
                                view.nodes = bt.nodes;
                                view.n_nodes = bt.n_nodes;
#pragma unroll
                                for (int k = 0; k < 9; ++k) view.rot[k] = bt.rot[k];
                                V3 used;
                                float tw;
                                bvh_trace<true>(view, rec, o, d1, 0, tw, slot, used);
                                break;
                            }
#pragma unroll
                            for (int j = 0; j < 8; ++j)
                                if (j == n_ex) gex[j] = slot;
                            ++n_ex;
                        }
                    }
                    hit_idx = slot;
                    closest = 0.0f;
                    if (timeline && blockIdx.x < (unsigned)kStatsSampleBlocks && lane == 0 && tl_sample_bvh < kStatsSampleSlots) {
                        stats_in[kStatsSampleLog + (blockIdx.x * kStatsSampleSlots + tl_sample_bvh) * 2 + 1] = (unsigned long long)wall_clock64();   // winners' chains checked
                        ++tl_sample_bvh;
                    }
                }, mv);
            } else if (!BVH && wave == 0) {
                long long tc_prev = stats ? (long long)clock64() : 0, tc_setup = 0, tc_trace = 0, tc_wait = 0;
                int tl_sample = 0;   // timeline mode: the first blocks log two time stamps per sample of their first shared tile
                render_pixel<true, false, true, !MAT, MAT>(p, x, ly, shade_n, state, rgb, lane_valid,
                             [&](V3, V3 d2, float& closest, int& hit_idx, auto&&... hit_branch_draws) {
                    ray_d[0][lane] = d2.x;
                    ray_d[1][lane] = d2.y;
                    ray_d[2][lane] = d2.z;
                    best_key[lane] = ~0ull;
                    long long tc0 = 0, tc1 = 0, tc2 = 0;
                    if (stats) { tc0 = (long long)clock64(); tc_setup += tc0 - tc_prev; }
                    __syncthreads();   // B1
                    (hit_branch_draws(), ...);   // list world, one Lambertian: on a copy of the stream, while the other waves trace
                    if (timeline && blockIdx.x < (unsigned)kStatsSampleBlocks && lane == 0 && tl_sample < kStatsSampleSlots) {
                        stats_in[kStatsSampleLog + (blockIdx.x * kStatsSampleSlots + tl_sample) * 2 + 0] = (unsigned long long)wall_clock64();   // draws done
                    }
                    if (stats) tc1 = (long long)clock64();
                    __syncthreads();   // B2
                    if (stats) { tc2 = (long long)clock64(); tc_trace += tc1 - tc0; tc_wait += tc2 - tc1; tc_prev = tc2; }
                    if (timeline && blockIdx.x < (unsigned)kStatsSampleBlocks && lane == 0 && tl_sample < kStatsSampleSlots) {
                        stats_in[kStatsSampleLog + (blockIdx.x * kStatsSampleSlots + tl_sample) * 2 + 1] = (unsigned long long)wall_clock64();   // slices done
                        ++tl_sample;
                    }
                    const unsigned long long kbest = best_key[lane];
                    hit_idx = (kbest == ~0ull) ? -1 : (int)(unsigned)(kbest & 0xffffffffull);
                }, mv);
                if (stats && lane == 0) {   // master-wave time split: (merge+shade+next setup), own trace, wait for the slowest slice
                    atomicAdd(&stats[32], (unsigned long long)tc_setup);
                    atomicAdd(&stats[33], (unsigned long long)tc_trace);
                    atomicAdd(&stats[34], (unsigned long long)tc_wait);
                    atomicAdd(&stats[35], 1ull);
                    atomicAdd(&stats[36], (unsigned long long)n_src);
                }
            } else {
                for (int smp = 0; smp < p.samples; ++smp) {
                    __syncthreads();   // B1
                    const V3 d2 = V3{ray_d[0][lane], ray_d[1][lane], ray_d[2][lane]};
                    float closest = FLT_MAX;
                    int hit_idx = -1;
                    trace_slice(d2, closest, hit_idx);
                    if (hit_idx >= 0) atomicMin(&best_key[lane], hit_key(closest, hit_idx));
                    __syncthreads();   // B2
                }
            }
            __syncthreads();   // partial buffers are reused by the next tile
            if (tid == 0) {
                unsigned nxt = 0xffffffffu;
                if (__hip_atomic_load(&w.counters[kCtrHeavyCursor], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < total_heavy)
                    nxt = atomicAdd(&w.counters[kCtrHeavyCursor], 1u);
                s_item = nxt;
            }
            __syncthreads();
            item = s_item;
            __syncthreads();
        }
    }
    const long long clk1 = stats ? (long long)clock64() : 0;
    const unsigned long long wall1 = timeline ? (unsigned long long)wall_clock64() : 0ull;

    // ---- light tiles: one wave per tile, lane = pixel
    // The light classes form one list, most survivors first; chunk k of kLightChunks owns the
    // items k, k + 64, k + 128, ... and has its own cursor (own cache line).  A wave works
    // through "its" chunk and then helps with whatever chunk still has tiles.
    // (bits 30..31 of split_packed: light tiles WITH geometry are dealt out as 1 << lsl row slabs too — for launches with far
    // more waves than such tiles, where one wave's 64-sample chain over a whole tile is the longest thing left; list world)
    const unsigned lsl = (BVH || FULL) ? 0u : (split_packed >> 30) & 3u;
    unsigned light_len[kLightClasses], n_light = 0;
#pragma unroll
    for (int c = 0; c < kLightClasses; ++c) {
        light_len[c] = w.counters[kCtrQueueLen + kHeavyClasses + c] << ((c < kLightClasses - 1) ? lsl : 0u);
        n_light += light_len[c];
    }
    // Items are dealt to the chunks in groups of kLightGroup consecutive queue positions: within a class the
    // queue follows the tile index, so a group is mostly a run of horizontally adjacent tiles, and the wave that
    // takes them one after the other writes their framebuffer rows back to back — 4 x 96 B of a row are three
    // full 128-B lines in the same L2 instead of four separately evicted partial ones.
    // (a launch with few tiles per CU — one rank's strips of a multi-GPU split — deals them ONE at a time, bits 27..29 of
    // split_packed: its waves far outnumber its tiles with geometry, and a wave that takes four neighbours with geometry
    // one after the other was the last to end, 250 us after waves that found only sky had gone: profiles/r3_tiled)
    const unsigned kLightGroup = FULL ? 4u : max(1u, (split_packed >> 27) & 7u);
    const unsigned n_groups = (n_light + kLightGroup - 1) / kLightGroup;
    int chunk = (int)((blockIdx.x * WAVES + wave) % kLightChunks);
    const int lane_k = lane;
    while (true) {
        // an opaque per-iteration copy of the lane id: LDS addresses formed from it cannot be hoisted out of the tile loop, where at
        // six waves per SIMD they were spilled to scratch (one v_mov per tile instead of a scratch reload per use)
        int lane = lane_k;
        asm volatile("" : "+v"(lane));
        if (CRT_TL_LIGHT && timeline && tl_light_t != 0ull && tl_light_end == 0ull) tl_light_end = (unsigned long long)wall_clock64();
        const unsigned c_len = ((n_groups + (unsigned)(kLightChunks - 1 - chunk)) / kLightChunks) * kLightGroup;   // positions of this chunk
        unsigned pos = 0xffffffffu;
        if (c_len != 0u) {
            if (lane == 0) pos = atomicAdd(&w.counters[kCtrLightCursor + chunk * kCtrLine], 1u);
            pos = __builtin_amdgcn_readfirstlane(pos);
        }
        if (pos >= c_len) {   // chunk exhausted (or empty): look at all cursors at once
            unsigned left = 0;
            if (lane < kLightChunks) {
                const unsigned len_l = ((n_groups + (unsigned)(kLightChunks - 1 - lane)) / kLightChunks) * kLightGroup;
                const unsigned cur = __hip_atomic_load(&w.counters[kCtrLightCursor + lane * kCtrLine], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                left = (cur < len_l) ? 1u : 0u;
            }
            const unsigned long long m = __builtin_amdgcn_ballot_w64(left != 0u);
            if (m == 0ull) break;
            const int r = chunk + 1;                                   // first candidate: the next chunk
            const unsigned long long rot = (r >= 64) ? m : ((m >> r) | (m << (64 - r)));
            chunk = (r + (int)__builtin_ctzll(rot)) & (kLightChunks - 1);
            continue;
        }
        unsigned item = ((pos / kLightGroup) * kLightChunks + (unsigned)chunk) * kLightGroup + pos % kLightGroup;
        if (item >= n_light) continue;                                 // the tail of the last group
        int lc = 0;
#pragma unroll
        for (int c = 0; c < kLightClasses - 1; ++c)
            if (lc == c && item >= light_len[c]) { item -= light_len[c]; lc = c + 1; }
        const unsigned isl = (lc < kLightClasses - 1) ? lsl : 0u;   // this item: slab `sub` of 1 << isl of its tile
        const int sub = (int)(item & ((1u << isl) - 1u)), sub_rows = 8 >> isl;
        const int tile_id = __builtin_amdgcn_readfirstlane((int)w.queue[(size_t)(kHeavyClasses + lc) * w.tiles_total + (item >> isl)]);
        const uint2 desc = w.tile_desc[tile_id];
        const int cnt = __builtin_amdgcn_readfirstlane((int)desc.y);
#if CRT_PRIO
        if (cnt != 0) __builtin_amdgcn_s_setprio(CRT_PRIO_GEOM); else __builtin_amdgcn_s_setprio(CRT_PRIO_SKY);
#endif
        if (CRT_TL_LIGHT && timeline && cnt != 0) {
            tl_light_t = (unsigned long long)wall_clock64();
            tl_light = ((tl_light + (1ull << 48)) & 0xffff000000000000ull) | ((unsigned long long)(unsigned)cnt << 32) | (unsigned)tile_id;
            tl_counts = 0ull;
            tl_light_end = 0ull;
        }
        const uint32_t* __restrict__ lst = w.pool + __builtin_amdgcn_readfirstlane(desc.x);
        const int x0 = (tile_id % w.tiles_x) * 8, ly0 = (tile_id / w.tiles_x) * 8;
        const int x = x0 + (lane & 7), ly = ly0 + (lane >> 3);
        // this item's pixels: on the image and inside its slab of rows (every other lane replays a pixel, unstored, without candidates)
        const bool lane_ok = x < p.nx && ly < p.row_count && (lane >> 3) >= sub * sub_rows && (lane >> 3) < (sub + 1) * sub_rows;
        // the whole list of a light tile fits one or two VGPRs (lane j holds positions j, 64+j)
        const int mine0 = (lane < cnt) ? (int)lst[lane] : 0;
        const int mine1 = (64 + lane < cnt) ? (int)lst[64 + lane] : 0;
        const int n0 = min(cnt, 64);
        Cand cand = {0u, 0u, 0u, 0u};
        if (BVH && desc.y != 0u) {
            const BvhTiles& bt = bvh;
            BvhView view;
            view.nodes = bt.nodes;
            view.n_nodes = bt.n_nodes;
#pragma unroll
            for (int k = 0; k < 9; ++k) view.rot[k] = bt.rot[k];
            const bool store = x < p.nx && ly < p.row_count;
            const int px = min(x, p.nx - 1), ply = min(ly, p.row_count - 1);   // lanes off the image replay a valid pixel, unstored
            const V3 o = V3{p.cam.origin[0], p.cam.origin[1], p.cam.origin[2]};
            // (a light tile never holds more than the heavy threshold, at most 64 entries — bin_alloc_kernel queues anything longer
            // as a shared tile — so one staging chunk serves it; rounds 1-2 carried two more paths here, for 65..256 and longer
            // lists, that no launch could reach and that cost the kernel registers)
            if (!VREC || desc.y == kNoList || cnt > kStage) {   // no list (pool overflow; or, never in a shipped launch, a light tile longer than the staging area): every sample walks the tree
                render_pixel<true, false, false, false, MAT>(p, px, ply, shade_n, state, rgb, store, [&](V3 d1, V3, float& closest, int& hit_idx) {
                    V3 used;
                    bvh_trace<true>(view, rec, o, d1, 0, closest, hit_idx, used);
                }, mv);
                continue;
            }
            stage_candidates<kStage>(lrec[wave], lqw[wave], lent[wave], p, cls, n_pad, rec, mine0, cnt, lane, px,
                                     global_row(p.row_begin, p.strip_rows, p.strip_step, ply), cand);
            // the root's and its children's boxes: proven for every ray of this tile?
            const unsigned top = __builtin_amdgcn_readfirstlane(
                tile_top_proven(bt, o, tile_cone(p, x0, global_row(p.row_begin, p.strip_rows, p.strip_step, ly0))));
            bool dead = false;
            if (lane < cnt) {
                // The tree's tie order instead of the slot — smaller wins (larger partner = later leaf, first in its leaf) —
                // with the entry's position in the low 6 bits, so the winner's staged data is one LDS read away.
                const int slot = mine0 & 0x7fffffff;
                lent[wave][lane] = (int)(((unsigned)mine0 & kCullSureBit) | ((unsigned)(0xffffff - bt.partner[slot]) << 6) | (unsigned)lane);
                lslot[wave][lane] = slot;
                stage_leaf(bt, slot, lflag[wave][lane], lbox0[wave][lane], lbox1[wave][lane]);
                dead = box_never_hit(lbox0[wave][lane], lbox1[wave][lane], o);
            }
            // entries whose leaf's box no ray from this origin passes (a flat box: an axis-aligned triangle, a ground plane) are no
            // candidates of any pixel: found as winners they were excluded and the walk repeated, sample after sample
            cand.lo &= ~(unsigned)__builtin_amdgcn_ballot_w64(dead);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (!store) cand.lo = cand.hi = 0u;
            unsigned bp = 0xffffffffu;
            if (WAVES <= 8) bp = build_pairs<kStage>(cand, lane, lpairs[wave], (unsigned)kPairCap);
            render_pixel<true, false, false, false, MAT>(p, px, ply, shade_n, state, rgb, store, [&](V3 d1, V3, float& closest, int& hit_idx) {
                // Only c3 (and d1) stay live across the walk: the chain's other elements are recomputed inside the rare branches
                // that need them (a lane off the hot chain; a winner whose chain of boxes needs node loads).
                V3 c3;
                const bool hot = hot_chain(d1, c3) && dbg_repeat_in != -77;   // -77: test hook, every lane takes the general path
                if (__builtin_amdgcn_ballot_w64(!hot) != 0ull) {
                    if (!hot) {   // off the hot chain (0.01 % of rays): this lane alone, with the literal chain element of every depth
                        const V3 c1 = rot0_exact(view.rot, d1);
                        const V3 c2 = chain_next<true>(view.rot, c1);
                        const V3 c3l = chain_next<true>(view.rot, c2);
                        const V3 c4 = chain_next<true>(view.rot, c3l);
                        const V3 c5 = rot0(view.rot, unit_vector(c4));
                        const int mode = same_bits(c4, c3l) ? 0 : (same_bits(c5, c4) ? 0 : (same_bits(c5, c3l) ? 1 : 2));
                        const unsigned masks[2] = {cand.lo, cand.hi};
                        unsigned gex[2] = {0u, 0u};
                        for (;;) {
                            int slot = -1;
                            const int q = general_winner(bt, rec, lst, cnt, masks, gex, 2, o, c1, c2, c3l, c4, mode, slot);
                            if (q < 0 || chain_ok_general(bt, slot, o, c1, c2, c3l, c4, mode)) { hit_idx = slot; break; }
                            for (int k = 0; k < cnt; ++k)
                                if ((int)(lst[k] & 0x7fffffffu) == slot) {
                                    if (k < 32) gex[0] |= 1u << k;
                                    else gex[1] |= 1u << (k - 32);
                                }
                        }
                    }
                }
                unsigned excl = 0u;
                for (bool first = true;; first = false) {
                    int q = -1;
                    if (first && bp != 0xffffffffu) {
                        lkey[wave][lane] = ~0ull;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        trace_pairs(lrec[wave], lent[wave], lpairs[wave], bp, lane, [&](int pix) {
                            return V3{__int_as_float(__builtin_amdgcn_ds_bpermute(pix << 2, __float_as_int(c3.x))),
                                      __int_as_float(__builtin_amdgcn_ds_bpermute(pix << 2, __float_as_int(c3.y))),
                                      __int_as_float(__builtin_amdgcn_ds_bpermute(pix << 2, __float_as_int(c3.z)))};
                        }, lkey[wave]);
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        const unsigned long long kb = lkey[wave][lane];
                        q = (kb == ~0ull) ? -1 : (int)(unsigned)(kb & 0xffffffffull);
                    } else {
                        Cand rest = cand;
                        rest.lo &= ~excl;
                        float tb = FLT_MAX;
                        trace_candidates<kStage>(lrec[wave], lent[wave], rest, c3, tb, q);
                    }
                    int slot = -1;
                    bool bad = false;
                    const int kw = q & 31;   // the winner's position in the staged list (kStage = 32)
                    if (q >= 0 && hot) {
                        // the winner's chain of boxes: its leaf's box with c3 (staged), and — where the tile's cone does not prove them,
                        // or the leaf sits right under the root — the root and its child on the leaf's side (node loads: rare)
                        slot = lslot[wave][kw];
#ifndef CRT_DBG_NOCHAIN   // (defined: timing experiment only — the chain of boxes is not tested, wrong pixels where it matters)
                        const int flag = lflag[wave][kw];
                        bool ok = (flag & 2) != 0 || leaf_box_hit(lbox0[wave][kw], lbox1[wave][kw], o, c3);
                        const bool rest = ok && chain_needs_top(flag, top);
                        if (__builtin_amdgcn_ballot_w64(rest) != 0ull && rest) ok = chain_top_ok(bt, slot, flag, o, d1, c3);
                        bad = !ok;
#endif
                    }
#ifdef CRT_TL_COUNTS   // diagnostic build: wave-samples of the wave's last geometry tile with an unreachable winner
                    if (CRT_TL_LIGHT && timeline) tl_counts += __builtin_amdgcn_ballot_w64(bad) != 0ull ? 1ull : 0ull;
#endif
                    if (__builtin_amdgcn_ballot_w64(bad) == 0ull) {
                        if (hot) hit_idx = slot;
                        closest = 0.0f;   // not used after the traversal
                        break;
                    }
                    if (bad) excl |= 1u << kw;   // this pixel's winner is not reachable through the tree: exclude it and look again
                }
            }, mv);
            continue;
        }
        // (from here on: the list world; a BVH-world tile that comes this far has no survivors — sky)
        if (!BVH && VREC && cnt > 0)   // all 64 lanes, before any lane leaves: lanes off the image classify a valid pixel
            stage_candidates<kStage>(lrec[wave], lqw[wave], lent[wave], p, cls, n_pad, rec, mine0, n0, lane, min(x, p.nx - 1),
                                     global_row(p.row_begin, p.strip_rows, p.strip_step, min(ly, p.row_count - 1)), cand);
        if (stats && VREC && cnt > 0) {   // diagnostics: per-pixel candidates vs the tile's list
            const unsigned pc = (unsigned)(__builtin_popcount(cand.lo) + __builtin_popcount(cand.hi));
            unsigned mx = pc, sm = pc;
#pragma unroll
            for (int ofs = 32; ofs > 0; ofs >>= 1) {
                mx = max(mx, (unsigned)__shfl_xor((int)mx, ofs, 64));
                sm += (unsigned)__shfl_xor((int)sm, ofs, 64);
            }
            if (lane == 0) {
                atomicAdd(&stats[37], 1ull);
                atomicAdd(&stats[38], (unsigned long long)cnt);
                atomicAdd(&stats[39], (unsigned long long)mx);
                atomicAdd(&stats[40], (unsigned long long)sm);
            }
        }
        // Balanced walk: the step count of trace_candidates is the LONGEST per-pixel list of the wave
        // while the mean is much shorter.  When the tile's (pixel, entry) pairs fit, they are laid out
        // once per tile, evenly over the lanes: every sample then takes ceil(pairs / 64) steps; a lane
        // fetches the ray of its pair's pixel with ds_bpermute and the results meet per pixel in LDS
        // as (ordered t, index) keys (atomicMin = the reference's accept rule).
        unsigned n_pairs = 0xffffffffu;
        if (!BVH && VREC && WAVES <= 8 && cnt > 0 && dbg_pairs != 0) {
            if (!lane_ok) cand.lo = cand.hi = 0u;   // lanes off the image replay a pixel: no pairs of their own
            n_pairs = build_pairs<kStage>(cand, lane, lpairs[wave], (unsigned)kPairCap);
        }
        const bool pair_mode = n_pairs != 0xffffffffu;
        if (stats && VREC && cnt > 0 && lane == 0) {   // diagnostics: did the pairs fit, how many
            if (pair_mode) atomicAdd(&stats[42], (unsigned long long)n_pairs);
            else atomicAdd(&stats[41], 1ull);
        }
        if (cnt == 0) {   // background only (most tiles of a typical frame)
            if (lane_ok) render_sky_pixel(p, x, ly, state, rgb);
        } else if (!BVH && pair_mode) {   // all 64 lanes stay active (ds_bpermute reads lanes); off-image lanes replay a valid pixel, unstored
            render_pixel<true, false, true, false, MAT>(p, min(x, p.nx - 1), min(ly, p.row_count - 1), shade_n, state, rgb, lane_ok,
                         [&](V3, V3 d2, float& closest, int& hit_idx) {
              for (int rep = 0; rep < dbg_repeat; ++rep) {
                lkey[wave][lane] = ~0ull;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                trace_pairs(lrec[wave], lent[wave], lpairs[wave], n_pairs, lane, [&](int pix) {   // the ray of another lane: ds_bpermute
                    return V3{__int_as_float(__builtin_amdgcn_ds_bpermute(pix << 2, __float_as_int(d2.x))),
                              __int_as_float(__builtin_amdgcn_ds_bpermute(pix << 2, __float_as_int(d2.y))),
                              __int_as_float(__builtin_amdgcn_ds_bpermute(pix << 2, __float_as_int(d2.z)))};
                }, lkey[wave]);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const unsigned long long kb = lkey[wave][lane];
                hit_idx = (kb == ~0ull) ? -1 : (int)(unsigned)(kb & 0xffffffffull);
                closest = 0.0f;   // not used after the traversal
              }
            }, mv);
        } else if (!BVH && lane_ok) {
            render_pixel<true, false, true, false, MAT>(p, x, ly, shade_n, state, rgb, true,
                         [&](V3, V3 d2, float& closest, int& hit_idx) {
              for (int rep = 0; rep < dbg_repeat; ++rep) {
                if (VREC) {
                    trace_candidates<kStage>(lrec[wave], lent[wave], cand, d2, closest, hit_idx);
                    walk<true>(rec, 64, cnt, [&](int k) { return __builtin_amdgcn_readlane(mine1, k - 64); }, d2, closest, hit_idx);
                } else {
                    walk<true>(rec, 0, min(cnt, 64), [&](int k) { return __builtin_amdgcn_readlane(mine0, k); }, d2, closest, hit_idx);
                    walk<true>(rec, 64, cnt, [&](int k) { return __builtin_amdgcn_readlane(mine1, k - 64); }, d2, closest, hit_idx);
                }
              }
            }, mv);
        }
    }
    if (stats && lane == 0) {   // cycle accounting (diagnostics): heavy phase per block, light phase per wave
        const long long clk2 = (long long)clock64();
        if (wave == 0) {
            atomicAdd(&stats[24], (unsigned long long)(clk1 - clk0));
            atomicMax(&stats[25], (unsigned long long)(clk1 - clk0));
        }
        atomicAdd(&stats[26], (unsigned long long)(clk2 - clk1));
        atomicMax(&stats[27], (unsigned long long)(clk2 - clk1));
        atomicMax(&stats[28], (unsigned long long)(clk2 - clk0));
    }
    if (timeline && lane == 0) {   // 100 MHz device-wide clock: when did this wave start, leave the heavy phase, end
        const unsigned long long wall2 = (unsigned long long)wall_clock64();   // plain stores to the wave's own slot: same-address
        const unsigned gw = blockIdx.x * WAVES + (unsigned)wave;              // atomics from 6 000 waves would cost more than the kernel
        if (gw < (unsigned)kStatsWaveSlots) {
            stats_in[kStatsWaveLog + 4 * gw + 0] = wall0;
            stats_in[kStatsWaveLog + 4 * gw + 1] = wall1;
            stats_in[kStatsWaveLog + 4 * gw + 2] = wall2;
            // the block's last shared tile: entries << 44 | tile << 20, and the wave's place (HW_REG_HW_ID: wave, SIMD, CU, SE)
            stats_in[kStatsWaveLog + 4 * gw + 3] = (tl_heavy << 20) | (unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xfffffu);
            if (CRT_TL_LIGHT) {
            stats_in[kStatsLightLog + 2 * gw + 0] = tl_light_t;
            stats_in[kStatsLightLog + 2 * gw + 1] = tl_light;
            // ... its duration in ticks (low 20 bits); CRT_TL_COUNTS builds: four 11-bit counts above
            stats_in[kStatsLightLog + 2 * kStatsWaveSlots + gw] = ((tl_light_end - tl_light_t) & 0xfffffull) | (tl_counts << 20);
            }
        }
    }
}

// Classification prepass: per triangle, the exact (fp64) affine coefficients of the four
// forms over d0(u,v) = (llc - o) + u*H + v*V, rounded to fp32, and the magnitudes M.
// FUSED (the per-frame prepass of the tiled variant, one launch): the exact record is derived from the
// base record right here (tri_prepass's arithmetic) and written out, and the launch also zeroes
// the binning scratch of this frame (tile counts and the counter block) — three launches less.
constexpr int kClsThreads = 64;
template <bool FUSED>
__global__ void __launch_bounds__(kClsThreads)
cls_prepass_kernel(const float* __restrict__ rec_in, const float* __restrict__ base, float* __restrict__ rec_out, int n, int n_pad,
                   RenderParams p, float4* __restrict__ cls, uint32_t* __restrict__ zero_a, int n_zero_a,
                   uint32_t* __restrict__ zero_b, int n_zero_b) {
    __shared__ double poly[2][2][10][kClsThreads];   // clip polygons: [ping-pong][x|y][vertex][lane], 20 KB
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (FUSED) {
        const int stride = (int)(gridDim.x * blockDim.x);
        for (int k = i; k < n_zero_a; k += stride) zero_a[k] = 0u;
        for (int k = i; k < n_zero_b; k += stride) zero_b[k] = 0u;
    }
    if (i >= n) return;
    const CameraPOD& cam = p.cam;
    float r[kRecFloats];
    if (FUSED) {
        float4 r4[4];
        derive_record(base + (size_t)i * kBaseFloats, cam.origin[0], cam.origin[1], cam.origin[2], r4);
        // (always stored: the overflow fallbacks — a tile without a list walks ALL triangles — and the tree walks read any record)
        float4* o = reinterpret_cast<float4*>(rec_out + (size_t)i * kRecFloats);
        o[0] = r4[0]; o[1] = r4[1]; o[2] = r4[2]; o[3] = r4[3];
        r[0] = r4[0].x; r[1] = r4[0].y; r[2] = r4[0].z; r[3] = r4[0].w; r[4] = r4[1].x; r[5] = r4[1].y; r[6] = r4[1].z; r[7] = r4[1].w;
        r[8] = r4[2].x; r[9] = r4[2].y; r[10] = r4[2].z; r[11] = r4[2].w; r[12] = r4[3].x; r[13] = r4[3].y; r[14] = r4[3].z; r[15] = r4[3].w;
    } else {
        const float4* ri = reinterpret_cast<const float4*>(rec_in + (size_t)i * kRecFloats);
        const float4 a = ri[0], b4 = ri[1], c4 = ri[2], d4 = ri[3];
        r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = b4.x; r[5] = b4.y; r[6] = b4.z; r[7] = b4.w;
        r[8] = c4.x; r[9] = c4.y; r[10] = c4.z; r[11] = c4.w; r[12] = d4.x; r[13] = d4.y; r[14] = d4.z; r[15] = d4.w;
    }
    const double nx = r[0], ny = r[1], nz = r[2];
    const double e2x = r[3], e2y = r[4], e2z = r[5];
    const double e1x = r[6], e1y = r[7], e1z = r[8];
    const double sx = r[9], sy = r[10], sz = r[11];
    const double qx = r[12], qy = r[13], qz = r[14];
    // a = e1.(d x e2) = d.(e2 x e1);  su = s.(d x e2) = d.(e2 x s);  dq = d.q
    const double gax = e2y * e1z - e2z * e1y, gay = e2z * e1x - e2x * e1z, gaz = e2x * e1y - e2y * e1x;
    const double gsx = e2y * sz - e2z * sy, gsy = e2z * sx - e2x * sz, gsz = e2x * sy - e2y * sx;
    const double p0x = (double)cam.llc[0] - (double)cam.origin[0];
    const double p0y = (double)cam.llc[1] - (double)cam.origin[1];
    const double p0z = (double)cam.llc[2] - (double)cam.origin[2];
    const double hx = cam.horizontal[0], hy = cam.horizontal[1], hz = cam.horizontal[2];
    const double vx = cam.vertical[0], vy = cam.vertical[1], vz = cam.vertical[2];
    double FA[5], FB[5], FC[5];   // fp64 coefficients of dn, a, su, dq, w
    auto form = [&](int k, double gx, double gy, double gz) {
        FA[k] = p0x * gx + p0y * gy + p0z * gz;
        FB[k] = hx * gx + hy * gy + hz * gz;
        FC[k] = vx * gx + vy * gy + vz * gz;
    };
    form(0, nx, ny, nz);
    form(1, gax, gay, gaz);
    form(2, gsx, gsy, gsz);
    form(3, qx, qy, qz);
    const float n1 = fabsf(r[0]) + fabsf(r[1]) + fabsf(r[2]);
    const float e11 = fabsf(r[6]) + fabsf(r[7]) + fabsf(r[8]);
    const float e21 = fabsf(r[3]) + fabsf(r[4]) + fabsf(r[5]);
    const float s1 = fabsf(r[9]) + fabsf(r[10]) + fabsf(r[11]);
    const float q1 = fabsf(r[12]) + fabsf(r[13]) + fabsf(r[14]);
    const float Ma = e11 * e21 * 1.001f;
    float M[5] = {n1 * 1.001f, Ma, s1 * e21 * 1.001f + Ma * 9.5367431640625e-7f, q1 * 1.001f + Ma * 9.5367431640625e-7f, 0.0f};
    float4 rec4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) rec4[k] = make_float4((float)FA[k], (float)FB[k], (float)FC[k], M[k]);
    // w exactly as the binning kernel forms it (from the rounded records), so that both tests
    // bound the same function
    FA[4] = (double)(rec4[1].x - rec4[2].x - rec4[3].x);
    FB[4] = (double)(rec4[1].y - rec4[2].y - rec4[3].y);
    FC[4] = (double)(rec4[1].z - rec4[2].z - rec4[3].z);
    M[4] = M[2] + M[3] + 1.25f * M[1];

    // (Round 4, measured and dropped: for a rank of a multi-GPU split, the binning's cell test on each full-width strip before the
    // clipping — every edge line of a triangle crosses a 3 840-pixel-wide cell somewhere, so interval bounds per form exclude
    // nothing; and clipping in a second kernel over a compacted list — 37 + 57 us against this kernel's 82 us at C5.)
    // Bounding box in (u,v) of every model position whose ray may be accepted: the screen
    // rectangle (widened by the position slack) clipped by the relaxed predicates
    //   dn >= -S_dn  and  [ a, su, dq, w >= -S   or   a, su, dq, w <= +S ],   S_X = K * M_X,
    // each a half-plane of the (u,v) plane.  fp64 Sutherland-Hodgman; the slacks are taken 1 %
    // larger and the box 2e-6 wider than computed, which covers the fp64 rounding by far.
    bool finite_all = true;
#pragma unroll
    for (int k = 0; k < 5; ++k) finite_all = finite_all && isfinite(FA[k]) && isfinite(FB[k]) && isfinite(FC[k]) && isfinite(M[k]);
    const double pad_u = (double)p.pos_du + 1.0e-6, pad_v = (double)p.pos_dv + 1.0e-6;
    double bu_lo = 1.0e30, bu_hi = -1.0e30, bv_lo = 1.0e30, bv_hi = -1.0e30;
    if (!finite_all) {
        bu_lo = -pad_u; bu_hi = 1.0 + pad_u; bv_lo = -pad_v; bv_hi = 1.0 + pad_v;
    } else {
        // corner values decide most triangles at once: a half-plane that excludes all four corners of
        // the (convex) screen rectangle excludes it entirely
        auto corners_below = [&](double sgn, int k) {   // sgn * F_k + S < 0 at every corner
            const double S = (double)p.K_slack * (double)M[k] * 1.01;
            const double c0 = sgn * FA[k] + S, cb = sgn * FB[k], cc = sgn * FC[k];
            const double u0 = -pad_u, u1 = 1.0 + pad_u, v0 = -pad_v, v1 = 1.0 + pad_v;
            return (c0 + u0 * cb + v0 * cc < 0.0) && (c0 + u1 * cb + v0 * cc < 0.0) &&
                   (c0 + u0 * cb + v1 * cc < 0.0) && (c0 + u1 * cb + v1 * cc < 0.0);
        };
        const bool culled = corners_below(1.0, 0);                     // dn < -S everywhere: back-facing on the whole screen
        for (int side = 0; side < 2 && !culled; ++side) {
            if (corners_below(side == 0 ? 1.0 : -1.0, 1)) continue;   // a has the other sign everywhere: this side is empty
            // polygon vertices live in LDS (vertex-major, lane-minor: conflict-free); in private arrays
            // the dynamic indexing would put them in scratch memory, ten times the latency
            int cur = 0, np = 4;
#define PX(b, j) poly[b][0][j][threadIdx.x]
#define PY(b, j) poly[b][1][j][threadIdx.x]
            PX(0, 0) = -pad_u;      PY(0, 0) = -pad_v;
            PX(0, 1) = 1.0 + pad_u; PY(0, 1) = -pad_v;
            PX(0, 2) = 1.0 + pad_u; PY(0, 2) = 1.0 + pad_v;
            PX(0, 3) = -pad_u;      PY(0, 3) = 1.0 + pad_v;
            for (int k = 0; k < 5 && np > 0; ++k) {
                const double sgn = (k == 0 || side == 0) ? 1.0 : -1.0;       // dn keeps its sign on both sides
                const double S = (double)p.K_slack * (double)M[k] * 1.01;
                const double ca = sgn * FA[k] + S, cb = sgn * FB[k], cc = sgn * FC[k];   // keep ca + u cb + v cc >= 0
                int m = 0;
                double xp = PX(cur, np - 1), yp = PY(cur, np - 1), fp = ca + xp * cb + yp * cc;
                for (int j = 0; j < np; ++j) {
                    const double xc = PX(cur, j), yc = PY(cur, j), fc = ca + xc * cb + yc * cc;
                    const bool in_c = fc >= 0.0, in_p = fp >= 0.0;
                    if (in_c != in_p) {
                        const double t = fp / (fp - fc);
                        PX(cur ^ 1, m) = xp + t * (xc - xp); PY(cur ^ 1, m) = yp + t * (yc - yp); ++m;
                    }
                    if (in_c) { PX(cur ^ 1, m) = xc; PY(cur ^ 1, m) = yc; ++m; }
                    xp = xc; yp = yc; fp = fc;
                }
                np = m;
                cur ^= 1;
            }
            for (int j = 0; j < np; ++j) {
                bu_lo = fmin(bu_lo, PX(cur, j)); bu_hi = fmax(bu_hi, PX(cur, j));
                bv_lo = fmin(bv_lo, PY(cur, j)); bv_hi = fmax(bv_hi, PY(cur, j));
            }
#undef PX
#undef PY
        }
    }
    const bool empty = bu_lo > bu_hi;
    cls[(size_t)4 * n_pad + i] = empty ? make_float4(1.0e30f, -1.0e30f, 1.0e30f, -1.0e30f)
                                       : make_float4((float)bu_lo - 2.0e-6f, (float)bu_hi + 2.0e-6f,
                                                     (float)bv_lo - 2.0e-6f, (float)bv_hi + 2.0e-6f);
    if (empty) return;   // back-facing or off the screen for every ray: the binning selects by the box first, nothing reads its forms
#pragma unroll
    for (int k = 0; k < 4; ++k) cls[(size_t)k * n_pad + i] = rec4[k];
}

hipError_t launch_cls_prepass(const float* d_rec, int n_tris, int n_pad, const RenderParams& p,
                              float4* d_cls, hipStream_t stream) {
    if (n_tris <= 0) return hipSuccess;
    dim3 block(kClsThreads), grid((unsigned)((n_tris + kClsThreads - 1) / kClsThreads));
    hipLaunchKernelGGL(cls_prepass_kernel<false>, grid, block, 0, stream, d_rec, (const float*)nullptr, (float*)nullptr, n_tris, n_pad, p,
                       d_cls, (uint32_t*)nullptr, 0, (uint32_t*)nullptr, 0);
    return hipGetLastError();
}

hipError_t launch_frame_prepass(const float* d_base, int n_tris, int n_pad, const RenderParams& p, float* d_rec, float4* d_cls,
                                const TiledWork& w, hipStream_t stream) {
    // enough blocks for the triangles and for zeroing the scratch in at most 16 words per thread
    const long long zero_words = (long long)w.tiles_total + kCtrWords;
    const unsigned blocks = (unsigned)std::max<long long>((n_tris + kClsThreads - 1) / kClsThreads, (zero_words + kClsThreads * 16 - 1) / (kClsThreads * 16));
    hipLaunchKernelGGL(cls_prepass_kernel<true>, dim3(std::max(blocks, 1u)), dim3(kClsThreads), 0, stream, (const float*)nullptr, d_base, d_rec,
                       n_tris, n_pad, p, d_cls, w.tile_count, w.tiles_total, w.counters, kCtrWords);
    return hipGetLastError();
}

hipError_t launch_render(const RenderParams& p, const float* d_rec, const float4* d_cls, int n_pad,
                         const float* d_shade_n, uint32_t* d_state, float* d_rgb, int variant,
                         const TiledWork* work, int n_cus, bool fast_rcp, unsigned long long* d_stats,
                         hipEvent_t ev_mid, LaunchCache* cache, bool prezeroed, hipStream_t stream, const BvhTiles* d_bvh_tiles,
                         bool bvh_force_general, const MatView* materials) {
    if (p.row_count <= 0 || p.nx <= 0) return hipSuccess;
    if (variant == 0 || d_cls == nullptr || work == nullptr) {
        dim3 block(256);
        dim3 grid((unsigned)((p.nx + 15) / 16), (unsigned)((p.row_count + 15) / 16));
        if (fast_rcp)
            hipLaunchKernelGGL(render_list_kernel<true>, grid, block, 0, stream, p, d_rec, d_shade_n, d_state, d_rgb);
        else
            hipLaunchKernelGGL(render_list_kernel<false>, grid, block, 0, stream, p, d_rec, d_shade_n, d_state, d_rgb);
        return hipGetLastError();
    }
    TiledWork w = *work;
    if (variant == 2) w.pool_cap = 4096;                 // test hook: force pool-full fallbacks
    if (variant == 4) w.log_cap = 16 * kLogSegments;   // (16 records per segment)                  // test hook: the binning log overflows, every tile walks the whole list
    // Blocks of 8 waves share a heavy tile (16-wave blocks have no LDS left for the balanced walk and
    // are slower at every launch size since it exists; kept for A/B through CRT_SHADE_GROUP).
    const int tiles_per_cu = w.tiles_total / std::max(n_cus, 1);
    int dbg_repeat = 1, group = 8;
    int occ = (group == 8) ? 6 : 8;   // measured optimum with the LDS-staged light tiles
#ifdef CRT_TUNING   // tuning builds only (make tuning): the shipped library reads no environment
    if (const char* e = getenv("CRT_SHADE_OCC")) occ = atoi(e);
    if (const char* e = getenv("CRT_SHADE_GROUP")) group = atoi(e);              // waves that share a heavy tile
    if (const char* e = getenv("CRT_DEBUG_WALK_REPEAT")) { dbg_repeat = atoi(e); if (dbg_repeat == 0) dbg_repeat = 1; }   // timing experiments only
#endif
    // tiles with more survivors than this are shared by all waves of a block (measured: a full frame, and one
    // rank's strips of a 2-, 4-, 8-way split on the bench scene)
    int heavy = (variant == 2 || variant == 5) ? 3 : (tiles_per_cu >= 96 ? kHeavyMin : (tiles_per_cu >= 24 ? kHeavyMinMedium : kHeavyMinSmall));
    bool vrec = variant != 3;   // light tiles: records staged in LDS (VGPR operands); variant 3 keeps the scalar-operand walk (A/B)
#ifdef CRT_TUNING
    if (const char* e = getenv("CRT_HEAVY_MIN")) heavy = std::max(1, std::min(atoi(e), 64));
    if (const char* e = getenv("CRT_SHADE_VREC")) vrec = atoi(e) != 0;
#endif
    using ShadeFn = void (*)(RenderParams, const float*, const float4*, int, const float*, uint32_t*, float*, TiledWork, unsigned long long*, int,
                             BvhTiles, unsigned, MatView);
    // log2 of the number of row slabs a heavy tile of each weight class (> 768, 512, 384, 256, 192, 128, 96, 64, heavy) is dealt
    // out as (shade_tiles_kernel: sub-items).  Measured (profiles/r3_tiled/sweep_split_*.txt): splitting the heaviest tiles
    // x2 .. x8 changes neither the full frame nor the slowest rank of an 8-way split, on either C3 mesh — what bounds a
    // small launch is the 64-sample chain of EVERY tile that holds geometry (about 2 us per sample whatever the list's
    // length), not the longest list; splitting the common classes only replicates the master wave's work (+20 .. +90 %).
    // So the default is no split — except for lists of more than 768 entries (the two ~1 980-entry pole tiles of the C5 frame end
    // its launch alone: 4.83 -> 4.06 ms with four slabs each, profiles/r3_tiled/c5_split.txt; round 4: with their slices as far pairs 0.3 ms).
    // The mechanism stays for the tuning build (CRT_SPLIT) and the test hook (variant 5).
    int split[kHeavyClasses] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    // (BVH mode: eight slabs of one row — its staging areas hold 32 entries, so the far pairs of a slice of a quarter tile would not fit)
    split[0] = d_bvh_tiles == nullptr ? 2 : 3;
    // ... and for launches with few tiles per CU (one rank's strips of an 8-way split), together with two slabs per light tile
    // that holds geometry (light_split below): slowest rank 0.309 -> 0.280 ms on C3, 0.407 -> 0.374 ms on C3B
    // (profiles/r3_tiled/sweep_light_split_*.txt) — the machine is idle enough there for the replicated ray set-up not to matter
    if (d_bvh_tiles == nullptr && tiles_per_cu < 24) { split[1] = split[2] = split[3] = split[4] = split[5] = 2; split[6] = 1; }
    if (variant == 5)   // test hook (crt_set_kernel_variant(10)): shared tiles from 4 entries on, dealt out as 2, 4 or 8 row slabs by class
        for (int c = 0; c < kHeavyClasses; ++c) split[c] = 1 + c % 3;
#ifdef CRT_TUNING
    if (const char* e = getenv("CRT_SPLIT")) {   // nine digits, e.g. CRT_SPLIT=222222100
        for (int c = 0; c < kHeavyClasses && e[c] >= '0' && e[c] <= '3'; ++c) split[c] = e[c] - '0';
    }
#endif
    unsigned split_packed = 0;
    for (int c = 0; c < kHeavyClasses; ++c) split_packed |= (unsigned)split[c] << (3 * c);
    // light tiles are dealt to the waves in groups of adjacent tiles (framebuffer rows written back to back: +3 % on a
    // full frame) — but one at a time when the launch is small (see shade_tiles_kernel)
    unsigned light_group = (tiles_per_cu >= 96 && variant != 5) ? 4u : 1u;
#ifdef CRT_TUNING
    if (const char* e = getenv("CRT_LIGHT_GROUP")) light_group = (unsigned)std::max(1, std::min(atoi(e), 7));
#endif
    split_packed |= light_group << 27;
    // light tiles with geometry as row slabs (bits 30..31): small launches only, list world (shade_tiles_kernel)
    unsigned light_split = (d_bvh_tiles == nullptr && tiles_per_cu < 24) ? 1u : 0u;
    if (variant == 5) light_split = 2u;   // test hook (crt_set_kernel_variant(10))
#ifdef CRT_TUNING
    if (const char* e = getenv("CRT_LIGHT_SPLIT")) light_split = (unsigned)std::max(0, std::min(atoi(e), 3));
#endif
    if (d_bvh_tiles != nullptr) light_split = 0u;
    split_packed |= light_split << 30;
    ShadeFn fn;
#ifdef CRT_TUNING   // block sizes and occupancies no shipped launch selects exist in tuning builds only (they were 7 of the 12
                    // instantiations of the library's largest kernel)
    if (group == 4) {   // experiment (CRT_SHADE_GROUP=4): half as many waves parked per shared tile, more blocks free for light tiles at once
        occ = 6;
        fn = shade_tiles_kernel<true, 6, 4>;
    } else if (group == 8) {
        occ = (occ == 4) ? 4 : 6;   // 8 waves per SIMD do not fit the LDS of the staged variant
        fn = vrec ? (occ == 6 ? shade_tiles_kernel<true, 6, 8> : shade_tiles_kernel<true, 4, 8>)
                  : (occ == 6 ? shade_tiles_kernel<false, 6, 8> : shade_tiles_kernel<false, 4, 8>);
    } else {
        group = 16;
        occ = (occ == 4) ? 4 : 8;
        fn = vrec ? (occ == 4 ? shade_tiles_kernel<true, 4, 16> : shade_tiles_kernel<true, 8, 16>)
                  : (occ == 4 ? shade_tiles_kernel<false, 4, 16> : shade_tiles_kernel<false, 8, 16>);
    }
#else
    fn = vrec ? shade_tiles_kernel<true, 6, 8> : shade_tiles_kernel<false, 6, 8>;
    (void)occ;
#endif
    // the full-frame policy as compile-time constants (shade_tiles_kernel<.., FULL>) when this launch's policy is exactly that
    unsigned full_packed = (4u << 27) | (d_bvh_tiles == nullptr ? 2u : 3u);
    const bool full = split_packed == full_packed && group == 8;
    if (full && d_bvh_tiles == nullptr && materials == nullptr && vrec) fn = shade_tiles_kernel<true, 6, 8, false, false, true>;
    if (d_bvh_tiles) {   // the tree as the world (shade_tiles_kernel<.., BVH = true>)
        group = 8;
        heavy = std::min(heavy, 32);   // its staging area holds 32 entries: a light tile must fit
#ifndef CRT_BVH_OCC
#define CRT_BVH_OCC 4   // round 4: the BVH mode stages 32 entries per wavefront (block 45 KB of LDS: three blocks per CU would fit) but at the 80 VGPRs of
                        // six waves per SIMD it spills into its sample loops: 128 VGPRs = two blocks per CU is 6 % faster on C3, equal on C5 (profiles/r4/bvh_occ_ab2.txt).
                        // (Rounds 2-3: 76 KB = two blocks = 4 waves/SIMD at 128 VGPRs; asked for 6 with that LDS the compiler gave up on
                        // the target and took 220 VGPRs = 2 waves/SIMD: sweep_bvh.txt)
#endif
        fn = materials ? shade_tiles_kernel<true, CRT_BVH_OCC, 8, true, true>
                       : (full ? shade_tiles_kernel<true, CRT_BVH_OCC, 8, true, false, true> : shade_tiles_kernel<true, CRT_BVH_OCC, 8, true>);
        if (bvh_force_general) dbg_repeat = -77;
    } else if (materials) {   // LambertShade with a material per triangle: the staged 8-wave kernel
        group = 8;
        fn = shade_tiles_kernel<true, 6, 8, false, true>;
    }
    // persistent blocks: exactly as many as are resident at once (block b starts with heavy item b);
    // the occupancy query is cached per launch context (device + stream), not per host thread
    hipError_t e = hipSuccess;
    if (cache->fn != (const void*)fn) {
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&cache->resident, fn, group * 64, 0);
        if (e != hipSuccess) return e;
        cache->fn = (const void*)fn;
    }
    const int resident = cache->resident;
    int n_blocks = n_cus * std::max(resident, 1);
#ifdef CRT_TUNING
    if (const char* ev = getenv("CRT_SHADE_GRID")) n_blocks = std::max(1, atoi(ev));
#endif
    n_blocks = std::max(1, std::min(n_blocks, w.tiles_total));
    e = launch_binning(p, d_cls, n_pad, w, heavy, (unsigned)n_blocks, d_stats, prezeroed, stream);
    if (e != hipSuccess) return e;
    if (ev_mid) {
        e = hipEventRecord(ev_mid, stream);   // splits the timing: [start, mid) = binning, [mid, stop) = shading
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(fn, dim3((unsigned)n_blocks), dim3((unsigned)group * 64), 0, stream, p, d_rec, d_cls, n_pad, d_shade_n,
                       d_state, d_rgb, w, d_stats, dbg_repeat, d_bvh_tiles ? *d_bvh_tiles : BvhTiles{}, split_packed, materials ? *materials : MatView{});
    return hipGetLastError();
}

// Audit of the binning (diagnostics/tests): brute force over ALL triangles for the rays of every
// 8x8 tile; every (ray, triangle) pair that passes Triangle::collision_detection — closest or
// not — must find its triangle in the tile's survivor list, and a "cull-sure" entry must never
// see dn < 0.  counters: [0] passing pairs, [1] pairs missing from the list, [2] cull-sure
// violations, [3] first offender (tile << 32 | triangle).
__global__ void __launch_bounds__(256)
audit_tiles_kernel(RenderParams p, const float* __restrict__ rec, const float4* __restrict__ cls, int n_pad,
                   const uint32_t* __restrict__ state, TiledWork w, unsigned long long* __restrict__ counters) {
    __shared__ volatile uint32_t member[4][64], sure_of[4][64];   // per wavefront: 2048-bit windows of the tile's list
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int tile_x = blockIdx.x * 2 + (wave & 1), tile_y = blockIdx.y * 2 + (wave >> 1);
    if (tile_x >= w.tiles_x || tile_y >= w.tiles_y) return;
    const int tile_id = tile_y * w.tiles_x + tile_x;
    const int xr = tile_x * 8 + (lane & 7), lyr = tile_y * 8 + (lane >> 3);
    const bool valid = xr < p.nx && lyr < p.row_count;
    const int x = valid ? xr : tile_x * 8, ly = valid ? lyr : tile_y * 8;   // off-image lanes replay pixel 0
    const int y = global_row(p.row_begin, p.strip_rows, p.strip_step, ly);
    const uint2 desc = w.tile_desc[tile_id];
    if (desc.y == kNoList) return;                    // pool was full: the tile walks every triangle anyway
    const uint32_t* __restrict__ lst = w.pool + desc.x;
    const int cnt = (int)desc.y;

    const long long si = (long long)ly * p.nx + x, stride = p.rng_count;
    Xorwow rs;
    rs.d = state[0 * stride + si]; rs.v0 = state[1 * stride + si]; rs.v1 = state[2 * stride + si];
    rs.v2 = state[3 * stride + si]; rs.v3 = state[4 * stride + si]; rs.v4 = state[5 * stride + si];
    const float fnx = (float)p.nx, fny = (float)p.ny;
    unsigned long long n_pass = 0, n_miss = 0, n_cull = 0;
    for (int smp = 0; smp < p.samples; ++smp) {
        // any jitter inside the pixel is a legitimate ray of the tile; the extreme corners are
        // included every fourth sample to stress the rectangle bounds
        float ju = xw_uniform(rs), jv = xw_uniform(rs);
        if ((smp & 3) == 3) { ju = (smp & 4) ? 1.0f : 1.1641532e-10f; jv = (smp & 8) ? 1.0f : 1.1641532e-10f; }
        const float u = ((float)x + ju) / fnx, v = ((float)y + jv) / fny;
        const V3 d2 = unit_vector(unit_vector(camera_direction(p.cam, u, v)));
        // lists are unordered: membership through a bitmap of 2048 triangle indices at a time
        for (int c0 = 0; c0 < p.n_tris; c0 += 2048) {
            member[wave][lane] = 0u;
            sure_of[wave][lane] = 0u;
            __builtin_amdgcn_wave_barrier();
            for (int k = lane; k < cnt; k += 64) {
                const uint32_t e = lst[k];
                const int rel = (int)(e & 0x7fffffffu) - c0;
                if (rel >= 0 && rel < 2048) {
                    atomicOr((unsigned*)&member[wave][rel >> 5], 1u << (rel & 31));
                    if (e & kCullSureBit) atomicOr((unsigned*)&sure_of[wave][rel >> 5], 1u << (rel & 31));
                }
            }
            __builtin_amdgcn_wave_barrier();
            const int c1 = min(c0 + 2048, p.n_tris);
            for (int i = c0; i < c1; ++i) {
                const Rec16 rc = load_rec(rec, i);
                const float dn = d2.x * rc.f[0] + d2.y * rc.f[1] + d2.z * rc.f[2];
                const float hx = (d2.y * rc.f[5] - d2.z * rc.f[4]);
                const float hy = (-(d2.x * rc.f[5] - d2.z * rc.f[3]));
                const float hz = (d2.x * rc.f[4] - d2.y * rc.f[3]);
                const float a = rc.f[6] * hx + rc.f[7] * hy + rc.f[8] * hz;
                const float f = 1.0f / a;
                const float uu = f * (rc.f[9] * hx + rc.f[10] * hy + rc.f[11] * hz);
                const float vv = f * (d2.x * rc.f[12] + d2.y * rc.f[13] + d2.z * rc.f[14]);
                const bool pass = !(dn < 0.0f) && !(a > -kTriEpsilon && a < kTriEpsilon) && !(uu < 0.0f || uu > 1.0f) &&
                                  !(vv < 0.0f || uu + vv > 1.0f);
                const int rel = i - c0;
                const bool in_list = ((member[wave][rel >> 5] >> (rel & 31)) & 1u) != 0u;
                const bool sure = ((sure_of[wave][rel >> 5] >> (rel & 31)) & 1u) != 0u;
                const unsigned long long mp = __builtin_amdgcn_ballot_w64(pass);
                n_pass += (unsigned long long)__builtin_popcountll(mp);
                if (mp != 0ull && cls != nullptr) {   // the shading kernel's per-pixel masks must keep the pair too
                    float4 f0 = cls[(size_t)0 * n_pad + i], f1 = cls[(size_t)1 * n_pad + i];
                    float4 f2 = cls[(size_t)2 * n_pad + i], f3 = cls[(size_t)3 * n_pad + i];
                    const float q4 = pixel_forms(p, f0, f1, f2, f3);
                    bool sure_px;
                    const bool may_px = pixel_may_hit(p, f0, f1, f2, f3, q4, x, y, sure_px);
                    const unsigned long long miss_px = __builtin_amdgcn_ballot_w64(pass && !may_px);
                    const unsigned long long cull_px = __builtin_amdgcn_ballot_w64(sure_px && dn < 0.0f);
                    n_miss += (unsigned long long)__builtin_popcountll(miss_px);
                    n_cull += (unsigned long long)__builtin_popcountll(cull_px);
                    if (miss_px != 0ull && lane == 0) atomicMax(&counters[3], ((unsigned long long)tile_id << 32) | (unsigned long long)i);
                }
                if (mp != 0ull && !in_list) {
                    n_miss += (unsigned long long)__builtin_popcountll(mp);
                    if (lane == 0) atomicMax(&counters[3], ((unsigned long long)tile_id << 32) | (unsigned long long)i);
                }
                if (sure) n_cull += (unsigned long long)__builtin_popcountll(__builtin_amdgcn_ballot_w64(dn < 0.0f));
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (lane == 0) {
        atomicAdd(&counters[0], n_pass);
        atomicAdd(&counters[1], n_miss);
        atomicAdd(&counters[2], n_cull);
    }
}

hipError_t launch_audit(const RenderParams& p, const float* d_rec, const float4* d_cls, int n_pad,
                        const uint32_t* d_state, const TiledWork* work, unsigned long long* d_counters,
                        hipStream_t stream) {
    TiledWork w = *work;
    hipError_t e = launch_binning(p, d_cls, n_pad, w, kHeavyMin, 0u, nullptr, false, stream);
    if (e != hipSuccess) return e;
    dim3 grid((unsigned)((w.tiles_x + 1) / 2), (unsigned)((w.tiles_y + 1) / 2));
    hipLaunchKernelGGL(audit_tiles_kernel, grid, dim3(256), 0, stream, p, d_rec, d_cls, n_pad, d_state, w, d_counters);
    return hipGetLastError();
}

// Self-check of reciprocal<true>: every float with biased exponent in [exp_lo, exp_hi], both
// signs, against the compiler's IEEE 1.0f / a.
__global__ void __launch_bounds__(256)
check_reciprocal_kernel(uint32_t exp_lo, uint32_t exp_hi, unsigned long long* __restrict__ bad) {
    unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long total = (unsigned long long)(exp_hi - exp_lo + 1) << 24;
    unsigned long long mine = 0;
    for (; i < total; i += (unsigned long long)gridDim.x * blockDim.x) {
        const uint32_t bits = ((uint32_t)(i & 1) << 31) | ((exp_lo + (uint32_t)(i >> 24)) << 23) | (uint32_t)((i >> 1) & 0x7fffff);
        const float a = __uint_as_float(bits);
        mine += (__float_as_uint(reciprocal<true>(a)) != __float_as_uint(1.0f / a)) ? 1ull : 0ull;
    }
    if (mine) atomicAdd(bad, mine);
}

// Self-checks of div_by_rcp (crt_debug_check_division).
// mode 1: exhaustive over significands — divisors b = 1.mb for mb in chunks [lo, hi] of 64
//         (2^17 divisors per chunk), numerators x = 1.mx for ALL 2^23 mx: 1.1e12 pairs per chunk,
//         7.04e13 for chunks 0..63.  Quotients cover both result binades.
// mode 2: every float divisor of the biased-exponent range [lo, hi], 64 pseudo-random numerators
//         each with exponents 0..60 below the divisor's and random signs (scaling / sign paths).
// mode 3: sqrt_normal against sqrtf for every positive float of the biased-exponent range.
// mode 4: the fused forms of xw_uniform / xw_uniform_2m1 against curand_uniform's literal product-then-sum
//         (x * 2^-32 + 2^-33, and 2.0f * that - 1.0f), for ALL 2^32 values of a draw.
__global__ void __launch_bounds__(256)
check_division_kernel(int mode, uint32_t lo, uint32_t hi, unsigned long long* __restrict__ bad) {
    unsigned long long mine = 0;
    if (mode == 4) {
        unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
        for (; i < (1ull << 32); i += (unsigned long long)gridDim.x * blockDim.x) {
            const uint32_t x = (uint32_t)i;
            const float k = 2.3283064e-10f;                                 // CURAND_2POW32_INV
            float prod = (float)x * k;
            asm volatile("" : "+v"(prod));                                  // the product as a value of its own: no contraction, whatever the flags
            const float u = prod + (k / 2.0f);                              // _curand_uniform
            const float m = 2.0f * u - 1.0f;                                // material.h:14, camera.h:9
            const float uf = __builtin_fmaf((float)x, k, k / 2.0f);         // xw_uniform
            const float k2 = 4.6566129e-10f;
            const float mf = __builtin_fmaf((float)x, k2, k2 / 2.0f) - 1.0f;   // xw_uniform_2m1
            mine += (__float_as_uint(u) != __float_as_uint(uf)) ? 1ull : 0ull;
            mine += (__float_as_uint(m) != __float_as_uint(mf)) ? 1ull : 0ull;
        }
    } else if (mode == 3) {   // sqrt_normal == sqrtf for every float with biased exponent in [lo, hi]
        unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
        const unsigned long long total = (unsigned long long)(hi - lo + 1) << 23;
        for (; i < total; i += (unsigned long long)gridDim.x * blockDim.x) {
            const float x = __uint_as_float(((lo + (uint32_t)(i >> 23)) << 23) | (uint32_t)(i & 0x7fffff));
            mine += (__float_as_uint(sqrt_normal(x)) != __float_as_uint(sqrtf(x))) ? 1ull : 0ull;
        }
    } else if (mode == 1) {
        const unsigned long long n_div = (unsigned long long)(hi - lo + 1) << 17;
        for (unsigned long long d = blockIdx.x; d < n_div; d += gridDim.x) {
            const uint32_t mb = (lo << 17) + (uint32_t)d;
            const float b = __uint_as_float(0x3f800000u | mb);
            const float y = reciprocal<true>(b);
            for (uint32_t mx = threadIdx.x; mx < (1u << 23); mx += blockDim.x) {
                const float x = __uint_as_float(0x3f800000u | mx);
                mine += (__float_as_uint(div_by_rcp(x, b, y)) != __float_as_uint(x / b)) ? 1ull : 0ull;
            }
        }
    } else {
        unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
        const unsigned long long total = (unsigned long long)(hi - lo + 1) << 24;
        for (; i < total; i += (unsigned long long)gridDim.x * blockDim.x) {
            const uint32_t eb = lo + (uint32_t)(i >> 24);
            const uint32_t bits = ((uint32_t)(i & 1) << 31) | (eb << 23) | (uint32_t)((i >> 1) & 0x7fffff);
            const float b = __uint_as_float(bits);
            const float y = reciprocal<true>(b);
            uint32_t h = bits * 2654435761u + 12345u;
            for (int k = 0; k < 64; ++k) {
                h = h * 1664525u + 1013904223u;
                const int ex = (int)eb - (int)((h >> 9) % 61u);
                if (ex <= 0) continue;
                const float x = __uint_as_float(((h & 1u) << 31) | ((uint32_t)ex << 23) | ((h >> 8) & 0x7fffffu));
                mine += (__float_as_uint(div_by_rcp(x, b, y)) != __float_as_uint(x / b)) ? 1ull : 0ull;
            }
        }
    }
    if (mine) atomicAdd(bad, mine);
}

hipError_t launch_check_reciprocal(int mode, uint32_t exp_lo, uint32_t exp_hi, unsigned long long* d_bad,
                                   hipStream_t stream) {
    if (mode == 0)
        hipLaunchKernelGGL(check_reciprocal_kernel, dim3(4096), dim3(256), 0, stream, exp_lo, exp_hi, d_bad);
    else
        hipLaunchKernelGGL(check_division_kernel, dim3(mode == 1 ? 32768 : 4096), dim3(256), 0, stream, mode, exp_lo, exp_hi, d_bad);
    return hipGetLastError();
}

// ------------------------------------------------------------------ vec3 arithmetic, one operation at a time
// crt_debug_vec3 (where = 1): the device functions the kernels above are built from, applied to arrays, so that
// tests/test_vec3_ref.py can hold them to vectors the reference's own core/vec3.h produced on the host
// (tests/golden/vec3_ref_vectors.npz).  Op codes of oracle/ref_vec3.cpp; 100.. are this file's fast forms, which must
// give the bits of the plain ones.
__global__ void __launch_bounds__(64)
vec3_ops_kernel(int op, long long n, const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ s,
                float* __restrict__ out) {
    const long long i0 = (long long)blockIdx.x * 64 + threadIdx.x;
    const long long i = i0 < n ? i0 : n - 1;            // every lane computes (the fast forms vote across the wavefront)
    const V3 A{a[3 * i], a[3 * i + 1], a[3 * i + 2]}, B{b[3 * i], b[3 * i + 1], b[3 * i + 2]};
    const float t = s[i];
    V3 R{0.0f, 0.0f, 0.0f};
    switch (op) {
        case 1: R = V3{A.x - B.x, A.y - B.y, A.z - B.z}; break;                       // derive_record: e1, e2, s (vec3.h:77-79)
        case 3: R = V3{t * A.x, t * A.y, t * A.z}; break;                             // vec3.h:93-99
        case 6: R.x = A.x * B.x + A.y * B.y + A.z * B.z; break;                       // dot as the exact tests spell it (vec3.h:109-111)
        case 7: R = V3{(A.y * B.z - A.z * B.y), (-(A.x * B.z - A.z * B.x)), (A.x * B.y - A.y * B.x)}; break;   // vec3.h:113-117
        case 8: R.x = sqrtf(A.x * A.x + A.y * A.y + A.z * A.z); break;                // vec3.h:38-40
        case 10: R = unit_vector(A); break;                                           // vec3.h:191-193
        case 11: {                                                                    // backgroundSky's lerp (render.h:45) = sky_from_unit:
            R = sky_from_unit(V3{0.0f, 2.0f * t - 1.0f, 0.0f});                       //   t = 0.5f * (y + 1.0f) with y = 2t - 1 exact for the t used
            break;
        }
        case 12:                                                                      // clip, vec3.h:119-131, as render_pixel spells it
            R.x = (A.x > 1.0f) ? 1.0f : ((A.x < 0.0f) ? 0.0f : A.x);
            R.y = (A.y > 1.0f) ? 1.0f : ((A.y < 0.0f) ? 0.0f : A.y);
            R.z = (A.z > 1.0f) ? 1.0f : ((A.z < 0.0f) ? 0.0f : A.z);
            break;
        case 13: {                                                                    // col /= float(ns), vec3.h:182-189, as render_pixel spells it
            const float k = (float)(1.0 / (double)t);
            R = V3{A.x * k, A.y * k, A.z * k};
            break;
        }
        case 100: R = unit_vector_fast(A); break;                                     // == op 10
        case 101: R.y = unit_vector_fast_y(A); break;                                 // == op 10's y
        case 102: R = sky_from_unit(A); break;                                        // lerp(0.5f * (A.y + 1.0f), vec3(1), vec3(0.5, 0.7, 1.0))
        default: break;
    }
    if (i0 < n) { out[3 * i] = R.x; out[3 * i + 1] = R.y; out[3 * i + 2] = R.z; }
}

hipError_t launch_vec3_ops(int op, long long n, const float* d_a, const float* d_b, const float* d_s, float* d_out, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(vec3_ops_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, op, n, d_a, d_b, d_s, d_out);
    return hipGetLastError();
}

// ------------------------------------------------------------------ multi-GPU reassembly
// Rank 0 after the gather: rank r's piece (its 32-row strips one after the other) starts at
// parts + r * part_stride; global strip s belongs to rank s % world and is that rank's strip
// s / world (only the image's last strip can be short, so every earlier strip of a rank is full).
// One element per thread, row-major over the destination: both sides are coalesced.  VEC: float4
// when a row (nx * 3 floats) and every base offset are multiples of four floats.
template <bool VEC>
__global__ void __launch_bounds__(256)
assemble_strips_kernel(const float* __restrict__ parts, long long part_stride, int row_elems, int ny,
                       int strip_rows, int world, float* __restrict__ full) {
    const long long per_row = VEC ? row_elems / 4 : row_elems;
    const long long total = per_row * ny;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(i / per_row);
        const long long c = i - (long long)y * per_row;
        const int s = y / strip_rows, r = s % world, k = s / world;
        const long long src_row = (long long)k * strip_rows + (y - s * strip_rows);
        if (VEC) {
            const float4* src = reinterpret_cast<const float4*>(parts + (long long)r * part_stride + src_row * row_elems);
            reinterpret_cast<float4*>(full + (long long)y * row_elems)[c] = src[c];
        } else {
            full[(long long)y * row_elems + c] = parts[(long long)r * part_stride + src_row * row_elems + c];
        }
    }
}

hipError_t launch_assemble_strips(const float* d_parts, long long part_stride, int nx, int ny, int strip_rows,
                                  int world, float* d_full, hipStream_t stream) {
    if (nx <= 0 || ny <= 0) return hipSuccess;
    const int row_elems = nx * 3;
    const bool vec = row_elems % 4 == 0 && part_stride % 4 == 0 && ((uintptr_t)d_parts & 15u) == 0 && ((uintptr_t)d_full & 15u) == 0;
    const long long total = (long long)(vec ? row_elems / 4 : row_elems) * ny;
    const unsigned blocks = (unsigned)std::min<long long>((total + 255) / 256, 256 * 32);
    if (vec)
        hipLaunchKernelGGL(assemble_strips_kernel<true>, dim3(blocks), dim3(256), 0, stream, d_parts, part_stride, row_elems, ny,
                           strip_rows, world, d_full);
    else
        hipLaunchKernelGGL(assemble_strips_kernel<false>, dim3(blocks), dim3(256), 0, stream, d_parts, part_stride, row_elems, ny,
                           strip_rows, world, d_full);
    return hipGetLastError();
}

#include "crt_shade_kernels.hpp"
#include "crt_debug_kernels.hpp"   // parity hooks of include/crt_debug.h: not on a render path

}  // namespace crt
