/*
 * crt_debug.h — diagnostics, A/B switches and timing hooks of libcrt_hip.so.
 *
 * NOT the drop-in surface.  include/crt.h is what a maintainer binds to replace the reference's host functions; the
 * entry points below exist for this repository's tests (tests/), measurement tools (tools/, bench.py) and parity
 * checks: kernel selection, counters, self-checks of the arithmetic, single rays / boxes through the kernels' device
 * functions, failure injection.  They are exported by the same library and follow crt.h's conventions (int return,
 * 0 = CRT_OK, negative = error, crt_last_error() has the text).
 */
#ifndef CRT_DEBUG_H
#define CRT_DEBUG_H

#include "crt.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Kernel selection (diagnostics / A-B benchmarking), per calling thread. 0 = auto (default:
 * tiled when the scene and camera are finite and |coordinate| <= 2^38, else brute force),
 * 1 = brute-force list walk, 2 = tiled, 3 = tiled with a tiny survivor pool and heavy-tile
 * threshold (test hook for the overflow fallback and the shared-tile path), 4 = tiled with
 * scalar-operand (SGPR) list walks instead of the LDS-staged records and per-pixel candidate
 * masks (A/B), 8 = tiled with a 16-record binning log (test hook for its overflow fallback), 10 = tiled with shared tiles from
 * 4 entries on, each dealt out as 2, 4 or 8 row slabs to different blocks (test hook of the sub-item path); BVH world only: 1 = the per-ray tree walk instead of the tiled kernels, 5 = the walk's
 * general loop for every wavefront, 6 = the same with every direction chain iterated from its
 * fourth element, 7 = the tiled kernels with every lane on their general per-lane path (test hooks
 * for the rare paths); crt_render_rows_ex with the recursive shader (default: the pair records, walked by the wavefront
 * together): 9 = the skip links of the flat node array, 11 = the pair records with one stack per lane (A/B), 12 = the
 * shared walk with a 96-item list (test hook: trips that overflow it are redone by the literal walk).
 * All variants produce identical pixels.  Returns the previous value. */
int crt_set_kernel_variant(int variant);

/* Diagnostics of the tiled variant, per calling thread.  out48 (nullable, 48 words) first
 * receives the counters accumulated since the last enable:
 *   binning: [0] wavefronts that found work, [1] regions they entered, [2] tiles box-tested
 *   (64 per triangle and region in the lane = tile mode), [3] (wavefront, tile) pairs with survivors,
 *   [4] tiles, [5] sum and [6] max of the tile survivor count, [7] pool overflows,
 *   [8..23] histogram of the tile survivor count (bucket b: 2^(b-1) <= n < 2^b),
 *   shading kernel cycles (s_memtime): [24] heavy phase summed over blocks, [25] its max,
 *   [26] light phase summed over waves, [27] its max, [28] max wave total,
 *   busiest binning wavefront: [29] regions entered, [30] tiles box-tested, [31] cycles;
 *   master wave of shared tiles: [32] set-up+merge+shade, [33] own trace, [34] wait at the
 *   second barrier (cycles summed over [35] tiles with [36] survivors in total);
 *   per-pixel candidate masks of the light tiles with geometry: [37] tiles, [38] their list
 *   entries, [39] sum of the longest per-pixel candidate list of each, [40] sum of all lists,
 *   [41] tiles whose (pixel, entry) pairs did not fit the pair buffer, [42] pairs of the others.
 * Then enable != 0 (re)starts counting, enable == 0 stops it.  Counting costs atomics and clock
 * reads that slow the kernels down (binning 7x, shading 1.4x); enable == 2 selects the TIMELINE
 * mode instead: the shading kernel runs at full speed and every wavefront stores three time
 * stamps of the device-wide 100 MHz clock (start, end of the shared-tile phase, end) and
 * (entries << 44 | tile index << 20 | HW_REG_HW_ID[19:0]) of its block's last shared tile, read with
 * crt_debug_wave_timeline (4 words per wavefront, index = block * waves per block + wave; returns
 * the number of wavefront slots copied, at most max_waves).  max_waves == -1 reads the per-sample
 * log of blocks 0..3 instead (their first shared tile, list world): 4 x 256 pairs of time stamps
 * (master's draws done, slices done); `out` then holds 2048 words.  max_waves == -2: per wavefront its last
 * light tile with geometry — start stamp; tile | entries << 32 | such tiles taken << 48 — 2 x 8192 words. */
int crt_debug_tiled_stats(int enable, uint64_t *out48);
int crt_debug_wave_timeline(uint64_t *out, int max_waves);

/* Audit of the tiled variant's binning (full nx x ny image on the current device): a brute-
 * force kernel tests every triangle against `samples` rays of every pixel (pixel-corner jitter
 * included) and checks that each (ray, triangle) pair passing Triangle::collision_detection —
 * closest hit or not — has its triangle in the tile's survivor list AND passes the shading
 * kernel's per-pixel candidate test for its pixel.  out4[0] = passing pairs seen, out4[1] = pairs
 * whose triangle is missing from the list or from the pixel's candidates (must be 0), out4[2] =
 * "cull-sure" verdicts (tile or pixel) that saw dot(d, normal) < 0 (must be 0), out4[3] = last
 * offender (tile << 32 | triangle). */
int crt_debug_audit_tiles(const crt_scene *scene, const crt_camera *camera, int nx, int ny,
                          int samples, uint64_t seed, uint64_t *out4);

/* Self-check: the kernels compute f = 1.0/a (shapes/triangle.h:80) as v_rcp_f32 + one Newton
 * step when the scene is bounded (|coordinate| <= 2^38, so 2^-20 <= |a| <= 2^84 wherever f
 * is used).  This compares that sequence with the IEEE quotient for EVERY float whose biased
 * exponent lies in [exp_lo, exp_hi], both signs, on the current device; *mismatches must be 0
 * for [27, 227] (2^-100 .. 2^100). */
int crt_debug_check_reciprocal(int exp_lo, int exp_hi, uint64_t *mismatches);
/* Same idea for the quotients of the shading kernel (`x / len` in unit_vector,
 * `(x + U) / float(nx)`), computed as q0 = x*y, r = fma(-q0, b, x), q = fma(r, y, q0) with
 * y = the reciprocal above.  mode 0 = the check above; mode 1 = exhaustive over significands:
 * divisors 1.mb for mb in chunks [exp_lo, exp_hi] of 64 (2^17 divisors each) against ALL 2^23
 * numerator significands (chunks 0..63 = 7.04e13 pairs = every pair of significands);
 * mode 2 = every float divisor with biased exponent in [exp_lo, exp_hi], 64 pseudo-random
 * numerators each, exponents up to 60 below the divisor's, both signs; mode 3 = the kernel's
 * square root of normal floats (v_sqrt_f32 + neighbour residuals) against sqrtf for every positive
 * float with biased exponent in [exp_lo, exp_hi]; mode 4 = the kernels' fused forms of curand_uniform and of
 * 2.0f * curand_uniform - 1.0f (the product x * 2^-32 is exact, so one fma rounds as the literal product-then-sum does) against
 * the literal forms for ALL 2^32 values of a draw (exp_lo, exp_hi ignored: pass 0, 0). */
int crt_debug_check_division(int mode, int exp_lo, int exp_hi, uint64_t *mismatches);

/* One core/vec3.h operation at a time on arrays (a, b, out: n x 3 floats, s: n floats; host pointers): where = 0 the
 * host layer's helpers (camera constructor, scene pack, BVH build: ops 1 sub, 3 s*a, 7 cross, 10 unit_vector,
 * 14 rotate), where = 1 the kernels' device functions (ops 1, 3, 6 dot, 7, 8 length, 10, 11 sky lerp of t = s,
 * 12 clip, 13 a /= s, and the fast forms 100 unit_vector_fast, 101 its y alone, 102 sky_from_unit).  Op codes are
 * those of oracle/ref_vec3.cpp, which wraps the reference's own core/vec3.h (replaces: core/vec3.h:73-217);
 * tests/test_vec3_ref.py compares bit for bit.  CRT_ERR_UNSUPPORTED for an operation that side does not have. */
int crt_debug_vec3(int where, int op, long long n, const float *a, const float *b, const float *s, float *out);

/* Duration in milliseconds of the last crt_render_rows main kernel launched through this
 * library on the calling thread, measured with HIP events on the launch stream.
 * Synchronises on that stream.  Negative on error. */
float crt_last_kernel_ms(void);

/* Same measurement split per kernel: ms3[0] = total, ms3[1] = binning kernel (0 for the
 * brute-force variant), ms3[2] = shading / brute-force kernel. */
int crt_last_kernel_times(float *ms3);

/* The three HIP events per frame behind crt_last_kernel_ms / crt_last_kernel_times (per calling
 * thread, default on).  Each is a marker packet between the frame's five kernels; a host that never
 * asks for kernel times can switch them off (bench.py does for its timed region: -7 us per frame).
 * While off, crt_last_kernel_* fail with CRT_ERR_INVALID.  Returns the previous value. */
int crt_set_kernel_timing(int enable);

/* Diagnostics: the tree as it is on the device.  boxes6: n_nodes x {min xyz, max xyz}; meta4:
 * n_nodes x {skip, first slot, triangle count (0 = inner node), depth}, nodes in preorder;
 * slot_triangle: n_tris list indices in leaf order.  Any pointer may be NULL. */
int crt_debug_bvh_export(const crt_scene *scene, float *boxes6, int *meta4, int *slot_triangle);

/* Test hook: the next `count` ncclSend calls made by this library (crt_gather_strips, crt_multi_render) report
 * ncclInternalError without reaching RCCL, so that the error paths of the exchange step can be exercised on one
 * device.  Returns the previous count; 0 switches it off. */
int crt_debug_inject_send_failure(int count);

/* ---- parity hooks against the reference's own device code (oracle/ref_hit.hip, tests/test_ref_hit.py) -------------
 * Closest hit of caller-supplied rays through the device functions the render kernels call (csrc/crt_debug_kernels.hpp).
 * direction: n_rays x 3, the ray as `render` hands it to (*world)->hit (un-normalised; core/render.h:75); origin: 3
 * floats shared by all rays (modes 0, 2: the pinhole camera's origin the per-frame records are derived for) or
 * n_rays x 3 (modes 1, 3, 4).  All arrays are HOST pointers.
 *   mode 0  list world, per-frame exact records + the exact_test walk (render_list_kernel; replaces
 *           hitable/hitable.h:29-38 twice, hitable/hitable_list.h:76-91, shapes/triangle.h:57-100)
 *   mode 1  list world, any origin, base records (tri_hit_base: the generic kernel's secondary rays)
 *   mode 2  BVH world, camera-origin records, skip-link walk (render_bvh_kernel; hitable/bvh.h:160-190, core/aabb.h:30-43)
 *   mode 3  BVH world, any origin, skip links;   mode 4  the same over the pair records (one stack per lane)
 * arith 0 = the literal IEEE forms, 1 = the fast forms (exact reciprocal / quotient sequences; CRT_ERR_UNSUPPORTED when
 * the scene is not bounded as those forms require — the render entry points then take the literal forms as well).
 * Outputs (each n_rays long; normal and point x 3): hit flag, rec.t, the LIST index of the winning triangle (-1 on a
 * miss; BVH worlds map the slot back), rec.normal (the shading normal), rec.p (modes 1, 3, 4; zeros otherwise). */
int crt_debug_trace_rays(const crt_scene *scene, int mode, int arith, int n_rays, const float *origin,
                         const float *direction, int *hit, float *t, int *index, float *normal, float *point);

/* AABB::hit(r, 0.001, FLT_MAX) (core/aabb.h:30-43) in the kernels' two forms on n (box, ray) pairs: form 0 the literal
 * loop (box_hit_ieee), form 1 the bounded-scene form (box_hit_fast).  box6: n x {min xyz, max xyz}; host pointers. */
int crt_debug_box_hit(int form, int n, const float *box6, const float *origin, const float *direction, int *out);
/* The refit kernel's box helpers on n boxes: op 0 surrounding_box(a, b) (core/aabb.h:58-66), op 1
 * Transform(0, 0, 1).TransformAABB(a) (hitable/transform.h:22-31).  where 0 = the host build's helpers
 * (crt_bvh.cpp), 1 = the device functions of bvh_refit_level_kernel. */
int crt_debug_box_op(int where, int op, int n, const float *a6, const float *b6, float *out6);

/* The BVH build's restatement of thrust::sort's sequential backend with BoxCompare's rule (hitable/bvh.h:9-45,86-92;
 * crt_bvh.cpp: thrust_sequential_sort) on a key array: order[0..n) is permuted in place.  Host only; pinned against
 * rocThrust's real thrust::sort by tests/test_ref_thrust.py. */
int crt_debug_thrust_sort(int n, const float *key, int *order);

/* Matrix k (0..63) of the table rng_init_kernel walks: T^(2^(67+k)), T = the GF(2) transition matrix of XORWOW's xorshift
 * part; rows[160][5], row b = image of state bit b (word b/32, bit b%32).  Host only; held to rocRAND's precomputed jump
 * matrices by tests/test_ref_xorwow.py. */
int crt_debug_xorwow_jump(int k, uint32_t *rows);

/* Test hook for the recycled-stream branch of the launch contexts: the next `count` context look-ups treat their stream
 * as a NEW stream behind an old handle (what a changed hipStreamGetId means on HIP >= 7.1 runtimes; PyTorch's bundled
 * 7.0 runtime has no hipStreamGetId, so without this hook the branch never runs there).  Returns the previous count. */
int crt_debug_force_stream_recycle(int count);

#ifdef __cplusplus
}
#endif
#endif /* CRT_DEBUG_H */
