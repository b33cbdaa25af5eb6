// crt_bvh_kernels.hpp — BVH world of the render kernel (included by crt_kernels.hip, namespace crt).
//
// Reference: BVHNode::collision_detection (hitable/bvh.h:160-190) under Hitable::hit
// (hitable/hitable.h:29-38), AABB::hit (core/aabb.h:30-43), launched as the same `render` kernel
// with world = BVHNode** (core/render.h:222); BVHNode::UpdateBVH (bvh.h:135-158) for the refit.
//
// What the reference does per ray is reproduced bit for bit; how it is done is not:
//   * The reference recurses through device-heap BVHNode objects (two virtual calls and a
//     TransformRay — normalisation, 18 sin/cos — per node).  Here the tree is a flat array in
//     preorder (host-built, crt_bvh.cpp) with skip links: a lane walks it without a stack — a node
//     whose box is hit is followed by its left child (the next node), anything else by the node
//     after its subtree.  The reference visits BOTH children of every hit node with the same
//     [t_min, t_max] = [0.001, FLT_MAX] and merges afterwards, so the set of visited nodes does not
//     depend on the order or on earlier hits: the stackless walk visits exactly that set.
//   * Every nesting level re-normalises the ray direction (Hitable::hit -> ScaleRay,
//     transform.h:50-54) and applies rotate(., 0) (vec3.h:200-217): the box of a node at depth L
//     sees F^(L+1)(d0), a leaf's triangles F^(L+3)(d0), F = rotate0 o unit_vector.  In float F
//     reaches a fixed point after three applications for 99.99 % of camera rays, and a 2-cycle
//     for most of the rest; the chain is computed once per ray (c1..c4, c5 when needed) and a
//     node picks its element by depth.  A wave whose lanes all have c4 == c3 runs the hot loop
//     (one direction, precomputed reciprocals); any other wave runs the general loop (per-node
//     selection, literal IEEE divisions, iteration from c4 for chains that do neither).
//   * Merge rule: `left.t < right.t ? left : right` up the tree (bvh.h:172-179) and the leaf
//     list's strict `<` (hitable_list.h:84) give: smallest t; among equal t the LAST leaf in
//     left-to-right order, inside a leaf the first triangle.  Accepted hits never carry a NaN t
//     (the leaf's `tmp_rec.t < closest_so_far` rejects it), so folding the leaves in preorder
//     with `keep best unless !(best.t < t)` is the same function.
//   * Leaf triangles are stored in leaf order ("slots"), so a leaf is a span of the per-frame
//     exact records (tri_prepass over the slot-ordered base array) and of the shading normals
//     (rotate(., 0) applied depth + 3 times on the host).
#pragma once

// node i = 2 float4:  n0 = (min.x, min.y, min.z, max.x),  n1 = (max.y, max.z, skip, info)
//   skip: preorder index of the node after this node's subtree (as int bits)
//   info: first slot (bits 0..23) | triangle count (24..25; 0 = inner node) | depth (26..30)
struct BvhView {
    const float4* nodes;
    int n_nodes;
    float rot[9];   // rows of rotate(., vec3(0)) as the reference computes them (host, crt_bvh.cpp)
    // pair records (bvh_walk_pairs): one per INNER node, preorder over the inner nodes, 4 float4 = 64 B:
    //   p0 = (L.min.xyz, L.max.x)  p1 = (L.max.yz, R.min.xy)  p2 = (R.min.z, R.max.xyz)  p3 = (refL, refR, coneL, coneR) as int bits
    //   ref: child is a leaf ? 0x80000000 | count << 24 | first slot : the child's pair index
    //   cone: the normal cone of the child's subtree (threshold byte | axis x, y, z as signed bytes; 0xff in the low byte: none)
    // null: the walks below use `nodes` only.
    const float4* pairs = nullptr;
    const int* partner = nullptr;   // BvhTiles::partner: the tie order as one key (larger wins among equal t)
#ifdef CRT_TUNING
    // tuning builds (crt_debug_tiled_stats around a generic-kernel launch): [0] trips [1] lanes in flight [2] walk steps
    // [3] lanes walking [4] rounds of triangle tests [5] lanes testing
    unsigned long long* walk_stats = nullptr;
    bool walk_count = false;   // set by the kernel: counters (stats mode 1), not only the time stamps (mode 2)
    unsigned* wave_steps = nullptr;   // set by the kernel (LDS): this wavefront's walk steps, for its timeline record
#endif
};
#ifdef CRT_TUNING
__device__ __forceinline__ void walk_stat(const BvhView& b, int word, unsigned long long mask) {
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    if (word == 2 && b.wave_steps && lane == __builtin_amdgcn_readfirstlane(lane)) ++*b.wave_steps;
    if (b.walk_count && lane == __builtin_amdgcn_readfirstlane(lane)) {   // the first ACTIVE lane
        atomicAdd(&b.walk_stats[word], 1ull);
        atomicAdd(&b.walk_stats[word + 1], (unsigned long long)__builtin_popcountll(mask));
    }
}
#else
#define walk_stat(b, word, mask) ((void)0)
#endif

// kPairStackDepth (crt_device.hpp): pending right children per lane (LDS); deeper trees take the skip-link walk

__device__ __forceinline__ V3 rot0(const float* m, V3 v) {   // vec3.h:215: ((origin * rotate_k).sum()) = (x*a + y*b) + z*c
    return V3{(v.x * m[0] + v.y * m[1]) + v.z * m[2], (v.x * m[3] + v.y * m[4]) + v.z * m[5],
              (v.x * m[6] + v.y * m[7]) + v.z * m[8]};
}

// rotate(v, 0) returns v bit for bit unless a component is zero (its sign can change) or not
// finite (NaN appears): only then is the literal product evaluated — for the whole wavefront,
// which is harmless for the lanes that did not need it.
__device__ __forceinline__ V3 rot0_exact(const float* m, V3 v) {
    const float lo = fminf(fminf(fabsf(v.x), fabsf(v.y)), fabsf(v.z)), hi = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fabsf(v.z));
    const bool plain = lo > 0.0f && hi < __builtin_inff();
    if (__builtin_amdgcn_ballot_w64(!plain) != 0ull) return rot0(m, v);
    return v;
}

// F = RotateRay(ScaleRay(.)) on the direction, identity transform: scale 1 (x / 1 == x), unit_vector, rotate(., 0).
template <bool FAST>
__device__ __forceinline__ V3 chain_next(const float* m, V3 c) {
    return rot0_exact(m, FAST ? unit_vector_fast(c) : unit_vector(c));
}

__device__ __forceinline__ bool same_bits(V3 a, V3 b) {
    return __float_as_uint(a.x) == __float_as_uint(b.x) && __float_as_uint(a.y) == __float_as_uint(b.y) &&
           __float_as_uint(a.z) == __float_as_uint(b.z);
}

// component-wise select of VALUES (a conditional expression on two struct lvalues selects their ADDRESSES and loads: the
// struct then has to live in scratch memory — 80 B per lane and 400 MB of write traffic per frame, measured)
__device__ __forceinline__ V3 sel3(bool c, V3 a, V3 b) {
    return V3{c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z};
}

// AABB::hit (core/aabb.h:30-43), literal: IEEE divisions, ffmin/ffmax as `a < b ? a : b`.
__device__ __forceinline__ bool box_hit_ieee(float4 n0, float4 n1, V3 o, V3 d) {
    const float mn[3] = {n0.x, n0.y, n0.z}, mx[3] = {n0.w, n1.x, n1.y};
    const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
    float t_min = 0.001f, t_max = FLT_MAX;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float qa = (mn[a] - oo[a]) / dd[a], qb = (mx[a] - oo[a]) / dd[a];
        const float t0 = qa < qb ? qa : qb;
        const float t1 = qa > qb ? qa : qb;
        t_min = t0 > t_min ? t0 : t_min;
        t_max = t1 < t_max ? t1 : t_max;
        if (t_max <= t_min) return false;
    }
    return true;
}

// The same test for bounded scenes and ordinary directions (every |d| component >= 2^-40, every
// coordinate difference 0 or >= 2^-80 in magnitude, <= 2^40: guaranteed by the host's scene check):
// the six quotients are div_by_rcp (bit-identical to IEEE, see there), all of them are finite, so
// ffmin/ffmax are v_min/v_max and — t_min only grows, t_max only shrinks — the three early exits
// are one comparison at the end.
__device__ __forceinline__ bool box_hit_fast(float4 n0, float4 n1, V3 o, V3 d, V3 y) {
    const float ax = div_by_rcp(n0.x - o.x, d.x, y.x), bx = div_by_rcp(n0.w - o.x, d.x, y.x);
    const float ay = div_by_rcp(n0.y - o.y, d.y, y.y), by = div_by_rcp(n1.x - o.y, d.y, y.y);
    const float az = div_by_rcp(n0.z - o.z, d.z, y.z), bz = div_by_rcp(n1.y - o.z, d.z, y.z);
    const float t_min = fmaxf(fmaxf(fminf(ax, bx), 0.001f), fmaxf(fminf(ay, by), fminf(az, bz)));
    const float t_max = fminf(fminf(fmaxf(ax, bx), FLT_MAX), fminf(fmaxf(ay, by), fmaxf(az, bz)));
    return !(t_max <= t_min);
}

// Triangle::collision_detection (shapes/triangle.h:57-100) for an arbitrary ray origin, from the base
// record (v0 v1 v2 n): secondary rays (crt_shade_kernels.hpp) do not start at the camera, so the
// per-frame exact records do not apply.
template <bool FAST>
__device__ __forceinline__ bool tri_hit_base(const float* __restrict__ b, V3 o, V3 d, float& t_out) {
    // the 48-B record as three 16-B loads (records start on 16-B boundaries: 12 floats each from a hipMalloc base): per-lane
    // records of incoherent rays cost one memory instruction per LOAD whatever its width — twelve dword loads were the
    // largest item of a secondary ray's traversal
    const float4* __restrict__ b4 = reinterpret_cast<const float4*>(b);
    const float4 ba = b4[0], bb = b4[1], bc = b4[2];   // v0.xyz v1.x | v1.yz v2.xy | v2.z n.xyz
    if (d.x * bc.y + d.y * bc.z + d.z * bc.w < 0.0f) return false;           // triangle.h:61
    const float e1x = ba.w - ba.x, e1y = bb.x - ba.y, e1z = bb.y - ba.z;      // triangle.h:69
    const float e2x = bb.z - ba.x, e2y = bb.w - ba.y, e2z = bc.x - ba.z;      // triangle.h:70
    const float hx = (d.y * e2z - d.z * e2y);                                 // triangle.h:71-72
    const float hy = (-(d.x * e2z - d.z * e2x));
    const float hz = (d.x * e2y - d.y * e2x);
    const float a = e1x * hx + e1y * hy + e1z * hz;
    if (a > -kTriEpsilon && a < kTriEpsilon) return false;                    // triangle.h:77
    const float f = FAST ? ((fabsf(a) >= 7.888609052210118e-31f && fabsf(a) <= 1.2676506002282294e30f) ? reciprocal<true>(a) : 1.0f / a)
                         : 1.0f / a;                                          // triangle.h:80 (fast form valid for 2^-100..2^100)
    const float sx = o.x - ba.x, sy = o.y - ba.y, sz = o.z - ba.z;            // triangle.h:81
    const float u = f * (sx * hx + sy * hy + sz * hz);                        // triangle.h:82
    if (u < 0.0f || u > 1.0f) return false;                                   // triangle.h:84
    const float qx = (sy * e1z - sz * e1y);                                   // triangle.h:87
    const float qy = (-(sx * e1z - sz * e1x));
    const float qz = (sx * e1y - sy * e1x);
    const float v = f * (d.x * qx + d.y * qy + d.z * qz);                     // triangle.h:88
    if (v < 0.0f || u + v > 1.0f) return false;                               // triangle.h:89
    t_out = f * (e2x * qx + e2y * qy + e2z * qz);                             // triangle.h:92
    return true;
}

// Triangle::collision_detection (shapes/triangle.h:57-100) on one slot's exact record, per lane
// (PRIMARY: the ray starts at the camera origin the records were derived for; else from the base record).
template <bool FAST, bool PRIMARY = true>
__device__ __forceinline__ bool tri_hit_lane(const float* __restrict__ rec, int slot, V3 o, V3 d, float& t_out) {
    if (!PRIMARY) return tri_hit_base<FAST>(rec + (size_t)slot * kBaseFloats, o, d, t_out);
    const float4* r = reinterpret_cast<const float4*>(rec + (size_t)slot * kRecFloats);
    const float4 ra = r[0], rb = r[1], rc = r[2], rd = r[3];   // n.xyz e2.x | e2.yz e1.xy | e1.z s.xyz | q.xyz tq
    if (d.x * ra.x + d.y * ra.y + d.z * ra.z < 0.0f) return false;            // triangle.h:61
    const float e2x = ra.w, e2y = rb.x, e2z = rb.y, e1x = rb.z, e1y = rb.w, e1z = rc.x;
    const float hx = (d.y * e2z - d.z * e2y);                                 // triangle.h:71-72
    const float hy = (-(d.x * e2z - d.z * e2x));
    const float hz = (d.x * e2y - d.y * e2x);
    const float a = e1x * hx + e1y * hy + e1z * hz;
    if (a > -kTriEpsilon && a < kTriEpsilon) return false;                    // triangle.h:77
    const float f = reciprocal<FAST>(a);                                      // triangle.h:80
    const float u = f * (rc.y * hx + rc.z * hy + rc.w * hz);                  // triangle.h:82
    if (u < 0.0f || u > 1.0f) return false;                                   // triangle.h:84
    const float v = f * (d.x * rd.x + d.y * rd.y + d.z * rd.z);               // triangle.h:88
    if (v < 0.0f || u + v > 1.0f) return false;                               // triangle.h:89
    t_out = f * rd.w;                                                         // triangle.h:92
    return true;
}

// A leaf's HitableList (1 or 2 triangles, hitable_list.h:76-91: strict <, t_max = FLT_MAX) folded
// into the running result with the tree's merge rule (bvh.h:172-179).
template <bool FAST, bool PRIMARY = true>
__device__ __forceinline__ void leaf_hit(const float* __restrict__ rec, int first, int count, V3 o, V3 d, float& best_t, int& best_slot) {
    float closest = FLT_MAX;
    int slot = -1;
    for (int j = 0; j < count; ++j) {
        float t;
        if (tri_hit_lane<FAST, PRIMARY>(rec, first + j, o, d, t) && t < closest) {
            closest = t;
            slot = first + j;
        }
    }
    if (slot >= 0 && !(best_slot >= 0 && best_t < closest)) {
        best_t = closest;
        best_slot = slot;
    }
}

// Hot loop: every lane of the wavefront has c4 == c3 and ordinary components.
template <bool PRIMARY = true>
__device__ __forceinline__ void bvh_walk_fast(const BvhView& b, const float* __restrict__ rec, V3 o, V3 c1, V3 c2, V3 c3,
                                              float& best_t, int& best_slot) {
    const V3 y3 = V3{reciprocal<true>(c3.x), reciprocal<true>(c3.y), reciprocal<true>(c3.z)};
    int i = 0;
    while (i < b.n_nodes) {
        const float4 n0 = b.nodes[2 * (size_t)i], n1 = b.nodes[2 * (size_t)i + 1];
        const int skip = __float_as_int(n1.z), info = __float_as_int(n1.w);
        const int depth = (info >> 26) & 31, count = (info >> 24) & 3, first = info & 0xffffff;
        V3 d = c3, y = y3;
        if (depth < 2) {   // the root sees F(d0), its children F^2(d0): three nodes of the whole tree
            d = sel3(depth == 0, c1, c2);
            y = V3{reciprocal<true>(d.x), reciprocal<true>(d.y), reciprocal<true>(d.z)};
        }
        const bool hit = box_hit_fast(n0, n1, o, d, y);
        if (hit && count != 0) leaf_hit<true, PRIMARY>(rec, first, count, o, c3, best_t, best_slot);   // triangles see F^(depth+3) = c3
        i = (hit && count == 0) ? i + 1 : skip;
    }
}

// The hot loop over PAIR records.  The skip-link walk above pays one dependent 32-B load per VISITED node — and the
// reference's rule visits both children of every node whose box is hit (bvh.h:160-190), 200 nodes for a typical ray of
// the 100 000-triangle mesh, box hit or not.  Here a hit inner node's record holds the boxes of BOTH children: one 64-B
// load, two slab tests, and only children that ARE hit are followed — half as many dependent loads, none for a child
// whose box is missed.  The set of leaves whose triangles are tested is the same (a leaf is tested iff its box and the
// boxes of all its ancestors are hit), each with the same direction, so the result is the same provided the merge rule
// is applied as a key instead of in preorder: smallest t; among equal t the larger partner[slot] (= the later leaf,
// and inside a leaf the first triangle: BvhTiles::partner).  A right child that waits while the left subtree is
// walked sits on a per-lane stack in LDS (stack[level * stack_stride]).
// Direction chains that are not at a fixed point by c3.  Camera rays of the reference camera almost always are (c4 == c3
// for 99.98 %), arbitrary directions are not: 1 % of the scattered rays of `shade` have c4 != c3 — a third of them reach the
// fixed point one step later (c5 == c4), half sit in a 2-cycle (c5 == c3) — so half of all wavefronts of secondary rays
// held such a lane and the whole wavefront took the general loop (IEEE divisions, chain element by iteration: 5x the
// time).  The pair walk therefore carries c3 AND c4 with their reciprocals and picks by depth:
//     k <= 1: c1    k == 2: c2    k == 3: c3    k >= 4: cyc && (k odd) ? c3 : c4        (chain_at's rule for modes 0 and 1)
// which serves every lane except those whose chain does neither by c5 (0.1 % of rays: they walk alone afterwards).
struct PairDirs {
    V3 c2, c3, c4, y2, y3, y4;
    bool cyc;     // 2-cycle c3, c4, c3, c4, ... (chain_at mode 1); otherwise c_k = c4 for k >= 4
};
__device__ __forceinline__ bool pair_uses_c3(const PairDirs& pd, int k) {   // k >= 3
    return k == 3 || (pd.cyc && (k & 1));
}

// The normal cone of a subtree (host: crt_bvh.cpp, cone_of): true when every stored normal n_i of the subtree has
// d . n_i < 0 with a margin of 1e-3 |d| |n_i| — whichever chain element (c3 or c4, a few ulp apart) its triangles see.
__device__ __forceinline__ bool cone_rejects(unsigned cone, V3 d) {
    const float qx = (float)((int)(cone << 16) >> 24), qy = (float)((int)(cone << 8) >> 24), qz = (float)((int)cone >> 24);
    const float t127 = (float)(cone & 0xffu) * (kConeScale * 127.0f);
    return (cone & 0xffu) != 0xffu && d.x * qx + d.y * qy + d.z * qz < -t127;
}

template <bool PRIMARY>
__device__ __forceinline__ void pair_tri(const BvhView& b, const float* __restrict__ rec, int slot, V3 o, V3 dt, float& best_t, int& best_slot,
                                         V3& dir_used) {
    float t;
    if (tri_hit_lane<true, PRIMARY>(rec, slot, o, dt, t) && t < FLT_MAX) {   // hitable_list.h:84 against t_max = FLT_MAX: NaN and +inf never pass
        if (best_slot < 0 || t < best_t || (t == best_t && b.partner[slot] > b.partner[best_slot])) {
            best_t = t;
            best_slot = slot;
            dir_used = dt;
        }
    }
}
template <bool PRIMARY>
__device__ __forceinline__ void pair_leaf(const BvhView& b, const float* __restrict__ rec, int ref, V3 o, V3 dt, float& best_t, int& best_slot,
                                          V3& dir_used) {
    const int first = ref & 0xffffff, count = (ref >> 24) & 3;   // 1 or 2 triangles (bvh.h:95-109)
    pair_tri<PRIMARY>(b, rec, first, o, dt, best_t, best_slot, dir_used);
    if (count == 2) pair_tri<PRIMARY>(b, rec, first + 1, o, dt, best_t, best_slot, dir_used);
}

// `take`: this lane walks here (its chain is served by PairDirs and its components are ordinary); the others keep their result.
// Depth first, one record per trip; a hit inner child that has to wait sits on the lane's stack (at most one entry per
// level).  Measured and dropped: two records per trip (the stack top together with the current node) — the LDS of a
// stack with two entries per level halves the resident blocks of the generic kernel, and that costs more (shade 19.8 ->
// 26.9 ms) than the second record in flight gains.  Returns false for a lane that ran out of stack (the caller walks it
// again with the general loop; the host only builds pair records for trees that fit).
//
// The triangles of hit leaves are NOT tested where they are found: with 64 lanes at 64 different nodes almost every trip
// has some lane at a leaf, and the triangle code (the larger half of a trip) ran for a handful of lanes each time — 20 of
// 64 lanes active per VALU cycle over the whole kernel (tools/pmc_lanes.sh).  A lane queues its triangles (slot + which
// chain element they see) at the top of its stack area and walks on; the wavefront tests queued triangles together, one
// per lane and round, when most lanes hold some (or a lane's area runs short, or the walk has ended).  Nothing observable
// depends on the order of the tests: every box-hit leaf is tested whatever has been found so far (the reference has no
// culling by the closest hit) and the merge rule below is a total order on (t, partner).
#ifndef CRT_DRAIN_HI
#define CRT_DRAIN_HI 40   // start a round of triangle tests when this many lanes hold queued triangles ...
#endif
#ifndef CRT_DRAIN_LO
#define CRT_DRAIN_LO 24   // ... and go on while this many still do
#endif
template <bool PRIMARY = true>
__device__ __forceinline__ bool bvh_walk_pairs(const BvhView& b, const float* __restrict__ rec, V3 o, V3 c1, const PairDirs& pd, bool take,
                                               int* __restrict__ stack, int stack_stride, float& best_t, int& best_slot, V3& dir_used) {
    bool alive = take, overflow = false;
    {   // the root: its own box sees c1 (depth 0, k = 1)
        const float4 n0 = b.nodes[0], n1 = b.nodes[1];
        const V3 y1 = V3{reciprocal<true>(c1.x), reciprocal<true>(c1.y), reciprocal<true>(c1.z)};
        alive = alive && box_hit_fast(n0, n1, o, c1, y1);
        const int info = __float_as_int(n1.w);
        if (((info >> 24) & 3) != 0) {   // a one-leaf tree (wave-uniform: the same node for every lane): its triangles see k = 3
            if (alive) pair_leaf<PRIMARY>(b, rec, (int)(0x80000000u | (unsigned)(info & 0x3ffffff)), o, pd.c3, best_t, best_slot, dir_used);
            return true;
        }
    }
    int cur = 0, sp = 0, depth = 0;   // depth of the pair's node: its children's boxes see k = depth + 2, a leaf child's triangles k = depth + 4
    int nq = 0;                       // queued triangles: stack[(kPairArea - 1 - j) * stride], j < nq
    // ONE loop with a wave-uniform condition; a lane that has run out of nodes is masked, the others go on.  (Written as a
    // per-lane `for (;;)` with break / pop branches the compiler nests the loops, and a lane that needs to pop waits for
    // every other lane of the wavefront to reach a dead end: 2x slower than the skip links, measured.)
    for (;;) {
        const bool walking = __builtin_amdgcn_ballot_w64(alive) != 0ull;
        // a trip needs room for one stack entry and four triangles; a lane without queue always has it (kPairArea - kPairStackDepth >= 5)
        int go_on = walking ? CRT_DRAIN_HI : 1;
        for (;;) {
            const unsigned long long pending = __builtin_amdgcn_ballot_w64(nq > 0);
            const unsigned long long tight = __builtin_amdgcn_ballot_w64(nq > 0 && sp + nq > kPairArea - 5);
            if (tight == 0ull && __builtin_popcountll(pending) < go_on) break;
            walk_stat(b, 4, pending);
            if (nq > 0) {
                --nq;
                const int item = stack[(kPairArea - 1 - nq) * stack_stride];
                pair_tri<PRIMARY>(b, rec, item & 0xffffff, o, sel3((item & 0x40000000) != 0, pd.c3, pd.c4), best_t, best_slot, dir_used);
            }
            go_on = walking ? CRT_DRAIN_LO : 1;
        }
        if (!walking) break;
        walk_stat(b, 2, __builtin_amdgcn_ballot_w64(alive));
        if (alive) {
            const float4* __restrict__ pr = b.pairs + 4 * (size_t)cur;
            const float4 p0 = pr[0], p1 = pr[1], p2 = pr[2], p3 = pr[3];
            const int k = depth + 2;
            const bool box3 = pair_uses_c3(pd, k), tri3 = pd.cyc && (k & 1);   // k + 2 >= 4 has k's parity
            const V3 d = sel3(k == 2, pd.c2, sel3(box3, pd.c3, pd.c4)), y = sel3(k == 2, pd.y2, sel3(box3, pd.y3, pd.y4));
            const V3 dt = sel3(tri3, pd.c3, pd.c4);
            const bool hit_l = box_hit_fast(p0, make_float4(p1.x, p1.y, 0.0f, 0.0f), o, d, y);
            const bool hit_r = box_hit_fast(make_float4(p1.z, p1.w, p2.x, p2.y), make_float4(p2.z, p2.w, 0.0f, 0.0f), o, d, y);
            const int ref_l = __float_as_int(p3.x), ref_r = __float_as_int(p3.y);
            // a child whose subtree's stored normals ALL have d . n < 0 beyond rounding holds no triangle that passes the
            // first test of Triangle::collision_detection (triangle.h:61): it is skipped like a missed box
            const bool try_l = hit_l && !cone_rejects(__float_as_uint(p3.z), dt), try_r = hit_r && !cone_rejects(__float_as_uint(p3.w), dt);
            const int flag = tri3 ? 0x40000000 : 0;
            if (try_l && ref_l < 0) {   // a leaf: 1 or 2 triangles (bvh.h:95-109)
                stack[(kPairArea - 1 - nq) * stack_stride] = (ref_l & 0xffffff) | flag;
                ++nq;
                if (((ref_l >> 24) & 3) == 2) {
                    stack[(kPairArea - 1 - nq) * stack_stride] = ((ref_l & 0xffffff) + 1) | flag;
                    ++nq;
                }
            }
            if (try_r && ref_r < 0) {
                stack[(kPairArea - 1 - nq) * stack_stride] = (ref_r & 0xffffff) | flag;
                ++nq;
                if (((ref_r >> 24) & 3) == 2) {
                    stack[(kPairArea - 1 - nq) * stack_stride] = ((ref_r & 0xffffff) + 1) | flag;
                    ++nq;
                }
            }
            const bool go_l = try_l && ref_l >= 0, go_r = try_r && ref_r >= 0;
            ++depth;
            if (go_l && go_r) {
                overflow = sp >= kPairStackDepth;
                if (!overflow) stack[sp * stack_stride] = ref_r | (depth << 26);
                ++sp;
            }
            int next = go_l ? ref_l : (go_r ? ref_r : -1);
            if (next < 0 && sp > 0) {
                --sp;
                const int e = stack[sp * stack_stride];
                next = e & 0x3ffffff;
                depth = (e >> 26) & 31;
            }
            alive = next >= 0 && !overflow;
            cur = next;
        }
    }
    return !overflow;
}

// General loop: chain element by depth, literal arithmetic.  mode: 0 = c_k is c4 for k > 4 (fixed
// point reached by c4 at the latest), 1 = 2-cycle (c5 == c3: odd k -> c3, even k -> c4), 2 = neither:
// iterate F from c4.
template <bool FAST>
__device__ __forceinline__ V3 chain_at(const float* m, int k, V3 c1, V3 c2, V3 c3, V3 c4, int mode) {
    if (k <= 1) return c1;
    if (k == 2) return c2;
    if (k == 3) return c3;
    if (k == 4 || mode == 0) return c4;
    if (mode == 1) return sel3((k & 1) != 0, c3, c4);
    V3 c = c4;
    for (int j = 4; j < k; ++j) c = rot0(m, unit_vector(c));   // per lane: no wave votes in here
    return c;
}

// dir_used: the direction the winning triangle was tested with (rec.p = o + t * that, triangle.h:95).
template <bool FAST, bool PRIMARY = true>
__device__ __forceinline__ void bvh_walk_general(const BvhView& b, const float* __restrict__ rec, V3 o, V3 c1, V3 c2, V3 c3,
                                                 V3 c4, int mode, float& best_t, int& best_slot, V3& dir_used) {
    int i = 0;
    while (i < b.n_nodes) {
        const float4 n0 = b.nodes[2 * (size_t)i], n1 = b.nodes[2 * (size_t)i + 1];
        const int skip = __float_as_int(n1.z), info = __float_as_int(n1.w);
        const int depth = (info >> 26) & 31, count = (info >> 24) & 3, first = info & 0xffffff;
        const bool hit = box_hit_ieee(n0, n1, o, chain_at<FAST>(b.rot, depth + 1, c1, c2, c3, c4, mode));
        if (hit && count != 0) {
            const V3 dt = chain_at<FAST>(b.rot, depth + 3, c1, c2, c3, c4, mode);
            const int before = best_slot;
            const float t_before = best_t;
            leaf_hit<FAST, PRIMARY>(rec, first, count, o, dt, best_t, best_slot);
            if (best_slot != before || __float_as_uint(best_t) != __float_as_uint(t_before)) dir_used = dt;
        }
        i = (hit && count == 0) ? i + 1 : skip;
    }
}

// (*world)->hit(r, 0.001, FLT_MAX, rec) with world = the BVH root, for the lane's ray (o, d0);
// d1 = unit_vector(d0).  FAST: bounded scene (crt_bvh.cpp: bvh_fast_ok).  force_general: test hook.
// PRIMARY: o is the camera origin and `rec` the per-frame exact records; else any origin and `rec` = the
// slot-ordered base records.  dir_used: direction of the winning triangle's test (for rec.p).
template <bool FAST, bool PRIMARY = true>
__device__ __forceinline__ void bvh_trace(const BvhView& b, const float* __restrict__ rec, V3 o, V3 d1, int force_mode,
                                          float& best_t, int& best_slot, V3& dir_used, int* __restrict__ stack = nullptr,
                                          int stack_stride = 0) {
    const V3 c1 = rot0_exact(b.rot, d1);
    const V3 c2 = chain_next<FAST>(b.rot, c1);
    const V3 c3 = chain_next<FAST>(b.rot, c2);
    const V3 c4 = chain_next<FAST>(b.rot, c3);
    best_t = FLT_MAX;
    best_slot = -1;
    const bool fixed3 = same_bits(c4, c3);
    // ordinary: every component of the directions is at least 2^-40 in magnitude (and finite)
    bool ordinary = FAST && force_mode == 0;
    if (ordinary) {
        const float lo = fminf(fminf(fminf(fabsf(c1.x), fabsf(c1.y)), fminf(fabsf(c1.z), fabsf(c2.x))),
                               fminf(fminf(fabsf(c2.y), fabsf(c2.z)), fminf(fminf(fabsf(c3.x), fabsf(c3.y)), fabsf(c3.z))));
        ordinary = lo >= 9.094947017729282e-13f && fminf(fminf(fabsf(c4.x), fabsf(c4.y)), fabsf(c4.z)) >= 9.094947017729282e-13f;
        if (!PRIMARY) {   // any origin: a coordinate difference below the fast division's range needs a tiny nonzero component
            const float ax = fabsf(o.x), ay = fabsf(o.y), az = fabsf(o.z);
            ordinary = ordinary && (ax == 0.0f || ax >= 1.3877787807814457e-17f) && (ay == 0.0f || ay >= 1.3877787807814457e-17f) &&
                       (az == 0.0f || az >= 1.3877787807814457e-17f) && fmaxf(fmaxf(ax, ay), az) <= 1099511627776.0f;
        }
    }
    dir_used = c3;
    const bool pairs = stack != nullptr && b.pairs != nullptr;
    if (!pairs && __builtin_amdgcn_ballot_w64(!(ordinary && fixed3)) == 0ull) {   // skip links, every lane on the hot chain
        bvh_walk_fast<PRIMARY>(b, rec, o, c1, c2, c3, best_t, best_slot);
        return;
    }
    // some lane is off the hot chain (or the pair walk serves them anyway): one more chain element decides how it goes on
    V3 c5 = c4;
    int mode = 0;
    if (__builtin_amdgcn_ballot_w64(!fixed3) != 0ull || force_mode != 0) {
        c5 = chain_next<FAST>(b.rot, c4);
        mode = fixed3 ? 0 : (same_bits(c5, c4) ? 0 : (same_bits(c5, c3) ? 1 : 2));
    }
    if (force_mode == 2) mode = 2;   // test hook: every lane iterates F from c4 (must give the same pixels)
    if (pairs) {
        PairDirs pd;
        pd.c2 = c2; pd.c3 = c3; pd.c4 = c4;
        pd.y2 = V3{reciprocal<true>(c2.x), reciprocal<true>(c2.y), reciprocal<true>(c2.z)};
        pd.y3 = V3{reciprocal<true>(c3.x), reciprocal<true>(c3.y), reciprocal<true>(c3.z)};
        pd.y4 = V3{reciprocal<true>(c4.x), reciprocal<true>(c4.y), reciprocal<true>(c4.z)};
        pd.cyc = mode == 1;
        bool take = ordinary && mode != 2;
        if (!bvh_walk_pairs<PRIMARY>(b, rec, o, c1, pd, take, stack, stack_stride, best_t, best_slot, dir_used)) {
            take = false;          // ran out of stack: start over below
            best_t = FLT_MAX;
            best_slot = -1;
        }
        if (__builtin_amdgcn_ballot_w64(!take) != 0ull) {   // the few lanes the pair walk cannot serve: alone, the literal way
            if (!take) bvh_walk_general<FAST, PRIMARY>(b, rec, o, c1, c2, c3, c4, mode, best_t, best_slot, dir_used);
        }
        return;
    }
    bvh_walk_general<FAST, PRIMARY>(b, rec, o, c1, c2, c3, c4, mode, best_t, best_slot, dir_used);
}

// The hot direction chain of the tiled kernels' BVH mode, per ray: c2 = F(c1), c3 = F(c2) with c1 = rotate(unit_vector(d0), 0)
// = d1.  For a ray whose components are all ordinary (finite, |component| >= 2^-40) rotate(., 0) is the identity bit for bit
// (x * 1 + y * (+-0) + z * (+-0) = x for x != 0) and unit_vector_fast is the IEEE result, so the chain is three guard-free
// normalisations; ONE test at the end covers it: |c3 components| >= 2^-39 (and finite) puts the same components of d1 and c2
// above 2^-40 — each normalisation moves a component by a few ulp of itself, so a tiny, zero or non-finite component stays
// what it was — and c4 == c3 says the chain has settled.  `hot` false: the caller recomputes the chain literally
// (rot0_exact, chain_next) for that lane; c3 must not be used for it.  Rounds 2-3 ran the guarded forms: four rotate
// guards, three range guards and a nine-component minimum per sample.
__device__ __forceinline__ bool hot_chain(V3 d1, V3& c3) {
    const V3 c2 = unit_vector_fast_ng(d1);
    c3 = unit_vector_fast_ng(c2);
    const V3 c4 = unit_vector_fast_ng(c3);
    const float lo = fminf(fminf(fabsf(c3.x), fabsf(c3.y)), fabsf(c3.z)), hi = fmaxf(fmaxf(fabsf(c3.x), fabsf(c3.y)), fabsf(c3.z));
    return same_bits(c4, c3) && lo >= 1.8189894035458565e-12f && hi <= 2.0f;   // 2^-39; (NaN fails both comparisons)
}

// ---- the chain of boxes for the tiled kernel's BVH mode ----------------------------------------------
// A triangle is reached through the tree iff the boxes of its leaf and of all the leaf's ancestors
// pass AABB::hit for this ray (each with the direction of its depth; on the hot chain c1 for the
// root, c2 for its children, c3 below).  Most of those boxes are far larger than an 8x8-pixel tile:
// TileCone bounds the directions of ALL rays of a tile, box_all_hit proves for a box that every
// such ray passes the slab test with a margin no rounding can eat, and those levels are skipped.

struct TileCone {      // per tile: component-wise bounds of c_k = F^k(d0) over every ray of the tile, k >= 1
    V3 lo, hi;
    bool usable;       // false: no bound (degenerate camera geometry): every level is tested
};

// (u, v) of a tile's rays lie in [x0/nx, (x0+8)/nx] x [y0/ny, (y0+8)/ny] up to rounding; one pixel of margin each way plus
// the camera's position slack covers the float evaluation of d0.  c_k is d0 / |d0| up to a few ulp.
__device__ __forceinline__ TileCone tile_cone(const RenderParams& p, int x0, int y0_global) {
    const CameraPOD& c = p.cam;
    const float u0 = ((float)x0 - 1.0f) * p.rnx - p.pos_du, u1 = ((float)x0 + 9.0f) * p.rnx + p.pos_du;
    const float v0 = ((float)y0_global - 1.0f) * p.rny - p.pos_dv, v1 = ((float)y0_global + 9.0f) * p.rny + p.pos_dv;
    float lo[3], hi[3], len_hi = 0.0f;
#pragma unroll
    for (int a = 0; a < 3; ++a) { lo[a] = 3.0e38f; hi[a] = -3.0e38f; }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float uu = (k & 1) ? u1 : u0, vv = (k & 2) ? v1 : v0;
        float d[3], l2 = 0.0f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            d[a] = ((c.llc[a] + uu * c.horizontal[a]) + vv * c.vertical[a]) - c.origin[a];
            lo[a] = fminf(lo[a], d[a]);
            hi[a] = fmaxf(hi[a], d[a]);
            l2 += d[a] * d[a];
        }
        len_hi = fmaxf(len_hi, sqrtf(l2));
    }
    float dc[3], lc2 = 0.0f, var = 0.0f;   // |d0| >= |d0(centre)| - (half extent of the affine map)
    const float uc = 0.5f * (u0 + u1), vc = 0.5f * (v0 + v1), hu = 0.5f * (u1 - u0), hv = 0.5f * (v1 - v0);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        dc[a] = ((c.llc[a] + uc * c.horizontal[a]) + vc * c.vertical[a]) - c.origin[a];
        lc2 += dc[a] * dc[a];
    }
    var = hu * p.cam_hlen + hv * p.cam_vlen;                         // |horizontal|, |vertical|: camera-only, formed on the host
    const float eps = p.cam_thr;                                     // 2^-20 mag: float evaluation of d0 (host)
    const float len_lo = (sqrtf(lc2) - var) * 0.99999f - 2.0f * eps;
    len_hi = len_hi * 1.00001f + 2.0f * eps;
    TileCone t;
    t.usable = len_lo > 0.5f * len_hi && len_hi < 1.0e30f && len_lo == len_lo;
    const float r_lo = 1.0f / len_hi, r_hi = 1.0f / fmaxf(len_lo, 1.0e-30f);
    float o_lo[3], o_hi[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float l = lo[a] - eps, h = hi[a] + eps;
        const float cl = fminf(l * r_lo, l * r_hi), ch = fmaxf(h * r_lo, h * r_hi);
        o_lo[a] = cl - (1.0e-6f + 1.0e-5f * fabsf(cl));             // c_k vs d0/|d0|: a few ulp; generous
        o_hi[a] = ch + (1.0e-6f + 1.0e-5f * fabsf(ch));
    }
    t.lo = V3{o_lo[0], o_lo[1], o_lo[2]};
    t.hi = V3{o_hi[0], o_hi[1], o_hi[2]};
    return t;
}

// One axis of the conservative slab test: bounds of t_enter (upper) and t_exit (lower) over every direction
// component in [dlo, dhi].  x0 = min - o, x1 = max - o.  false: no bound for this axis.
__device__ __forceinline__ bool slab_bounds(float x0, float x1, float dlo, float dhi, float& enter_up, float& exit_lo) {
    if (dlo > 1.0e-4f) {                 // all directions positive: enter = x0 / d, exit = x1 / d, monotone in d
        enter_up = fmaxf(x0 / dlo, x0 / dhi);
        exit_lo = fminf(x1 / dlo, x1 / dhi);
        return true;
    }
    if (dhi < -1.0e-4f) {                // all negative: enter = x1 / d, exit = x0 / d
        enter_up = fmaxf(x1 / dlo, x1 / dhi);
        exit_lo = fminf(x0 / dlo, x0 / dhi);
        return true;
    }
    // The interval touches 0: the origin must lie strictly inside the slab; then enter < 0 < exit for either sign
    // (and -inf / +inf for a zero component), the extremes being at the interval's ends.
    const float m = 1.0e-5f * (fabsf(x0) + fabsf(x1)) + 1.0e-30f;
    if (!(x0 < -m && x1 > m)) return false;
    enter_up = -1.0e-30f;
    exit_lo = 3.0e38f;
    if (dhi > 0.0f) { exit_lo = fminf(exit_lo, x1 / dhi); }
    if (dlo < 0.0f) { exit_lo = fminf(exit_lo, x0 / dlo); }
    return true;
}

// Every ray of the tile (directions within the cone, origin o) passes AABB::hit on this box, with a relative
// margin of 1e-4 on the slab parameters — four orders of magnitude above the rounding of the per-ray test.
__device__ __forceinline__ bool box_all_hit(float4 n0, float4 n1, V3 o, const TileCone& t) {
    if (!t.usable) return false;
    float ex, lx, ey, ly, ez, lz;
    if (!slab_bounds(n0.x - o.x, n0.w - o.x, t.lo.x, t.hi.x, ex, lx)) return false;
    if (!slab_bounds(n0.y - o.y, n1.x - o.y, t.lo.y, t.hi.y, ey, ly)) return false;
    if (!slab_bounds(n0.z - o.z, n1.y - o.z, t.lo.z, t.hi.z, ez, lz)) return false;
    const float enter = fmaxf(fmaxf(ex, ey), fmaxf(ez, 0.001f));
    const float leave = fminf(fminf(lx, ly), lz);
    const float margin = 1.0e-4f * (fabsf(enter) + fabsf(leave)) + 1.0e-6f;
    return leave - enter > margin && leave < 1.0e30f;   // NaN anywhere -> false
}

#ifndef CRT_CHAIN_BATCH
#define CRT_CHAIN_BATCH 4
#endif
constexpr int kChainBatch = CRT_CHAIN_BATCH;   // boxes tested per round of loads in chain_ok

// The chain test for one lane's winner `slot`: levels [first_level, depth] of its path (the levels above are
// proven for the whole tile).  Loads come in rounds: the path's node indices, then up to kChainBatch nodes at once.
__device__ __forceinline__ bool chain_ok(const BvhTiles& bt, int slot, int tile_first_level, const int* __restrict__ tile_path, V3 o, V3 c1,
                                         V3 c2, V3 c3) {
    const int info = bt.slot_leaf[slot];
    const int ord = info & 0xffffff, depth = (info >> 24) & 31;
    const int* __restrict__ path = bt.anc + (size_t)ord * bt.anc_stride;
    // the first level to test: where this path leaves the tile's reference path, or the reference's first unproven level
    int first = tile_first_level;
    for (int k = 0; k < tile_first_level && k <= depth; k += 4) {
        const int4 ids = *reinterpret_cast<const int4*>(path + k);
        const int4 ref = *reinterpret_cast<const int4*>(tile_path + k);
        int d = 4;
        if (ids.w != ref.w) d = 3;
        if (ids.z != ref.z) d = 2;
        if (ids.y != ref.y) d = 1;
        if (ids.x != ref.x) d = 0;
        if (d < 4) { first = min(first, k + d); break; }
    }
    const V3 y3 = V3{reciprocal<true>(c3.x), reciprocal<true>(c3.y), reciprocal<true>(c3.z)};
    bool ok = true;
    for (int k0 = first; k0 <= depth && ok; k0 += kChainBatch) {
        int node[kChainBatch];
        float4 a[kChainBatch], b[kChainBatch];
#pragma unroll
        for (int j = 0; j < kChainBatch; ++j) node[j] = path[min(k0 + j, depth)];
#pragma unroll
        for (int j = 0; j < kChainBatch; ++j) {
            a[j] = bt.nodes[2 * (size_t)node[j]];
            b[j] = bt.nodes[2 * (size_t)node[j] + 1];
        }
#pragma unroll
        for (int j = 0; j < kChainBatch; ++j) {
            const int k = min(k0 + j, depth);
            V3 d = c3, y = y3;
            if (k < 2) {
                d = sel3(k == 0, c1, c2);
                y = V3{reciprocal<true>(d.x), reciprocal<true>(d.y), reciprocal<true>(d.z)};
            }
            ok = ok && box_hit_fast(a[j], b[j], o, d, y);
        }
    }
    return ok;
}

// The same for a lane off the hot chain (its direction chain has not settled by c3, or a component is tiny): every
// level with the chain element of its depth, literal IEEE slab test, no level skipped.  Per lane, no wave votes.
__device__ __forceinline__ bool chain_ok_general(const BvhTiles& bt, int slot, V3 o, V3 c1, V3 c2, V3 c3, V3 c4, int mode) {
    const int info = bt.slot_leaf[slot];
    const int ord = info & 0xffffff, depth = (info >> 24) & 31;
    const int* __restrict__ path = bt.anc + (size_t)ord * bt.anc_stride;
    for (int k = 0; k <= depth; ++k) {
        const int node = path[k];
        if (!box_hit_ieee(bt.nodes[2 * (size_t)node], bt.nodes[2 * (size_t)node + 1], o, chain_at<true>(bt.rot, k + 1, c1, c2, c3, c4, mode)))
            return false;
    }
    return true;
}

// The winner of one lane off the hot chain among the entries `lst[0..cnt)` whose bit is set in its candidate masks
// (mask word j covers entries 32 j .. 32 j + 31), minus the excluded ones: each entry is tested with the chain element of
// ITS leaf's depth (F^(depth + 3)), from the exact record in global memory.  Returns the tie key q (or -1) and the slot.
__device__ __forceinline__ int general_winner(const BvhTiles& bt, const float* __restrict__ rec, const uint32_t* __restrict__ lst, int cnt,
                                              const unsigned* masks, const unsigned* excl, int n_words, V3 o, V3 c1, V3 c2, V3 c3, V3 c4,
                                              int mode, int& slot_out) {
    float best_t = FLT_MAX;
    int best_q = -1, best_slot = -1;
    for (int j = 0; j < n_words; ++j) {
        unsigned m = masks[j] & ~excl[j];
        while (m != 0u) {
            const int k = 32 * j + __builtin_ctz(m);
            m &= m - 1u;
            if (k >= cnt) break;
            const int slot = (int)(lst[k] & 0x7fffffffu);
            const int depth = (bt.slot_leaf[slot] >> 24) & 31;
            float t;
            if (!tri_hit_lane<true, true>(rec, slot, o, chain_at<true>(bt.rot, depth + 3, c1, c2, c3, c4, mode), t)) continue;
            const int q = 0x7fffffff - bt.partner[slot];
            if (t < best_t || (t == best_t && q < best_q)) {
                best_t = t;
                best_q = q;
                best_slot = slot;
            }
        }
    }
    slot_out = best_slot;
    return best_q;
}

// The same over the WHOLE list of a tile (no candidate masks), minus up to 8 excluded slots: heavy tiles, whose masks
// live in other wavefronts.
__device__ __forceinline__ int general_winner_all(const BvhTiles& bt, const float* __restrict__ rec, const uint32_t* __restrict__ lst, int cnt,
                                                  const int* excl, int n_excl, V3 o, V3 c1, V3 c2, V3 c3, V3 c4, int mode, int& slot_out) {
    float best_t = FLT_MAX;
    int best_q = -1, best_slot = -1;
    for (int k = 0; k < cnt; ++k) {
        const int slot = (int)(lst[k] & 0x7fffffffu);
        bool skip = false;
        for (int j = 0; j < n_excl; ++j) skip = skip || excl[j] == slot;
        if (skip) continue;
        const int depth = (bt.slot_leaf[slot] >> 24) & 31;
        float t;
        if (!tri_hit_lane<true, true>(rec, slot, o, chain_at<true>(bt.rot, depth + 3, c1, c2, c3, c4, mode), t)) continue;
        const int q = 0x7fffffff - bt.partner[slot];
        if (t < best_t || (t == best_t && q < best_q)) {
            best_t = t;
            best_q = q;
            best_slot = slot;
        }
    }
    slot_out = best_slot;
    return best_q;
}

// ---- the chain of boxes of a winner on the hot chain, without walking it (round 4) ------------------------------------
// AABB::hit (aabb.h:30-42) per axis: q(face) = fl(fl(face - o) / d), t0 = ffmin(q(min), q(max)), t1 = ffmax(..); t_min = ffmax(t0,
// t_min); t_max = ffmin(t1, t_max); false as soon as t_max <= t_min.  Both roundings are monotone in `face`, so for two boxes
// L inside B (min_B <= min_L, max_L <= max_B on every axis) and the SAME origin and direction bits, the computed slab of B
// contains the computed slab of L on every axis: t0_B <= t0_L and t1_B >= t1_L, hence after every axis t_min_B <= t_min_L and
// t_max_B >= t_max_L — if L is not rejected, B is not (all values finite: hot lane, bounded scene; BVHNode::hit hands the same
// t_min = 0.001, t_max = FLT_MAX down the tree, bvh.h:160-190).  The boxes of a leaf's ancestors contain the leaf's box
// EXACTLY: surrounding_box is min / max of the children's bounds (aabb.h:58-66) and TransformAABB with rotation 0, position 0
// returns its argument (transform.h:22-31; crt_bvh.cpp transform_aabb; tests/test_bvh.py checks the nesting bit for bit), at
// build time and after every refit.  And on the hot chain every node at depth >= 2 sees the same direction c3.  So for a
// leaf at depth >= 2:   the leaf's box passes with c3   ==>   every ancestor at depth >= 2 passes,
// and the chain test is the leaf's box (staged next to the entry) + the root with c1 + the root's child on the leaf's side
// with c2 — three nodes of the whole tree, which the tile's cone proves for all its rays almost everywhere.  Rounds 1-3
// loaded and tested every unproven level per winner (chain_ok), or proved the chain from a hit point strictly inside the
// leaf's box — which fails wherever the surface faces the camera along an axis (thin boxes, the hit next to a face of EVERY
// box around it): the tiles there were the last of the frame to end.

// Bits 0..2: every ray of the tile passes the root's box / its left child's / its right child's (box_all_hit with the
// tile's cone); 0 when the root is a leaf.
__device__ __forceinline__ unsigned tile_top_proven(const BvhTiles& bt, V3 o, const TileCone& cone) {
    const float4 r0 = bt.nodes[0], r1 = bt.nodes[1];
    if (((__float_as_int(r1.w) >> 24) & 3) != 0) return 0u;
    const float4 l0 = bt.nodes[2], l1 = bt.nodes[3];
    const int right = __float_as_int(l1.z);
    const float4 q0 = bt.nodes[2 * (size_t)right], q1 = bt.nodes[2 * (size_t)right + 1];
    return (box_all_hit(r0, r1, o, cone) ? 1u : 0u) | (box_all_hit(l0, l1, o, cone) ? 2u : 0u) | (box_all_hit(q0, q1, o, cone) ? 4u : 0u);
}

// Staged per list entry: the box of its leaf and flag = (leaf under the root's RIGHT child) | (leaf at depth < 2) << 1.
__device__ __forceinline__ void stage_leaf(const BvhTiles& bt, int slot, int& flag, float4& box0, float2& box1) {
    const int info = bt.slot_leaf[slot];
    const int depth = (info >> 24) & 31;
    const int* __restrict__ path = bt.anc + (size_t)(info & 0xffffff) * bt.anc_stride;
    const int leaf = path[depth];
    flag = ((depth >= 1 && path[1] != 1) ? 1 : 0) | (depth < 2 ? 2 : 0);
    const float4 l0 = bt.nodes[2 * (size_t)leaf], l1 = bt.nodes[2 * (size_t)leaf + 1];
    box0 = l0;
    box1 = make_float2(l1.x, l1.y);
}

// A box no ray from `o` passes, whatever its direction: min == max on an axis (an axis-aligned triangle; a ground plane) and o
// off that plane — both quotients of the axis are the same finite or infinite value q, so t_min >= q >= t_max after it
// (aabb.h:40).  With o ON the plane the quotients are 0 / d: zero, or NaN for d = 0, which ffmin / ffmax drop: not decided here.
__device__ __forceinline__ bool box_never_hit(float4 box0, float2 box1, V3 o) {
    return (box0.x == box0.w && o.x != box0.x) || (box0.y == box1.x && o.y != box0.y) || (box0.z == box1.y && o.z != box0.z);
}

// Hot lane, leaf at depth >= 2: its box with c3.
__device__ __forceinline__ bool leaf_box_hit(float4 box0, float2 box1, V3 o, V3 c3) {
    const V3 y3 = V3{reciprocal<true>(c3.x), reciprocal<true>(c3.y), reciprocal<true>(c3.z)};
    return box_hit_fast(box0, make_float4(box1.x, box1.y, 0.0f, 0.0f), o, c3, y3);
}
// Does this entry need more than its leaf's box?  (top: tile_top_proven)
__device__ __forceinline__ bool chain_needs_top(int flag, unsigned top) {
    return (flag & 2) != 0 || (top & 1u) == 0u || (top & ((flag & 1) ? 4u : 2u)) == 0u;
}
// The rest of the chain of a hot lane whose leaf passed: the root with c1 = d1 and its child on the leaf's side with
// c2 = F(d1) (rotate(., 0) is the identity on a hot lane's directions); a leaf at depth < 2: the literal chain.
__device__ __forceinline__ bool chain_top_ok(const BvhTiles& bt, int slot, int flag, V3 o, V3 d1, V3 c3) {
    const V3 c2 = unit_vector_fast_ng(d1);
    if (flag & 2) return chain_ok(bt, slot, 0, nullptr, o, d1, c2, c3);
    const float4 r0 = bt.nodes[0], r1 = bt.nodes[1];
    const int child = (flag & 1) ? __float_as_int(bt.nodes[3].z) : 1;
    const float4 m0 = bt.nodes[2 * (size_t)child], m1 = bt.nodes[2 * (size_t)child + 1];
    const V3 y1 = V3{reciprocal<true>(d1.x), reciprocal<true>(d1.y), reciprocal<true>(d1.z)};
    const V3 y2 = V3{reciprocal<true>(c2.x), reciprocal<true>(c2.y), reciprocal<true>(c2.z)};
    return box_hit_fast(r0, r1, o, d1, y1) && box_hit_fast(m0, m1, o, c2, y2);
}
// The same for a winner that is not staged (the unstaged rest of a shared tile's slice): leaf and flag from the tables.
__device__ __forceinline__ bool chain_ok_hot(const BvhTiles& bt, int slot, unsigned top, V3 o, V3 d1, V3 c3) {
    int flag;
    float4 b0;
    float2 b1;
    stage_leaf(bt, slot, flag, b0, b1);
    if ((flag & 2) == 0 && !leaf_box_hit(b0, b1, o, c3)) return false;
    return !chain_needs_top(flag, top) || chain_top_ok(bt, slot, flag, o, d1, c3);
}

// The render kernel with world = the BVH: lane = pixel, wavefront = 8x8 tile, block = 16x16.
template <bool FAST>
__global__ void __launch_bounds__(256)
render_bvh_kernel(RenderParams p, BvhView b, const float* __restrict__ rec, const float* __restrict__ shade_n,
                  uint32_t* __restrict__ state, float* __restrict__ rgb, int force_mode) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int x = blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7);
    const int ly = blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3);
    const bool valid = x < p.nx && ly < p.row_count;                  // render.h:109
    if (__builtin_amdgcn_ballot_w64(valid) == 0ull) return;
    const V3 o = V3{p.cam.origin[0], p.cam.origin[1], p.cam.origin[2]};
    // lanes off the image replay a valid pixel (unstored): the traversal votes need whole wavefronts.  Primary rays of one
    // tile walk the same nodes: the skip-link walk (no pair records: measured 2.5 against 2.7 ms per 4-spp C3 frame)
    render_pixel<FAST, false, false>(p, min(x, p.nx - 1), min(ly, p.row_count - 1), shade_n, state, rgb, valid,
                                     [&](V3 d1, V3, float& closest, int& hit_idx) {
        V3 dir_used;
        bvh_trace<FAST>(b, rec, o, d1, force_mode, closest, hit_idx, dir_used);
    });
}

hipError_t launch_render_bvh(const RenderParams& p, const float4* d_nodes, int n_nodes, const float rot[9], const float* d_rec,
                             const float* d_shade_n, uint32_t* d_state, float* d_rgb, bool fast, int force_mode,
                             hipStream_t stream) {
    if (p.row_count <= 0 || p.nx <= 0) return hipSuccess;
    BvhView b;
    b.nodes = d_nodes;
    b.n_nodes = n_nodes;
    for (int k = 0; k < 9; ++k) b.rot[k] = rot[k];
    dim3 block(256), grid((unsigned)((p.nx + 15) / 16), (unsigned)((p.row_count + 15) / 16));
    if (fast)
        hipLaunchKernelGGL(render_bvh_kernel<true>, grid, block, 0, stream, p, b, d_rec, d_shade_n, d_state, d_rgb, force_mode);
    else
        hipLaunchKernelGGL(render_bvh_kernel<false>, grid, block, 0, stream, p, b, d_rec, d_shade_n, d_state, d_rgb, force_mode);
    return hipGetLastError();
}

// ------------------------------------------------------------------ refit (BVHNode::UpdateBVH, bvh.h:135-158)
// One launch per tree level, deepest first (the reference recurses in one device thread).
//   leaf:  childList->bounding_box(0, 1, box) (hitable_list.h:94-118): surrounding_box over the
//          triangles' GetBV = Triangle::bounding_box (triangle.h:102-115) + TransformAABB (transform.h:22-31)
//   inner: surrounding_box(right->GetBV, left->GetBV), GetBV = the child's box + TransformAABB
struct BoxF {
    float mn[3], mx[3];
};

__device__ __forceinline__ BoxF transform_aabb(const float* m, BoxF bx) {   // transform.h:22-31 with rotation 0, position 0
    const V3 rmin = rot0(m, V3{bx.mn[0], bx.mn[1], bx.mn[2]}), rmax = rot0(m, V3{bx.mx[0], bx.mx[1], bx.mx[2]});
    BoxF o;
    o.mn[0] = fminf(rmin.x, rmax.x) + 0.0f; o.mn[1] = fminf(rmin.y, rmax.y) + 0.0f; o.mn[2] = fminf(rmin.z, rmax.z) + 0.0f;
    o.mx[0] = fmaxf(rmin.x, rmax.x) + 0.0f; o.mx[1] = fmaxf(rmin.y, rmax.y) + 0.0f; o.mx[2] = fmaxf(rmin.z, rmax.z) + 0.0f;
    return o;
}

__device__ __forceinline__ BoxF surrounding(BoxF a, BoxF b) {   // aabb.h:58-66
    BoxF o;
#pragma unroll
    for (int k = 0; k < 3; ++k) { o.mn[k] = fminf(a.mn[k], b.mn[k]); o.mx[k] = fmaxf(a.mx[k], b.mx[k]); }
    return o;
}

struct RotM {
    float m[9];
};

__global__ void __launch_bounds__(256)
bvh_refit_level_kernel(float4* __restrict__ nodes, const int* __restrict__ level_nodes, int n_level,
                       const float* __restrict__ base /* slot order */, RotM rot) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_level) return;
    const int i = level_nodes[j];
    const float4 n1 = nodes[2 * (size_t)i + 1];
    const int info = __float_as_int(n1.w);
    const int count = (info >> 24) & 3, first = info & 0xffffff;
    BoxF box;
    if (count != 0) {
        for (int t = 0; t < count; ++t) {
            const float* v = base + (size_t)(first + t) * kBaseFloats;
            BoxF tb;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                tb.mn[k] = fminf(v[k], fminf(v[3 + k], v[6 + k]));
                tb.mx[k] = fmaxf(v[k], fmaxf(v[3 + k], v[6 + k]));
            }
            tb = transform_aabb(rot.m, tb);
            box = (t == 0) ? tb : surrounding(box, tb);
        }
    } else {
        const int l = i + 1;
        const int r = __float_as_int(nodes[2 * (size_t)l + 1].z);   // the left child's skip = the right child
        const float4 l0 = nodes[2 * (size_t)l], l1 = nodes[2 * (size_t)l + 1], r0 = nodes[2 * (size_t)r], r1 = nodes[2 * (size_t)r + 1];
        BoxF bl = {{l0.x, l0.y, l0.z}, {l0.w, l1.x, l1.y}}, br = {{r0.x, r0.y, r0.z}, {r0.w, r1.x, r1.y}};
        box = surrounding(transform_aabb(rot.m, br), transform_aabb(rot.m, bl));   // bvh.h:155: (box_right, box_left)
    }
    nodes[2 * (size_t)i] = make_float4(box.mn[0], box.mn[1], box.mn[2], box.mx[0]);
    nodes[2 * (size_t)i + 1] = make_float4(box.mx[1], box.mx[2], n1.z, n1.w);
}

// Pair records from the flat nodes (after the build's upload and after every refit): lane = inner node; the boxes of its two
// children are copied next to each other (bvh_walk_pairs); the references p3 are static (host, crt_bvh.cpp).
__global__ void __launch_bounds__(256)
bvh_pairs_kernel(const float4* __restrict__ nodes, float4* __restrict__ pairs, const int* __restrict__ pair_flat, int n_pairs) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_pairs) return;
    const int l = pair_flat[2 * (size_t)j], r = pair_flat[2 * (size_t)j + 1];
    const float4 l0 = nodes[2 * (size_t)l], l1 = nodes[2 * (size_t)l + 1], r0 = nodes[2 * (size_t)r], r1 = nodes[2 * (size_t)r + 1];
    pairs[4 * (size_t)j + 0] = l0;                                          // L.min.xyz, L.max.x
    pairs[4 * (size_t)j + 1] = make_float4(l1.x, l1.y, r0.x, r0.y);         // L.max.yz, R.min.xy
    pairs[4 * (size_t)j + 2] = make_float4(r0.z, r0.w, r1.x, r1.y);         // R.min.z, R.max.xyz
}

hipError_t launch_bvh_pairs(const float4* d_nodes, float4* d_pairs, const int* d_pair_flat, int n_pairs, hipStream_t stream) {
    if (n_pairs <= 0 || d_pairs == nullptr) return hipSuccess;
    hipLaunchKernelGGL(bvh_pairs_kernel, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, stream, d_nodes, d_pairs, d_pair_flat, n_pairs);
    return hipGetLastError();
}

hipError_t launch_bvh_refit_level(float4* d_nodes, const int* d_level_nodes, int n_level, const float* d_base_slots,
                                  const float rot[9], hipStream_t stream) {
    if (n_level <= 0) return hipSuccess;
    RotM r;
    for (int k = 0; k < 9; ++k) r.m[k] = rot[k];
    hipLaunchKernelGGL(bvh_refit_level_kernel, dim3((unsigned)((n_level + 255) / 256)), dim3(256), 0, stream, d_nodes, d_level_nodes,
                       n_level, d_base_slots, r);
    return hipGetLastError();
}
