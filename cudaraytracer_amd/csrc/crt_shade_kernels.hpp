// crt_shade_kernels.hpp — the generic render kernel: any of the reference's three shaders on either
// world with per-triangle materials (included by crt_kernels.hip after crt_bvh_kernels.hpp).
//
// The reference selects the shader by (un)commenting one of core/render.h:119-121:
//     shade          core/render.h:48-67   the recursive path tracer (depth <= max_depth)
//     LambertShade   core/render.h:70-87   the live one
//     shade_normal   core/render.h:90-103
// and gives every triangle a Material* (shapes/triangle.h:54): Lambertian, Metal, Dielectric,
// DiffuseLight (material/material.h:57-161) over ConstantTexture (material/texture.h:12-22).
// LambertShade on the single-Lambertian scene of add_mesh_withNormal is served by the specialised
// kernels (tiled list path, BVH walk); everything else runs here:
//   * persistent wavefronts, lane = one pixel at a time (taken from a queue over the launch's pixels); a pixel's samples and
//     bounces are serial (RNG order), the BVH world's tree walks are shared by the wavefront (bvh_walk_coop);
//   * `shade` is recursive in the reference: emitted + attenuation * shade(scattered, depth - 1).
//     A material that scatters never emits (only DiffuseLight emits, and it never scatters), so a
//     level that recursed contributes  0 + attenuation * (deeper result);  the bounce loop keeps the
//     material index of every level in LDS and folds the levels back innermost first, which is the
//     reference's evaluation order operation for operation;
//   * secondary rays start anywhere: the BVH world walks the same flat tree with the triangle
//     test evaluated from the base records (tri_hit_base); the list world walks all triangles
//     with a wave-uniform index (scalar loads), as the reference's list does.
#pragma once

constexpr int kMaxDepthLevels = 32;   // shade(): max_depth + 1 levels of material indices per lane in LDS

// MaterialPOD: crt_device.hpp

struct ShadeScene {
    const float* base;            // n x 12 base records: list order (list world) or slot order (BVH world)
    const float* shade_n;         // n x 3 shading normals, same order
    const int* material_of;       // per triangle, same order; nullptr = all 0
    const MaterialPOD* materials;
    int n_tris;
    float rot[9];
};

struct HitInfo {
    float t;
    int idx;       // triangle (list order) or slot (BVH world); -1 = miss
    V3 p, normal;  // rec.p, rec.normal
};

// The list world for any ray: (*world)->hit = two nested Hitable::hit levels (list, triangle): the
// triangles see F^2(d); strict `<` in index order (hitable_list.h:76-91).  The triangle index is
// wave-uniform: base records arrive through scalar loads.
template <bool FAST>
__device__ __forceinline__ void list_trace_any(const ShadeScene& sc, V3 o, V3 d1, float& best_t, int& best_idx, V3& dir_used) {
    const V3 c1 = rot0_exact(sc.rot, d1);
    const V3 c2 = chain_next<FAST>(sc.rot, c1);
    dir_used = c2;
    best_t = FLT_MAX;
    best_idx = -1;
    for (int i = 0; i < sc.n_tris; ++i) {
        float t;
        if (tri_hit_base<FAST>(sc.base + (size_t)i * kBaseFloats, o, c2, t) && t < best_t) {
            best_t = t;
            best_idx = i;
        }
    }
}


// ---- the pair walk, shared by the wavefront -------------------------------------------------------------------
// With one ray per lane and a private stack (bvh_walk_pairs) a trip lasts as long as the LONGEST walk of its 64 rays —
// 195 steps for rays inside the mesh where the mean is 30 — with 28 of 64 lanes walking (tools/walk_stats.py), and a
// pixel behind glass needs 36 such trips one after the other: the launch waited 10 ms for those chains with the chip
// idle.  Here the wavefront's rays sit in LDS and the WORK is shared: two lists per wavefront, node items (ray, pair
// record) and triangle items (ray, slot); every round the 64 lanes take the 64 newest items of one list, whichever rays
// they belong to, and push what the tests yield.  A trip lasts (sum of its rays' steps) / 64 rounds, the last ray of a
// pixel gets all 64 lanes.  Nothing observable depends on the order: every leaf whose box and whose ancestors' boxes
// are hit is tested, each with the direction of its depth (the reference has no culling by the closest hit), and the
// results meet in one 64-bit key per ray under atomicMin — smallest t, among equal t the larger partner (bvh_walk_pairs'
// merge rule, a total order).
constexpr int kCoopNodeCap = 1024;   // node items per wavefront (newest first: at most 64 per level of the tree pending — more: the trip is redone by the literal walk)
constexpr int kCoopTriCap = 576;     // triangle items: fewer than 64 before a round of nodes, at most 8 more per lane
constexpr int kCoopRayVecs = 5;      // float4 per ray: (o, cyc) c3 y3 c4 y4
constexpr int kCoopWaveBytes = 64 * kCoopRayVecs * 16 + 64 * 8 + kCoopNodeCap * 4 + kCoopTriCap * 4;
struct CoopLds {
    float4* rays;                // [64][kCoopRayVecs]
    unsigned long long* best;    // [64] (ordered t, tie order) keys
    unsigned* nodes;             // [cap]  ray << 26 | class << 24 | pair index
    unsigned* tris;              // [kCoopTriCap]  ray << 26 | sees c3 << 25 | place in its leaf << 23 | slot
    int cap;                     // kCoopNodeCap (a test variant: 96, so that trips are abandoned)
};
struct PairStep {                // what one pair record yields for one ray
    bool go_l = false, go_r = false;   // inner children to follow
    int ref_l = 0, ref_r = 0;
    int cnt_l = 0, cnt_r = 0;          // triangles of hit leaf children (0: not a leaf, or missed)
};
struct PairRec {
    float4 p0, p1, p2, p3;
};
__device__ __forceinline__ PairRec pair_load(const BvhView& b, unsigned cur) {
    const float4* __restrict__ pr = b.pairs + 4 * (size_t)cur;
    return PairRec{pr[0], pr[1], pr[2], pr[3]};
}
// `has`: the lane holds an item (otherwise the result is empty, whatever the operands)
__device__ __forceinline__ PairStep pair_eval(const PairRec& r, bool has, V3 o, V3 d, V3 y, V3 dt) {
    const bool hit_l = has && box_hit_fast(r.p0, make_float4(r.p1.x, r.p1.y, 0.0f, 0.0f), o, d, y);
    const bool hit_r = has && box_hit_fast(make_float4(r.p1.z, r.p1.w, r.p2.x, r.p2.y), make_float4(r.p2.z, r.p2.w, 0.0f, 0.0f), o, d, y);
    PairStep s;
    s.ref_l = __float_as_int(r.p3.x);
    s.ref_r = __float_as_int(r.p3.y);
    // normal cones as in bvh_walk_pairs: a child none of whose triangles can pass triangle.h:61 is skipped like a missed box
    const bool try_l = hit_l && !cone_rejects(__float_as_uint(r.p3.z), dt), try_r = hit_r && !cone_rejects(__float_as_uint(r.p3.w), dt);
    s.go_l = try_l && s.ref_l >= 0;
    s.go_r = try_r && s.ref_r >= 0;
    s.cnt_l = (try_l && s.ref_l < 0) ? ((s.ref_l >> 24) & 3) : 0;
    s.cnt_r = (try_r && s.ref_r < 0) ? ((s.ref_r >> 24) & 3) : 0;
    return s;
}
// one node item of the shared walk: its ray's data from LDS (class -> which chain elements its boxes and triangles see)
struct CoopItem {
    unsigned ray, child_class;
    bool tri3;
    V3 o, d, y, dt;
};
__device__ __forceinline__ CoopItem coop_item(const CoopLds& L, unsigned item) {
    CoopItem c;
    c.ray = item >> 26;
    const unsigned cls = (item >> 24) & 3u;   // 1: depth 1 (k = 3)   2: even depth >= 2   3: odd depth >= 3
    const float4 ro = L.rays[c.ray * kCoopRayVecs];
    const bool cyc = ro.w != 0.0f;
    const bool odd = cls != 2u;                    // k = depth + 2 is odd
    const bool box3 = cls == 1u || (cyc && odd);   // pair_uses_c3
    c.tri3 = cyc && odd;                           // the triangles of a leaf child see k + 2
    const float4 rd = L.rays[c.ray * kCoopRayVecs + (box3 ? 1 : 3)], ry = L.rays[c.ray * kCoopRayVecs + (box3 ? 2 : 4)];
    const float4 rt = L.rays[c.ray * kCoopRayVecs + (c.tri3 ? 1 : 3)];
    c.o = V3{ro.x, ro.y, ro.z}; c.d = V3{rd.x, rd.y, rd.z}; c.y = V3{ry.x, ry.y, ry.z}; c.dt = V3{rt.x, rt.y, rt.z};
    c.child_class = cls == 2u ? 3u : 2u;
    return c;
}
__device__ __forceinline__ unsigned lanes_below(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}
// the whole wavefront: lanes without an item pass an empty PairStep
__device__ __forceinline__ void coop_push(const CoopLds& L, unsigned ray, unsigned child_class, bool tri3, const PairStep& s, unsigned& nn, unsigned& nt) {
    const unsigned long long ml = __builtin_amdgcn_ballot_w64(s.go_l), mr = __builtin_amdgcn_ballot_w64(s.go_r);
    const unsigned head = (ray << 26) | (child_class << 24);
    if (s.go_r) L.nodes[nn + lanes_below(mr)] = head | (unsigned)s.ref_r;
    const unsigned nr = (unsigned)__builtin_popcountll(mr);
    if (s.go_l) L.nodes[nn + nr + lanes_below(ml)] = head | (unsigned)s.ref_l;
    nn += nr + (unsigned)__builtin_popcountll(ml);
    const int c = s.cnt_l + s.cnt_r;   // 0..4
    const unsigned long long b0 = __builtin_amdgcn_ballot_w64((c & 1) != 0), b1 = __builtin_amdgcn_ballot_w64((c & 2) != 0),
                             b2 = __builtin_amdgcn_ballot_w64((c & 4) != 0);
    unsigned at = nt + lanes_below(b0) + 2u * lanes_below(b1) + 4u * lanes_below(b2);
    const unsigned thead = (ray << 26) | (tri3 ? 1u << 25 : 0u);
    if (s.cnt_l >= 1) L.tris[at++] = thead | (s.cnt_l == 2 ? 1u << 23 : 0u) | (unsigned)(s.ref_l & 0x7fffff);
    if (s.cnt_l == 2) L.tris[at++] = thead | (2u << 23) | (unsigned)((s.ref_l & 0x7fffff) + 1);
    if (s.cnt_r >= 1) L.tris[at++] = thead | (s.cnt_r == 2 ? 1u << 23 : 0u) | (unsigned)(s.ref_r & 0x7fffff);
    if (s.cnt_r == 2) L.tris[at++] = thead | (2u << 23) | (unsigned)((s.ref_r & 0x7fffff) + 1);
    nt += (unsigned)__builtin_popcountll(b0) + 2u * (unsigned)__builtin_popcountll(b1) + 4u * (unsigned)__builtin_popcountll(b2);
}
__device__ __forceinline__ void coop_sync() {   // LDS traffic between the lanes of ONE wavefront
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Called by ALL lanes of the wavefront.  `take`: this lane has a ray for this walk (ordinary components, chain served by
// PairDirs).  Returns false when the node list ran full: no lane has a result then (the caller walks the rays one by one).
template <bool PRIMARY>
__device__ __forceinline__ bool bvh_walk_coop(const BvhView& b, const float* __restrict__ rec, V3 o, V3 c1, const PairDirs& pd, bool take,
                                              const CoopLds& L, float& best_t, int& best_slot, V3& dir_used) {
    const unsigned lane = lanes_below(~0ull);
    bool alive = take;
    {   // the root: its own box sees c1 (depth 0, k = 1); a one-leaf tree has no pair records and does not come here
        const float4 n0 = b.nodes[0], n1 = b.nodes[1];
        const V3 y1 = V3{reciprocal<true>(c1.x), reciprocal<true>(c1.y), reciprocal<true>(c1.z)};
        alive = alive && box_hit_fast(n0, n1, o, c1, y1);
    }
    if (take) {
        float4* r = L.rays + lane * kCoopRayVecs;
        r[0] = make_float4(o.x, o.y, o.z, pd.cyc ? 1.0f : 0.0f);
        r[1] = make_float4(pd.c3.x, pd.c3.y, pd.c3.z, 0.0f);
        r[2] = make_float4(pd.y3.x, pd.y3.y, pd.y3.z, 0.0f);
        r[3] = make_float4(pd.c4.x, pd.c4.y, pd.c4.z, 0.0f);
        r[4] = make_float4(pd.y4.x, pd.y4.y, pd.y4.z, 0.0f);
    }
    L.best[lane] = ~0ull;
    unsigned nn = 0, nt = 0;   // wave-uniform list lengths
    {   // the root's record by the rays' own lanes, from registers: its children's boxes see c2 (k = 2), triangles of a leaf child k = 4
        const PairStep s = pair_eval(pair_load(b, 0u), alive, o, pd.c2, pd.y2, pd.c4);
        coop_push(L, lane, 1u, false, s, nn, nt);
    }
    coop_sync();
    bool ok = true;
    while (nn != 0u || nt != 0u) {
        if (nt >= 64u || nn == 0u) {   // a round of triangles
            const unsigned n = min(nt, 64u);
            nt -= n;
            walk_stat(b, 4, __builtin_amdgcn_ballot_w64(lane < n));
            if (lane < n) {
                const unsigned item = L.tris[nt + lane];
                const unsigned ray = item >> 26, slot = item & 0x7fffffu, place = (item >> 23) & 3u;
                const bool tri3 = (item & (1u << 25)) != 0u;
                const float4 ro = L.rays[ray * kCoopRayVecs], rd = L.rays[ray * kCoopRayVecs + (tri3 ? 1 : 3)];
                float t;
                if (tri_hit_lane<true, PRIMARY>(rec, (int)slot, V3{ro.x, ro.y, ro.z}, V3{rd.x, rd.y, rd.z}, t) && t < FLT_MAX) {   // hitable_list.h:84 against t_max = FLT_MAX
                    // BvhTiles::partner without the table: a leaf of two triangles swaps them, a single one is its own
                    const unsigned partner = place == 0u ? slot : (place == 1u ? slot + 1u : slot - 1u);
                    atomicMin(&L.best[ray], hit_key(t, (int)(((0xffffffu - partner) << 3) | (place << 1) | (tri3 ? 1u : 0u))));
                }
            }
        } else {                        // a round of pair records: the 128 newest, two per lane (both records in flight together)
            const unsigned n = min(nn, 128u);
            if (nn + n > (unsigned)L.cap) {
                walk_stat(b, 6, 0ull);   // [6] abandoned trips
                ok = false;
                break;
            }
            nn -= n;
            const bool has_a = lane < n, has_b = lane + 64u < n;
            walk_stat(b, 2, __builtin_amdgcn_ballot_w64(has_a));
            if (n > 64u) walk_stat(b, 2, __builtin_amdgcn_ballot_w64(has_b));
            // a lane without an item reads item 0's place and the root's record (hot), and its result is masked
            const unsigned item_a = L.nodes[has_a ? nn + lane : 0u], item_b = L.nodes[has_b ? nn + 64u + lane : 0u];
            const PairRec rec_a = pair_load(b, has_a ? (item_a & 0xffffffu) : 0u), rec_b = pair_load(b, has_b ? (item_b & 0xffffffu) : 0u);
            const CoopItem ca = coop_item(L, has_a ? item_a : 0u), cb = coop_item(L, has_b ? item_b : 0u);
            const PairStep sa = pair_eval(rec_a, has_a, ca.o, ca.d, ca.y, ca.dt);
            const PairStep sb = pair_eval(rec_b, has_b, cb.o, cb.d, cb.y, cb.dt);
            coop_push(L, ca.ray, ca.child_class, ca.tri3, sa, nn, nt);
            if (n > 64u) coop_push(L, cb.ray, cb.child_class, cb.tri3, sb, nn, nt);
        }
        coop_sync();
    }
    coop_sync();
    if (ok && take) {
        const unsigned long long key = L.best[lane];
        if (key != ~0ull) {
            const unsigned low = (unsigned)key, place = (low >> 1) & 3u, partner = 0xffffffu - (low >> 3);
            const unsigned slot = place == 0u ? partner : (place == 1u ? partner - 1u : partner + 1u);
            const V3 dt = sel3((low & 1u) != 0u, pd.c3, pd.c4);
            float t = FLT_MAX;
            (void)tri_hit_lane<true, PRIMARY>(rec, (int)slot, o, dt, t);   // the winner's own t, bit for bit (the key folds -0 into +0)
            best_t = t;
            best_slot = (int)slot;
            dir_used = dt;
        }
    }
    coop_sync();   // the next walk rewrites rays and keys
    return ok;
}

// ---- ONE ray whose direction chain has not settled, walked by the whole wavefront ------------------------------------
// 0.1 % of scattered rays have a chain that is neither fixed by c4 nor the 2-cycle (c3, c4): the element a node sees then
// depends on its depth beyond what an item's class says, and rounds 2-3 walked such a ray ALONE over the skip links (the
// literal loop: ~200 dependent node loads with 63 lanes idle, about 0.2 ms per ray — 30 % of the `shade` frame, measured by
// walking them as if settled: 5.7 -> 4.0 ms).  Here the wavefront walks that ray level by level: all items of a round are
// pair records of the SAME depth L, so their boxes see c_(L+2) and the triangles of their leaf children c_(L+4) — wave-
// uniform directions, advanced by one normalisation per level (chain_at's mode 2: F iterated from c4; on ordinary
// components unit_vector_fast and the skipped rotate(., 0) are the literal results bit for bit).  Same visited set, same
// tests and the same merge key as every other walk.  o, c1..c4 are wave-uniform (broadcast from the ray's lane).
// Returns false (nothing changed) when a level does not fit the lists or an element leaves the ordinary range: the
// caller then walks the ray the literal way.
template <bool PRIMARY>
__device__ __forceinline__ bool bvh_walk_one(const BvhView& b, const float* __restrict__ rec, V3 o, V3 c1, V3 c2, V3 c3, V3 c4,
                                             const CoopLds& L, float& t_out, int& slot_out, V3& dir_out) {
    const unsigned lane = lanes_below(~0ull);
    t_out = FLT_MAX;
    slot_out = -1;
    dir_out = c3;
    {
        const float4 n0 = b.nodes[0], n1 = b.nodes[1];
        if (((__float_as_int(n1.w) >> 24) & 3) != 0) return false;   // a one-leaf tree has no pair records
        const V3 y1 = V3{reciprocal<true>(c1.x), reciprocal<true>(c1.y), reciprocal<true>(c1.z)};
        if (!box_hit_fast(n0, n1, o, c1, y1)) return true;            // the root's box sees c1: a miss
    }
    const unsigned half = (unsigned)L.cap / 2u;                       // two lists of pair indices: this level's and the next one's
    V3 dB = c2, dMid = c3, dT = c4;                                   // the window c_(L+2), c_(L+3), c_(L+4) at L = 0
    unsigned cur = 0u, n_cur = 1u;
    if (lane == 0u) L.nodes[0] = 0u;
    unsigned long long my_key = ~0ull;                                // this lane's best hit so far
    float my_t = FLT_MAX;
    int my_slot = -1;
    V3 my_dir = c3;
    coop_sync();
    bool ok = true;
    while (n_cur != 0u) {
        const V3 yB = V3{reciprocal<true>(dB.x), reciprocal<true>(dB.y), reciprocal<true>(dB.z)};
        unsigned n_next = 0u, nt = 0u;
        for (unsigned base = 0u; base < n_cur; base += 64u) {
            const bool has = base + lane < n_cur;
            const unsigned pair = has ? L.nodes[cur * half + base + lane] : 0u;
            walk_stat(b, 2, __builtin_amdgcn_ballot_w64(has));
            const PairStep s = pair_eval(pair_load(b, pair), has, o, dB, yB, dT);
            if (n_next + 128u > half || nt + 256u > (unsigned)kCoopTriCap) { ok = false; break; }
            const unsigned long long ml = __builtin_amdgcn_ballot_w64(s.go_l), mr = __builtin_amdgcn_ballot_w64(s.go_r);
            unsigned* next = L.nodes + (cur ^ 1u) * half;
            if (s.go_r) next[n_next + lanes_below(mr)] = (unsigned)s.ref_r;
            const unsigned nr = (unsigned)__builtin_popcountll(mr);
            if (s.go_l) next[n_next + nr + lanes_below(ml)] = (unsigned)s.ref_l;
            n_next += nr + (unsigned)__builtin_popcountll(ml);
            const int c = s.cnt_l + s.cnt_r;   // 0..4 triangles of hit leaf children
            const unsigned long long b0 = __builtin_amdgcn_ballot_w64((c & 1) != 0), b1 = __builtin_amdgcn_ballot_w64((c & 2) != 0),
                                     b2 = __builtin_amdgcn_ballot_w64((c & 4) != 0);
            unsigned at = nt + lanes_below(b0) + 2u * lanes_below(b1) + 4u * lanes_below(b2);
            if (s.cnt_l >= 1) L.tris[at++] = (s.cnt_l == 2 ? 1u << 23 : 0u) | (unsigned)(s.ref_l & 0x7fffff);
            if (s.cnt_l == 2) L.tris[at++] = (2u << 23) | (unsigned)((s.ref_l & 0x7fffff) + 1);
            if (s.cnt_r >= 1) L.tris[at++] = (s.cnt_r == 2 ? 1u << 23 : 0u) | (unsigned)(s.ref_r & 0x7fffff);
            if (s.cnt_r == 2) L.tris[at++] = (2u << 23) | (unsigned)((s.ref_r & 0x7fffff) + 1);
            nt += (unsigned)__builtin_popcountll(b0) + 2u * (unsigned)__builtin_popcountll(b1) + 4u * (unsigned)__builtin_popcountll(b2);
        }
        if (!ok) break;
        coop_sync();
        for (unsigned base = 0u; base < nt; base += 64u) {            // this level's leaf children: their triangles see c_(L+4)
            walk_stat(b, 4, __builtin_amdgcn_ballot_w64(base + lane < nt));
            if (base + lane < nt) {
                const unsigned item = L.tris[base + lane], slot = item & 0x7fffffu, place = (item >> 23) & 3u;
                float t;
                if (tri_hit_lane<true, PRIMARY>(rec, (int)slot, o, dT, t) && t < FLT_MAX) {   // hitable_list.h:84 against t_max = FLT_MAX
                    const unsigned partner = place == 0u ? slot : (place == 1u ? slot + 1u : slot - 1u);
                    const unsigned long long key = hit_key(t, (int)(((0xffffffu - partner) << 3) | (place << 1)));
                    if (key < my_key) { my_key = key; my_t = t; my_slot = (int)slot; my_dir = dT; }
                }
            }
        }
        // the next level: c_(L+5) = F(c_(L+4)); every element must stay ordinary (it moves by ulps: it does)
        dB = dMid;
        dMid = dT;
        dT = chain_next<true>(b.rot, dT);
        if (!(fminf(fminf(fabsf(dT.x), fabsf(dT.y)), fabsf(dT.z)) >= 9.094947017729282e-13f)) { ok = false; break; }
        cur ^= 1u;
        n_cur = n_next;
        coop_sync();
    }
    coop_sync();
    if (!ok) return false;
    // the ray's winner: the smallest key over the lanes; its raw t and direction from the lane that found it
    if (lane == 0u) L.best[0] = ~0ull;
    coop_sync();
    if (my_key != ~0ull) atomicMin(&L.best[0], my_key);
    coop_sync();
    const unsigned long long kmin = L.best[0];
    if (kmin != ~0ull && my_key == kmin) {   // (keys are unique per triangle)
        L.rays[0] = make_float4(my_t, __int_as_float(my_slot), my_dir.x, my_dir.y);
        L.rays[1] = make_float4(my_dir.z, 0.0f, 0.0f, 0.0f);
    }
    coop_sync();
    if (kmin != ~0ull) {
        const float4 r0 = L.rays[0], r1 = L.rays[1];
        t_out = r0.x;
        slot_out = __float_as_int(r0.y);
        dir_out = V3{r0.z, r0.w, r1.x};
    }
    coop_sync();
    return true;
}

__device__ __forceinline__ V3 bcast3(V3 v, int src) {   // lane src's vector, wave-uniform
    return V3{__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.x), src)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.y), src)),
              __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.z), src))};
}

// bvh_trace (crt_bvh_kernels.hpp) for the shared walk: called by all lanes; `active`: this lane has a ray.
template <bool FAST>
__device__ __forceinline__ void bvh_trace_coop(const BvhView& b, const float* __restrict__ rec, V3 o, V3 d1, int force_mode, bool active,
                                               const CoopLds& L, float& best_t, int& best_slot, V3& dir_used) {
    const V3 c1 = rot0_exact(b.rot, d1);
    const V3 c2 = chain_next<FAST>(b.rot, c1);
    const V3 c3 = chain_next<FAST>(b.rot, c2);
    const V3 c4 = chain_next<FAST>(b.rot, c3);
    best_t = FLT_MAX;
    best_slot = -1;
    dir_used = c3;
    const bool fixed3 = same_bits(c4, c3);
    bool ordinary = FAST && force_mode == 0;   // every component at least 2^-40 in magnitude (and finite), the origin in the fast division's range
    if (ordinary) {
        const float lo = fminf(fminf(fminf(fabsf(c1.x), fabsf(c1.y)), fminf(fabsf(c1.z), fabsf(c2.x))),
                               fminf(fminf(fabsf(c2.y), fabsf(c2.z)), fminf(fminf(fabsf(c3.x), fabsf(c3.y)), fabsf(c3.z))));
        ordinary = lo >= 9.094947017729282e-13f && fminf(fminf(fabsf(c4.x), fabsf(c4.y)), fabsf(c4.z)) >= 9.094947017729282e-13f;
        const float ax = fabsf(o.x), ay = fabsf(o.y), az = fabsf(o.z);
        ordinary = ordinary && (ax == 0.0f || ax >= 1.3877787807814457e-17f) && (ay == 0.0f || ay >= 1.3877787807814457e-17f) &&
                   (az == 0.0f || az >= 1.3877787807814457e-17f) && fmaxf(fmaxf(ax, ay), az) <= 1099511627776.0f;
    }
    V3 c5 = c4;
    int mode = 0;
    if (__builtin_amdgcn_ballot_w64(active && !fixed3) != 0ull || force_mode != 0) {
        c5 = chain_next<FAST>(b.rot, c4);
        mode = fixed3 ? 0 : (same_bits(c5, c4) ? 0 : (same_bits(c5, c3) ? 1 : 2));
    }
    if (force_mode == 2) mode = 2;   // test hook: every lane iterates F from c4 (must give the same pixels)
    PairDirs pd;
    pd.c2 = c2; pd.c3 = c3; pd.c4 = c4;
    pd.y2 = V3{reciprocal<true>(c2.x), reciprocal<true>(c2.y), reciprocal<true>(c2.z)};
    pd.y3 = V3{reciprocal<true>(c3.x), reciprocal<true>(c3.y), reciprocal<true>(c3.z)};
    pd.y4 = V3{reciprocal<true>(c4.x), reciprocal<true>(c4.y), reciprocal<true>(c4.z)};
    pd.cyc = mode == 1;
    bool take = active && ordinary && mode != 2;
    if (!bvh_walk_coop<false>(b, rec, o, c1, pd, take, L, best_t, best_slot, dir_used)) take = false;   // list full: nobody has a result
    // rays with ordinary components whose chain has not settled (mode 2): one at a time, the whole wavefront on each (bvh_walk_one)
    for (unsigned long long irr = __builtin_amdgcn_ballot_w64(active && ordinary && mode == 2); irr != 0ull; irr &= irr - 1ull) {
        const int src = (int)__builtin_ctzll(irr);
        float t1;
        int s1;
        V3 d1u;
        if (bvh_walk_one<false>(b, rec, bcast3(o, src), bcast3(c1, src), bcast3(c2, src), bcast3(c3, src), bcast3(c4, src), L, t1, s1, d1u)) {
            if ((int)lanes_below(~0ull) == src) {
                best_t = t1;
                best_slot = s1;
                dir_used = d1u;
                take = true;
            }
        }
    }
    if (__builtin_amdgcn_ballot_w64(active && !take) != 0ull) {   // what is left (non-ordinary components, lists that ran full): alone, the literal way
        if (active && !take) bvh_walk_general<FAST, false>(b, rec, o, c1, c2, c3, c4, mode, best_t, best_slot, dir_used);
    }
}

// Called by ALL lanes of the wavefront; `active`: this lane has a ray.  use_coop: the shared pair walk (its LDS: coop).
// (coop by reference and not behind a pointer that may be null: the struct then stays in registers and its members keep
// their LDS address space — through a pointer the walk's LDS traffic became FLAT instructions.)
template <bool FAST, bool BVH>
__device__ __forceinline__ HitInfo trace_any(const ShadeScene& sc, const BvhView& b, V3 o, V3 d, int force_mode, int* __restrict__ stack,
                                             bool active, bool use_coop, const CoopLds& coop) {
    HitInfo h;
    h.t = FLT_MAX;
    h.idx = -1;
    V3 used = V3{0.0f, 0.0f, 0.0f};
    if (BVH && use_coop) {
        // Hitable::hit at the world level: ScaleRay normalises (transform.h:50-54)
        const V3 d1 = FAST ? unit_vector_fast(d) : unit_vector(d);
        bvh_trace_coop<FAST>(b, sc.base, o, d1, force_mode, active, coop, h.t, h.idx, used);
    } else if (active) {
        const V3 d1 = FAST ? unit_vector_fast(d) : unit_vector(d);
        if (BVH) bvh_trace<FAST, false>(b, sc.base, o, d1, force_mode, h.t, h.idx, used, stack, 256);
        else list_trace_any<FAST>(sc, o, d1, h.t, h.idx, used);
    }
    if (active && h.idx >= 0) {
        h.p = V3{o.x + h.t * used.x, o.y + h.t * used.y, o.z + h.t * used.z};   // rec.p = r.point_at_t(rec.t), triangle.h:95
        const float* n = sc.shade_n + (size_t)h.idx * 3;
        h.normal = V3{n[0], n[1], n[2]};
    }
    return h;
}

__device__ __forceinline__ float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

__device__ __forceinline__ V3 random_in_unit_sphere_dev(Xorwow& rs) {   // material.h:11-19, draws in x, y, z order
    V3 p;
    do {
        p.x = xw_uniform_2m1(rs);
        p.y = xw_uniform_2m1(rs);
        p.z = xw_uniform_2m1(rs);
    } while (p.x * p.x + p.y * p.y + p.z * p.z >= 1.0f);
    return p;
}

__device__ __forceinline__ V3 reflect_dev(V3 v, V3 n) {   // material.h:6-8: v - 2.0f * dot(v, n) * n
    const float k = 2.0f * dot3(v, n);
    return V3{v.x - k * n.x, v.y - k * n.y, v.z - k * n.z};
}

// schlick, material.h:22-26; pow(x, 5.0f) by the rule DESIGN.md states for this path (fifth power in
// double, rounded once; CUDA's powf is not reproducible off-device: PARITY UNPINNED).
__device__ __forceinline__ float schlick_dev(float cosine, float ref_idx) {
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    const double p = (double)(1.0f - cosine), p2 = p * p, p4 = p2 * p2;
    return r0 + (1.0f - r0) * (float)(p4 * p);
}

// Material::scatter (material.h:57-161).  r_o/r_d: r_in; returns false when the material does not scatter.
__device__ __forceinline__ bool scatter_dev(const MaterialPOD& m, V3 r_d, const HitInfo& rec, V3& attenuation, V3& s_o, V3& s_d, Xorwow& rs) {
    if (m.kind == 0) {          // Lambertian, material.h:61-71
        const V3 rnd = random_in_unit_sphere_dev(rs);
        const V3 target = V3{(rec.p.x + rec.normal.x) + rnd.x, (rec.p.y + rec.normal.y) + rnd.y, (rec.p.z + rec.normal.z) + rnd.z};
        s_o = rec.p;
        s_d = V3{target.x - rec.p.x, target.y - rec.p.y, target.z - rec.p.z};
        attenuation = V3{m.albedo[0], m.albedo[1], m.albedo[2]};
        return true;
    }
    if (m.kind == 1) {          // Metal, material.h:83-92
        const float fuzz = m.param < 1 ? m.param : 1;
        const V3 reflected = reflect_dev(unit_vector(r_d), rec.normal);
        const V3 rnd = random_in_unit_sphere_dev(rs);
        s_o = rec.p;
        s_d = V3{reflected.x + fuzz * rnd.x, reflected.y + fuzz * rnd.y, reflected.z + fuzz * rnd.z};
        attenuation = V3{m.albedo[0], m.albedo[1], m.albedo[2]};
        return dot3(s_d, rec.normal) > 0.0f;
    }
    if (m.kind == 2) {          // Dielectric, material.h:103-141
        const float ref_idx = m.param;
        const V3 reflected = reflect_dev(r_d, rec.normal);
        attenuation = V3{1.0f, 1.0f, 1.0f};
        V3 outward_normal, refracted = V3{0.0f, 0.0f, 0.0f};
        float ni_over_nt, reflect_prob, cosine;
        const float len = sqrtf(r_d.x * r_d.x + r_d.y * r_d.y + r_d.z * r_d.z);
        if (dot3(r_d, rec.normal) > 0.0f) {
            outward_normal = V3{-rec.normal.x, -rec.normal.y, -rec.normal.z};
            ni_over_nt = ref_idx;
            cosine = dot3(r_d, rec.normal) / len;
            cosine = sqrtf(1.0f - ref_idx * ref_idx * (1.0f - cosine * cosine));
        } else {
            outward_normal = rec.normal;
            ni_over_nt = 1.0f / ref_idx;
            cosine = -dot3(r_d, rec.normal) / len;
        }
        // refract(), material.h:29-39
        const V3 uv = unit_vector(r_d);
        const float dt = dot3(uv, outward_normal);
        const float discriminant = 1.0f - ni_over_nt * ni_over_nt * (1.0f - dt * dt);
        if (discriminant > 0.0f) {
            const float sq = sqrtf(discriminant);
            refracted = V3{ni_over_nt * (uv.x - dt * outward_normal.x) - sq * outward_normal.x,
                           ni_over_nt * (uv.y - dt * outward_normal.y) - sq * outward_normal.y,
                           ni_over_nt * (uv.z - dt * outward_normal.z) - sq * outward_normal.z};
            reflect_prob = schlick_dev(cosine, ref_idx);
        } else {
            reflect_prob = 1.0f;
        }
        s_o = rec.p;
        s_d = (xw_uniform(rs) < reflect_prob) ? reflected : refracted;
        return true;
    }
    return false;               // DiffuseLight, material.h:147-156
}

__device__ __forceinline__ V3 sky_of(V3 d) {   // backgroundSky(d), render.h:41-46
    return sky_from_unit(unit_vector(d));
}

// shader: 0 LambertShade, 1 shade, 2 shade_normal.  Persistent blocks of four wavefronts.
template <bool FAST, bool BVH>
#ifndef CRT_GENERIC_OCC
#define CRT_GENERIC_OCC 3   // waves per SIMD the register allocation aims at: the shared walk's LDS admits three blocks per CU
#endif
#if CRT_GENERIC_OCC
__global__ void __launch_bounds__(256, CRT_GENERIC_OCC)
#else
__global__ void __launch_bounds__(256)
#endif
render_generic_kernel(RenderParams p, ShadeScene sc, BvhView b, uint32_t* __restrict__ state, float* __restrict__ rgb,
                      int shader, int max_depth, int force_mode, unsigned* __restrict__ pixel_queue, int walk_mode, int coop_cap,
                      unsigned chunk_log, unsigned perm_mul) {
    // dynamic LDS: [the pair walk's area: walk_mode 2: kCoopWaveBytes per wavefront (bvh_walk_coop); 1: kPairArea x 256 ints
    // (bvh_walk_pairs, one stack per lane); 0: none] [material indices of shade's levels, max(1, max_depth) x 256
    // (crt_scene_set_materials: at most 65 536 materials)]
    extern __shared__ float4 dyn_lds[];
    const int pair_bytes = !BVH ? 0 : (walk_mode == 2 ? 4 * kCoopWaveBytes : (walk_mode == 1 ? kPairArea * 256 * 4 : 0));
    int* const stack = (BVH && walk_mode == 1) ? reinterpret_cast<int*>(dyn_lds) + threadIdx.x : nullptr;
    unsigned short* const level_mat = reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(dyn_lds) + pair_bytes) + threadIdx.x;
    CoopLds coop_lds;
    {
        char* base = reinterpret_cast<char*>(dyn_lds) + (threadIdx.x >> 6) * kCoopWaveBytes;
        coop_lds.rays = reinterpret_cast<float4*>(base);
        coop_lds.best = reinterpret_cast<unsigned long long*>(base + 64 * kCoopRayVecs * 16);
        coop_lds.nodes = reinterpret_cast<unsigned*>(base + 64 * kCoopRayVecs * 16 + 64 * 8);
        coop_lds.tris = coop_lds.nodes + kCoopNodeCap;
        coop_lds.cap = coop_cap;
    }
    const bool use_coop = BVH && walk_mode == 2;
    const long long stride = p.rng_count;
    const V3 cam_o = V3{p.cam.origin[0], p.cam.origin[1], p.cam.origin[2]};
    const float fnx = (float)p.nx, fny = (float)p.ny;
    // Persistent wavefronts, lane = one pixel AT A TIME: the pixels of the launch are a queue (64 positions per 8x8 tile, in
    // the order of launch_render_generic), a wavefront takes as many positions as it has free lanes, with one atomic.
    // A pixel's work is its own (its stream, its rays, its colour), whichever lane runs it, so pixels and
    // states are the same bits as with one fixed pixel per lane — but a wavefront no longer lasts as long as its longest
    // pixel (4 samples x 9 rays through glass against 4 rays of sky: the launch used to end with a few wavefronts of such
    // pixels on an idle chip).  Every wavefront leaves when the queue is exhausted and its lanes are done.
    const unsigned tiles_x = (unsigned)(p.nx + 7) >> 3, tiles_y = (unsigned)(p.row_count + 7) >> 3;
    const unsigned total = tiles_x * tiles_y * 64u;                     // the host checks that this fits 32 bits
    unsigned pool_next = 0, pool_end = 0;                               // wave-uniform: the positions this wavefront holds
    bool exhausted = false;                                             // wave-uniform: the queue has nothing left
#ifdef CRT_TUNING
    b.walk_count = b.walk_stats && b.walk_stats[kStatsTimeline] == 0ull;
    __shared__ unsigned tl_steps[4];
    if ((threadIdx.x & 63) == 0) tl_steps[threadIdx.x >> 6] = 0u;
    b.wave_steps = b.walk_stats ? &tl_steps[threadIdx.x >> 6] : nullptr;
    unsigned tl_trips = 0, tl_pixels = 0;
    const unsigned long long tl0 = b.walk_stats ? (unsigned long long)wall_clock64() : 0ull;   // 100 MHz
    unsigned long long tl1 = 0ull, tl_few = 0ull, tl_one = 0ull;   // first time with fewer than 16 pixels / with one pixel left (after the queue ran dry)
    unsigned tl_trips_few = 0;
    unsigned long long tl_walk = 0ull;   // time inside trace_any
#endif
    int x = 0, ly = 0, y = 0;
    long long si = 0;
    bool have = false;                                                  // this lane holds a pixel
    Xorwow rs;
    rs.d = rs.v0 = rs.v1 = rs.v2 = rs.v3 = rs.v4 = 0u;
    float colx = 0.0f, coly = 0.0f, colz = 0.0f;
    // One loop over RAYS, not over samples: every trip, each lane that has a path in flight traces its next ray —
    // the primary ray of its next sample or the next bounce of `shade` — so a lane whose path ended does not wait for
    // the longest path of the wavefront before it starts its next sample.  A pixel's own order of operations — u, v,
    // lens draws, then per bounce the scatter draws — is unchanged.  The wave votes inside the traversals are taken
    // over the lanes that trace in that trip.
    int smp = 0, depth = 0, levels = 0;
    bool in_flight = false;
    V3 r_o = cam_o, r_d = V3{0.0f, 0.0f, 0.0f};
    for (;;) {
        if (have && !in_flight && smp >= p.samples) {
            // col /= float(ns); sqrt; clip (render.h:123-128, vec3.h:182-189,119-131)
            const float k = (float)(1.0 / (double)(float)p.samples);
            colx *= k; coly *= k; colz *= k;
            colx = sqrtf(colx); coly = sqrtf(coly); colz = sqrtf(colz);
            colx = (colx > 1.0f) ? 1.0f : ((colx < 0.0f) ? 0.0f : colx);
            coly = (coly > 1.0f) ? 1.0f : ((coly < 0.0f) ? 0.0f : coly);
            colz = (colz > 1.0f) ? 1.0f : ((colz < 0.0f) ? 0.0f : colz);
            const size_t o = (size_t)si * 3;
            rgb[o + 0] = colx; rgb[o + 1] = coly; rgb[o + 2] = colz;
            state[0 * stride + si] = rs.d; state[1 * stride + si] = rs.v0; state[2 * stride + si] = rs.v1;
            state[3 * stride + si] = rs.v2; state[4 * stride + si] = rs.v3; state[5 * stride + si] = rs.v4;
            have = false;
        }
        for (;;) {                                                      // free lanes take the next pixels of the queue
            const unsigned long long want = __builtin_amdgcn_ballot_w64(!have);
            if (want == 0ull || exhausted) break;
            if (pool_next == pool_end) {
                // as many positions as lanes are free, no more: a position held back for a lane that is still busy would start late
                const unsigned n_want = (unsigned)__builtin_popcountll(want);
                unsigned base = 0;
                if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u) base = atomicAdd(pixel_queue, n_want);
                base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
                if (base >= total) {
                    exhausted = true;
#ifdef CRT_TUNING
                    if (b.walk_stats) tl1 = (unsigned long long)wall_clock64();
#endif
                    continue;
                }
                pool_next = base;
                pool_end = min(base + n_want, total);
            }
            const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(want >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)want, 0u));
            const unsigned q = pool_next + rank;
            if (!have && q < pool_end) {
                // the queue is dealt out in chunks of 1 << chunk_log positions (whole rows of a tile) in a scattered order (chunk * mul mod
                // n_chunks, mul coprime to n_chunks: host): the pixels a wavefront holds come from all over the frame, so the long
                // pixels of one region (glass: 36 rays in sequence) are spread over all wavefronts instead of filling a few
                const unsigned chunk = ((q >> chunk_log) * perm_mul) % (total >> chunk_log);
                const unsigned within = q & ((1u << chunk_log) - 1u);
                const unsigned tile = chunk >> (6 - chunk_log), sub = chunk & ((1u << (6 - chunk_log)) - 1u);
                const unsigned tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
                const unsigned in_tile = (sub << chunk_log) + within;   // 0..63, row-major in the 8x8 tile
                x = (int)(txi * 8u + (in_tile & 7u));
                ly = (int)(tyi * 8u + (in_tile >> 3));
                if (x < p.nx && ly < p.row_count) {                     // render.h:109; positions off the image are dropped
                    y = global_row(p.row_begin, p.strip_rows, p.strip_step, ly);
                    si = (long long)ly * p.nx + x;
                    rs.d = state[0 * stride + si]; rs.v0 = state[1 * stride + si]; rs.v1 = state[2 * stride + si];
                    rs.v2 = state[3 * stride + si]; rs.v3 = state[4 * stride + si]; rs.v4 = state[5 * stride + si];
                    colx = coly = colz = 0.0f;
                    smp = 0;
                    have = true;
#ifdef CRT_TUNING
                    tl_pixels += 1u;   // per lane; summed below
#endif
                }
            }
            pool_next = min(pool_next + (unsigned)__builtin_popcountll(want), pool_end);
        }
        if (have && !in_flight && smp < p.samples) {                    // render.h:115-118: the next sample of this pixel
            const float u = ((float)x + xw_uniform(rs)) / fnx;          // render.h:116
            const float v = ((float)y + xw_uniform(rs)) / fny;          // render.h:117
            if (y == 0 && x == 0) {                                     // get_ray's lens sample from state[0] (render.h:118): DESIGN.md §1
                float px, py;
                do {
                    px = xw_uniform_2m1(rs);
                    py = xw_uniform_2m1(rs);
                } while (px * px + py * py + 0.0f * 0.0f >= 1.0f);
            }
            r_o = cam_o;
            r_d = camera_direction(p.cam, u, v);
            depth = max_depth;
            levels = 0;
            in_flight = true;
            ++smp;
        }
        if (__builtin_amdgcn_ballot_w64(in_flight) == 0ull) {
            if (exhausted && __builtin_amdgcn_ballot_w64(have) == 0ull) {
#ifdef CRT_TUNING
                if (b.walk_stats && (threadIdx.x & 63) == 0) {   // [8] latest "queue empty" [9] latest end [10] ~earliest start [11..13] sums: start..empty, start..end, waves
                    const unsigned long long tl2 = (unsigned long long)wall_clock64();
                    atomicMax(&b.walk_stats[8], tl1); atomicMax(&b.walk_stats[9], tl2); atomicMax(&b.walk_stats[10], ~tl0);
                    atomicAdd(&b.walk_stats[11], tl1 - tl0); atomicAdd(&b.walk_stats[12], tl2 - tl0); atomicAdd(&b.walk_stats[13], 1ull);
                    atomicAdd(&b.walk_stats[14], tl_walk);   // [14] time inside the walks, summed over wavefronts
                    const unsigned slot = blockIdx.x * 4u + (threadIdx.x >> 6);
                    if (slot < (unsigned)kStatsWaveSlots) {   // per wavefront: start, queue empty, end, steps << 40 | trips << 20 | pixels
                        unsigned long long* w = b.walk_stats + kStatsWaveLog + 4 * (size_t)slot;
                        w[0] = tl0; w[1] = tl1; w[2] = tl2;
                        if (slot < (unsigned)(kStatsSampleBlocks * kStatsSampleSlots)) {   // (the sample log's words, unused by this kernel)
                            b.walk_stats[kStatsSampleLog + 2 * (size_t)slot] = tl_few ? tl_few : tl2;
                            b.walk_stats[kStatsSampleLog + 2 * (size_t)slot + 1] = ((tl_one ? tl_one : tl2) << 16) | (unsigned long long)(tl_few ? tl_trips - tl_trips_few : 0u);
                        }
                        w[3] = ((unsigned long long)tl_steps[threadIdx.x >> 6] << 40) | ((unsigned long long)tl_trips << 20) | tl_pixels;
                    }
                }
#endif
                break;
            }
            continue;                                                   // pixels to store (samples == 0) or positions off the image
        }
        walk_stat(b, 0, __builtin_amdgcn_ballot_w64(in_flight));
#ifdef CRT_TUNING
        ++tl_trips;
        if (b.walk_stats && exhausted) {
            const int live = __builtin_popcountll(__builtin_amdgcn_ballot_w64(have));
            if (live < 16 && tl_few == 0ull) { tl_few = (unsigned long long)wall_clock64(); tl_trips_few = tl_trips; }
            if (live <= 1 && tl_one == 0ull) tl_one = (unsigned long long)wall_clock64();
        }
#endif
#ifdef CRT_TUNING
        const unsigned long long tl_w0 = b.walk_stats ? (unsigned long long)wall_clock64() : 0ull;
#endif
        const HitInfo rec = trace_any<FAST, BVH>(sc, b, r_o, r_d, force_mode, stack, in_flight, use_coop, coop_lds);   // all lanes: the shared walk needs them
#ifdef CRT_TUNING
        if (b.walk_stats) tl_walk += (unsigned long long)wall_clock64() - tl_w0;
#endif
        if (in_flight) {
            V3 L = V3{0.0f, 0.0f, 0.0f};
            bool done = true;
            if (shader == 1) {
                // shade(r, world, depth, state), render.h:48-67.  Going down: one level per scattering hit (its material
                // index is kept); the chain ends with the sky (miss), or with emitted + 0.1 (depth exhausted, or the
                // material did not scatter).
                if (rec.idx < 0) {
                    L = sky_of(r_d);                                    // render.h:65
                } else {
                    const int mi = sc.material_of ? sc.material_of[rec.idx] : 0;
                    const MaterialPOD m = sc.materials[mi];
                    V3 att, s_o, s_d;
                    if (depth > 0 && scatter_dev(m, r_d, rec, att, s_o, s_d, rs)) {   // render.h:58
                        level_mat[levels * 256] = (unsigned short)mi;
                        ++levels;
                        --depth;
                        r_o = s_o;
                        r_d = s_d;
                        done = false;
                    } else {
                        const V3 emitted = (m.kind == 3) ? V3{m.albedo[0], m.albedo[1], m.albedo[2]} : V3{0.0f, 0.0f, 0.0f};
                        L = V3{emitted.x + 0.1f, emitted.y + 0.1f, emitted.z + 0.1f};   // render.h:62
                    }
                }
                if (done) {
                    // Coming back up: emitted + attenuation * (deeper result), render.h:59; a material that
                    // scattered is not a light, so emitted = vec3(0,0,0) (material.h:49-53).
                    for (int k = levels - 1; k >= 0; --k) {
                        const MaterialPOD m = sc.materials[level_mat[k * 256]];
                        const V3 att = (m.kind == 2) ? V3{1.0f, 1.0f, 1.0f} : V3{m.albedo[0], m.albedo[1], m.albedo[2]};
                        L = V3{0.0f + att.x * L.x, 0.0f + att.y * L.y, 0.0f + att.z * L.z};
                    }
                }
            } else if (rec.idx < 0) {
                L = sky_of(r_d);                                        // render.h:85,101
            } else if (shader == 2) {
                L = rec.normal;                                         // shade_normal, render.h:98
            } else {                                                    // LambertShade, render.h:75-83
                const MaterialPOD m = sc.materials[sc.material_of ? sc.material_of[rec.idx] : 0];
                const V3 emitted = (m.kind == 3) ? V3{m.albedo[0], m.albedo[1], m.albedo[2]} : V3{0.0f, 0.0f, 0.0f};
                V3 att = V3{0.0f, 0.0f, 0.0f}, s_o, s_d;                 // DiffuseLight::scatter leaves attenuation unwritten: 0 (as the oracle)
                (void)scatter_dev(m, r_d, rec, att, s_o, s_d, rs);
                float t = dot3(r_d, rec.normal);
                if (t < 0.0f) t = 0.0f;
                const V3 sky = sky_of(r_d);
                L = V3{0.2f * ((t * att.x) * sky.x) + emitted.x, 0.2f * ((t * att.y) * sky.y) + emitted.y,
                       0.2f * ((t * att.z) * sky.z) + emitted.z};
            }
            if (done) {
                colx += L.x;                                            // render.h:119-121
                coly += L.y;
                colz += L.z;
                in_flight = false;
            }
        }
    }
}

// ---- the order of the pixel queue ------------------------------------------------------------------------------
// A pixel's cost is set by what its rays meet: behind glass `shade` follows 9 rays per sample one after the other, on a
// Lambertian 2, on the sky 1 — and a wavefront's trip lasts as long as the sum of its rays' walks, so 64 pixels behind glass
// in one wavefront are the longest chain there is.  The queue is therefore dealt out in chunks of 4 pixels of a row in a
// scattered order (chunk c at place c * mul mod n_chunks): what a wavefront holds comes from all over the frame, expensive
// regions are spread over all wavefronts and over the whole launch.  Measured against it on one box (tools/ab_env.sh) and
// dropped: orders from a probe ray per 8x8 tile, per 4x4 block or per chunk (the material at its first hit; or only whether
// the root's box is met: sky last) — most expensive first from one cursor, dealt like cards over the wavefronts' first
// pools, or class 0 dealt statically (wavefront w renders S[w], S[w + W], ...): 7.7-11.6 ms against 5.9 ms — a queue that
// starts with the expensive pixels fills EVERY wavefront with them, their trips are the longest possible, and the cheap nine
// tenths of the frame wait behind them.
hipError_t launch_render_generic(const RenderParams& p, const float* d_base, const float* d_shade_n, const int* d_material_of,
                                 const void* d_materials, int n_tris, const float4* d_nodes, int n_nodes, const float rot[9],
                                 uint32_t* d_state, float* d_rgb, bool bvh, bool fast, int shader, int max_depth, int force_mode,
                                 hipStream_t stream, const float4* d_pairs, const int* d_partner, int walk_mode, unsigned* d_pixel_queue, int n_cus,
                                 unsigned long long* d_stats, LaunchCache* cache) {
    if (p.row_count <= 0 || p.nx <= 0) return hipSuccess;
    ShadeScene sc;
    sc.base = d_base; sc.shade_n = d_shade_n; sc.material_of = d_material_of;
    sc.materials = static_cast<const MaterialPOD*>(d_materials);
    sc.n_tris = n_tris;
    BvhView b;
    b.nodes = d_nodes; b.n_nodes = n_nodes;
    if (!bvh || d_pairs == nullptr) walk_mode = 0;
    int coop_cap = kCoopNodeCap;
    if (walk_mode == 3) {   // test hook: the shared walk with a list so short that trips are abandoned
        walk_mode = 2;
        coop_cap = 96;
    }
    b.pairs = walk_mode ? d_pairs : nullptr; b.partner = d_partner;
#ifdef CRT_TUNING
    b.walk_stats = d_stats;
#endif
    for (int k = 0; k < 9; ++k) { sc.rot[k] = rot[k]; b.rot[k] = rot[k]; }
    auto fn = bvh ? (fast ? render_generic_kernel<true, true> : render_generic_kernel<false, true>)
                  : (fast ? render_generic_kernel<true, false> : render_generic_kernel<false, false>);
    const size_t lds = (walk_mode == 2 ? (size_t)4 * kCoopWaveBytes : (walk_mode == 1 ? (size_t)kPairArea * 256 * sizeof(int) : 0)) +
                       (size_t)std::min(std::max(1, max_depth), kMaxDepthLevels) * 256 * sizeof(unsigned short);
    // persistent blocks: as many as are resident at once, fewer for small frames
    const unsigned long long tiles = (unsigned long long)((p.nx + 7) / 8) * (unsigned long long)((p.row_count + 7) / 8);
    if (tiles * 64ull >= 0xffff0000ull) return hipErrorInvalidValue;   // the queue cursor is 32 bits (a 65 000 x 65 000 frame)
    int per_cu = 0;
    hipError_t e = hipSuccess;
    if (cache && cache->fn == (const void*)fn && cache->lds == lds) {
        per_cu = cache->resident;
    } else {
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, lds);
        if (e != hipSuccess) return e;
        if (cache) { cache->fn = (const void*)fn; cache->lds = lds; cache->resident = per_cu; }
    }
    unsigned resident = (unsigned)std::max(1, n_cus) * (unsigned)std::max(1, per_cu);
#ifdef CRT_TUNING
    if (const char* ev = getenv("CRT_GENERIC_BLOCKS")) resident = (unsigned)std::max(1, atoi(ev));   // resident blocks in total
#endif
    dim3 block(256), grid((unsigned)std::min<unsigned long long>((tiles + 3) / 4, resident));
    e = hipMemsetAsync(d_pixel_queue, 0, kGenericQueueWords * sizeof(unsigned), stream);   // [0] the queue's cursor
    if (e != hipSuccess) return e;
    // the order of the pixel queue: chunks of 4 positions, chunk c at place c * mul mod n_chunks (tools/sweep_queue.sh)
    unsigned chunk_log = kGenericChunkLog, perm_mul = 1;
#ifdef CRT_TUNING
    if (const char* ev = getenv("CRT_QUEUE_CHUNK_LOG")) chunk_log = (unsigned)std::min(6, std::max(0, atoi(ev)));
#endif
    {
        const unsigned long long n_chunks = tiles * 64ull >> chunk_log;
        static const unsigned primes[] = {1999u, 1997u, 1993u, 1987u, 997u, 991u, 499u, 491u, 251u, 241u, 127u, 113u, 61u, 59u};   // the first that does not divide n_chunks and keeps c * mul in 32 bits
        unsigned first = 0;
#ifdef CRT_TUNING
        if (const char* ev = getenv("CRT_QUEUE_MUL")) first = (unsigned)std::max(0, atoi(ev));   // 0: the table; else this multiplier if coprime
        if (first > 1 && n_chunks % first != 0ull && n_chunks * first < 0xffffffffull) perm_mul = first;
#endif
        // only for the shared walk: with one walk per lane (skip links, one stack per lane) neighbouring pixels in one wavefront
        // keep the walks' loads coherent (LambertShade through this kernel over the skip links: 2.5 ms in order, 5.5 ms scattered)
        if (perm_mul == 1 && first != 1 && walk_mode == 2)
            for (unsigned pr : primes)
                if (n_chunks % pr != 0ull && n_chunks * pr < 0xffffffffull) {
                    perm_mul = pr;
                    break;
                }
    }
    hipLaunchKernelGGL(fn, grid, block, lds, stream, p, sc, b, d_state, d_rgb, shader, max_depth, force_mode, d_pixel_queue, walk_mode, coop_cap,
                       chunk_log, perm_mul);
    return hipGetLastError();
}
