// crt_bvh.cpp — host side of the BVH world (include/crt.h "BVH" section).
//
//   create_BVHfromList -> create_BVH<<<1,1>>> -> BVHNode::BVHNode   createScene.h:42-44,212-219; hitable/bvh.h:76-125
//        -> crt_scene_build_bvh   (the reference builds in ONE device thread with thrust::sort;
//           here the host builds the same tree and uploads it flat, in preorder with skip links)
//   Update_BVH -> UpdateBVH<<<1,1>>> -> BVHNode::UpdateBVH           createScene.h:46-58; bvh.h:135-158
//        -> crt_scene_refit_bvh   (one small launch per tree level instead of one recursive thread)
//   writeCSV                                                          Loader/CSVWriter.h:8-32; kernel.cu:61,101
//        -> crt_write_timing_csv
// The render entry points for the BVH world live in crt_api.cpp (crt_render_rows_bvh).
//
// thrust::sort inside a device thread resolves to Thrust's sequential backend; its comparator
// here (BoxCompare, bvh.h:9-45) returns true on equal keys, which is not a strict weak ordering,
// so the order it produces is a property of the algorithm: the iterative stable merge sort of
// thrust/system/detail/sequential (insertion sort per 32 elements, then pairwise merges through a
// temporary array).  That algorithm is restated below; this boundary is PARITY UNPINNED.
#include "crt_internal.hpp"

#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <string>

using namespace crt_host;

namespace {

// ---- cuRAND XORWOW on the host: only stream 0 (subsequence 0, no skip) is ever needed here.
struct HostXorwow {
    uint32_t d, v[5];
};
HostXorwow xw_init_stream0(uint64_t seed) {   // curand_init(seed, 0, 0): the seed scrambling of rng_init_kernel
    const uint32_t s0 = (uint32_t)seed ^ 0xaad26b49u, s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
    const uint32_t t0 = 1099087573u * s0, t1 = 2591861531u * s1;
    HostXorwow s;
    s.d = 6615241u + t1 + t0;
    s.v[0] = 123456789u + t0;
    s.v[1] = 362436069u ^ t0;
    s.v[2] = 521288629u + t1;
    s.v[3] = 88675123u ^ t1;
    s.v[4] = 5783321u + t0;
    return s;
}
float xw_uniform(HostXorwow& s) {             // curand_uniform: x * 2^-32 + 2^-33
    const uint32_t t = s.v[0] ^ (s.v[0] >> 2);
    s.v[0] = s.v[1]; s.v[1] = s.v[2]; s.v[2] = s.v[3]; s.v[3] = s.v[4];
    s.v[4] = (s.v[4] ^ (s.v[4] << 4)) ^ (t ^ (t << 1));
    s.d += 362437u;
    const uint32_t x = s.v[4] + s.d;
    const float k = 2.3283064e-10f;
    return (float)x * k + (k / 2.0f);
}

// ---- rotate(., vec3(0)) as a matrix (crt_host::rotation_zero_matrix): the entries are 1, +0 and -0.
H3 rot_apply(const float* m, H3 v) {          // vec3.h:215: (origin * rotate_k).sum()
    return H3{(v.x * m[0] + v.y * m[1]) + v.z * m[2], (v.x * m[3] + v.y * m[4]) + v.z * m[5],
              (v.x * m[6] + v.y * m[7]) + v.z * m[8]};
}

struct Box {
    float mn[3], mx[3];
};
Box transform_aabb(const float* m, Box b) {   // Transform::TransformAABB, transform.h:22-31 (rotation 0, position 0)
    const H3 rmin = rot_apply(m, H3{b.mn[0], b.mn[1], b.mn[2]}), rmax = rot_apply(m, H3{b.mx[0], b.mx[1], b.mx[2]});
    Box o;
    o.mn[0] = fminf(rmin.x, rmax.x) + 0.0f; o.mn[1] = fminf(rmin.y, rmax.y) + 0.0f; o.mn[2] = fminf(rmin.z, rmax.z) + 0.0f;
    o.mx[0] = fmaxf(rmin.x, rmax.x) + 0.0f; o.mx[1] = fmaxf(rmin.y, rmax.y) + 0.0f; o.mx[2] = fmaxf(rmin.z, rmax.z) + 0.0f;
    return o;
}
Box surrounding(Box a, Box b) {               // surrounding_box, aabb.h:58-66
    Box o;
    for (int k = 0; k < 3; ++k) { o.mn[k] = fminf(a.mn[k], b.mn[k]); o.mx[k] = fmaxf(a.mx[k], b.mx[k]); }
    return o;
}
Box triangle_bv(const float* m, const float* v) {   // Triangle::bounding_box (triangle.h:102-115) + GetBV's TransformAABB (hitable.h:45-50)
    Box b;
    for (int k = 0; k < 3; ++k) {
        b.mn[k] = fminf(v[k], fminf(v[3 + k], v[6 + k]));
        b.mx[k] = fmaxf(v[k], fmaxf(v[3 + k], v[6 + k]));
    }
    return transform_aabb(m, b);
}

// ---- Thrust's sequential stable merge sort on triangle indices, comparator BoxCompare
bool box_compare(const float* key, int a, int b) {   // bvh.h:11-41: `val1 - val2 < 0.0 ? false : true`
    const float val1 = key[a], val2 = key[b];
    if (val1 - val2 < 0.0) return false;
    return true;
}
void insertion_sort(int* first, int* last, const float* key) {   // sequential/insertion_sort.h
    if (first == last) return;
    for (int* i = first + 1; i != last; ++i) {
        const int tmp = *i;
        if (box_compare(key, tmp, *first)) {
            std::copy_backward(first, i, i + 1);
            *first = tmp;
        } else {
            int *j = i, *k = i - 1;
            while (box_compare(key, tmp, *k)) { *j = *k; j = k; --k; }
            *j = tmp;
        }
    }
}
void merge(const int* first1, const int* last1, const int* first2, const int* last2, int* result, const float* key) {   // sequential/merge.inl
    while (first1 != last1 && first2 != last2) {
        if (box_compare(key, *first2, *first1)) *result++ = *first2++;
        else *result++ = *first1++;
    }
    result = std::copy(first1, last1, result);
    std::copy(first2, last2, result);
}
void thrust_sequential_sort(int* first, int n, const float* key, int* temp) {   // iterative_stable_merge_sort
    int partition = 32;
    for (int at = 0; at < n; at += partition) insertion_sort(first + at, first + std::min(n, at + partition), key);
    if (n <= partition) return;
    bool ping = true;
    for (; partition < n; partition *= 2, ping = !ping) {
        const int* src = ping ? first : temp;
        int* dst = ping ? temp : first;
        for (int at = 0; at < n; at += 2 * partition) {
            const int mid = std::min(n, at + partition), end = std::min(n, mid + partition);
            merge(src + at, src + mid, src + mid, src + end, dst + at, key);
        }
    }
    if (!ping) std::copy(temp, temp + n, first);
}

struct HostNode {
    Box box;
    int skip = 0, first = 0, count = 0, depth = 0;   // count 0 = inner node; first = first slot of a leaf
};

struct Builder {
    const float* base;            // n x 12, list order
    const float* rotm;
    std::vector<int> order;       // the list's pointer array: sorted in place span by span
    std::vector<float> key[3];    // box.min() of every triangle's GetBV
    std::vector<int> temp;
    std::vector<HostNode> nodes;
    HostXorwow rng;
    int max_depth = 0;

    int construct(int first, int n, int depth) {   // BVHNode::BVHNode(l, n, 0, 1, state), bvh.h:76-125
        const int me = (int)nodes.size();
        nodes.emplace_back();
        max_depth = std::max(max_depth, depth);
        const int axis = (int)(3 * xw_uniform(rng));                 // bvh.h:84
        thrust_sequential_sort(order.data() + first, n, key[axis == 0 ? 0 : (axis == 1 ? 1 : 2)].data(), temp.data());
        if (n == 1 || n == 2) {                                      // bvh.h:95-109: childList = the span, box = childList->GetBV
            Box b = triangle_bv(rotm, base + (size_t)order[first] * crt::kBaseFloats);
            if (n == 2) b = surrounding(b, triangle_bv(rotm, base + (size_t)order[first + 1] * crt::kBaseFloats));
            HostNode& nd = nodes[me];
            nd.box = transform_aabb(rotm, b);                         // GetBV's TransformAABB of the childList
            nd.first = first;
            nd.count = n;
            nd.depth = depth;
        } else {
            const int l = construct(first, n / 2, depth + 1);         // bvh.h:111
            const int r = construct(first + n / 2, n - n / 2, depth + 1);
            HostNode& nd = nodes[me];
            nd.box = surrounding(transform_aabb(rotm, nodes[l].box), transform_aabb(rotm, nodes[r].box));   // bvh.h:116-121
            nd.first = first;
            nd.count = 0;
            nd.depth = depth;
        }
        nodes[me].skip = (int)nodes.size();
        return me;
    }
};

}  // namespace

// crt_debug.h hooks onto the build's own helpers (tests/test_ref_thrust.py, tests/test_ref_hit.py)
namespace crt_host {
void debug_thrust_sort(int* order, int n, const float* key) {
    std::vector<int> temp((size_t)std::max(n, 1));
    thrust_sequential_sort(order, n, key, temp.data());
}
void debug_host_box_op(int op, const float* a6, const float* b6, float* out6) {
    float m[9];
    rotation_zero_matrix(m);
    Box a, b;
    for (int k = 0; k < 3; ++k) { a.mn[k] = a6[k]; a.mx[k] = a6[3 + k]; b.mn[k] = b6[k]; b.mx[k] = b6[3 + k]; }
    const Box r = op == 0 ? surrounding(a, b) : transform_aabb(m, a);
    for (int k = 0; k < 3; ++k) { out6[k] = r.mn[k]; out6[3 + k] = r.mx[k]; }
}
}  // namespace crt_host

crt_bvh::~crt_bvh() {
    if (d_nodes) (void)hipFree(d_nodes);
    if (d_pairs) (void)hipFree(d_pairs);
    if (d_pair_flat) (void)hipFree(d_pair_flat);
    if (d_base_slots) (void)hipFree(d_base_slots);
    if (d_shade_n_slots) (void)hipFree(d_shade_n_slots);
    if (d_idx_slots) (void)hipFree(d_idx_slots);
    if (d_level_nodes) (void)hipFree(d_level_nodes);
    if (d_material_slots) (void)hipFree(d_material_slots);
    if (d_partner) (void)hipFree(d_partner);
    if (d_slot_leaf) (void)hipFree(d_slot_leaf);
    if (d_anc) (void)hipFree(d_anc);
}

extern "C" {

int crt_scene_build_bvh(crt_scene* s, crt_rng* rng, uint64_t seed) {
    if (!s) return fail(CRT_ERR_INVALID, "crt_scene_build_bvh: null scene");
    if (s->n_tris < 1)
        return fail(CRT_ERR_UNSUPPORTED, "crt_scene_build_bvh: empty list (the reference's BVHNode constructor recurses without end, bvh.h:110-112)");
    if (s->n_tris > 0xffffff) return fail(CRT_ERR_UNSUPPORTED, "crt_scene_build_bvh: more than 2^24 - 1 triangles");
    if (!have_device()) return fail(CRT_ERR_NO_DEVICE, "crt_scene_build_bvh: no HIP device");
    int device = 0;
    HIP_TRY(hipGetDevice(&device));
    if (device != s->device || (rng && rng->device != device))
        return fail(CRT_ERR_INVALID, "crt_scene_build_bvh: current device is %d, the scene lives on device %d", device, s->device);
    const int n = s->n_tris;
    Builder b;
    b.base = s->h_base.data();
    float rotm[9];
    rotation_zero_matrix(rotm);
    b.rotm = rotm;
    // The split axes come from stream 0 (create_BVH passes the base of the curandState array,
    // createScene.h:42-44,215): the stream continues from where it is when `rng` holds pixel 0.
    const bool have_stream0 = rng && rng->first == 0 && rng->count > 0;
    if (have_stream0) {
        // stream 0's words are read here and written back below: every render that uses these states — on whatever
        // stream, non-blocking ones included — has to be done first
        HIP_TRY(hipDeviceSynchronize());
        uint32_t w[6];
        for (int k = 0; k < 6; ++k)
            HIP_TRY(hipMemcpy(&w[k], rng->d_state + (size_t)k * rng->count, sizeof(uint32_t), hipMemcpyDeviceToHost));
        b.rng.d = w[0];
        for (int k = 0; k < 5; ++k) b.rng.v[k] = w[1 + k];
    } else {
        b.rng = xw_init_stream0(seed);
    }
    b.order.resize(n);
    b.temp.resize(n);
    for (int a = 0; a < 3; ++a) b.key[a].resize(n);
    for (int i = 0; i < n; ++i) {
        b.order[i] = i;
        const Box bx = triangle_bv(rotm, b.base + (size_t)i * crt::kBaseFloats);   // ah->GetBV(0, 0, box_left), bvh.h:19
        for (int a = 0; a < 3; ++a) b.key[a][i] = bx.mn[a];
    }
    b.nodes.reserve((size_t)2 * n);
    b.construct(0, n, 0);
    if (b.max_depth > 31) return fail(CRT_ERR_UNSUPPORTED, "crt_scene_build_bvh: tree depth %d", b.max_depth);

    crt_bvh* t = new (std::nothrow) crt_bvh;
    if (!t) return fail(CRT_ERR_NOMEM, "crt_scene_build_bvh: out of memory");
    t->n_nodes = (int)b.nodes.size();
    t->max_depth = b.max_depth;
    memcpy(t->rot, rotm, sizeof rotm);
    t->slot_tri = b.order;
    // flat nodes + per-level lists + slot-ordered triangle data
    std::vector<float4> flat((size_t)2 * t->n_nodes);
    std::vector<std::vector<int>> levels((size_t)b.max_depth + 1);
    std::vector<int> slot_depth(n, 0);
    for (int i = 0; i < t->n_nodes; ++i) {
        const HostNode& nd = b.nodes[i];
        const int info = (nd.count ? nd.first : 0) | (nd.count << 24) | (nd.depth << 26);
        float skip_f, info_f;
        memcpy(&skip_f, &nd.skip, 4);
        memcpy(&info_f, &info, 4);
        flat[2 * (size_t)i] = make_float4(nd.box.mn[0], nd.box.mn[1], nd.box.mn[2], nd.box.mx[0]);
        flat[2 * (size_t)i + 1] = make_float4(nd.box.mx[1], nd.box.mx[2], skip_f, info_f);
        levels[nd.depth].push_back(i);
        for (int k = 0; k < nd.count; ++k) slot_depth[nd.first + k] = nd.depth;
    }
    // tiled primary rays: tie order, leaf of every slot, path of every leaf
    std::vector<int> partner(n), slot_leaf(n), path;
    t->anc_stride = (b.max_depth + 1 + 3) / 4 * 4;
    std::vector<int> anc;
    {
        int n_leaves = 0;
        for (int i = 0; i < t->n_nodes; ++i) n_leaves += b.nodes[i].count != 0;
        anc.assign((size_t)n_leaves * t->anc_stride, -1);
        int ord = 0;
        path.clear();
        for (int i = 0; i < t->n_nodes; ++i) {   // preorder: the node at depth d replaces the path's entry d
            const HostNode& nd = b.nodes[i];
            path.resize(nd.depth);
            path.push_back(i);
            if (nd.count == 0) continue;
            for (int k = 0; k <= nd.depth; ++k) anc[(size_t)ord * t->anc_stride + k] = path[k];
            for (int k = 0; k < nd.count; ++k) {
                const int slot = nd.first + k;
                slot_leaf[slot] = ord | (nd.depth << 24);
                partner[slot] = nd.count == 2 ? nd.first + (1 - k) : slot;
            }
            ++ord;
        }
    }
    // pair records (crt_bvh_kernels.hpp: bvh_walk_pairs): one per inner node, preorder over the inner nodes; the references and
    // the normal cones are static, the boxes are copied from the flat nodes on the device (after this upload and after every refit)
    std::vector<float4> pairs;
    std::vector<int> pair_flat;
    // one stack entry per level (bvh_walk_pairs); 23 bits of slot and 24 of pair index in the items of the shared walk (bvh_walk_coop)
    if (b.max_depth + 1 <= crt::kPairStackDepth && n <= (1 << 23)) {
        std::vector<int> pair_of((size_t)t->n_nodes, -1), first_slot((size_t)t->n_nodes + 1, n);
        int n_inner = 0;
        for (int i = 0; i < t->n_nodes; ++i)
            if (b.nodes[i].count == 0) pair_of[i] = n_inner++;
        for (int i = t->n_nodes - 1; i >= 0; --i) first_slot[i] = b.nodes[i].count ? b.nodes[i].first : first_slot[i + 1];
        // unit stored normals in slot order (what Triangle::collision_detection's first test multiplies the direction with,
        // triangle.h:61); non-finite or zero normals never satisfy `dot < 0` for sure: their subtrees get no cone
        std::vector<double> un((size_t)n * 3);
        std::vector<char> bad((size_t)n, 0);
        for (int slot = 0; slot < n; ++slot) {
            const float* nr = s->h_base.data() + (size_t)b.order[slot] * crt::kBaseFloats + 9;
            const double len = std::sqrt((double)nr[0] * nr[0] + (double)nr[1] * nr[1] + (double)nr[2] * nr[2]);
            // ... nor do normals whose float products d_k * n_k can underflow or overflow: the reference's float dot product of
            // a subnormal-sized normal rounds to +-0 (`< 0` false: the triangle IS tested) whatever the exact sign says
            if (!(len >= 7.888609052210118e-31) || !(len <= 1.2676506002282294e30)) { bad[slot] = 1; continue; }   // 2^-100 .. 2^100
            for (int k = 0; k < 3; ++k) un[(size_t)slot * 3 + k] = nr[k] / len;
        }
        // Normal cone of a subtree as one word: axis as three signed bytes / 127, threshold byte T = byte * kConeScale.  The walk
        // skips the subtree when  d . (qx, qy, qz) < -T * 127:  every stored normal n_i then has d . n_i < -1e-3 |d| |n_i|,
        // three orders of magnitude beyond the rounding of the reference's float dot product, so its first test rejects every
        // triangle of the subtree (none of them could be the hit, and nothing else of a rejected triangle is observable).
        auto cone_of = [&](int node) -> unsigned {
            const int s0 = first_slot[node], s1 = (b.nodes[node].skip < t->n_nodes) ? first_slot[b.nodes[node].skip] : n;
            double ax = 0.0, ay = 0.0, az = 0.0;
            for (int k = s0; k < s1; ++k) {
                if (bad[k]) return 0xffu;
                ax += un[(size_t)k * 3]; ay += un[(size_t)k * 3 + 1]; az += un[(size_t)k * 3 + 2];
            }
            const double al = std::sqrt(ax * ax + ay * ay + az * az);
            if (!(al > 1e-9 * (double)(s1 - s0))) return 0xffu;
            const int qx = (int)std::lround(ax / al * 127.0), qy = (int)std::lround(ay / al * 127.0), qz = (int)std::lround(az / al * 127.0);
            const double dx = qx / 127.0, dy = qy / 127.0, dz = qz / 127.0, dl = std::sqrt(dx * dx + dy * dy + dz * dz);
            if (!(dl > 0.5)) return 0xffu;
            double min_cos = 1.0;
            for (int k = s0; k < s1; ++k)
                min_cos = std::min(min_cos, (un[(size_t)k * 3] * dx + un[(size_t)k * 3 + 1] * dy + un[(size_t)k * 3 + 2] * dz) / dl);
            if (!(min_cos > 0.05)) return 0xffu;                                    // half angle towards 90 degrees: nothing to gain
            const double sin_t = std::sqrt(std::max(0.0, 1.0 - min_cos * min_cos));
            const double T = (sin_t + 1.0e-3) * 1.0001 * dl + 2.0e-5;               // |d| <= 1 + 1e-5; float evaluation of the dot product
            const int byte = (int)std::ceil(T / crt::kConeScale);
            if (byte > 254) return 0xffu;
            return (unsigned)byte | ((unsigned)(qx & 0xff) << 8) | ((unsigned)(qy & 0xff) << 16) | ((unsigned)(qz & 0xff) << 24);
        };
        pairs.assign((size_t)4 * n_inner, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
        pair_flat.assign((size_t)2 * n_inner, 0);
        auto ref_of = [&](int node) {
            const HostNode& nd = b.nodes[node];
            return nd.count ? (int)(0x80000000u | ((unsigned)nd.count << 24) | (unsigned)nd.first) : pair_of[node];
        };
        for (int i = 0; i < t->n_nodes; ++i) {
            if (b.nodes[i].count != 0) continue;
            const int l = i + 1, r = b.nodes[l].skip;   // the left child follows its parent; its skip link is the right child
            const int refs[4] = {ref_of(l), ref_of(r), (int)cone_of(l), (int)cone_of(r)};
            float4 p3;
            memcpy(&p3, refs, sizeof p3);
            pairs[(size_t)4 * pair_of[i] + 3] = p3;
            pair_flat[(size_t)2 * pair_of[i]] = l;
            pair_flat[(size_t)2 * pair_of[i] + 1] = r;
        }
        t->n_pairs = n_inner;
    }
    std::vector<int> level_nodes;
    t->level_offset.assign(1, 0);
    for (const std::vector<int>& lv : levels) {
        level_nodes.insert(level_nodes.end(), lv.begin(), lv.end());
        t->level_offset.push_back((int)level_nodes.size());
    }
    std::vector<float> base_slots((size_t)n * crt::kBaseFloats), shade_slots((size_t)n * 3);
    std::vector<int> idx_slots((size_t)n * 3);
    bool tiny = false;   // a nonzero coordinate below 2^-56: differences could fall below the fast division's range
    for (int slot = 0; slot < n; ++slot) {
        const int tri = b.order[slot];
        const float* src = s->h_base.data() + (size_t)tri * crt::kBaseFloats;
        memcpy(&base_slots[(size_t)slot * crt::kBaseFloats], src, crt::kBaseFloats * sizeof(float));
        memcpy(&idx_slots[(size_t)slot * 3], &s->h_idx[(size_t)tri * 3], 3 * sizeof(int));
        for (int k = 0; k < 9; ++k) tiny = tiny || (src[k] != 0.0f && fabsf(src[k]) < 1.3877787807814457e-17f);
        // rec.normal = rotate(rec.normal, rotation) once per Hitable::hit level on the way out
        // (hitable.h:36): the triangle, the leaf's list, the leaf node and its depth ancestors
        H3 nrm = h3(src + 9);
        for (int k = 0; k < slot_depth[slot] + 3; ++k) nrm = rot_apply(rotm, nrm);
        shade_slots[(size_t)slot * 3] = nrm.x; shade_slots[(size_t)slot * 3 + 1] = nrm.y; shade_slots[(size_t)slot * 3 + 2] = nrm.z;
    }
    t->coords_not_tiny = !tiny;
    hipError_t e = hipMalloc(&t->d_nodes, flat.size() * sizeof(float4));
    if (e == hipSuccess) e = hipMalloc(&t->d_base_slots, base_slots.size() * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&t->d_shade_n_slots, shade_slots.size() * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&t->d_idx_slots, idx_slots.size() * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&t->d_level_nodes, level_nodes.size() * sizeof(int));
    if (e == hipSuccess) e = hipMemcpy(t->d_nodes, flat.data(), flat.size() * sizeof(float4), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(t->d_base_slots, base_slots.data(), base_slots.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(t->d_shade_n_slots, shade_slots.data(), shade_slots.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(t->d_idx_slots, idx_slots.data(), idx_slots.size() * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(t->d_level_nodes, level_nodes.data(), level_nodes.size() * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess && t->n_pairs > 0) {
        e = hipMalloc(&t->d_pairs, pairs.size() * sizeof(float4));
        if (e == hipSuccess) e = hipMalloc(&t->d_pair_flat, pair_flat.size() * sizeof(int));
        if (e == hipSuccess) e = hipMemcpy(t->d_pairs, pairs.data(), pairs.size() * sizeof(float4), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(t->d_pair_flat, pair_flat.data(), pair_flat.size() * sizeof(int), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = crt::launch_bvh_pairs(t->d_nodes, t->d_pairs, t->d_pair_flat, t->n_pairs, nullptr);
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    }
    if (e == hipSuccess) e = hipMalloc(&t->d_partner, partner.size() * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&t->d_slot_leaf, slot_leaf.size() * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&t->d_anc, anc.size() * sizeof(int));
    if (e == hipSuccess) e = hipMemcpy(t->d_partner, partner.data(), partner.size() * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(t->d_slot_leaf, slot_leaf.data(), slot_leaf.size() * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(t->d_anc, anc.data(), anc.size() * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        crt::BvhTiles view;
        view.nodes = t->d_nodes; view.n_nodes = t->n_nodes; view.partner = t->d_partner; view.slot_leaf = t->d_slot_leaf;
        view.anc = t->d_anc; view.anc_stride = t->anc_stride;
        memcpy(view.rot, rotm, sizeof rotm);
        t->tiles_view = view;
    }
    if (e == hipSuccess && have_stream0) {   // the stream moved on by one draw per node
        const uint32_t w[6] = {b.rng.d, b.rng.v[0], b.rng.v[1], b.rng.v[2], b.rng.v[3], b.rng.v[4]};
        for (int k = 0; k < 6 && e == hipSuccess; ++k)
            e = hipMemcpy(rng->d_state + (size_t)k * rng->count, &w[k], sizeof(uint32_t), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        delete t;
        return fail(CRT_ERR_HIP, "crt_scene_build_bvh: %s", hipGetErrorString(e));
    }
    delete s->bvh;
    s->bvh = t;
    return upload_bvh_materials(s);
}

int crt_scene_refit_bvh(crt_scene* s, void* stream_) {
    if (!s || !s->bvh) return fail(CRT_ERR_INVALID, "crt_scene_refit_bvh: the scene has no BVH (crt_scene_build_bvh)");
    int device = 0;
    HIP_TRY(hipGetDevice(&device));
    if (device != s->device)
        return fail(CRT_ERR_INVALID, "crt_scene_refit_bvh: current device is %d, the scene lives on device %d", device, s->device);
    crt_bvh* t = s->bvh;
    for (int level = t->max_depth; level >= 0; --level) {   // children before parents (UpdateBVH recurses first, bvh.h:142-143)
        const int at = t->level_offset[level], cnt = t->level_offset[level + 1] - at;
        HIP_TRY(crt::launch_bvh_refit_level(t->d_nodes, t->d_level_nodes + at, cnt, t->d_base_slots, t->rot, (hipStream_t)stream_));
    }
    HIP_TRY(crt::launch_bvh_pairs(t->d_nodes, t->d_pairs, t->d_pair_flat, t->n_pairs, (hipStream_t)stream_));   // the pair records carry copies of the boxes
    return CRT_OK;
}

int crt_scene_bvh_node_count(const crt_scene* s) { return (s && s->bvh) ? s->bvh->n_nodes : CRT_ERR_INVALID; }

int crt_debug_bvh_export(const crt_scene* s, float* boxes6, int* meta4, int* slot_triangle) {
    if (!s || !s->bvh) return fail(CRT_ERR_INVALID, "crt_debug_bvh_export: the scene has no BVH");
    const crt_bvh* t = s->bvh;
    std::vector<float4> flat((size_t)2 * t->n_nodes);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(flat.data(), t->d_nodes, flat.size() * sizeof(float4), hipMemcpyDeviceToHost));
    for (int i = 0; i < t->n_nodes; ++i) {
        const float4 n0 = flat[2 * (size_t)i], n1 = flat[2 * (size_t)i + 1];
        if (boxes6) {
            float* o = boxes6 + 6 * (size_t)i;
            o[0] = n0.x; o[1] = n0.y; o[2] = n0.z; o[3] = n0.w; o[4] = n1.x; o[5] = n1.y;
        }
        if (meta4) {
            int skip, info;
            memcpy(&skip, &n1.z, 4);
            memcpy(&info, &n1.w, 4);
            int* o = meta4 + 4 * (size_t)i;
            o[0] = skip; o[1] = info & 0xffffff; o[2] = (info >> 24) & 3; o[3] = (info >> 26) & 31;
        }
    }
    if (slot_triangle) memcpy(slot_triangle, t->slot_tri.data(), t->slot_tri.size() * sizeof(int));
    return CRT_OK;
}

int crt_write_timing_csv(const char* path, int n_frames, const int* frame, const double* rendering,
                         const double* update, double build) {
    // kernel.cu:61 header row, kernel.cu:18,37 build row {"", "", "", build}, core/render.h:232 frame rows
    // {frame, rendering, update, ""}; fields are std::to_string(double) = "%f"; writeCSV (CSVWriter.h:8-32)
    // joins with ',' and ends every row with '\n'.
    if (!path || n_frames < 0 || (n_frames > 0 && (!frame || !rendering || !update)))
        return fail(CRT_ERR_INVALID, "crt_write_timing_csv: bad argument");
    FILE* f = fopen(path, "w");
    if (!f) return fail(CRT_ERR_IO, "crt_write_timing_csv: cannot open %s", path);
    fprintf(f, "frame,rendering,update,build\n");
    fprintf(f, ",,,%f\n", build);
    for (int i = 0; i < n_frames; ++i) fprintf(f, "%d,%f,%f,\n", frame[i], rendering[i], update[i]);
    if (fclose(f) != 0) return fail(CRT_ERR_IO, "crt_write_timing_csv: write to %s failed", path);
    return CRT_OK;
}

}  // extern "C"
