// crt_debug_kernels.hpp — diagnostics kernels behind include/crt_debug.h (included by crt_kernels.hip last).
//
// Nothing here is on a render path.  These kernels call the SAME device functions the render kernels call — the
// list walk over the per-frame exact records (walk / exact_test), the any-origin triangle test (tri_hit_base,
// list_trace_any), the tree walks (bvh_trace with skip links or pair records), the two forms of AABB::hit and the
// refit's box helpers — for caller-supplied rays and boxes, so that tests/test_ref_hit.py can hold them to the
// reference's own hitable/*.h, shapes/triangle.h and core/aabb.h running on the same GPU (oracle/ref_hit.hip).
#pragma once

struct DebugRays {
    const float* origin;      // 3 floats (shared_origin) or n x 3
    const float* direction;   // n x 3: the ray as render() hands it to world->hit (un-normalised)
    int n, shared_origin;
    int* hit;
    float* t;
    int* index;               // list index (list world) or slot (BVH world)
    float* normal;            // n x 3 rec.normal
    float* point;             // n x 3 rec.p (modes with an any-origin test; zeros otherwise)
};

// mode 0: list world, camera-origin records (`rec` = tri_prepass for the shared origin): what render_list_kernel runs
// mode 1: list world, any origin (base records): what render_generic_kernel's list world runs
// mode 2: BVH world, camera-origin records in slot order, skip links: render_bvh_kernel's walk
// mode 3: BVH world, any origin, skip links;  mode 4: the same over the pair records with one LDS stack per lane
template <bool FAST>
__global__ void __launch_bounds__(256)
debug_trace_rays_kernel(DebugRays q, int mode, const float* __restrict__ rec, ShadeScene sc, BvhView b) {
    __shared__ int stack[kPairArea * 256];
    const int i_raw = blockIdx.x * 256 + threadIdx.x;
    const int i = min(i_raw, q.n - 1);            // lanes past the end replay the last ray (the walks vote over whole wavefronts)
    const bool store = i_raw < q.n;
    const float* op = q.origin + (q.shared_origin ? 0 : 3 * (size_t)i);
    const V3 o = V3{op[0], op[1], op[2]};
    const V3 d0 = V3{q.direction[3 * (size_t)i], q.direction[3 * (size_t)i + 1], q.direction[3 * (size_t)i + 2]};
    float best_t = FLT_MAX;
    int best = -1;
    V3 used = V3{0.0f, 0.0f, 0.0f};
    bool have_point = false;
    if (mode == 0) {
        const V3 d1 = FAST ? unit_vector_fast(d0) : unit_vector(d0);      // world list's ScaleRay, then the triangle's
        const V3 d2 = FAST ? unit_vector_fast(d1) : unit_vector(d1);
        const int n_tris = sc.n_tris;
        walk<FAST>(rec, 0, n_tris, [](int k) { return k; }, d2, best_t, best);
    } else if (mode == 1) {
        const V3 d1 = FAST ? unit_vector_fast(d0) : unit_vector(d0);
        list_trace_any<FAST>(sc, o, d1, best_t, best, used);
        have_point = true;
    } else if (mode == 2) {
        const V3 d1 = FAST ? unit_vector_fast(d0) : unit_vector(d0);
        bvh_trace<FAST, true>(b, rec, o, d1, 0, best_t, best, used);
    } else {
        const V3 d1 = FAST ? unit_vector_fast(d0) : unit_vector(d0);
        bvh_trace<FAST, false>(b, sc.base, o, d1, 0, best_t, best, used, mode == 4 ? stack + threadIdx.x : nullptr, 256);
        have_point = true;
    }
    if (!store) return;
    q.hit[i] = best >= 0 ? 1 : 0;
    q.t[i] = best_t;
    q.index[i] = best;
    V3 nrm = V3{0.0f, 0.0f, 0.0f}, pt = V3{0.0f, 0.0f, 0.0f};
    if (best >= 0) {
        const float* n = sc.shade_n + (size_t)best * 3;
        nrm = V3{n[0], n[1], n[2]};
        if (have_point) pt = V3{o.x + best_t * used.x, o.y + best_t * used.y, o.z + best_t * used.z};   // triangle.h:95
    }
    q.normal[3 * (size_t)i] = nrm.x; q.normal[3 * (size_t)i + 1] = nrm.y; q.normal[3 * (size_t)i + 2] = nrm.z;
    q.point[3 * (size_t)i] = pt.x; q.point[3 * (size_t)i + 1] = pt.y; q.point[3 * (size_t)i + 2] = pt.z;
}

hipError_t launch_debug_trace_rays(int mode, bool fast, const float* d_origin, int shared_origin, const float* d_direction, int n,
                                   const float* d_rec, const float* d_base, const float* d_shade_n, int n_tris,
                                   const float4* d_nodes, int n_nodes, const float4* d_pairs, const int* d_partner,
                                   const float rot[9], int* d_hit, float* d_t, int* d_index, float* d_normal, float* d_point,
                                   hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    DebugRays q{d_origin, d_direction, n, shared_origin, d_hit, d_t, d_index, d_normal, d_point};
    ShadeScene sc;
    sc.base = d_base; sc.shade_n = d_shade_n; sc.material_of = nullptr; sc.materials = nullptr; sc.n_tris = n_tris;
    BvhView b;
    b.nodes = d_nodes; b.n_nodes = n_nodes;
    b.pairs = mode == 4 ? d_pairs : nullptr;
    b.partner = d_partner;
    for (int k = 0; k < 9; ++k) { sc.rot[k] = rot[k]; b.rot[k] = rot[k]; }
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (fast) hipLaunchKernelGGL(debug_trace_rays_kernel<true>, grid, block, 0, stream, q, mode, d_rec, sc, b);
    else hipLaunchKernelGGL(debug_trace_rays_kernel<false>, grid, block, 0, stream, q, mode, d_rec, sc, b);
    return hipGetLastError();
}

// AABB::hit(r, 0.001, FLT_MAX) (core/aabb.h:30-43) in the kernels' two forms: form 0 box_hit_ieee (the literal loop),
// form 1 box_hit_fast (div_by_rcp quotients, one comparison at the end; valid on bounded scenes and ordinary directions
// — the caller supplies such inputs).  box6: n x {min xyz, max xyz}.
__global__ void __launch_bounds__(256)
debug_box_hit_kernel(int form, int n, const float* __restrict__ box6, const float* __restrict__ origin,
                     const float* __restrict__ direction, int* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* bx = box6 + 6 * (size_t)i;
    const float4 n0 = make_float4(bx[0], bx[1], bx[2], bx[3]);
    const float4 n1 = make_float4(bx[4], bx[5], 0.0f, 0.0f);
    const V3 o = V3{origin[3 * (size_t)i], origin[3 * (size_t)i + 1], origin[3 * (size_t)i + 2]};
    const V3 d = V3{direction[3 * (size_t)i], direction[3 * (size_t)i + 1], direction[3 * (size_t)i + 2]};
    if (form == 0) out[i] = box_hit_ieee(n0, n1, o, d) ? 1 : 0;
    else out[i] = box_hit_fast(n0, n1, o, d, V3{reciprocal<true>(d.x), reciprocal<true>(d.y), reciprocal<true>(d.z)}) ? 1 : 0;
}

// the refit's box helpers: op 0 surrounding(a, b) (aabb.h:58-66), op 1 transform_aabb(rot0, a) (transform.h:22-31 with
// rotation 0, position 0)
__global__ void __launch_bounds__(256)
debug_box_op_kernel(int op, int n, const float* __restrict__ a6, const float* __restrict__ b6, RotM rot, float* __restrict__ out6) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    BoxF a, b, r;
    for (int c = 0; c < 3; ++c) {
        a.mn[c] = a6[6 * (size_t)i + c]; a.mx[c] = a6[6 * (size_t)i + 3 + c];
        b.mn[c] = b6[6 * (size_t)i + c]; b.mx[c] = b6[6 * (size_t)i + 3 + c];
    }
    r = op == 0 ? surrounding(a, b) : transform_aabb(rot.m, a);
    for (int c = 0; c < 3; ++c) { out6[6 * (size_t)i + c] = r.mn[c]; out6[6 * (size_t)i + 3 + c] = r.mx[c]; }
}

hipError_t launch_debug_box(int what, int form, int n, const float* d_a6, const float* d_b6, const float* d_origin,
                            const float* d_direction, const float rot[9], int* d_flag, float* d_out6, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (what == 0) {
        hipLaunchKernelGGL(debug_box_hit_kernel, grid, block, 0, stream, form, n, d_a6, d_origin, d_direction, d_flag);
    } else {
        RotM r;
        for (int k = 0; k < 9; ++k) r.m[k] = rot[k];
        hipLaunchKernelGGL(debug_box_op_kernel, grid, block, 0, stream, form, n, d_a6, d_b6, r, d_out6);
    }
    return hipGetLastError();
}
