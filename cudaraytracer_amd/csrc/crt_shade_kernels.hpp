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
//   * lane = pixel, the sample loop and the bounce loop are serial per lane (RNG order);
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

template <bool FAST, bool BVH>
__device__ __forceinline__ HitInfo trace_any(const ShadeScene& sc, const BvhView& b, V3 o, V3 d, int force_mode, int* __restrict__ stack) {
    // Hitable::hit at the world level: ScaleRay normalises (transform.h:50-54)
    const V3 d1 = FAST ? unit_vector_fast(d) : unit_vector(d);
    HitInfo h;
    V3 used;
    if (BVH) bvh_trace<FAST, false>(b, sc.base, o, d1, force_mode, h.t, h.idx, used, stack, 256);
    else list_trace_any<FAST>(sc, o, d1, h.t, h.idx, used);
    if (h.idx >= 0) {
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

// shader: 0 LambertShade, 1 shade, 2 shade_normal.  256 threads = four 8x8 pixel tiles.
template <bool FAST, bool BVH>
#ifndef CRT_GENERIC_OCC
#define CRT_GENERIC_OCC 4   // waves per SIMD the register allocation aims at (the pair walk sits at 129 VGPRs without a target: 3 waves)
#endif
#if CRT_GENERIC_OCC
__global__ void __launch_bounds__(256, CRT_GENERIC_OCC)
#else
__global__ void __launch_bounds__(256)
#endif
render_generic_kernel(RenderParams p, ShadeScene sc, BvhView b, uint32_t* __restrict__ state, float* __restrict__ rgb,
                      int shader, int max_depth, int force_mode) {
    __shared__ unsigned short level_mat[kMaxDepthLevels][256];   // material indices (crt_scene_set_materials: at most 65 536 materials)
    extern __shared__ int pair_stack[];   // bvh_walk_pairs: kPairStackDepth x 256 pending right children (launched with it only when the pair records are used)
    int* const stack = (BVH && b.pairs != nullptr) ? &pair_stack[threadIdx.x] : nullptr;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int xr = blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7);
    const int lyr = blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3);
    const bool valid = xr < p.nx && lyr < p.row_count;                 // render.h:109
    if (__builtin_amdgcn_ballot_w64(valid) == 0ull) return;
    const int x = min(xr, p.nx - 1), ly = min(lyr, p.row_count - 1);    // lanes off the image replay a valid pixel (unstored)
    const int y = global_row(p.row_begin, p.strip_rows, p.strip_step, ly);
    const long long pixel_index = (long long)y * p.nx + x;
    const long long si = (long long)ly * p.nx + x, stride = p.rng_count;
    Xorwow rs;
    rs.d = state[0 * stride + si]; rs.v0 = state[1 * stride + si]; rs.v1 = state[2 * stride + si];
    rs.v2 = state[3 * stride + si]; rs.v3 = state[4 * stride + si]; rs.v4 = state[5 * stride + si];
    const V3 cam_o = V3{p.cam.origin[0], p.cam.origin[1], p.cam.origin[2]};
    const float fnx = (float)p.nx, fny = (float)p.ny;
    float colx = 0.0f, coly = 0.0f, colz = 0.0f;
    // One loop over RAYS, not over samples: every trip, each lane that has a path in flight traces its next ray —
    // the primary ray of its next sample or the next bounce of `shade` — so a lane whose path ended does not wait for
    // the longest path of the wavefront before it starts its next sample (the wavefront's time is the longest SUM of
    // a pixel's path lengths instead of the sum over samples of the longest path; the launch used to wait for its
    // longest chains of bounces: DESIGN.md §4.8).  A pixel's own order of operations — u, v, lens draws, then per
    // bounce the scatter draws — is unchanged, and its stream is its own, so pixels and states are the same bits.
    // The wave votes inside the traversals are taken over the lanes that trace in that trip.
    int smp = 0, depth = 0, levels = 0;
    bool in_flight = false;
    V3 r_o = cam_o, r_d = V3{0.0f, 0.0f, 0.0f};
    for (;;) {
        if (!in_flight && smp < p.samples) {                            // render.h:115-118: the next sample of this pixel
            const float u = ((float)x + xw_uniform(rs)) / fnx;          // render.h:116
            const float v = ((float)y + xw_uniform(rs)) / fny;          // render.h:117
            if (pixel_index == 0) {                                     // get_ray's lens sample from state[0] (render.h:118): DESIGN.md §1
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
        if (__builtin_amdgcn_ballot_w64(in_flight) == 0ull) break;
        if (in_flight) {
            const HitInfo rec = trace_any<FAST, BVH>(sc, b, r_o, r_d, force_mode, stack);
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
                        level_mat[levels][threadIdx.x] = (unsigned short)mi;
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
                        const MaterialPOD m = sc.materials[level_mat[k][threadIdx.x]];
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
    // col /= float(ns); sqrt; clip (render.h:123-128, vec3.h:182-189,119-131)
    const float k = (float)(1.0 / (double)(float)p.samples);
    colx *= k; coly *= k; colz *= k;
    colx = sqrtf(colx); coly = sqrtf(coly); colz = sqrtf(colz);
    colx = (colx > 1.0f) ? 1.0f : ((colx < 0.0f) ? 0.0f : colx);
    coly = (coly > 1.0f) ? 1.0f : ((coly < 0.0f) ? 0.0f : coly);
    colz = (colz > 1.0f) ? 1.0f : ((colz < 0.0f) ? 0.0f : colz);
    if (!valid) return;
    const size_t o = ((size_t)ly * p.nx + x) * 3;
    rgb[o + 0] = colx; rgb[o + 1] = coly; rgb[o + 2] = colz;
    state[0 * stride + si] = rs.d; state[1 * stride + si] = rs.v0; state[2 * stride + si] = rs.v1;
    state[3 * stride + si] = rs.v2; state[4 * stride + si] = rs.v3; state[5 * stride + si] = rs.v4;
}

hipError_t launch_render_generic(const RenderParams& p, const float* d_base, const float* d_shade_n, const int* d_material_of,
                                 const void* d_materials, int n_tris, const float4* d_nodes, int n_nodes, const float rot[9],
                                 uint32_t* d_state, float* d_rgb, bool bvh, bool fast, int shader, int max_depth, int force_mode,
                                 hipStream_t stream, const float4* d_pairs, const int* d_partner) {
    if (p.row_count <= 0 || p.nx <= 0) return hipSuccess;
    ShadeScene sc;
    sc.base = d_base; sc.shade_n = d_shade_n; sc.material_of = d_material_of;
    sc.materials = static_cast<const MaterialPOD*>(d_materials);
    sc.n_tris = n_tris;
    BvhView b;
    b.nodes = d_nodes; b.n_nodes = n_nodes;
    b.pairs = d_pairs; b.partner = d_partner;
    for (int k = 0; k < 9; ++k) { sc.rot[k] = rot[k]; b.rot[k] = rot[k]; }
    dim3 block(256), grid((unsigned)((p.nx + 15) / 16), (unsigned)((p.row_count + 15) / 16));
    auto fn = bvh ? (fast ? render_generic_kernel<true, true> : render_generic_kernel<false, true>)
                  : (fast ? render_generic_kernel<true, false> : render_generic_kernel<false, false>);
    const size_t lds = (bvh && d_pairs != nullptr) ? (size_t)kPairStackDepth * 256 * sizeof(int) : 0;
    hipLaunchKernelGGL(fn, grid, block, lds, stream, p, sc, b, d_state, d_rgb, shader, max_depth, force_mode);
    return hipGetLastError();
}
