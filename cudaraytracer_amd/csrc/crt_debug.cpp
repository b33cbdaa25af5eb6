// crt_debug.cpp — parity hooks of include/crt_debug.h: single rays and boxes through the kernels' own device functions
// (csrc/crt_debug_kernels.hpp), the BVH build's sort on a key array.  Nothing here is on a render path.
#include "crt_internal.hpp"

#include <algorithm>
#include <cmath>
#include <vector>

using namespace crt_host;

namespace crt {
hipError_t launch_debug_trace_rays(int mode, bool fast, const float* d_origin, int shared_origin, const float* d_direction, int n,
                                   const float* d_rec, const float* d_base, const float* d_shade_n, int n_tris,
                                   const float4* d_nodes, int n_nodes, const float4* d_pairs, const int* d_partner,
                                   const float rot[9], int* d_hit, float* d_t, int* d_index, float* d_normal, float* d_point,
                                   hipStream_t stream);
hipError_t launch_debug_box(int what, int form, int n, const float* d_a6, const float* d_b6, const float* d_origin,
                            const float* d_direction, const float rot[9], int* d_flag, float* d_out6, hipStream_t stream);
}  // namespace crt

namespace {
struct DevMem {   // one allocation, freed on scope exit
    void* p = nullptr;
    ~DevMem() { if (p) (void)hipFree(p); }
};
}  // namespace

extern "C" {

int crt_debug_trace_rays(const crt_scene* scene, int mode, int arith, int n_rays, const float* origin, const float* direction,
                         int* hit, float* t, int* index, float* normal, float* point) {
    if (!scene || mode < 0 || mode > 4 || arith < 0 || arith > 1 || n_rays < 0 ||
        (n_rays > 0 && (!origin || !direction || !hit || !t || !index || !normal || !point)))
        return fail(CRT_ERR_INVALID, "crt_debug_trace_rays: bad argument");
    if (!have_device()) return fail(CRT_ERR_NO_DEVICE, "crt_debug_trace_rays: no HIP device");
    int device = 0;
    HIP_TRY(hipGetDevice(&device));
    if (scene->device != device) return fail(CRT_ERR_INVALID, "crt_debug_trace_rays: the scene lives on device %d", scene->device);
    const bool bvh = mode >= 2;
    if (bvh && !scene->bvh) return fail(CRT_ERR_INVALID, "crt_debug_trace_rays: the scene has no BVH (crt_scene_build_bvh)");
    if (mode == 4 && !scene->bvh->d_pairs) return fail(CRT_ERR_UNSUPPORTED, "crt_debug_trace_rays: the tree has no pair records");
    if (n_rays == 0 || scene->n_tris == 0) {
        for (int i = 0; i < n_rays; ++i) { hit[i] = 0; t[i] = 3.402823466e+38f; index[i] = -1; }
        std::fill(normal, normal + 3 * (size_t)n_rays, 0.0f);
        std::fill(point, point + 3 * (size_t)n_rays, 0.0f);
        return CRT_OK;
    }
    const bool shared = mode == 0 || mode == 2;
    if (arith == 1) {   // the conditions under which the render entry points choose the fast forms (crt_api.cpp: render_impl)
        bool ok = scene->tiled_safe && (!bvh || scene->bvh->coords_not_tiny);
        const size_t no = shared ? 3 : 3 * (size_t)n_rays;
        for (size_t k = 0; k < no && ok; ++k) {
            const float o = origin[k];
            ok = std::isfinite(o) && fabsf(o) <= 274877906944.0f && (o == 0.0f || fabsf(o) >= 1.3877787807814457e-17f);
        }
        if (!ok) return fail(CRT_ERR_UNSUPPORTED, "crt_debug_trace_rays: scene or origin outside the range of the fast forms");
    }
    const size_t n = (size_t)n_rays, no = shared ? 3 : 3 * n;
    DevMem in, out, rec;
    HIP_TRY(hipMalloc(&in.p, (no + 3 * n) * sizeof(float)));
    HIP_TRY(hipMalloc(&out.p, (n * 2 + n + 6 * n) * sizeof(float)));
    float* d_origin = static_cast<float*>(in.p);
    float* d_dir = d_origin + no;
    int* d_hit = static_cast<int*>(out.p);
    int* d_index = d_hit + n;
    float* d_t = reinterpret_cast<float*>(d_index + n);
    float* d_normal = d_t + n;
    float* d_point = d_normal + 3 * n;
    HIP_TRY(hipMemcpy(d_origin, origin, no * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_dir, direction, 3 * n * sizeof(float), hipMemcpyHostToDevice));
    const crt_bvh* tr = bvh ? scene->bvh : nullptr;
    const float* d_base = tr ? tr->d_base_slots : scene->d_base;
    const float* d_shade_n = tr ? tr->d_shade_n_slots : scene->d_shade_n;
    if (shared) {   // the per-frame exact records for this origin (the same prepass the render entry points launch)
        HIP_TRY(hipMalloc(&rec.p, (size_t)scene->n_tris * crt::kRecFloats * sizeof(float)));
        HIP_TRY(crt::launch_tri_prepass(d_base, scene->n_tris, origin, static_cast<float*>(rec.p), nullptr));
    }
    float rot[9];
    rotation_zero_matrix(rot);
    HIP_TRY(crt::launch_debug_trace_rays(mode, arith == 1, d_origin, shared ? 1 : 0, d_dir, n_rays, static_cast<const float*>(rec.p),
                                         d_base, d_shade_n, scene->n_tris, tr ? tr->d_nodes : nullptr, tr ? tr->n_nodes : 0,
                                         tr ? tr->d_pairs : nullptr, tr ? tr->d_partner : nullptr, tr ? tr->rot : rot, d_hit, d_t,
                                         d_index, d_normal, d_point, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(hit, d_hit, n * sizeof(int), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(index, d_index, n * sizeof(int), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(t, d_t, n * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(normal, d_normal, 3 * n * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(point, d_point, 3 * n * sizeof(float), hipMemcpyDeviceToHost));
    if (tr)
        for (size_t i = 0; i < n; ++i)
            if (index[i] >= 0) index[i] = tr->slot_tri[(size_t)index[i]];
    return CRT_OK;
}

int crt_debug_box_hit(int form, int n, const float* box6, const float* origin, const float* direction, int* out) {
    if (form < 0 || form > 1 || n < 0 || (n > 0 && (!box6 || !origin || !direction || !out)))
        return fail(CRT_ERR_INVALID, "crt_debug_box_hit: bad argument");
    if (!have_device()) return fail(CRT_ERR_NO_DEVICE, "crt_debug_box_hit: no HIP device");
    if (n == 0) return CRT_OK;
    const size_t m = (size_t)n;
    DevMem mem;
    HIP_TRY(hipMalloc(&mem.p, (6 * m + 3 * m + 3 * m + m) * sizeof(float)));
    float* d_box = static_cast<float*>(mem.p);
    float *d_o = d_box + 6 * m, *d_d = d_o + 3 * m;
    int* d_out = reinterpret_cast<int*>(d_d + 3 * m);
    HIP_TRY(hipMemcpy(d_box, box6, 6 * m * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_o, origin, 3 * m * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_d, direction, 3 * m * sizeof(float), hipMemcpyHostToDevice));
    float rot[9];
    rotation_zero_matrix(rot);
    HIP_TRY(crt::launch_debug_box(0, form, n, d_box, nullptr, d_o, d_d, rot, d_out, nullptr, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, d_out, m * sizeof(int), hipMemcpyDeviceToHost));
    return CRT_OK;
}

int crt_debug_box_op(int where, int op, int n, const float* a6, const float* b6, float* out6) {
    if (where < 0 || where > 1 || op < 0 || op > 1 || n < 0 || (n > 0 && (!a6 || !b6 || !out6)))
        return fail(CRT_ERR_INVALID, "crt_debug_box_op: bad argument");
    if (where == 0) {
        for (int i = 0; i < n; ++i) debug_host_box_op(op, a6 + 6 * (size_t)i, b6 + 6 * (size_t)i, out6 + 6 * (size_t)i);
        return CRT_OK;
    }
    if (!have_device()) return fail(CRT_ERR_NO_DEVICE, "crt_debug_box_op: no HIP device");
    if (n == 0) return CRT_OK;
    const size_t m = (size_t)n;
    DevMem mem;
    HIP_TRY(hipMalloc(&mem.p, 18 * m * sizeof(float)));
    float* d_a = static_cast<float*>(mem.p);
    float *d_b = d_a + 6 * m, *d_out = d_b + 6 * m;
    HIP_TRY(hipMemcpy(d_a, a6, 6 * m * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_b, b6, 6 * m * sizeof(float), hipMemcpyHostToDevice));
    float rot[9];
    rotation_zero_matrix(rot);
    HIP_TRY(crt::launch_debug_box(1, op, n, d_a, d_b, nullptr, nullptr, rot, nullptr, d_out, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out6, d_out, 6 * m * sizeof(float), hipMemcpyDeviceToHost));
    return CRT_OK;
}

int crt_debug_thrust_sort(int n, const float* key, int* order) {
    if (n < 0 || (n > 0 && (!key || !order))) return fail(CRT_ERR_INVALID, "crt_debug_thrust_sort: bad argument");
    debug_thrust_sort(order, n, key);
    return CRT_OK;
}

int crt_debug_xorwow_jump(int k, uint32_t* rows) {
    if (k < 0 || k >= crt::kXwJumps || !rows) return fail(CRT_ERR_INVALID, "crt_debug_xorwow_jump: bad argument");
    const std::vector<uint32_t>& jt = jump_table();
    const size_t words = (size_t)crt::kXwBits * crt::kXwWords;
    std::copy(jt.begin() + (size_t)k * words, jt.begin() + (size_t)(k + 1) * words, rows);
    return CRT_OK;
}

int crt_debug_force_stream_recycle(int count) {
    const int prev = g_force_stream_recycle;
    g_force_stream_recycle = count < 0 ? 0 : count;
    return prev;
}

}  // extern "C"
