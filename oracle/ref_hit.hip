// oracle/ref_hit.hip — TEST INFRASTRUCTURE, not part of the product.
//
// Runs the reference's OWN hit path on the GPU: hitable/hitable.h, hitable/transform.h, hitable/hitable_list.h,
// shapes/triangle.h, core/aabb.h and core/ray.h are `__device__` code that includes nothing but <float.h>, core/vec3.h
// and each other, so the image's HIP compiler builds them for gfx950 as they are.  They are included from where they
// lie under /root/reference (-I passed by oracle/Makefile: target `ref`); nothing is copied, nothing is stood in for;
// the output goes to oracle/_ref/libref_hit.so, which travels to the GPU box as a binary like libref_vec3.so.
// The kernels below are this file's own: they build the object graph with device `new` the way the reference's
// add_mesh_withNormal does (createScene.h:175-190: HitableList(n), one Transform(0,0,1) per Triangle, reversed
// winding, the 6-argument Triangle constructor; the world list's Transform is reset to identity, the rule bvh.h:81
// applies and DESIGN §1 fixes for the list path), then call the reference's member functions for caller-supplied
// rays and boxes and store what they return.
//
// What it pins (tests/test_ref_hit.py, -m gpu): Hitable::hit + Transform::TransformRay (rows a5, a7),
// HitableList::collision_detection (a8), Triangle::collision_detection (a9), HitRecord (a10), and f1's box arithmetic
// AABB::hit, surrounding_box, Triangle::bounding_box through GetBV, Transform::TransformAABB — against the oracle's
// restatement (o_world_hit, o_bvh_*) and the product's kernels.  NOT covered (their headers need cuRAND / Thrust /
// cuda_runtime.h, absent from this image): BVHNode's constructor and merge rule, the materials, the camera, `render`.
//
// The winning triangle's index is not part of HitRecord; every Triangle gets `Material* = (Material*)(i + 1)`, a
// value the hit path copies into rec.mat_ptr and never dereferences (triangle.h:95), and the kernel reports it.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>
#include <math.h>
#include "hitable/hitable_list.h"
#include "shapes/triangle.h"

namespace {

struct RefScene {
    HitableList* world;      // depth 0: the list add_mesh_withNormal builds
    Hitable*     nested;     // the same list wrapped in `levels` one-element HitableLists (each level = one more Hitable::hit)
    int          levels;
    int          n;
};

__global__ void build_kernel(RefScene* sc, const float* points, const float* idx3f, const float* normals, int n_tris,
                             int index_base, int levels, int* err) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    HitableList* list = new HitableList(n_tris);                                   // createScene.h:179
    if (!list || !list->list || !list->transform) { *err = 1; return; }
    list->transform->ResetTransform();                                             // identity world transform (bvh.h:81's rule)
    for (int i = 0; i < n_tris; i++) {
        vec3 idx(idx3f[3 * i], idx3f[3 * i + 1], idx3f[3 * i + 2]);
        const int i0 = int(idx[0]) - index_base, i1 = int(idx[1]) - index_base, i2 = int(idx[2]) - index_base;
        vec3 v[3] = { vec3(points[3 * i2], points[3 * i2 + 1], points[3 * i2 + 2]),
                      vec3(points[3 * i1], points[3 * i1 + 1], points[3 * i1 + 2]),
                      vec3(points[3 * i0], points[3 * i0 + 1], points[3 * i0 + 2]) };   // createScene.h:183
        Transform* transform = new Transform(vec3(0), vec3(0), vec3(1));           // :184
        Material* tag = (Material*)(uintptr_t)(i + 1);
        Triangle* tri = normals
            ? new Triangle(v, vec3(normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]), tag, false, transform, false)  // :185
            : new Triangle(v, tag, false, transform, false);                       // triangle.h:8-19
        if (!transform || !tri) { *err = 2; return; }
        list->list[i] = tri;
    }
    Hitable* top = list;
    for (int l = 0; l < levels; l++) {                                             // one more Hitable::hit level each
        HitableList* wrap = new HitableList(1);
        if (!wrap || !wrap->list || !wrap->transform) { *err = 3; return; }
        wrap->transform->ResetTransform();
        wrap->list[0] = top;
        top = wrap;
    }
    sc->world = list; sc->nested = top; sc->levels = levels; sc->n = n_tris;
    *err = 0;
}

// one thread = one ray through (*world)->hit(r, 0.001, FLT_MAX, rec, frame)   (core/render.h:75)
__global__ void hit_kernel(const RefScene* sc, int nested, int n_rays, const float* origin, const float* direction,
                           int* hit_out, float* t_out, int* index_out, float* normal_out, float* p_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rays) return;
    Ray r(vec3(origin[3 * i], origin[3 * i + 1], origin[3 * i + 2]),
          vec3(direction[3 * i], direction[3 * i + 1], direction[3 * i + 2]));
    HitRecord rec;
    rec.t = 0.0f; rec.normal = vec3(0.0f, 0.0f, 0.0f); rec.p = vec3(0.0f, 0.0f, 0.0f); rec.mat_ptr = nullptr;
    Hitable* w = nested ? sc->nested : (Hitable*)sc->world;
    const bool h = w->hit(r, 0.001f, FLT_MAX, rec, 0);
    hit_out[i] = h ? 1 : 0;
    t_out[i] = rec.t;
    index_out[i] = int((uintptr_t)rec.mat_ptr) - 1;
    for (int c = 0; c < 3; c++) { normal_out[3 * i + c] = rec.normal[c]; p_out[3 * i + c] = rec.p[c]; }
}

// one thread = one (box, ray): AABB::hit(r, t_min, t_max)   (core/aabb.h:30-43)
__global__ void aabb_hit_kernel(int n, const float* bmin, const float* bmax, const float* origin, const float* direction,
                                const float* t_min, const float* t_max, int* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    AABB box(vec3(bmin[3 * i], bmin[3 * i + 1], bmin[3 * i + 2]), vec3(bmax[3 * i], bmax[3 * i + 1], bmax[3 * i + 2]));
    Ray r(vec3(origin[3 * i], origin[3 * i + 1], origin[3 * i + 2]),
          vec3(direction[3 * i], direction[3 * i + 1], direction[3 * i + 2]));
    out[i] = box.hit(r, t_min[i], t_max[i]) ? 1 : 0;
}

// op 0: surrounding_box(box a, box b) (aabb.h:58-66); op 1: Transform(position = b.min, rotation = b.max, scale 1)
// .TransformAABB(box a) (transform.h:22-31)
__global__ void aabb_op_kernel(int op, int n, const float* amin, const float* amax, const float* bmin, const float* bmax,
                               float* omin, float* omax) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    AABB a(vec3(amin[3 * i], amin[3 * i + 1], amin[3 * i + 2]), vec3(amax[3 * i], amax[3 * i + 1], amax[3 * i + 2]));
    vec3 p(bmin[3 * i], bmin[3 * i + 1], bmin[3 * i + 2]), q(bmax[3 * i], bmax[3 * i + 1], bmax[3 * i + 2]);
    AABB o;
    if (op == 0) o = surrounding_box(a, AABB(p, q));
    else { Transform t(p, q, vec3(1)); o = a; t.TransformAABB(o); }
    for (int c = 0; c < 3; c++) { omin[3 * i + c] = o.min()[c]; omax[3 * i + c] = o.max()[c]; }
}

// one thread = one triangle of the scene: Hitable::GetBV = Triangle::bounding_box + TransformAABB (hitable.h:45-50,
// triangle.h:102-115); which == 1: the world list's GetBV (HitableList::bounding_box, hitable_list.h:94-118) by thread 0
__global__ void bv_kernel(const RefScene* sc, int which, float* omin, float* omax, int* flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (which == 1) {
        if (i != 0) return;
        AABB box;
        flag[0] = sc->world->GetBV(0.0f, 1.0f, box) ? 1 : 0;
        for (int c = 0; c < 3; c++) { omin[c] = box.min()[c]; omax[c] = box.max()[c]; }
        return;
    }
    if (i >= sc->n) return;
    AABB box;
    flag[i] = sc->world->list[i]->GetBV(0.0f, 1.0f, box) ? 1 : 0;
    for (int c = 0; c < 3; c++) { omin[3 * i + c] = box.min()[c]; omax[3 * i + c] = box.max()[c]; }
}

// createScene.h:99-110 update_pose writes new vertices with Triangle::SetVertices (triangle.h:40-45)
__global__ void set_vertices_kernel(RefScene* sc, const float* points, const float* idx3f, int index_base) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= sc->n) return;
    vec3 idx(idx3f[3 * i], idx3f[3 * i + 1], idx3f[3 * i + 2]);
    const int i0 = int(idx[0]) - index_base, i1 = int(idx[1]) - index_base, i2 = int(idx[2]) - index_base;
    vec3 v[3] = { vec3(points[3 * i2], points[3 * i2 + 1], points[3 * i2 + 2]),
                  vec3(points[3 * i1], points[3 * i1 + 1], points[3 * i1 + 2]),
                  vec3(points[3 * i0], points[3 * i0 + 1], points[3 * i0 + 2]) };
    ((Triangle*)sc->world->list[i])->SetVertices(v);
}

template <class T> struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    bool alloc(size_t count) { n = count; return hipMalloc((void**)&p, (count ? count : 1) * sizeof(T)) == hipSuccess; }
    bool up(const T* h) { return hipMemcpy(p, h, n * sizeof(T), hipMemcpyHostToDevice) == hipSuccess; }
    bool down(T* h) const { return hipMemcpy(h, p, n * sizeof(T), hipMemcpyDeviceToHost) == hipSuccess; }
    ~DevBuf() { if (p) (void)hipFree(p); }
};

struct Handle { RefScene* d_scene; int n_tris; int levels; };

}  // namespace

extern "C" {

// Device objects are never freed (the reference leaks them too, SURVEY §8b "Ownership"); one process builds a few scenes.
void* ref_hit_scene_create(const float* points, int n_points, const float* idx3f, int n_tris, const float* normals,
                           int index_base, int levels) {
    if (n_tris < 0 || n_points <= 0 || levels < 0) return nullptr;
    for (int i = 0; i < 3 * n_tris; i++) {
        const int k = int(idx3f[i]) - index_base;
        if (k < 0 || k >= n_points) return nullptr;
    }
    static bool heap_set = false;
    if (!heap_set) {   // the reference's own limits (kernel.cu:65,67): device `new` needs heap, the virtual hit() chain needs stack
        (void)hipDeviceSetLimit(hipLimitMallocHeapSize, size_t(1) << 30);
        (void)hipDeviceSetLimit(hipLimitStackSize, 16384);
        (void)hipGetLastError();
        heap_set = true;
    }
    DevBuf<float> dp, di, dn;
    DevBuf<int> derr;
    if (!dp.alloc(size_t(3) * n_points) || !dp.up(points) || !di.alloc(size_t(3) * n_tris) || !di.up(idx3f) || !derr.alloc(1))
        return nullptr;
    if (normals && (!dn.alloc(size_t(3) * n_tris) || !dn.up(normals))) return nullptr;
    RefScene* d_scene = nullptr;
    if (hipMalloc((void**)&d_scene, sizeof(RefScene)) != hipSuccess) return nullptr;
    int err = -1;
    (void)hipMemcpy(derr.p, &err, sizeof(int), hipMemcpyHostToDevice);
    build_kernel<<<1, 1>>>(d_scene, dp.p, di.p, normals ? dn.p : nullptr, n_tris, index_base, levels, derr.p);
    if (hipDeviceSynchronize() != hipSuccess || !derr.down(&err) || err != 0) { (void)hipFree(d_scene); return nullptr; }
    return new Handle{d_scene, n_tris, levels};
}

void ref_hit_scene_free(void* h) {
    Handle* s = (Handle*)h;
    if (!s) return;
    (void)hipFree(s->d_scene);
    delete s;
}

int ref_hit_scene_set_vertices(void* h, const float* points, int n_points, const float* idx3f, int index_base) {
    Handle* s = (Handle*)h;
    if (!s) return -1;
    DevBuf<float> dp, di;
    if (!dp.alloc(size_t(3) * n_points) || !dp.up(points) || !di.alloc(size_t(3) * s->n_tris) || !di.up(idx3f)) return -2;
    set_vertices_kernel<<<(s->n_tris + 63) / 64, 64>>>(s->d_scene, dp.p, di.p, index_base);
    return hipDeviceSynchronize() == hipSuccess ? 0 : -3;
}

// nested = 0: the world list; 1: the list behind `levels` more Hitable::hit wrappers
int ref_hit_rays(void* h, int nested, int n_rays, const float* origin, const float* direction, int* hit_out, float* t_out,
                 int* index_out, float* normal_out, float* p_out) {
    Handle* s = (Handle*)h;
    if (!s || n_rays < 0) return -1;
    if (n_rays == 0) return 0;
    DevBuf<float> dorg, ddir, dt, dnrm, dpp;
    DevBuf<int> dh, didx;
    const size_t n = (size_t)n_rays;
    if (!dorg.alloc(3 * n) || !dorg.up(origin) || !ddir.alloc(3 * n) || !ddir.up(direction) || !dt.alloc(n) ||
        !dnrm.alloc(3 * n) || !dpp.alloc(3 * n) || !dh.alloc(n) || !didx.alloc(n))
        return -2;
    hit_kernel<<<(n_rays + 63) / 64, 64>>>(s->d_scene, nested, n_rays, dorg.p, ddir.p, dh.p, dt.p, didx.p, dnrm.p, dpp.p);
    if (hipDeviceSynchronize() != hipSuccess) return -3;
    return (dh.down(hit_out) && dt.down(t_out) && didx.down(index_out) && dnrm.down(normal_out) && dpp.down(p_out)) ? 0 : -4;
}

int ref_aabb_hit(int n, const float* bmin, const float* bmax, const float* origin, const float* direction,
                 const float* t_min, const float* t_max, int* out) {
    if (n <= 0) return n == 0 ? 0 : -1;
    DevBuf<float> a, b, o, d, t0, t1;
    DevBuf<int> r;
    const size_t m = (size_t)n;
    if (!a.alloc(3 * m) || !a.up(bmin) || !b.alloc(3 * m) || !b.up(bmax) || !o.alloc(3 * m) || !o.up(origin) ||
        !d.alloc(3 * m) || !d.up(direction) || !t0.alloc(m) || !t0.up(t_min) || !t1.alloc(m) || !t1.up(t_max) || !r.alloc(m))
        return -2;
    aabb_hit_kernel<<<(n + 63) / 64, 64>>>(n, a.p, b.p, o.p, d.p, t0.p, t1.p, r.p);
    if (hipDeviceSynchronize() != hipSuccess) return -3;
    return r.down(out) ? 0 : -4;
}

int ref_aabb_op(int op, int n, const float* amin, const float* amax, const float* bmin, const float* bmax, float* omin,
                float* omax) {
    if (n <= 0 || op < 0 || op > 1) return n == 0 ? 0 : -1;
    DevBuf<float> a, b, c, d, e, f;
    const size_t m = (size_t)n;
    if (!a.alloc(3 * m) || !a.up(amin) || !b.alloc(3 * m) || !b.up(amax) || !c.alloc(3 * m) || !c.up(bmin) ||
        !d.alloc(3 * m) || !d.up(bmax) || !e.alloc(3 * m) || !f.alloc(3 * m))
        return -2;
    aabb_op_kernel<<<(n + 63) / 64, 64>>>(op, n, a.p, b.p, c.p, d.p, e.p, f.p);
    if (hipDeviceSynchronize() != hipSuccess) return -3;
    return (e.down(omin) && f.down(omax)) ? 0 : -4;
}

// which = 0: per-triangle GetBV (omin/omax: n_tris x 3, flag: n_tris); 1: the list's GetBV (3 + 3 floats, 1 flag)
int ref_hit_bounding_boxes(void* h, int which, float* omin, float* omax, int* flag) {
    Handle* s = (Handle*)h;
    if (!s || which < 0 || which > 1) return -1;
    const size_t m = which ? 1 : (size_t)s->n_tris;
    if (m == 0) return 0;
    DevBuf<float> a, b;
    DevBuf<int> f;
    if (!a.alloc(3 * m) || !b.alloc(3 * m) || !f.alloc(m)) return -2;
    bv_kernel<<<which ? 1 : (s->n_tris + 63) / 64, 64>>>(s->d_scene, which, a.p, b.p, f.p);
    if (hipDeviceSynchronize() != hipSuccess) return -3;
    return (a.down(omin) && b.down(omax) && f.down(flag)) ? 0 : -4;
}

}  // extern "C"
