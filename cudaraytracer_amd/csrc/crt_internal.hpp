// crt_internal.hpp — host-side internals shared by the translation units of libcrt_hip.so
// (crt_api.cpp, crt_bvh.cpp, crt_multi.cpp).  Not part of the ABI.
#pragma once

#include "../../include/crt.h"
#include "../../include/crt_debug.h"
#include "crt_device.hpp"

#include <math.h>
#include <stdint.h>

#include <mutex>
#include <vector>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define fail(...) crt::set_error(__VA_ARGS__)

#define HIP_TRY(expr)                                                                  \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess)                                                          \
            return fail(CRT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                           \
    } while (0)

namespace crt_host {

// Launch context: everything a render launch writes besides its outputs — the per-frame derived
// triangle records (exact + classification), the scratch of the tiled variant, the timing events,
// the cached occupancy — lives here, one context per (device, stream).  Work enqueued on one
// stream is ordered by the stream, so launches of one context never overlap; launches on
// different streams or devices use different contexts and share nothing but read-only scene data.
// Lifetime: created at the first launch on (device, stream); buffers grow with the frames and scenes it sees; freed by
// crt_stream_release (which a host calls before it destroys the stream; crt_multi_free does for its own streams).
// A hipStream_t is a pointer and the runtime may hand a destroyed stream's address to a new one: every context
// remembers hipStreamGetId of its stream, and a handle that comes back with another id is a NEW stream — its
// context is reset (device synchronised once, events and cached occupancy dropped; the scratch buffers carry no
// state from one frame to the next and are kept).
struct Context {
    unsigned long long id = 0;       // unique per context (crt_last_kernel_* look their context up by id)
    int device = 0;
    hipStream_t stream = nullptr;
    unsigned long long stream_id = 0;   // hipStreamGetId at creation
    std::mutex mu;                   // host-side: two threads enqueueing on the same stream
    int pins = 0;                    // holders of a pointer from context_for (guarded by the context table's mutex)
    crt::TiledWork w{};
    size_t pool_entries = 0, tiles = 0, log_entries = 0;
    int n_cus = 0;
    float* d_rec = nullptr;          // n x 16 derived records (tri_prepass output)
    float4* d_cls = nullptr;         // 5 planes x n_pad classification records (cls_prepass output)
    size_t rec_cap = 0, cls_cap = 0; // in triangles
    hipEvent_t ev_start = nullptr, ev_stop = nullptr, ev_mid = nullptr;
    bool ev_has_mid = false, ev_valid = false;
    crt::LaunchCache cache, cache_bvh, cache_generic;
    unsigned* d_pixel_queue = nullptr;   // render_generic_kernel's queue cursor (kGenericQueueWords)
};

Context* context_for(int device, hipStream_t stream);   // PINNED: pair every call with unpin_context (CtxPin)
void unpin_context(Context* c);
struct CtxPin {                      // scope guard of a pinned context
    Context* c;
    explicit CtxPin(Context* c_) : c(c_) {}
    ~CtxPin() { unpin_context(c); }
    CtxPin(const CtxPin&) = delete;
    CtxPin& operator=(const CtxPin&) = delete;
};
// frees the context of (device, stream) if there is one (the stream is synchronised first); CRT_OK either way
int release_context(int device, hipStream_t stream);
void release_onecall(int device);   // what crt_render keeps per device
int ensure_records(Context& cx, int n_tris, int n_pad);
int ensure_workspace(Context& ws, int nx, int rows, int n_tris);
bool have_device();
void rotation_zero_matrix(float m[9]);
const std::vector<uint32_t>& jump_table();   // crt_api.cpp: the 64 XORWOW jump matrices rng_init_kernel walks
void debug_thrust_sort(int* order, int n, const float* key);                     // crt_bvh.cpp: the build's sort
void debug_host_box_op(int op, const float* a6, const float* b6, float* out6);   // crt_bvh.cpp: the build's box helpers
extern int g_force_stream_recycle;                                               // crt_api.cpp: crt_debug_force_stream_recycle

// ---------------------------------------------------------------- host vec3 helpers
struct H3 {
    float x, y, z;
};
inline H3 h3(const float* p) { return H3{p[0], p[1], p[2]}; }
inline H3 hsub(H3 a, H3 b) { return H3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline H3 hmul(float t, H3 v) { return H3{t * v.x, t * v.y, t * v.z}; }
inline H3 hcross(H3 a, H3 b) {   // core/vec3.h:113-117
    return H3{(a.y * b.z - a.z * b.y), (-(a.x * b.z - a.z * b.x)), (a.x * b.y - a.y * b.x)};
}
inline H3 hunit(H3 v) {          // core/vec3.h:191-193
    float len = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    return H3{v.x / len, v.y / len, v.z / len};
}
// rotate(origin, rotation) of core/vec3.h:200-217, literally (used with rotation 0 only).
inline H3 hrotate(H3 o, H3 rot) {
    float radiansX = (float)((M_PI / 180.) * rot.x);
    float sin_X = sinf(radiansX), cos_X = cosf(radiansX);
    float radiansY = (float)((M_PI / 180.) * rot.y);
    float sin_Y = sinf(radiansY), cos_Y = cosf(radiansY);
    float radiansZ = (float)(-(M_PI / 180.) * rot.z);
    float sin_Z = sinf(radiansZ), cos_Z = cosf(radiansZ);
    H3 r0{cos_Y * cos_Z, -cos_Y * sin_Z, sin_Y};
    H3 r1{sin_X * sin_Y * cos_Z + cos_X * sin_Z, -sin_X * sin_Y * sin_Z + cos_X * cos_Z, -sin_X * cos_Y};
    H3 r2{-cos_X * sin_Y * cos_Z + sin_X * sin_Z, cos_X * sin_Y * sin_Z + sin_X * cos_Z, cos_X * cos_Y};
    return H3{o.x * r0.x + o.y * r0.y + o.z * r0.z, o.x * r1.x + o.y * r1.y + o.z * r1.z,
              o.x * r2.x + o.y * r2.y + o.z * r2.z};
}


}  // namespace crt_host

// BVH world of a scene (crt_bvh.cpp): the tree flat on the device, triangles in leaf order.
struct crt_bvh {
    int n_nodes = 0, max_depth = 0;
    float4* d_nodes = nullptr;          // 2 float4 per node, preorder (crt_bvh_kernels.hpp)
    float4* d_pairs = nullptr;          // 4 float4 per INNER node: both children's boxes + references (bvh_walk_pairs); null for trees deeper than kPairStackDepth
    int* d_pair_flat = nullptr;         // per pair: the flat node indices of its two children (bvh_pairs_kernel copies their boxes)
    int n_pairs = 0;
    float* d_base_slots = nullptr;      // n x 12 base records in slot order
    float* d_shade_n_slots = nullptr;   // n x 3 shading normals in slot order (rotate(., 0) applied depth + 3 times)
    int* d_idx_slots = nullptr;         // n x 3 vertex indices in slot order (update_pose)
    int* d_material_slots = nullptr;    // per-slot material index (null while the scene has a single material)
    // primary rays through the tiled kernels (crt_kernels.hip: shade_tiles_kernel<.., BVH = true>)
    int* d_partner = nullptr;           // per slot: the slot with the two slots of a 2-triangle leaf swapped (tie order, see BvhTiles)
    int* d_slot_leaf = nullptr;         // per slot: leaf ordinal | leaf depth << 24
    int* d_anc = nullptr;               // per leaf ordinal: node indices root..leaf, anc_stride ints (-1 padded)
    int anc_stride = 0;
    crt::BvhTiles tiles_view{};         // what the tiled kernels' BVH mode needs of the tree (device pointers; passed by value)
    int* d_level_nodes = nullptr;       // node indices grouped by depth
    std::vector<int> level_offset;      // level d = d_level_nodes[level_offset[d] .. level_offset[d + 1])
    std::vector<int> slot_tri;          // slot -> triangle index of the list
    float rot[9] = {0};                 // rotate(., 0) matrix
    bool coords_not_tiny = false;       // every nonzero vertex coordinate is >= 2^-56 in magnitude
    ~crt_bvh();
};

struct crt_scene {
    int n_tris = 0;
    int device = 0;
    float albedo[3] = {0, 0, 0};
    float* d_base = nullptr;      // n x 12: v0 v1 v2 n
    float* d_shade_n = nullptr;   // n x 3: normal after the two rotate() calls of Hitable::hit
    int n_pad = 0;                // triangle count rounded up to 64 (plane stride of the classification records)
    bool tiled_safe = false;      // all coordinates finite and |c| <= 2^38
    int n_points = 0;             // update_pose: size of the point array the indices refer to
    int* d_idx = nullptr;         // n x 3 vertex indices, 0-based (as given: idx[0], idx[1], idx[2])
    float* d_points = nullptr;    // n_points x 3 staging of the current pose
    float* h_points = nullptr;    // pinned host copy of the caller's array (the caller's buffer is free again on return)
    hipEvent_t pose_copied = nullptr;   // the last H2D copy out of h_points has finished
    std::vector<float> h_base;    // host mirror of d_base (BVH build)
    std::vector<int> h_idx;       // host mirror of d_idx
    crt_bvh* bvh = nullptr;       // crt_scene_build_bvh
    // materials (crt_scene_set_materials); default: one Lambertian(albedo), createScene.h:180
    std::vector<crt_material> h_materials;
    std::vector<int> h_material_of;   // per triangle (empty = all 0)
    void* d_materials = nullptr;      // crt_material[n]
    int* d_material_of = nullptr;     // list order (null = all 0)
    bool single_lambertian = true;    // what the specialised LambertShade kernels assume
};

// (re)uploads the scene's per-slot material indices after a BVH build or a material change
int upload_bvh_materials(crt_scene* s);

struct crt_rng {
    long long first = 0, count = 0;
    int nx = 0, strip_rows = 1, strip_step = 1;   // layout (crt_rng_create_strips); contiguous when step == 1
    int device = 0;
    uint32_t* d_state = nullptr;  // SoA 6 x count: d, v0..v4
    uint32_t* d_jump = nullptr;
};

