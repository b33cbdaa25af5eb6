// crt_api.cpp — the C ABI of libcrt_hip.so (include/crt.h): host side of the list-render
// path.  Mirrors the reference's host functions (paths relative to CudaTest/src):
//   SetCurandState            core/deviceManage.h:130-136   -> crt_rng_create / crt_rng_reset
//   init_camera/create_camera createScene.h:142-165          -> crt_camera_make
//   create_FBXMesh            createScene.h:192-210          -> crt_scene_from_mesh
//   parseObjByName            Loader/obj_loader.h:20-133     -> crt_parse_obj / crt_scene_from_obj
//   renderListAnimation       core/render.h:159-189          -> crt_render / crt_render_rows
//   WritePng                  core/render.h:135-157          -> crt_to_rgba8 / crt_write_png
// There is no CPU compute path in this library: without a HIP device every compute entry
// point returns CRT_ERR_NO_DEVICE.
#include "crt_internal.hpp"

#include <dlfcn.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

namespace {
thread_local std::string g_err;
}

int crt::set_error(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

namespace {

thread_local int g_variant = 0;
thread_local bool g_timing = true;   // crt_set_kernel_timing: record the HIP events behind crt_last_kernel_times
thread_local unsigned long long* g_d_stats = nullptr;   // device counters when stats are enabled



}  // namespace

namespace crt_host {

namespace {
std::mutex g_ctx_mu;
std::condition_variable g_ctx_cv;     // release_context waits here until nobody holds a pin on its context
std::vector<Context*> g_contexts;
unsigned long long g_next_ctx_id = 1;
thread_local unsigned long long g_last_ctx_id = 0;   // crt_last_kernel_ms / crt_last_kernel_times: last launch of this thread

// hipStreamGetId is a hip_7.1 symbol: a process whose HIP runtime is older (PyTorch 2.10's bundled ROCm 7.0 copy) does not
// have it, so it is looked up at run time; without it every stream reports id 0 and a recycled handle is not noticed
// (its context is then simply reused, as in round 2).
unsigned long long stream_id_of(hipStream_t stream) {
    using Fn = hipError_t (*)(hipStream_t, unsigned long long*);
    static const Fn fn = reinterpret_cast<Fn>(dlsym(RTLD_DEFAULT, "hipStreamGetId"));
    unsigned long long id = 0;
    if (!fn) return 0;
    if (fn(stream, &id) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return id;
}

void free_context_memory(Context* c) {   // device memory and events of a context whose stream is idle
    if (c->d_rec) (void)hipFree(c->d_rec);
    if (c->d_cls) (void)hipFree(c->d_cls);
    if (c->w.log) (void)hipFree(c->w.log);
    if (c->w.pool) (void)hipFree(c->w.pool);
    if (c->w.tile_desc) (void)hipFree(c->w.tile_desc);
    if (c->w.queue) (void)hipFree(c->w.queue);
    if (c->w.tile_count) (void)hipFree(c->w.tile_count);
    if (c->w.counters) (void)hipFree(c->w.counters);
    if (c->d_pixel_queue) (void)hipFree(c->d_pixel_queue);
    if (c->ev_start) (void)hipEventDestroy(c->ev_start);
    if (c->ev_stop) (void)hipEventDestroy(c->ev_stop);
    if (c->ev_mid) (void)hipEventDestroy(c->ev_mid);
}
}  // namespace

int g_force_stream_recycle = 0;   // crt_debug_force_stream_recycle (guarded by g_ctx_mu where it is consumed)

// Returns the context PINNED (Context::pins, under g_ctx_mu): release_context does not free a pinned context, so the
// pointer stays valid until unpin_context — also for a caller that is still waiting for Context::mu.
Context* context_for(int device, hipStream_t stream) {
    const unsigned long long sid = stream_id_of(stream);
    std::lock_guard<std::mutex> lock(g_ctx_mu);
    for (Context* c : g_contexts)
        if (c->device == device && c->stream == stream) {
            bool recycled = c->stream_id != sid;
            if (g_force_stream_recycle > 0 && c->pins == 0) {
                --g_force_stream_recycle;
                recycled = true;
            }
            ++c->pins;
            if (recycled) {
                // the handle of a destroyed stream, handed out again: a new stream.  Nothing of the old one may still be
                // running into the scratch (one device synchronisation, once per recycled handle); time stamps and the
                // cached occupancy are dropped, the buffers are scratch without state between frames and stay.
                (void)hipDeviceSynchronize();
                c->stream_id = sid;
                c->id = g_next_ctx_id++;
                c->ev_valid = false;
                c->ev_has_mid = false;
                c->cache = crt::LaunchCache{};
                c->cache_bvh = crt::LaunchCache{};
                c->cache_generic = crt::LaunchCache{};
            }
            return c;
        }
    Context* c = new (std::nothrow) Context;
    if (!c) return nullptr;
    c->id = g_next_ctx_id++;
    c->device = device;
    c->stream = stream;
    c->stream_id = sid;
    c->pins = 1;
    g_contexts.push_back(c);
    return c;
}

void unpin_context(Context* c) {
    if (!c) return;
    std::lock_guard<std::mutex> lock(g_ctx_mu);
    if (--c->pins == 0) g_ctx_cv.notify_all();
}

int release_context(int device, hipStream_t stream) {
    Context* c = nullptr;
    {
        std::unique_lock<std::mutex> lock(g_ctx_mu);
        for (size_t i = 0; i < g_contexts.size(); ++i)
            if (g_contexts[i]->device == device && g_contexts[i]->stream == stream) {
                c = g_contexts[i];
                g_contexts.erase(g_contexts.begin() + (long)i);   // from here on nobody can look it up ...
                break;
            }
        if (c) g_ctx_cv.wait(lock, [&] { return c->pins == 0; });   // ... and whoever already had has let go of it
    }
    if (!c) return CRT_OK;
    hipError_t e;
    {
        std::lock_guard<std::mutex> lock(c->mu);   // a launch that is being enqueued on this stream finishes enqueueing first
        e = (c->stream_id == stream_id_of(stream)) ? hipStreamSynchronize(stream) : hipDeviceSynchronize();
        free_context_memory(c);
    }
    delete c;
    if (e != hipSuccess) return fail(CRT_ERR_HIP, "crt_stream_release: %s", hipGetErrorString(e));
    return CRT_OK;
}

// the context crt_last_kernel_* refer to: still alive?  (crt_stream_release may have freed it)
static Context* last_context_locked_lookup() {   // pinned like context_for's
    std::lock_guard<std::mutex> lock(g_ctx_mu);
    for (Context* c : g_contexts)
        if (c->id == g_last_ctx_id) {
            ++c->pins;
            return c;
        }
    return nullptr;
}

int ensure_records(Context& cx, int n_tris, int n_pad) {
    const size_t want = (size_t)std::max(n_tris, 1), want_pad = (size_t)std::max(n_pad, 64);
    if (want > cx.rec_cap) {
        if (cx.d_rec) (void)hipFree(cx.d_rec);
        cx.d_rec = nullptr; cx.rec_cap = 0;
        HIP_TRY(hipMalloc(&cx.d_rec, want * crt::kRecFloats * sizeof(float)));
        cx.rec_cap = want;
    }
    if (want_pad > cx.cls_cap) {
        if (cx.d_cls) (void)hipFree(cx.d_cls);
        cx.d_cls = nullptr; cx.cls_cap = 0;
        HIP_TRY(hipMalloc(&cx.d_cls, (size_t)crt::kClsPlanes * want_pad * sizeof(float4)));
        cx.cls_cap = want_pad;
    }
    return CRT_OK;
}

int ensure_workspace(Context& ws, int nx, int rows, int n_tris) {
    const int tiles_x = (nx + 7) / 8, tiles_y = (rows + 7) / 8;
    const size_t tiles = (size_t)tiles_x * tiles_y;
    // Sized from the frame and the scene (round 2 had floors of 16 M pool entries and 4 M log records = 160 MB per
    // stream whatever the frame).  The pool holds the tile lists: never more than triangles x tiles entries; 64 per
    // tile or 8 per triangle is 20x what the bench frames need (C3: sum of lists 104 k for 32 400 tiles).  The log
    // holds one record per (wavefront of 64 triangles, tile) pair with survivors: never more than wavefronts x tiles.
    // A frame that overflows either falls back to the whole-list walk for the affected tiles (variants 3 and 8 test
    // that path): slower, same pixels.
    const size_t tris = (size_t)std::max(n_tris, 1), waves = (tris + 63) / 64;
    const size_t want_pool = std::min<size_t>(std::min<size_t>(std::max<size_t>(std::max<size_t>(tiles * 64, tris * 8), (size_t)1 << 18),
                                                               tris * tiles + 64), (size_t)1 << 30);
    if (!ws.n_cus) HIP_TRY(hipDeviceGetAttribute(&ws.n_cus, hipDeviceAttributeMultiprocessorCount, ws.device));
    if (!ws.w.counters) {
        HIP_TRY(hipMalloc(&ws.w.counters, crt::kCtrWords * sizeof(uint32_t)));
        ws.w.counters64 = reinterpret_cast<unsigned long long*>(ws.w.counters);
    }
    // (the log is cut into kLogSegments segments, a wavefront appends to the one its block index selects: the bound for a tiny
    // scene — every wavefront, every tile — holds per segment)
    const size_t want_log = std::min<size_t>(std::min<size_t>(std::max<size_t>(std::max<size_t>(tiles * 16, waves * 32), (size_t)1 << 16),
                                                              (waves * tiles + 256) * crt::kLogSegments), (size_t)1 << 27);
    if (want_log > ws.log_entries) {
        if (ws.w.log) (void)hipFree(ws.w.log);
        ws.w.log = nullptr;
        ws.log_entries = 0;
        HIP_TRY(hipMalloc(&ws.w.log, want_log * sizeof(crt::BinRecord)));
        ws.log_entries = want_log;
    }
    ws.w.log_cap = (uint32_t)ws.log_entries;
    if (want_pool > ws.pool_entries) {
        if (ws.w.pool) (void)hipFree(ws.w.pool);
        ws.w.pool = nullptr;
        ws.pool_entries = 0;
        HIP_TRY(hipMalloc(&ws.w.pool, want_pool * sizeof(uint32_t)));
        ws.pool_entries = want_pool;
    }
    if (tiles > ws.tiles) {
        if (ws.w.tile_desc) (void)hipFree(ws.w.tile_desc);
        if (ws.w.queue) (void)hipFree(ws.w.queue);
        ws.w.tile_desc = nullptr;
        ws.w.queue = nullptr;
        if (ws.w.tile_count) (void)hipFree(ws.w.tile_count);
        ws.w.tile_count = nullptr;
        ws.tiles = 0;
        HIP_TRY(hipMalloc(&ws.w.tile_desc, tiles * sizeof(uint2)));
        HIP_TRY(hipMalloc(&ws.w.queue, crt::kClasses * tiles * sizeof(uint32_t)));
        HIP_TRY(hipMalloc(&ws.w.tile_count, tiles * sizeof(uint32_t)));
        ws.tiles = tiles;
    }
    ws.w.pool_cap = (uint32_t)std::min<size_t>(ws.pool_entries, 0xfffffff0u);
    ws.w.tiles_x = tiles_x;
    ws.w.tiles_y = tiles_y;
    ws.w.tiles_total = (int)tiles;
    return CRT_OK;
}

// Slacks of the tiled classification, from the camera alone.
// Position slack: the reference evaluates d0 = llc + u*H + v*V - o in float; every component is
// within delta0 = 2^-21 * mag of the exact affine model (4 roundings of intermediates <= mag =
// |llc|+|H|+|V|+|o|, with a factor 2 to spare).  With B = [H V c], c = model(0.5, 0.5), and
// delta = B (alpha, beta, gamma)^T:
//   model(u,v) + delta = (1+gamma) * model(u + du', v + dv'),  du' = (alpha - gamma (u - 0.5)) / (1 + gamma)
// so the float direction is a positive multiple of the model direction of a point at most
// (du, dv) away — the binning widens every rectangle by that much.
// Value slack K (times the 1-norm product M of a form): normalisation and float evaluation of
// the dot/cross chains, <= 2^-21.1 |d0| M, budgeted 2^-19 |d0|; fp32 rounding of the affine
// coefficients and of the classification FMAs, <= 2^-22 mag0 M, budgeted 2^-21 mag0
// (mag0 = |llc-o|_1 + |H|_1 + |V|_1 >= |A|/M, |B|/M, |C|/M).
// false: ill-conditioned camera (the caller then uses the brute-force kernel).
bool camera_slacks(const crt_camera* cam, float* du, float* dv, float* K) {
    double H[3], V[3], c[3], p0[3], mag = 0.0, mag0 = 0.0;
    for (int k = 0; k < 3; ++k) {
        H[k] = cam->horizontal[k];
        V[k] = cam->vertical[k];
        p0[k] = (double)cam->lower_left_corner[k] - (double)cam->origin[k];
        c[k] = p0[k] + 0.5 * H[k] + 0.5 * V[k];
        mag = std::max(mag, std::fabs((double)cam->lower_left_corner[k]) + std::fabs(H[k]) + std::fabs(V[k]) +
                                std::fabs((double)cam->origin[k]));
        mag0 += std::fabs(p0[k]) + std::fabs(H[k]) + std::fabs(V[k]);
    }
    const double delta0 = mag * 4.76837158203125e-7;   // 2^-21
    auto cross = [](const double* a, const double* b, double* o) {
        o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
    };
    double r0[3], r1[3], r2[3];   // rows of B^-1 times det
    cross(V, c, r0);
    cross(c, H, r1);
    cross(H, V, r2);
    const double det = H[0] * r0[0] + H[1] * r0[1] + H[2] * r0[2];
    if (!std::isfinite(det) || det == 0.0) return false;
    auto norm1 = [&](const double* r) { return (std::fabs(r[0]) + std::fabs(r[1]) + std::fabs(r[2])) / std::fabs(det); };
    const double a = norm1(r0) * delta0, b = norm1(r1) * delta0, g = norm1(r2) * delta0;
    if (!std::isfinite(a) || !std::isfinite(b) || !std::isfinite(g) || g > 0.25) return false;
    const double su = (a + 0.5 * g) / (1.0 - g) * 1.01, sv = (b + 0.5 * g) / (1.0 - g) * 1.01;
    if (!(su < 1.0) || !(sv < 1.0)) return false;      // a slack of the whole screen: nothing to gain
    // |d0_float| over the screen: the model is affine, its norm is largest at a corner
    double len_hi = 0.0;
    for (int corner = 0; corner < 4; ++corner) {
        double n2 = 0.0;
        for (int k = 0; k < 3; ++k) {
            const double d = p0[k] + ((corner & 1) ? H[k] : 0.0) + ((corner & 2) ? V[k] : 0.0);
            n2 += d * d;
        }
        len_hi = std::max(len_hi, std::sqrt(n2));
    }
    len_hi = (len_hi + 1.7320508075688772 * delta0) * 1.0001;
    const double k_slack = (1.9073486328125e-6 * len_hi + 4.76837158203125e-7 * mag0 * 1.0001) * 1.001;   // 2^-19, 2^-21
    if (!std::isfinite(k_slack) || k_slack > 1.0e30) return false;
    *du = (float)su * 1.000001f;
    *dv = (float)sv * 1.000001f;
    *K = (float)k_slack * 1.000001f;
    return true;
}

bool have_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return n > 0;
}

// ---------------------------------------------------------------- XORWOW jump matrices
// J_k = T^(2^(67+k)) for k = 0..63, T = GF(2) transition matrix of the xorshift part of
// cuRAND's XORWOW (row-vector convention: row b = image of state bit b).
struct Mat {
    uint32_t row[crt::kXwBits][crt::kXwWords];
};

void mat_vec(const uint32_t* vec, const Mat& m, uint32_t* out) {
    uint32_t r[crt::kXwWords] = {0, 0, 0, 0, 0};
    for (int w = 0; w < crt::kXwWords; ++w) {
        uint32_t word = vec[w];
        while (word) {
            int b = __builtin_ctz(word);
            word &= word - 1;
            const uint32_t* row = m.row[w * 32 + b];
            for (int k = 0; k < crt::kXwWords; ++k) r[k] ^= row[k];
        }
    }
    for (int k = 0; k < crt::kXwWords; ++k) out[k] = r[k];
}

void mat_square(Mat& m) {
    Mat* t = new Mat;
    for (int r = 0; r < crt::kXwBits; ++r) mat_vec(m.row[r], m, t->row[r]);
    m = *t;
    delete t;
}

const std::vector<uint32_t>& jump_table() {
    static std::vector<uint32_t> table;
    static std::once_flag once;
    std::call_once(once, [] {
        Mat* m = new Mat;
        for (int b = 0; b < crt::kXwBits; ++b) {
            uint32_t v[crt::kXwWords] = {0, 0, 0, 0, 0};
            v[b / 32] = 1u << (b % 32);
            uint32_t t = v[0] ^ (v[0] >> 2);
            v[0] = v[1]; v[1] = v[2]; v[2] = v[3]; v[3] = v[4];
            v[4] = (v[4] ^ (v[4] << 4)) ^ (t ^ (t << 1));
            for (int k = 0; k < crt::kXwWords; ++k) m->row[b][k] = v[k];
        }
        for (int i = 0; i < 67; ++i) mat_square(*m);
        table.resize((size_t)crt::kXwJumps * crt::kXwBits * crt::kXwWords);
        for (int k = 0; k < crt::kXwJumps; ++k) {
            memcpy(&table[(size_t)k * crt::kXwBits * crt::kXwWords], m->row, sizeof(Mat));
            if (k + 1 < crt::kXwJumps) mat_square(*m);
        }
        delete m;
    });
    return table;
}

void rotation_zero_matrix(float m[9]) {   // rotate(., vec3(0)) as a matrix: rows of vec3.h:200-213 with sin/cos of 0
    const float radiansX = (float)((M_PI / 180.) * 0.0f), sin_X = sinf(radiansX), cos_X = cosf(radiansX);
    const float radiansY = (float)((M_PI / 180.) * 0.0f), sin_Y = sinf(radiansY), cos_Y = cosf(radiansY);
    const float radiansZ = (float)(-(M_PI / 180.) * 0.0f), sin_Z = sinf(radiansZ), cos_Z = cosf(radiansZ);
    m[0] = cos_Y * cos_Z; m[1] = -cos_Y * sin_Z; m[2] = sin_Y;
    m[3] = sin_X * sin_Y * cos_Z + cos_X * sin_Z; m[4] = -sin_X * sin_Y * sin_Z + cos_X * cos_Z; m[5] = -sin_X * cos_Y;
    m[6] = -cos_X * sin_Y * cos_Z + sin_X * sin_Z; m[7] = cos_X * sin_Y * sin_Z + sin_X * cos_Z; m[8] = cos_X * cos_Y;
}

}  // namespace crt_host

using namespace crt_host;

int upload_bvh_materials(crt_scene* s) {
    if (!s->bvh) return CRT_OK;
    crt_bvh* t = s->bvh;
    if (t->d_material_slots) { (void)hipFree(t->d_material_slots); t->d_material_slots = nullptr; }
    if (s->h_material_of.empty()) return CRT_OK;
    std::vector<int> slots((size_t)s->n_tris);
    for (int k = 0; k < s->n_tris; ++k) slots[k] = s->h_material_of[t->slot_tri[k]];
    HIP_TRY(hipMalloc(&t->d_material_slots, slots.size() * sizeof(int)));
    HIP_TRY(hipMemcpy(t->d_material_slots, slots.data(), slots.size() * sizeof(int), hipMemcpyHostToDevice));
    return CRT_OK;
}

extern "C" {

int crt_abi_version(void) { return CRT_ABI_VERSION; }

const char* crt_last_error(void) { return g_err.c_str(); }

int crt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

void crt_camera_make(crt_camera* out, const float lookfrom_[3], const float lookat_[3],
                     const float vup_[3], float vfov, float aspect, float aperture,
                     float focus_dist) {
    // Camera::Camera, core/camera.h:18-38
    H3 lookfrom = h3(lookfrom_), lookat = h3(lookat_), vup = h3(vup_);
    out->lens_radius = aperture / 2.0f;
    float theta = (float)(vfov * M_PI / 180.0f);   // double product/quotient, one rounding
    float half_height = tanf(theta / 2.0f);
    float half_width = half_height * aspect;
    H3 origin = lookfrom;
    H3 z = hunit(hsub(lookfrom, lookat));
    H3 x = hunit(hcross(vup, z));
    H3 y = hcross(z, x);
    H3 llc = hsub(hsub(hsub(origin, hmul(half_width * focus_dist, x)), hmul(half_height * focus_dist, y)),
                  hmul(focus_dist, z));
    H3 hor = hmul(2.0f * half_width * focus_dist, x);
    H3 ver = hmul(2.0f * half_height * focus_dist, y);
    auto put = [](float* d, H3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; };
    put(out->lower_left_corner, llc);
    put(out->horizontal, hor);
    put(out->vertical, ver);
    put(out->origin, origin);
    put(out->x, x);
    put(out->y, y);
    put(out->z, z);
}

crt_scene* crt_scene_from_mesh(const float* points, int n_points, const float* idx3f, int n_tris,
                               const float* face_normals, int index_base, const float albedo[3]) {
    if (!points || !idx3f || !albedo || n_points <= 0 || n_tris < 0) {
        fail(CRT_ERR_INVALID, "crt_scene_from_mesh: null pointer or non-positive size");
        return nullptr;
    }
    if (!have_device()) {
        fail(CRT_ERR_NO_DEVICE, "crt_scene_from_mesh: no HIP device");
        return nullptr;
    }
    float max_abs = 0.0f;
    bool finite = true;
    std::vector<float> base((size_t)std::max(n_tris, 1) * crt::kBaseFloats);
    std::vector<float> shade((size_t)std::max(n_tris, 1) * 3);
    std::vector<int> idx0((size_t)std::max(n_tris, 1) * 3);
    for (int i = 0; i < n_tris; ++i) {
        int id[3];
        for (int k = 0; k < 3; ++k) {
            id[k] = (int)idx3f[3 * (size_t)i + k] - index_base;   // int(idx[k]), createScene.h:183
            if (id[k] < 0 || id[k] >= n_points) {
                fail(CRT_ERR_INVALID, "crt_scene_from_mesh: triangle %d references vertex %d of %d", i,
                     id[k], n_points);
                return nullptr;
            }
        }
        for (int k = 0; k < 3; ++k) idx0[3 * (size_t)i + k] = id[k];
        float* b = &base[(size_t)i * crt::kBaseFloats];
        for (int k = 0; k < 3; ++k)   // v = {points[idx[2]], points[idx[1]], points[idx[0]]}
            memcpy(b + 3 * k, points + 3 * (size_t)id[2 - k], 3 * sizeof(float));
        H3 n;
        if (face_normals) {           // 6-argument Triangle ctor, shapes/triangle.h:21-29
            n = h3(face_normals + 3 * (size_t)i);
        } else {                      // 5-argument ctor, shapes/triangle.h:14-17
            H3 e1 = hsub(h3(b + 3), h3(b));
            H3 e2 = hsub(h3(b + 6), h3(b));
            n = hunit(hcross(e1, e2));
        }
        b[9] = n.x; b[10] = n.y; b[11] = n.z;
        for (int k = 0; k < 9; ++k) {
            finite = finite && std::isfinite(b[k]);
            max_abs = std::max(max_abs, fabsf(b[k]));
        }
        // rec.normal = rotate(rec.normal, transform->rotation) at the triangle and again
        // at the world list (hitable/hitable.h:36), both with rotation 0.
        H3 sn = hrotate(hrotate(n, H3{0, 0, 0}), H3{0, 0, 0});
        shade[3 * (size_t)i] = sn.x; shade[3 * (size_t)i + 1] = sn.y; shade[3 * (size_t)i + 2] = sn.z;
    }
    crt_scene* s = new (std::nothrow) crt_scene;
    if (!s) { fail(CRT_ERR_NOMEM, "crt_scene_from_mesh: out of memory"); return nullptr; }
    s->n_tris = n_tris;
    s->n_pad = (std::max(n_tris, 1) + 63) / 64 * 64;
    s->tiled_safe = finite && max_abs <= 274877906944.0f;   // 2^38
    memcpy(s->albedo, albedo, sizeof s->albedo);
    hipError_t e = hipGetDevice(&s->device);
    size_t nalloc = (size_t)std::max(n_tris, 1);
    if (e == hipSuccess) e = hipMalloc(&s->d_base, nalloc * crt::kBaseFloats * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&s->d_shade_n, nalloc * 3 * sizeof(float));
    s->n_points = n_points;
    s->h_base = base;
    s->h_idx = idx0;
    crt_material lam;
    lam.kind = CRT_LAMBERTIAN;
    memcpy(lam.albedo, albedo, sizeof lam.albedo);
    lam.param = 0.0f;
    s->h_materials.assign(1, lam);   // new Lambertian(new ConstantTexture(albedo)), createScene.h:180
    if (e == hipSuccess) e = hipMalloc(&s->d_materials, sizeof(crt_material));
    if (e == hipSuccess) e = hipMemcpy(s->d_materials, &lam, sizeof lam, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&s->d_idx, nalloc * 3 * sizeof(int));
    if (e == hipSuccess) e = hipMemcpy(s->d_idx, idx0.data(), nalloc * 3 * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(s->d_base, base.data(), nalloc * crt::kBaseFloats * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(s->d_shade_n, shade.data(), nalloc * 3 * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        fail(CRT_ERR_HIP, "crt_scene_from_mesh: %s", hipGetErrorString(e));
        crt_scene_free(s);
        return nullptr;
    }
    return s;
}

int crt_parse_obj(const char* path, float* points, float* idx3f, int* n_points, int* n_tris) {
    // parseObjByName (Loader/obj_loader.h:20-111) reads `v x y z` and `f a b c` with `ss >> float`
    // and nothing else; files it can read give the same arrays here.  Beyond it (SURVEY §8f rank 4,
    // a deliberate extension — the reference's extraction stops at the first '/'): face tokens
    // `a/b`, `a//c`, `a/b/c` (the leading vertex index is used), negative (relative) indices, and
    // polygons with more than three vertices (fan: (v0, vk, vk+1)), which is what makes
    // CudaTest/objects/cbox_smallbox.obj and Meshlab/Blender exports loadable.
    if (!path || !n_points || !n_tris) return fail(CRT_ERR_INVALID, "crt_parse_obj: null argument");
    FILE* f = fopen(path, "r");
    if (!f) return fail(CRT_ERR_IO, "crt_parse_obj: cannot open %s", path);
    int np = 0, nt = 0;
    char line[4096];
    std::vector<float> face;
    while (fgets(line, sizeof line, f)) {
        const char* p = line;
        while (*p == ' ' || *p == '\t') ++p;
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
            float v[3] = {0, 0, 0};
            p += 1;
            for (int k = 0; k < 3; ++k) {   // `ss >> point[0] >> point[1] >> point[2]`, obj_loader.h:49
                char* end = nullptr;
                v[k] = strtof(p, &end);
                if (end == p) { v[k] = 0.0f; break; }
                p = end;
            }
            if (points) memcpy(points + 3 * (size_t)np, v, sizeof v);
            ++np;
        } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            face.clear();
            p += 1;
            for (;;) {
                while (*p == ' ' || *p == '\t') ++p;
                char* end = nullptr;
                float a = strtof(p, &end);
                if (end == p) break;
                if (a < 0.0f) a = (float)np + a + 1.0f;   // relative index: -1 is the latest vertex
                face.push_back(a);                        // kept as floats, 1-based (obj_loader.h:62-64)
                p = end;
                while (*p && *p != ' ' && *p != '\t' && *p != '\n' && *p != '\r') ++p;   // skip "/vt/vn"
            }
            while (face.size() < 3) face.push_back(0.0f);   // a failed extraction leaves 0 (C++11 operator>>)
            for (size_t k = 1; k + 1 < face.size(); ++k) {
                if (idx3f) {
                    float* o = idx3f + 3 * (size_t)nt;
                    o[0] = face[0]; o[1] = face[k]; o[2] = face[k + 1];
                }
                ++nt;
            }
        }
    }
    fclose(f);
    *n_points = np;
    *n_tris = nt;
    if (points && np > 0) {
        // computeMean + centering, Loader/obj_loader.h:120-133 (fp32 running sum, true divide)
        float mx = 0, my = 0, mz = 0;
        for (int i = 0; i < np; ++i) { mx += points[3 * (size_t)i]; my += points[3 * (size_t)i + 1]; mz += points[3 * (size_t)i + 2]; }
        float fn = (float)np;
        mx = mx / fn; my = my / fn; mz = mz / fn;
        for (int i = 0; i < np; ++i) { points[3 * (size_t)i] -= mx; points[3 * (size_t)i + 1] -= my; points[3 * (size_t)i + 2] -= mz; }
    }
    return CRT_OK;
}

crt_scene* crt_scene_from_obj(const char* path, float scale, const float albedo[3]) {
    int np = 0, nt = 0;
    if (crt_parse_obj(path, nullptr, nullptr, &np, &nt) != CRT_OK) return nullptr;
    if (np <= 0) { fail(CRT_ERR_IO, "crt_scene_from_obj: %s has no vertices", path); return nullptr; }
    std::vector<float> pts((size_t)np * 3), idx((size_t)std::max(nt, 1) * 3);
    if (crt_parse_obj(path, pts.data(), idx.data(), &np, &nt) != CRT_OK) return nullptr;
    for (size_t i = 0; i < pts.size(); ++i) pts[i] *= scale;   // points[i] *= 100.0, obj_loader.h:164
    return crt_scene_from_mesh(pts.data(), np, idx.data(), nt, nullptr, 1, albedo);
}

int crt_scene_update_pose(crt_scene* s, const float* new_points, int n_points, void* stream_) {
    // updateFBXObj's device half (createScene.h:99-110,126-140): H2D copy of the new point
    // positions, then every triangle re-reads its vertices; normals are kept.
    if (!s || !new_points) return fail(CRT_ERR_INVALID, "crt_scene_update_pose: null argument");
    if (n_points != s->n_points)
        return fail(CRT_ERR_INVALID, "crt_scene_update_pose: %d points, the scene was built from %d", n_points, s->n_points);
    if (!have_device()) return fail(CRT_ERR_NO_DEVICE, "crt_scene_update_pose: no HIP device");
    hipStream_t stream = (hipStream_t)stream_;
    bool finite = true, tiny = false;
    float max_abs = 0.0f;
    for (size_t k = 0; k < (size_t)n_points * 3; ++k) {
        finite = finite && std::isfinite(new_points[k]);
        max_abs = std::max(max_abs, fabsf(new_points[k]));
        tiny = tiny || (new_points[k] != 0.0f && fabsf(new_points[k]) < 1.3877787807814457e-17f);   // 2^-56: box_hit_fast's range (crt_bvh.cpp)
    }
    int device = 0;
    HIP_TRY(hipGetDevice(&device));
    if (device != s->device)
        return fail(CRT_ERR_INVALID, "crt_scene_update_pose: current device is %d, the scene lives on device %d", device, s->device);
    const size_t bytes = (size_t)n_points * 3 * sizeof(float);
    if (!s->d_points) HIP_TRY(hipMalloc(&s->d_points, bytes));
    if (!s->h_points) {
        HIP_TRY(hipHostMalloc(&s->h_points, bytes, hipHostMallocDefault));
        HIP_TRY(hipEventCreateWithFlags(&s->pose_copied, hipEventDisableTiming));
    } else {
        HIP_TRY(hipEventSynchronize(s->pose_copied));   // the previous pose has left the staging buffer
    }
    memcpy(s->h_points, new_points, bytes);
    HIP_TRY(hipMemcpyAsync(s->d_points, s->h_points, bytes, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipEventRecord(s->pose_copied, stream));
    HIP_TRY(crt::launch_update_pose(s->d_points, s->d_idx, s->n_tris, s->d_base, stream));
    if (s->bvh)   // the BVH world keeps its triangles in leaf order
        HIP_TRY(crt::launch_update_pose(s->d_points, s->bvh->d_idx_slots, s->n_tris, s->bvh->d_base_slots, stream));
    for (int i = 0; i < s->n_tris; ++i)   // host mirror (a later crt_scene_build_bvh sees the current pose)
        for (int k = 0; k < 3; ++k)
            memcpy(&s->h_base[(size_t)i * crt::kBaseFloats + 3 * k], new_points + 3 * (size_t)s->h_idx[3 * (size_t)i + (2 - k)], 3 * sizeof(float));
    s->tiled_safe = finite && max_abs <= 274877906944.0f;   // 2^38, as at creation (here over all points)
    if (s->bvh) s->bvh->coords_not_tiny = !tiny;            // as at the build (there over the triangles' vertices)
    return CRT_OK;
}

int crt_scene_triangle_count(const crt_scene* s) { return s ? s->n_tris : CRT_ERR_INVALID; }

void crt_scene_free(crt_scene* s) {
    if (!s) return;
    if (s->d_base) (void)hipFree(s->d_base);
    if (s->d_shade_n) (void)hipFree(s->d_shade_n);
    if (s->d_idx) (void)hipFree(s->d_idx);
    if (s->d_points) (void)hipFree(s->d_points);
    if (s->h_points) (void)hipHostFree(s->h_points);
    if (s->pose_copied) (void)hipEventDestroy(s->pose_copied);
    if (s->d_materials) (void)hipFree(s->d_materials);
    if (s->d_material_of) (void)hipFree(s->d_material_of);
    delete s->bvh;
    delete s;
}

namespace {
// rows of the strips {row_begin/strip_rows + k*strip_step} that lie inside the image
int strip_row_count(int ny, int row_begin, int strip_rows, int strip_step) {
    int rows = 0;
    for (long long y0 = row_begin; y0 < ny; y0 += (long long)strip_rows * strip_step)
        rows += (int)std::min<long long>(strip_rows, ny - y0);
    return rows;
}
}  // namespace

crt_rng* crt_rng_create_strips(int nx, int ny, int first_strip, int strip_rows, int strip_step, uint64_t seed) {
    if (nx <= 0 || ny <= 0 || first_strip < 0 || strip_rows <= 0 || strip_step <= 0 ||
        (strip_step > 1 && strip_rows % 32 != 0) || (long long)first_strip * strip_rows >= ny) {
        fail(CRT_ERR_INVALID, "crt_rng_create_strips: bad strip layout (strip_rows must be a multiple of 32)");
        return nullptr;
    }
    const int row_begin = first_strip * strip_rows;
    const int rows = strip_row_count(ny, row_begin, strip_rows, strip_step);
    if (!have_device()) { fail(CRT_ERR_NO_DEVICE, "crt_rng_create_strips: no HIP device"); return nullptr; }
    crt_rng* r = new (std::nothrow) crt_rng;
    if (!r) { fail(CRT_ERR_NOMEM, "crt_rng_create_strips: out of memory"); return nullptr; }
    r->first = (long long)row_begin * nx;
    r->count = (long long)rows * nx;
    r->nx = nx;
    r->strip_rows = strip_rows;
    r->strip_step = strip_step;
    const std::vector<uint32_t>& jt = jump_table();
    hipError_t e = hipGetDevice(&r->device);
    if (e == hipSuccess) e = hipMalloc(&r->d_state, (size_t)r->count * 6 * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&r->d_jump, jt.size() * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemcpy(r->d_jump, jt.data(), jt.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        fail(CRT_ERR_HIP, "crt_rng_create_strips: %s", hipGetErrorString(e));
        crt_rng_free(r);
        return nullptr;
    }
    if (crt_rng_reset(r, seed) != CRT_OK) { crt_rng_free(r); return nullptr; }
    return r;
}

crt_rng* crt_rng_create(int64_t first_pixel, int64_t count, uint64_t seed) {
    if (first_pixel < 0 || count <= 0) { fail(CRT_ERR_INVALID, "crt_rng_create: bad pixel range"); return nullptr; }
    if (!have_device()) { fail(CRT_ERR_NO_DEVICE, "crt_rng_create: no HIP device"); return nullptr; }
    crt_rng* r = new (std::nothrow) crt_rng;
    if (!r) { fail(CRT_ERR_NOMEM, "crt_rng_create: out of memory"); return nullptr; }
    r->first = first_pixel;
    r->count = count;
    r->nx = 1;   // contiguous range: element i is global pixel first_pixel + i
    const std::vector<uint32_t>& jt = jump_table();
    hipError_t e = hipGetDevice(&r->device);
    if (e == hipSuccess) e = hipMalloc(&r->d_state, (size_t)count * 6 * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&r->d_jump, jt.size() * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemcpy(r->d_jump, jt.data(), jt.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        fail(CRT_ERR_HIP, "crt_rng_create: %s", hipGetErrorString(e));
        crt_rng_free(r);
        return nullptr;
    }
    if (crt_rng_reset(r, seed) != CRT_OK) { crt_rng_free(r); return nullptr; }
    return r;
}

int crt_rng_reset(crt_rng* r, uint64_t seed) {
    if (!r) return fail(CRT_ERR_INVALID, "crt_rng_reset: null handle");
    HIP_TRY(crt::launch_rng_init(r->d_state, r->first, r->count, r->nx, r->strip_rows, r->strip_step, seed,
                                 r->d_jump, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    return CRT_OK;
}

int crt_rng_export(const crt_rng* r, uint32_t* out6) {
    if (!r || !out6) return fail(CRT_ERR_INVALID, "crt_rng_export: null argument");
    std::vector<uint32_t> soa((size_t)r->count * 6);
    HIP_TRY(hipMemcpy(soa.data(), r->d_state, soa.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (long long i = 0; i < r->count; ++i)
        for (int k = 0; k < 6; ++k) out6[(size_t)i * 6 + k] = soa[(size_t)k * r->count + i];
    return CRT_OK;
}

void crt_rng_free(crt_rng* r) {
    if (!r) return;
    if (r->d_state) (void)hipFree(r->d_state);
    if (r->d_jump) (void)hipFree(r->d_jump);
    delete r;
}

int crt_set_kernel_variant(int variant) {
    int prev = g_variant;
    g_variant = variant;
    return prev;
}

int crt_set_kernel_timing(int enable) {
    const int prev = g_timing ? 1 : 0;
    g_timing = enable != 0;
    return prev;
}

namespace {
// world: 0 = the triangle list (renderListAnimation, render.h:179), 1 = the BVH over it (renderBVHAnimation, render.h:222)
int render_impl(const crt_scene* scene, const crt_camera* camera, crt_rng* rng, int nx, int ny, int samples,
                int row_begin, int row_count, int strip_rows, int strip_step, float* d_rgb, void* stream_, int world = 0,
                int shader = 0, int max_depth = 0);
}

int crt_render_rows(const crt_scene* scene, const crt_camera* camera, crt_rng* rng, int nx, int ny,
                    int samples, int max_depth, int frame_index, int row_begin, int row_count,
                    float* d_rgb, void* stream_) {
    (void)max_depth;
    (void)frame_index;
    if (nx <= 0 || ny <= 0 || samples < 0 || row_begin < 0 || row_count <= 0 || row_begin + row_count > ny)
        return fail(CRT_ERR_INVALID, "crt_render_rows: bad image/row range (nx=%d ny=%d rows %d+%d)", nx, ny, row_begin, row_count);
    return render_impl(scene, camera, rng, nx, ny, samples, row_begin, row_count, 32, 1, d_rgb, stream_);
}

int crt_render_strips(const crt_scene* scene, const crt_camera* camera, crt_rng* rng, int nx, int ny,
                      int samples, int max_depth, int frame_index, int first_strip, int strip_rows,
                      int strip_step, float* d_rgb, void* stream_) {
    (void)max_depth;
    (void)frame_index;
    if (nx <= 0 || ny <= 0 || samples < 0 || first_strip < 0 || strip_rows <= 0 || strip_step <= 0 ||
        (strip_step > 1 && strip_rows % 32 != 0) || (long long)first_strip * strip_rows >= ny)
        return fail(CRT_ERR_INVALID, "crt_render_strips: bad strip layout (first %d, rows %d, step %d, ny %d)",
                    first_strip, strip_rows, strip_step, ny);
    const int row_begin = first_strip * strip_rows;
    return render_impl(scene, camera, rng, nx, ny, samples, row_begin,
                       strip_row_count(ny, row_begin, strip_rows, strip_step), strip_rows, strip_step, d_rgb, stream_);
}

int crt_render_rows_bvh(const crt_scene* scene, const crt_camera* camera, crt_rng* rng, int nx, int ny,
                        int samples, int max_depth, int frame_index, int row_begin, int row_count,
                        float* d_rgb, void* stream_) {
    (void)max_depth;
    (void)frame_index;
    if (nx <= 0 || ny <= 0 || samples < 0 || row_begin < 0 || row_count <= 0 || row_begin + row_count > ny)
        return fail(CRT_ERR_INVALID, "crt_render_rows_bvh: bad image/row range (nx=%d ny=%d rows %d+%d)", nx, ny, row_begin, row_count);
    return render_impl(scene, camera, rng, nx, ny, samples, row_begin, row_count, 32, 1, d_rgb, stream_, 1);
}

int crt_render_strips_bvh(const crt_scene* scene, const crt_camera* camera, crt_rng* rng, int nx, int ny,
                          int samples, int max_depth, int frame_index, int first_strip, int strip_rows,
                          int strip_step, float* d_rgb, void* stream_) {
    (void)max_depth;
    (void)frame_index;
    if (nx <= 0 || ny <= 0 || samples < 0 || first_strip < 0 || strip_rows <= 0 || strip_step <= 0 ||
        (strip_step > 1 && strip_rows % 32 != 0) || (long long)first_strip * strip_rows >= ny)
        return fail(CRT_ERR_INVALID, "crt_render_strips_bvh: bad strip layout (first %d, rows %d, step %d, ny %d)",
                    first_strip, strip_rows, strip_step, ny);
    const int row_begin = first_strip * strip_rows;
    return render_impl(scene, camera, rng, nx, ny, samples, row_begin,
                       strip_row_count(ny, row_begin, strip_rows, strip_step), strip_rows, strip_step, d_rgb, stream_, 1);
}

int crt_render_rows_ex(const crt_scene* scene, const crt_camera* camera, crt_rng* rng, int nx, int ny,
                       int samples, int max_depth, int frame_index, int row_begin, int row_count,
                       int world, int shader, float* d_rgb, void* stream_) {
    (void)frame_index;
    if (nx <= 0 || ny <= 0 || samples < 0 || row_begin < 0 || row_count <= 0 || row_begin + row_count > ny)
        return fail(CRT_ERR_INVALID, "crt_render_rows_ex: bad image/row range (nx=%d ny=%d rows %d+%d)", nx, ny, row_begin, row_count);
    if ((world != CRT_WORLD_LIST && world != CRT_WORLD_BVH) || shader < CRT_SHADER_LAMBERT || shader > CRT_SHADER_NORMAL)
        return fail(CRT_ERR_INVALID, "crt_render_rows_ex: world %d / shader %d", world, shader);
    if (shader == CRT_SHADER_SHADE && (max_depth < 0 || max_depth > crt::kMaxShadeDepth))
        return fail(CRT_ERR_UNSUPPORTED, "crt_render_rows_ex: max_depth %d (0..%d)", max_depth, crt::kMaxShadeDepth);
    return render_impl(scene, camera, rng, nx, ny, samples, row_begin, row_count, 32, 1, d_rgb, stream_, world, shader, max_depth);
}

int crt_render_strips_ex(const crt_scene* scene, const crt_camera* camera, crt_rng* rng, int nx, int ny,
                         int samples, int max_depth, int frame_index, int first_strip, int strip_rows, int strip_step,
                         int world, int shader, float* d_rgb, void* stream_) {
    (void)frame_index;
    if (nx <= 0 || ny <= 0 || samples < 0 || first_strip < 0 || strip_rows <= 0 || strip_step <= 0 ||
        (strip_step > 1 && strip_rows % 32 != 0) || (long long)first_strip * strip_rows >= ny)
        return fail(CRT_ERR_INVALID, "crt_render_strips_ex: bad strip layout (first %d, rows %d, step %d, ny %d)",
                    first_strip, strip_rows, strip_step, ny);
    if ((world != CRT_WORLD_LIST && world != CRT_WORLD_BVH) || shader < CRT_SHADER_LAMBERT || shader > CRT_SHADER_NORMAL)
        return fail(CRT_ERR_INVALID, "crt_render_strips_ex: world %d / shader %d", world, shader);
    if (shader == CRT_SHADER_SHADE && (max_depth < 0 || max_depth > crt::kMaxShadeDepth))
        return fail(CRT_ERR_UNSUPPORTED, "crt_render_strips_ex: max_depth %d (0..%d)", max_depth, crt::kMaxShadeDepth);
    const int row_begin = first_strip * strip_rows;
    return render_impl(scene, camera, rng, nx, ny, samples, row_begin, strip_row_count(ny, row_begin, strip_rows, strip_step),
                       strip_rows, strip_step, d_rgb, stream_, world, shader, max_depth);
}

int crt_scene_set_materials(crt_scene* s, const crt_material* table, int n_materials, const int* triangle_material) {
    if (!s || !table || n_materials < 1) return fail(CRT_ERR_INVALID, "crt_scene_set_materials: null or empty table");
    if (n_materials > 65536) return fail(CRT_ERR_UNSUPPORTED, "crt_scene_set_materials: %d materials (the kernels keep 16-bit material indices)", n_materials);
    for (int k = 0; k < n_materials; ++k)
        if (table[k].kind < CRT_LAMBERTIAN || table[k].kind > CRT_DIFFUSE_LIGHT)
            return fail(CRT_ERR_INVALID, "crt_scene_set_materials: material %d has kind %d", k, table[k].kind);
    if (triangle_material)
        for (int i = 0; i < s->n_tris; ++i)
            if (triangle_material[i] < 0 || triangle_material[i] >= n_materials)
                return fail(CRT_ERR_INVALID, "crt_scene_set_materials: triangle %d uses material %d of %d", i, triangle_material[i], n_materials);
    if (!have_device()) return fail(CRT_ERR_NO_DEVICE, "crt_scene_set_materials: no HIP device");
    int device = 0;
    HIP_TRY(hipGetDevice(&device));
    if (device != s->device)
        return fail(CRT_ERR_INVALID, "crt_scene_set_materials: current device is %d, the scene lives on device %d", device, s->device);
    HIP_TRY(hipDeviceSynchronize());   // renders in flight still read the old table
    void* d_tab = nullptr;
    int* d_of = nullptr;
    HIP_TRY(hipMalloc(&d_tab, (size_t)n_materials * sizeof(crt_material)));
    hipError_t e = hipMemcpy(d_tab, table, (size_t)n_materials * sizeof(crt_material), hipMemcpyHostToDevice);
    if (e == hipSuccess && triangle_material && s->n_tris > 0) {
        e = hipMalloc(&d_of, (size_t)s->n_tris * sizeof(int));
        if (e == hipSuccess) e = hipMemcpy(d_of, triangle_material, (size_t)s->n_tris * sizeof(int), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        (void)hipFree(d_tab);
        if (d_of) (void)hipFree(d_of);
        return fail(CRT_ERR_HIP, "crt_scene_set_materials: %s", hipGetErrorString(e));
    }
    if (s->d_materials) (void)hipFree(s->d_materials);
    if (s->d_material_of) (void)hipFree(s->d_material_of);
    s->d_materials = d_tab;
    s->d_material_of = d_of;
    s->h_materials.assign(table, table + n_materials);
    s->h_material_of.clear();
    if (triangle_material) s->h_material_of.assign(triangle_material, triangle_material + s->n_tris);
    // the specialised LambertShade kernels assume ONE Lambertian; its colour is the scene's albedo
    bool one = table[0].kind == CRT_LAMBERTIAN;
    for (int i = 0; one && triangle_material && i < s->n_tris; ++i) one = triangle_material[i] == 0;
    s->single_lambertian = one;
    if (one) memcpy(s->albedo, table[0].albedo, sizeof s->albedo);
    return upload_bvh_materials(s);
}

int crt_strip_rows(int ny, int first_strip, int strip_rows, int strip_step) {
    if (ny <= 0 || first_strip < 0 || strip_rows <= 0 || strip_step <= 0) return CRT_ERR_INVALID;
    return strip_row_count(ny, first_strip * strip_rows, strip_rows, strip_step);
}

namespace {
int render_impl(const crt_scene* scene, const crt_camera* camera, crt_rng* rng, int nx, int ny, int samples,
                int row_begin, int row_count, int strip_rows, int strip_step, float* d_rgb, void* stream_, int world,
                int shader, int max_depth) {
    if (!scene || !camera || !rng || !d_rgb) return fail(CRT_ERR_INVALID, "crt_render_rows: null argument");
    // LambertShade with a material per triangle rides the tiled kernels too (the material only changes what a hit draws
    // and returns, crt_kernels.hip: render_pixel<.., MAT>) when they are eligible; variant 1 keeps the generic kernel (A/B)
    const bool lambert_materials = (shader == CRT_SHADER_LAMBERT && !scene->single_lambertian) || shader == CRT_SHADER_NORMAL;
    bool generic = shader != CRT_SHADER_LAMBERT || !scene->single_lambertian;
    if (world == 1 && !scene->bvh) return fail(CRT_ERR_INVALID, "crt_render_rows_bvh: the scene has no BVH (crt_scene_build_bvh)");
    if ((long long)nx * ny > 0x7fffffffLL)
        return fail(CRT_ERR_INVALID, "crt_render_rows: nx*ny exceeds the reference's int pixel_index (core/render.h:111)");
    if (camera->lens_radius != 0.0f)
        return fail(CRT_ERR_UNSUPPORTED,
                    "crt_render_rows: aperture != 0 is a data race on state[0] in the reference (core/render.h:118)");
    long long first = (long long)row_begin * nx, count = (long long)row_count * nx;
    const bool layout_ok = rng->first == first && rng->count == count &&
                           (strip_step == 1 ? rng->strip_step == 1
                                            : (rng->strip_step == strip_step && rng->strip_rows == strip_rows && rng->nx == nx));
    if (!layout_ok)
        return fail(CRT_ERR_INVALID, "crt_render_rows: rng covers pixels [%lld,+%lld) step %d, the rows need [%lld,+%lld) step %d",
                    rng->first, rng->count, rng->strip_step, first, count, strip_step);
    hipStream_t stream = (hipStream_t)stream_;
    int device = 0;
    HIP_TRY(hipGetDevice(&device));
    if (scene->device != device || rng->device != device)
        return fail(CRT_ERR_INVALID, "crt_render_rows: current device is %d but the scene lives on device %d and the RNG states on device %d",
                    device, scene->device, rng->device);
    Context* cxp = context_for(device, stream);
    if (!cxp) return fail(CRT_ERR_NOMEM, "crt_render_rows: out of memory");
    CtxPin pin(cxp);
    Context& cx = *cxp;
    std::lock_guard<std::mutex> lock(cx.mu);
    g_last_ctx_id = cxp->id;
    {
        int rc = ensure_records(cx, scene->n_tris, scene->n_pad);
        if (rc != CRT_OK) return rc;
    }

    if (!cx.ev_start) {
        HIP_TRY(hipEventCreate(&cx.ev_start));
        HIP_TRY(hipEventCreate(&cx.ev_stop));
        HIP_TRY(hipEventCreate(&cx.ev_mid));
    }
    const bool timing = g_timing;   // three marker packets per frame: a host that never asks for kernel times can switch them off
    if (timing) HIP_TRY(hipEventRecord(cx.ev_start, stream));   // the per-frame prepasses count as binning time

    // public variant numbering (crt_set_kernel_variant): 0 auto, 1 brute force, 2 tiled,
    // 3 tiled with tiny list capacities (exercises the overflow fallbacks)
    bool cam_ok = true;
    for (int k = 0; k < 3; ++k)
        cam_ok = cam_ok && std::isfinite(camera->lower_left_corner[k]) && std::isfinite(camera->horizontal[k]) &&
                 std::isfinite(camera->vertical[k]) && std::isfinite(camera->origin[k]) &&
                 fabsf(camera->origin[k]) <= 274877906944.0f && fabsf(camera->lower_left_corner[k]) <= 274877906944.0f;
    float pos_du = 0.0f, pos_dv = 0.0f, k_slack = 0.0f;
    cam_ok = cam_ok && camera_slacks(camera, &pos_du, &pos_dv, &k_slack);
    const bool tiled_ok = scene->tiled_safe && cam_ok && scene->n_tris > 0;
    crt::MatView mat_view;
    const crt::MatView* materials = nullptr;
    if (lambert_materials && tiled_ok) {
        // list world: any tiled variant with staged records; BVH world: only where the tiled mode below is taken (its
        // conditions, restated: fast arithmetic of the traversal, variant 0 or 2)
        bool ok = world == 0 ? (g_variant == 0 || g_variant == 2 || g_variant == 3 || g_variant == 8 || g_variant == 10) : (g_variant == 0 || g_variant == 2);
        if (world == 1) {
            ok = ok && scene->bvh->coords_not_tiny && scene->n_tris <= (1 << 22);
            for (int k = 0; k < 3; ++k) {
                const float o = camera->origin[k];
                ok = ok && std::isfinite(o) && fabsf(o) <= 274877906944.0f && (o == 0.0f || fabsf(o) >= 1.3877787807814457e-17f);
            }
        }
        if (ok) {
            mat_view.material_of = world == 1 ? scene->bvh->d_material_slots : scene->d_material_of;
            mat_view.materials = static_cast<const crt::MaterialPOD*>(scene->d_materials);
            mat_view.normal_shader = shader == CRT_SHADER_NORMAL ? 1 : 0;
            materials = &mat_view;
            generic = false;
        }
    }
    int kernel_variant = 0;
    if (g_variant == 0 || g_variant == 2) kernel_variant = tiled_ok ? 1 : 0;
    else if (g_variant == 3) kernel_variant = tiled_ok ? 2 : 0;
    else if (g_variant == 4) kernel_variant = tiled_ok ? 3 : 0;   // tiled, scalar-operand walk for light tiles too (A/B)
    else if (g_variant == 8) kernel_variant = tiled_ok ? 4 : 0;   // tiled with a 16-record binning log (overflow fallback)
    else if (g_variant == 10) kernel_variant = tiled_ok ? 5 : 0;  // tiled, shared tiles from 4 entries on, dealt out as row slabs (sub-items)

    crt::RenderParams p;
    memcpy(p.cam.llc, camera->lower_left_corner, 12);
    memcpy(p.cam.horizontal, camera->horizontal, 12);
    memcpy(p.cam.vertical, camera->vertical, 12);
    memcpy(p.cam.origin, camera->origin, 12);
    memcpy(p.albedo, scene->albedo, 12);
    p.nx = nx; p.ny = ny; p.samples = samples;
    p.row_begin = row_begin; p.row_count = row_count;
    p.strip_rows = strip_rows; p.strip_step = strip_step;
    p.n_tris = scene->n_tris;
    p.rng_count = count;
    p.rnx = 1.0f / (float)nx;
    p.rny = 1.0f / (float)ny;
    p.pos_du = pos_du; p.pos_dv = pos_dv; p.K_slack = k_slack;
    {
        float mag = 0.0f;   // the order the kernel summed them in (any order would do: the value feeds a conservative guard only)
        for (int k = 0; k < 3; ++k) mag += fabsf(p.cam.llc[k]) + fabsf(p.cam.horizontal[k]) + fabsf(p.cam.vertical[k]) + fabsf(p.cam.origin[k]);
        p.cam_mag = mag;
        p.cam_thr = mag * 9.5367431640625e-7f;   // 2^-20 mag
        // camera-only terms of per-tile computations (pixel_half_extents, tile_cone), in the kernels' own float expressions
        p.pix_hu = (0.5f * p.rnx) * 1.00001f + 4.0e-7f + p.pos_du;
        p.pix_hv = (0.5f * p.rny) * 1.00001f + 4.0e-7f + p.pos_dv;
        float h2 = 0.0f, w2 = 0.0f;
        for (int k = 0; k < 3; ++k) { h2 += p.cam.horizontal[k] * p.cam.horizontal[k]; w2 += p.cam.vertical[k] * p.cam.vertical[k]; }
        p.cam_hlen = sqrtf(h2);
        p.cam_vlen = sqrtf(w2);
    }

    // BVH world, LambertShade on one Lambertian, where the tiled kernels do not apply (more than 2^22 triangles; a camera outside
    // the binning's range) or variant 1 asks for a per-ray walk: the generic kernel's walk shared by the wavefront is twice as
    // fast as render_bvh_kernel (1.3 against 2.4 ms per 4-spp C3 frame) — which stays for trees without pair records, for the
    // literal arithmetic of unbounded scenes and for the test hooks 5 and 6.
    if (!generic && world == 1 && scene->bvh->d_pairs != nullptr && scene->tiled_safe && scene->bvh->coords_not_tiny &&
        (g_variant == 1 || ((g_variant == 0 || g_variant == 2) && !(tiled_ok && scene->n_tris <= (1 << 22)))))
        generic = true;   // (a camera the fast arithmetic does not cover sends the generic kernel to its literal loops: below)
    if (generic) {
        // any shader, any materials (crt_shade_kernels.hpp); fast arithmetic under the same conditions as below
        bool fast = scene->tiled_safe && (world == 0 || scene->bvh->coords_not_tiny);
        for (int k = 0; k < 3; ++k) {
            const float o = camera->origin[k];
            fast = fast && std::isfinite(o) && fabsf(o) <= 274877906944.0f && (o == 0.0f || fabsf(o) >= 1.3877787807814457e-17f);
            fast = fast && std::isfinite(camera->lower_left_corner[k]) && std::isfinite(camera->horizontal[k]) && std::isfinite(camera->vertical[k]);
        }
        const int force_mode = g_variant == 5 ? 1 : (g_variant == 6 ? 2 : 0);
        float rot[9];
        rotation_zero_matrix(rot);
        const crt_bvh* t = world == 1 ? scene->bvh : nullptr;
        if (!cx.d_pixel_queue) HIP_TRY(hipMalloc(&cx.d_pixel_queue, crt::kGenericQueueWords * sizeof(unsigned)));
        if (!cx.n_cus) HIP_TRY(hipDeviceGetAttribute(&cx.n_cus, hipDeviceAttributeMultiprocessorCount, cx.device));
        HIP_TRY(crt::launch_render_generic(p, t ? t->d_base_slots : scene->d_base, t ? t->d_shade_n_slots : scene->d_shade_n,
                                           t ? t->d_material_slots : scene->d_material_of, scene->d_materials, scene->n_tris,
                                           t ? t->d_nodes : nullptr, t ? t->n_nodes : 0, rot, rng->d_state, d_rgb, world == 1, fast,
                                           shader, max_depth, force_mode, stream,
                                           // the pair records, walked by the wavefront together (variant 9: skip links; 11: one stack
                                           // per lane; 12: the shared walk with a short list)
                                           t ? t->d_pairs : nullptr, t ? t->d_partner : nullptr,
                                           g_variant == 9 ? 0 : (g_variant == 11 ? 1 : (g_variant == 12 ? 3 : 2)),
                                           cx.d_pixel_queue, cx.n_cus, g_d_stats, &cx.cache_generic));
        cx.ev_has_mid = false;
        if (timing) HIP_TRY(hipEventRecord(cx.ev_stop, stream));
        cx.ev_valid = timing;
        return CRT_OK;
    }
    if (world == 1) {
        // fast arithmetic of the traversal: bounded scene and origin, no nonzero coordinate below 2^-56
        // (crt_bvh_kernels.hpp: box_hit_fast); otherwise the literal IEEE loop
        bool fast = scene->tiled_safe && scene->bvh->coords_not_tiny;
        for (int k = 0; k < 3; ++k) {
            const float o = camera->origin[k];
            fast = fast && std::isfinite(o) && fabsf(o) <= 274877906944.0f && (o == 0.0f || fabsf(o) >= 1.3877787807814457e-17f);
            fast = fast && std::isfinite(camera->lower_left_corner[k]) && std::isfinite(camera->horizontal[k]) && std::isfinite(camera->vertical[k]);
        }
        // Primary rays ride the tiled kernels (binning over the slot-ordered triangles, the tree only for every
        // pixel's winner: shade_tiles_kernel<.., BVH>) whenever the list world's tiled variant is eligible and
        // the fast arithmetic holds; variant 1 (and the test hooks 5, 6) select the per-ray tree walk.
        // (variant 10: test hook — shared tiles from 4 entries on, dealt out as row slabs, like the list world's)
        if (fast && tiled_ok && scene->n_tris <= (1 << 22) && (g_variant == 0 || g_variant == 2 || g_variant == 7 || g_variant == 10)) {   // (22 bits of tie order in the keys)
            int rc = ensure_workspace(cx, nx, row_count, scene->n_tris);
            if (rc != CRT_OK) return rc;
            HIP_TRY(crt::launch_frame_prepass(scene->bvh->d_base_slots, scene->n_tris, scene->n_pad, p, cx.d_rec, cx.d_cls, cx.w, stream));
            HIP_TRY(crt::launch_render(p, cx.d_rec, cx.d_cls, scene->n_pad, scene->bvh->d_shade_n_slots, rng->d_state, d_rgb, g_variant == 10 ? 5 : 1, &cx.w,
                                       cx.n_cus, true, g_d_stats, timing ? cx.ev_mid : nullptr, &cx.cache_bvh, true, stream,
                                       &scene->bvh->tiles_view, g_variant == 7, materials));
            cx.ev_has_mid = true;
            if (timing) HIP_TRY(hipEventRecord(cx.ev_stop, stream));
            cx.ev_valid = timing;
            return CRT_OK;
        }
        HIP_TRY(crt::launch_tri_prepass(scene->bvh->d_base_slots, scene->n_tris, camera->origin, cx.d_rec, stream));
        const int force_mode = g_variant == 5 ? 1 : (g_variant == 6 ? 2 : 0);
        HIP_TRY(crt::launch_render_bvh(p, scene->bvh->d_nodes, scene->bvh->n_nodes, scene->bvh->rot, cx.d_rec,
                                       scene->bvh->d_shade_n_slots, rng->d_state, d_rgb, fast, force_mode, stream));
        cx.ev_has_mid = false;
        if (timing) HIP_TRY(hipEventRecord(cx.ev_stop, stream));
        cx.ev_valid = timing;
        return CRT_OK;
    }
    if (kernel_variant != 0) {   // one launch: exact + classification records, scratch zeroed
        int rc = ensure_workspace(cx, nx, row_count, scene->n_tris);
        if (rc != CRT_OK) return rc;
        HIP_TRY(crt::launch_frame_prepass(scene->d_base, scene->n_tris, scene->n_pad, p, cx.d_rec, cx.d_cls, cx.w, stream));
    } else {
        HIP_TRY(crt::launch_tri_prepass(scene->d_base, scene->n_tris, camera->origin, cx.d_rec, stream));
    }
    HIP_TRY(crt::launch_render(p, cx.d_rec, cx.d_cls, scene->n_pad, scene->d_shade_n, rng->d_state,
                               d_rgb, kernel_variant, kernel_variant ? &cx.w : nullptr, cx.n_cus,
                               scene->tiled_safe && cam_ok, g_d_stats, (kernel_variant && timing) ? cx.ev_mid : nullptr, &cx.cache,
                               kernel_variant != 0, stream, nullptr, false, materials));
    cx.ev_has_mid = kernel_variant != 0;
    if (timing) HIP_TRY(hipEventRecord(cx.ev_stop, stream));
    cx.ev_valid = timing;
    return CRT_OK;
}
}  // namespace

int crt_debug_audit_tiles(const crt_scene* scene, const crt_camera* camera, int nx, int ny, int samples,
                          uint64_t seed, uint64_t* out4) {
    // Brute-force audit of the binning on the current device (full image): see audit_tiles_kernel.
    if (!scene || !camera || !out4 || nx <= 0 || ny <= 0 || samples < 0)
        return fail(CRT_ERR_INVALID, "crt_debug_audit_tiles: bad argument");
    if (!have_device()) return fail(CRT_ERR_NO_DEVICE, "crt_debug_audit_tiles: no HIP device");
    float pos_du = 0.0f, pos_dv = 0.0f, k_slack = 0.0f;
    if (!scene->tiled_safe || scene->n_tris <= 0 || !camera_slacks(camera, &pos_du, &pos_dv, &k_slack))
        return fail(CRT_ERR_UNSUPPORTED, "crt_debug_audit_tiles: the scene is not eligible for the tiled kernels");
    crt_rng* rng = crt_rng_create(0, (int64_t)nx * ny, seed);
    if (!rng) return CRT_ERR_HIP;
    crt::RenderParams p;
    memcpy(p.cam.llc, camera->lower_left_corner, 12);
    memcpy(p.cam.horizontal, camera->horizontal, 12);
    memcpy(p.cam.vertical, camera->vertical, 12);
    memcpy(p.cam.origin, camera->origin, 12);
    memcpy(p.albedo, scene->albedo, 12);
    p.nx = nx; p.ny = ny; p.samples = samples;
    p.row_begin = 0; p.row_count = ny;
    p.strip_rows = 32; p.strip_step = 1;
    p.n_tris = scene->n_tris;
    p.rng_count = (long long)nx * ny;
    p.rnx = 1.0f / (float)nx;
    p.rny = 1.0f / (float)ny;
    p.pos_du = pos_du; p.pos_dv = pos_dv; p.K_slack = k_slack;
    {
        float mag = 0.0f;   // the order the kernel summed them in (any order would do: the value feeds a conservative guard only)
        for (int k = 0; k < 3; ++k) mag += fabsf(p.cam.llc[k]) + fabsf(p.cam.horizontal[k]) + fabsf(p.cam.vertical[k]) + fabsf(p.cam.origin[k]);
        p.cam_mag = mag;
        p.cam_thr = mag * 9.5367431640625e-7f;   // 2^-20 mag
        // camera-only terms of per-tile computations (pixel_half_extents, tile_cone), in the kernels' own float expressions
        p.pix_hu = (0.5f * p.rnx) * 1.00001f + 4.0e-7f + p.pos_du;
        p.pix_hv = (0.5f * p.rny) * 1.00001f + 4.0e-7f + p.pos_dv;
        float h2 = 0.0f, w2 = 0.0f;
        for (int k = 0; k < 3; ++k) { h2 += p.cam.horizontal[k] * p.cam.horizontal[k]; w2 += p.cam.vertical[k] * p.cam.vertical[k]; }
        p.cam_hlen = sqrtf(h2);
        p.cam_vlen = sqrtf(w2);
    }
    unsigned long long* d = nullptr;
    int device = 0;
    Context* cxp = (hipGetDevice(&device) == hipSuccess) ? context_for(device, nullptr) : nullptr;
    if (!cxp) { crt_rng_free(rng); return fail(CRT_ERR_HIP, "crt_debug_audit_tiles: no launch context"); }
    CtxPin pin(cxp);
    Context& cx = *cxp;
    std::lock_guard<std::mutex> lock(cx.mu);
    int rc = ensure_workspace(cx, nx, ny, scene->n_tris);
    if (rc == CRT_OK) rc = ensure_records(cx, scene->n_tris, scene->n_pad);
    hipError_t e = (rc == CRT_OK) ? hipMalloc(&d, 4 * sizeof(unsigned long long)) : hipErrorUnknown;
    if (e == hipSuccess) e = hipMemset(d, 0, 4 * sizeof(unsigned long long));
    if (e == hipSuccess) e = crt::launch_tri_prepass(scene->d_base, scene->n_tris, camera->origin, cx.d_rec, nullptr);
    if (e == hipSuccess) e = crt::launch_cls_prepass(cx.d_rec, scene->n_tris, scene->n_pad, p, cx.d_cls, nullptr);
    if (e == hipSuccess) e = crt::launch_audit(p, cx.d_rec, cx.d_cls, scene->n_pad, rng->d_state, &cx.w, d, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    unsigned long long h[4] = {0, 0, 0, 0};
    if (e == hipSuccess) e = hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    if (d) (void)hipFree(d);
    crt_rng_free(rng);
    if (rc != CRT_OK) return rc;
    if (e != hipSuccess) return fail(CRT_ERR_HIP, "crt_debug_audit_tiles: %s", hipGetErrorString(e));
    for (int k = 0; k < 4; ++k) out4[k] = h[k];
    return CRT_OK;
}

int crt_debug_tiled_stats(int enable, uint64_t* out48) {   // out: kStatsWords = 48 words
    // Diagnostics: per-thread device counters filled by the tiled kernels (include/crt.h).  enable=1 allocates/zeroes,
    // enable=2 the same in timeline mode, enable=0 frees; out48 (nullable) receives the current values first.
    if (out48) {
        memset(out48, 0, crt::kStatsWords * sizeof(uint64_t));
        if (g_d_stats) {
            HIP_TRY(hipDeviceSynchronize());
            HIP_TRY(hipMemcpy(out48, g_d_stats, crt::kStatsWords * sizeof(uint64_t), hipMemcpyDeviceToHost));
        }
    }
    if (enable) {
        if (!g_d_stats) HIP_TRY(hipMalloc(&g_d_stats, crt::kStatsTotalWords * sizeof(uint64_t)));
        HIP_TRY(hipMemset(g_d_stats, 0, crt::kStatsTotalWords * sizeof(uint64_t)));
        if (enable == 2) {   // timeline mode: only wave time stamps, the kernel's timing stays what it is
            const uint64_t one = 1ull;
            HIP_TRY(hipMemcpy(g_d_stats + crt::kStatsTimeline, &one, sizeof(one), hipMemcpyHostToDevice));
        }
    } else if (g_d_stats) {
        (void)hipFree(g_d_stats);
        g_d_stats = nullptr;
    }
    return CRT_OK;
}

int crt_debug_wave_timeline(uint64_t* out, int max_waves) {
    // Timeline mode of crt_debug_tiled_stats: three 100 MHz time stamps (start, end of the shared-tile phase, end) and
    // (entries << 32 | tile) of the block's last shared tile per wavefront of the last shade_tiles_kernel launch,
    // 4 * max_waves words; returns the number of slots copied.
    if (!out || (max_waves <= 0 && max_waves != -1 && max_waves != -2)) return fail(CRT_ERR_INVALID, "crt_debug_wave_timeline: bad arguments");
    if (!g_d_stats) return fail(CRT_ERR_INVALID, "crt_debug_wave_timeline: stats are not enabled");
    if (max_waves == -1) {   // the per-sample log of the first blocks instead: 2 * kStatsSampleBlocks * kStatsSampleSlots words
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(out, g_d_stats + crt::kStatsSampleLog, (size_t)2 * crt::kStatsSampleBlocks * crt::kStatsSampleSlots * sizeof(uint64_t),
                          hipMemcpyDeviceToHost));
        return crt::kStatsSampleBlocks * crt::kStatsSampleSlots;
    }
    if (max_waves == -2) {   // per wavefront: its last light tile with geometry (start stamp; tile | entries << 32 | such tiles taken << 48)
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(out, g_d_stats + crt::kStatsLightLog, (size_t)3 * crt::kStatsWaveSlots * sizeof(uint64_t), hipMemcpyDeviceToHost));
        return crt::kStatsWaveSlots;
    }
    const int n = std::min(max_waves, crt::kStatsWaveSlots);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, g_d_stats + crt::kStatsWaveLog, (size_t)n * 4 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return n;
}

int crt_debug_check_division(int mode, int exp_lo, int exp_hi, uint64_t* mismatches);

int crt_debug_check_reciprocal(int exp_lo, int exp_hi, uint64_t* mismatches) {
    return crt_debug_check_division(0, exp_lo, exp_hi, mismatches);
}

int crt_debug_check_division(int mode, int exp_lo, int exp_hi, uint64_t* mismatches) {
    // Exhaustive self-check of the 3-instruction reciprocal used by the kernels on bounded
    // scenes: all floats with biased exponent in [exp_lo, exp_hi], both signs.
    if (!mismatches || exp_lo < 0 || exp_hi > 255 || exp_lo > exp_hi || mode < 0 || mode > 4)
        return fail(CRT_ERR_INVALID, "crt_debug_check_reciprocal: bad range");
    if (!have_device()) return fail(CRT_ERR_NO_DEVICE, "crt_debug_check_reciprocal: no HIP device");
    unsigned long long* d = nullptr;
    HIP_TRY(hipMalloc(&d, sizeof(unsigned long long)));
    hipError_t e = hipMemset(d, 0, sizeof(unsigned long long));
    if (mode == 1) {   // significand chunks: one launch (about half a second) per chunk
        if (exp_hi > 63) { (void)hipFree(d); return fail(CRT_ERR_INVALID, "crt_debug_check_division: mode 1 takes chunks 0..63"); }
        for (int c = exp_lo; c <= exp_hi && e == hipSuccess; ++c) {
            e = crt::launch_check_reciprocal(1, (uint32_t)c, (uint32_t)c, d, nullptr);
            if (e == hipSuccess) e = hipDeviceSynchronize();
        }
    } else {
        if (e == hipSuccess) e = crt::launch_check_reciprocal(mode, (uint32_t)exp_lo, (uint32_t)exp_hi, d, nullptr);
        if (e == hipSuccess) e = hipDeviceSynchronize();
    }
    unsigned long long h = 0;
    if (e == hipSuccess) e = hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(CRT_ERR_HIP, "crt_debug_check_reciprocal: %s", hipGetErrorString(e));
    *mismatches = h;
    return CRT_OK;
}

int crt_debug_vec3(int where, int op, long long n, const float* a, const float* b, const float* s, float* out) {
    if (n < 0 || (n > 0 && (!a || !b || !s || !out))) return fail(CRT_ERR_INVALID, "crt_debug_vec3: null array");
    if (where == 0) {   // the host helpers of crt_internal.hpp (camera constructor, scene pack, BVH build)
        for (long long i = 0; i < n; ++i) {
            const H3 A = h3(a + 3 * i), B = h3(b + 3 * i);
            H3 R{0.0f, 0.0f, 0.0f};
            switch (op) {
                case 1: R = hsub(A, B); break;
                case 3: R = hmul(s[i], A); break;
                case 7: R = hcross(A, B); break;
                case 10: R = hunit(A); break;
                case 14: R = hrotate(A, B); break;
                default: return fail(CRT_ERR_UNSUPPORTED, "crt_debug_vec3: the host layer has no helper for op %d", op);
            }
            out[3 * i] = R.x; out[3 * i + 1] = R.y; out[3 * i + 2] = R.z;
        }
        return CRT_OK;
    }
    if (where != 1) return fail(CRT_ERR_INVALID, "crt_debug_vec3: where must be 0 (host helpers) or 1 (device functions)");
    switch (op) {
        case 1: case 3: case 6: case 7: case 8: case 10: case 11: case 12: case 13: case 100: case 101: case 102: break;
        default: return fail(CRT_ERR_UNSUPPORTED, "crt_debug_vec3: the kernels have no device function for op %d", op);
    }
    if (!have_device()) return fail(CRT_ERR_NO_DEVICE, "crt_debug_vec3: no HIP device");
    if (n == 0) return CRT_OK;
    float* d = nullptr;
    const size_t n3 = (size_t)n * 3;
    HIP_TRY(hipMalloc(&d, (3 * n3 + (size_t)n) * sizeof(float)));
    float *d_a = d, *d_b = d + n3, *d_out = d + 2 * n3, *d_s = d + 3 * n3;
    hipError_t e = hipMemcpy(d_a, a, n3 * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_b, b, n3 * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_s, s, (size_t)n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = crt::launch_vec3_ops(op, n, d_a, d_b, d_s, d_out, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out, d_out, n3 * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(CRT_ERR_HIP, "crt_debug_vec3: %s", hipGetErrorString(e));
    return CRT_OK;
}

int crt_last_kernel_times(float* ms3) {
    // ms3[0] = all render kernels of the last crt_render_rows of this thread, ms3[1] = prepasses +
    // binning, ms3[2] = shading kernel (tiled variant) or the brute-force kernel; HIP events on the
    // launch stream of that call.
    if (!ms3) return fail(CRT_ERR_INVALID, "crt_last_kernel_times: null output");
    Context* cx = last_context_locked_lookup();
    CtxPin pin(cx);
    if (!cx || !cx->ev_valid) return fail(CRT_ERR_INVALID, "crt_last_kernel_times: no timed kernel launched on this thread (crt_set_kernel_timing)");
    std::lock_guard<std::mutex> lock(cx->mu);
    HIP_TRY(hipEventSynchronize(cx->ev_stop));
    HIP_TRY(hipEventElapsedTime(&ms3[0], cx->ev_start, cx->ev_stop));
    ms3[1] = 0.0f;
    ms3[2] = ms3[0];
    if (cx->ev_has_mid) {
        HIP_TRY(hipEventElapsedTime(&ms3[1], cx->ev_start, cx->ev_mid));
        HIP_TRY(hipEventElapsedTime(&ms3[2], cx->ev_mid, cx->ev_stop));
    }
    return CRT_OK;
}

float crt_last_kernel_ms(void) {
    float ms3[3];
    if (crt_last_kernel_times(ms3) != CRT_OK) return -1.0f;
    return ms3[0];
}

}  // extern "C"

namespace {
// What one crt_render call needs besides the scene, kept from call to call per device: the RNG states of the frame, a
// pristine copy of them right after SetCurandState(seed) (restoring it is a device-to-device copy, 0.04 ms at
// 1080p, instead of the 0.36 ms skip-ahead kernel — the reference runs SetCurandState once per program,
// kernel.cu:69-71, and a call of crt_render is defined as "that, then one frame"), and the device colour buffer.
// Round 2 allocated, initialised and freed all three on every call.  Freed by crt_stream_release(NULL).
struct OneCall {
    int device = -1, nx = 0, ny = 0;
    uint64_t seed = 0;
    crt_rng* rng = nullptr;
    uint32_t* d_pristine = nullptr;
    float* d_rgb = nullptr;
};
std::mutex g_onecall_mu;
std::vector<OneCall> g_onecall;

void onecall_free(OneCall& o) {
    crt_rng_free(o.rng);
    if (o.d_pristine) (void)hipFree(o.d_pristine);
    if (o.d_rgb) (void)hipFree(o.d_rgb);
    o = OneCall{};
}
}  // namespace

void crt_host::release_onecall(int device) {
    std::lock_guard<std::mutex> lock(g_onecall_mu);
    for (size_t i = 0; i < g_onecall.size(); ++i)
        if (g_onecall[i].device == device) {
            onecall_free(g_onecall[i]);
            g_onecall.erase(g_onecall.begin() + (long)i);
            return;
        }
}

extern "C" {

int crt_render(int nx, int ny, int samples, int max_depth, int frame_index, const crt_scene* scene,
               const crt_camera* camera, uint64_t seed, float* rgb_out) {
    if (!rgb_out) return fail(CRT_ERR_INVALID, "crt_render: null output");
    if (nx <= 0 || ny <= 0) return fail(CRT_ERR_INVALID, "crt_render: bad image size");
    if (!have_device()) return fail(CRT_ERR_NO_DEVICE, "crt_render: no HIP device");
    int device = 0;
    HIP_TRY(hipGetDevice(&device));
    std::lock_guard<std::mutex> lock(g_onecall_mu);   // one crt_render at a time per process: it owns the default stream's frame
    OneCall* o = nullptr;
    for (OneCall& c : g_onecall)
        if (c.device == device) o = &c;
    if (!o) {
        g_onecall.emplace_back();
        o = &g_onecall.back();
        o->device = device;
    }
    const size_t bytes = (size_t)nx * ny * 3 * sizeof(float), state_bytes = (size_t)nx * ny * 6 * sizeof(uint32_t);
    if (o->nx != nx || o->ny != ny || !o->rng) {
        const int dev = o->device;
        onecall_free(*o);
        o->device = dev;
        o->rng = crt_rng_create(0, (int64_t)nx * ny, seed);   // SetCurandState
        if (!o->rng) return CRT_ERR_HIP;
        hipError_t e = hipMalloc(&o->d_rgb, bytes);
        if (e == hipSuccess) e = hipMalloc(&o->d_pristine, state_bytes);
        if (e == hipSuccess) e = hipMemcpy(o->d_pristine, o->rng->d_state, state_bytes, hipMemcpyDeviceToDevice);
        if (e != hipSuccess) { const int d2 = o->device; onecall_free(*o); o->device = d2; return fail(CRT_ERR_HIP, "crt_render: %s", hipGetErrorString(e)); }
        o->nx = nx; o->ny = ny; o->seed = seed;
    } else if (o->seed != seed) {
        int rc = crt_rng_reset(o->rng, seed);
        if (rc != CRT_OK) return rc;
        HIP_TRY(hipMemcpy(o->d_pristine, o->rng->d_state, state_bytes, hipMemcpyDeviceToDevice));
        o->seed = seed;
    } else {
        HIP_TRY(hipMemcpyAsync(o->rng->d_state, o->d_pristine, state_bytes, hipMemcpyDeviceToDevice, nullptr));
    }
    int rc = crt_render_rows(scene, camera, o->rng, nx, ny, samples, max_depth, frame_index, 0, ny, o->d_rgb, nullptr);
    if (rc == CRT_OK) {
        hipError_t e = hipMemcpy(rgb_out, o->d_rgb, bytes, hipMemcpyDeviceToHost);   // ordered after the kernels of the null stream
        if (e != hipSuccess) rc = fail(CRT_ERR_HIP, "crt_render: %s", hipGetErrorString(e));
    }
    return rc;
}

int crt_stream_release(void* stream) {
    if (!have_device()) return fail(CRT_ERR_NO_DEVICE, "crt_stream_release: no HIP device");
    int device = 0;
    HIP_TRY(hipGetDevice(&device));
    if (!stream) release_onecall(device);
    return release_context(device, (hipStream_t)stream);
}

void crt_to_rgba8(const float* rgb, int nx, int ny, unsigned char* rgba) {
    // WritePng, core/render.h:138-146: row i of the PNG is buffer row ny-1-i;
    // channel = char(255.99 * c) -> double product, truncation, low 8 bits; NaN -> 0.
    for (int i = 0; i < ny; ++i)
        for (int j = 0; j < nx; ++j) {
            size_t src = ((size_t)(ny - 1 - i) * nx + j) * 3, dst = ((size_t)i * nx + j) * 4;
            for (int c = 0; c < 3; ++c) {
                double d = 255.99 * (double)rgb[src + c];
                int iv = (d != d) ? 0 : (int)d;
                rgba[dst + c] = (unsigned char)iv;
            }
            rgba[dst + 3] = 255;
        }
}

}  // extern "C"
