// crt_multi.cpp — the multi-GPU half of the C ABI (include/crt.h "multi-GPU" section).
//
// The reference is single-GPU (SURVEY.md §2 "Parallelism": one process, one device, default
// stream), so there is no reference interface to mirror; the split follows north_star /
// pixels are independent (RNG stream = GLOBAL pixel index, core/deviceManage.h:
// 126-127), triangles are replicated, strips of 32 image rows are dealt round-robin to the
// devices, and the only exchange step is ONE grouped RCCL send/recv of the float3 pieces to
// rank 0 (rccl.h: ncclSend / ncclRecv inside ncclGroupStart / ncclGroupEnd), followed by a
// reassembly kernel there.  No collective touches the render itself.
//
// Two ways in:
//   crt_gather_strips   one process (or thread) per GPU; the caller owns the ncclComm_t
//   crt_multi_*         one process, n_gpus devices: ncclCommInitAll, one stream and one host
//                       worker thread per device (the driver shape of core/render.h:159-189)
// RCCL is bound at first use with dlopen (librccl.so.1 — the copy the process already holds, if
// any — so the library itself loads on hosts without RCCL and fails loudly only here).
#include "../../include/crt.h"
#include "../../include/crt_debug.h"
#include "crt_device.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <string.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#define fail(...) crt::set_error(__VA_ARGS__)

namespace {

struct Rccl {
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

const Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = nullptr;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (h) break;
        }
        if (!h) return;
        auto sym = [&](const char* n) { return dlsym(h, n); };
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        r.ok = r.CommInitAll && r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv && r.GetErrorString;
    });
    return r;
}

#define NCCL_TRY(expr)                                                                          \
    do {                                                                                        \
        ncclResult_t r_ = (expr);                                                               \
        if (r_ != ncclSuccess)                                                                  \
            return fail(CRT_ERR_HIP, "%s failed: %s", #expr, rccl().GetErrorString(r_));        \
    } while (0)
#define HIP_TRY(expr)                                                                           \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(CRT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// Test hook (crt_debug_inject_send_failure): the next `count` ncclSend calls of this library report ncclInternalError
// WITHOUT reaching RCCL — the error paths of the exchange step must close the group, drain the streams and restore
// the caller's device (tests/test_gpu_multi.py).
std::atomic<int> g_fail_sends{0};

ncclResult_t send_or_injected_failure(const Rccl& R, const void* buf, size_t count, int peer, ncclComm_t comm, hipStream_t stream) {
    int left = g_fail_sends.load();
    while (left > 0 && !g_fail_sends.compare_exchange_weak(left, left - 1)) {}
    if (left > 0) return ncclInternalError;
    return R.Send(buf, count, ncclFloat, peer, comm, stream);
}

// First error of a grouped exchange: recorded, not returned, so that ncclGroupEnd always runs.
struct FirstError {
    ncclResult_t rc = ncclSuccess;
    const char* what = "";
    void note(ncclResult_t r, const char* w) { if (r != ncclSuccess && rc == ncclSuccess) { rc = r; what = w; } }
    bool ok() const { return rc == ncclSuccess; }
};

bool layout_ok(int nx, int ny, int strip_rows, int world) {
    return nx > 0 && ny > 0 && world > 0 && strip_rows > 0 && (world == 1 || strip_rows % 32 == 0);
}

}  // namespace

extern "C" {

int64_t crt_strip_part_stride(int nx, int ny, int strip_rows, int world) {
    if (!layout_ok(nx, ny, strip_rows, world)) return CRT_ERR_INVALID;
    int rows = crt_strip_rows(ny, 0, strip_rows, world);   // rank 0 owns the first strip of every round: the largest piece
    int64_t elems = (int64_t)rows * nx * 3;
    return (elems + 3) / 4 * 4;
}

int crt_assemble_strips(const float* d_parts, int64_t part_stride, int nx, int ny, int strip_rows, int world,
                        float* d_full, void* stream) {
    if (!d_parts || !d_full || !layout_ok(nx, ny, strip_rows, world) || part_stride < crt_strip_part_stride(nx, ny, strip_rows, world) - 3)
        return fail(CRT_ERR_INVALID, "crt_assemble_strips: bad argument");
    if (crt_device_count() < 1) return fail(CRT_ERR_NO_DEVICE, "crt_assemble_strips: no HIP device");
    HIP_TRY(crt::launch_assemble_strips(d_parts, part_stride, nx, ny, strip_rows, world, d_full, (hipStream_t)stream));
    return CRT_OK;
}

int crt_gather_strips(const float* d_piece, int nx, int ny, int strip_rows, int rank, int world, float* d_parts,
                      int64_t part_stride, float* d_full, void* nccl_comm, void* stream_) {
    if (!layout_ok(nx, ny, strip_rows, world) || rank < 0 || rank >= world || !nccl_comm)
        return fail(CRT_ERR_INVALID, "crt_gather_strips: bad layout, rank or communicator");
    const int my_rows = ((long long)rank * strip_rows < ny) ? crt_strip_rows(ny, rank, strip_rows, world) : 0;
    if ((my_rows > 0 && !d_piece) || (rank == 0 && (!d_parts || !d_full)))
        return fail(CRT_ERR_INVALID, "crt_gather_strips: null buffer");
    if (rank == 0 && part_stride < crt_strip_part_stride(nx, ny, strip_rows, world) - 3)
        return fail(CRT_ERR_INVALID, "crt_gather_strips: part_stride %lld is smaller than the largest piece", (long long)part_stride);
    if (crt_device_count() < 1) return fail(CRT_ERR_NO_DEVICE, "crt_gather_strips: no HIP device");
    const Rccl& R = rccl();
    if (!R.ok) return fail(CRT_ERR_UNSUPPORTED, "crt_gather_strips: RCCL (librccl.so.1) cannot be loaded: %s", dlerror());
    ncclComm_t comm = (ncclComm_t)nccl_comm;
    hipStream_t stream = (hipStream_t)stream_;
    // Every rank, the root included, sends its piece to rank 0; the root posts one receive per rank.  An error inside
    // the group is recorded and the group is CLOSED all the same: a thread that returned between GroupStart and
    // GroupEnd would leave every later RCCL call of that thread swallowed by the open group.
    NCCL_TRY(R.GroupStart());
    FirstError err;
    if (my_rows > 0) err.note(send_or_injected_failure(R, d_piece, (size_t)my_rows * nx * 3, 0, comm, stream), "ncclSend");
    if (rank == 0 && err.ok())
        for (int r = 0; r < world && err.ok(); ++r) {
            if ((long long)r * strip_rows >= ny) break;   // ranks beyond the last strip own nothing
            const int rows = crt_strip_rows(ny, r, strip_rows, world);
            err.note(R.Recv(d_parts + (size_t)r * part_stride, (size_t)rows * nx * 3, ncclFloat, r, comm, stream), "ncclRecv");
        }
    err.note(R.GroupEnd(), "ncclGroupEnd");
    if (!err.ok()) {
        (void)hipStreamSynchronize(stream);   // whatever part of the exchange was enqueued has left the stream when this returns
        return fail(CRT_ERR_HIP, "crt_gather_strips: %s failed: %s", err.what, R.GetErrorString(err.rc));
    }
    if (rank == 0) HIP_TRY(crt::launch_assemble_strips(d_parts, part_stride, nx, ny, strip_rows, world, d_full, stream));
    return CRT_OK;
}

int crt_debug_inject_send_failure(int count) {
    const int prev = g_fail_sends.exchange(count < 0 ? 0 : count);
    return prev;
}

static_assert(sizeof(ncclUniqueId) == CRT_COMM_ID_BYTES, "CRT_COMM_ID_BYTES must be sizeof(ncclUniqueId) (rccl.h: NCCL_UNIQUE_ID_BYTES)");

int crt_comm_unique_id(void* id_bytes) {
    if (!id_bytes) return fail(CRT_ERR_INVALID, "crt_comm_unique_id: null output");
    const Rccl& R = rccl();
    if (!R.ok) return fail(CRT_ERR_UNSUPPORTED, "crt_comm_unique_id: RCCL (librccl.so.1) cannot be loaded");
    ncclUniqueId id;
    NCCL_TRY(R.GetUniqueId(&id));
    memcpy(id_bytes, &id, sizeof id);
    return CRT_OK;
}

int crt_comm_init_rank(int world, const void* id_bytes, int rank, void** nccl_comm_out) {
    if (!id_bytes || !nccl_comm_out || world < 1 || rank < 0 || rank >= world)
        return fail(CRT_ERR_INVALID, "crt_comm_init_rank: bad argument (world %d, rank %d)", world, rank);
    if (crt_device_count() < 1) return fail(CRT_ERR_NO_DEVICE, "crt_comm_init_rank: no HIP device");
    const Rccl& R = rccl();
    if (!R.ok) return fail(CRT_ERR_UNSUPPORTED, "crt_comm_init_rank: RCCL (librccl.so.1) cannot be loaded");
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof id);
    ncclComm_t comm = nullptr;
    NCCL_TRY(R.CommInitRank(&comm, world, id, rank));
    *nccl_comm_out = comm;
    return CRT_OK;
}

int crt_comm_destroy(void* nccl_comm) {
    if (!nccl_comm) return CRT_OK;
    const Rccl& R = rccl();
    if (!R.ok) return fail(CRT_ERR_UNSUPPORTED, "crt_comm_destroy: RCCL (librccl.so.1) cannot be loaded");
    NCCL_TRY(R.CommDestroy((ncclComm_t)nccl_comm));
    return CRT_OK;
}

}  // extern "C"

// ---------------------------------------------------------------- one process, n_gpus devices
struct crt_multi {
    int n = 0, nx = 0, ny = 0;
    std::vector<int> dev;
    std::vector<crt_scene*> scene;
    std::vector<crt_rng*> rng;
    std::vector<float*> piece;          // per device: its strips, one after the other
    std::vector<hipStream_t> stream;
    std::vector<ncclComm_t> comm;
    float *parts = nullptr, *full = nullptr;   // device 0
    int64_t part_stride = 0;
    // one host worker per device: enqueueing a frame costs tens of microseconds per device,
    // which a single thread would serialise
    std::vector<std::thread> worker;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    unsigned long long generation = 0;
    int pending = 0;
    bool quit = false;
    // the job of the current generation
    const crt_camera* job_camera = nullptr;
    const float* job_points = nullptr;
    int job_samples = 0, job_max_depth = 0, job_frame = 0, job_n_points = 0;
    int world = CRT_WORLD_LIST, shader = CRT_SHADER_LAMBERT;   // crt_multi_configure
    uint64_t seed = 0;
    // a configure job (crt_multi_configure): every worker sets its replica's materials and (BVH world) builds its tree
    bool job_configure = false;
    const crt_material* job_table = nullptr;
    int job_n_materials = 0;
    const int* job_triangle_material = nullptr;
    int job_variant = 0, job_timing = 1;   // the CALLER's crt_set_kernel_variant / crt_set_kernel_timing (per-thread settings; the workers are other threads)
    std::vector<int> job_rc;
    std::vector<std::string> job_err;
};

namespace {

constexpr int kStripRows = 32;

void worker_main(crt_multi* m, int g) {
    (void)hipSetDevice(m->dev[g]);
    unsigned long long seen = 0;
    for (;;) {
        {
            std::unique_lock<std::mutex> lock(m->mu);
            m->cv_go.wait(lock, [&] { return m->quit || m->generation != seen; });
            if (m->quit) return;
            seen = m->generation;
        }
        int rc = CRT_OK;
        crt_set_kernel_variant(m->job_variant);
        crt_set_kernel_timing(m->job_timing);
        if (m->job_configure) {
            if (m->job_table) rc = crt_scene_set_materials(m->scene[g], m->job_table, m->job_n_materials, m->job_triangle_material);
            // the same private stream-0 sequence on every replica (rng = NULL): identical trees (host build, deterministic)
            if (rc == CRT_OK && m->world == CRT_WORLD_BVH && crt_scene_bvh_node_count(m->scene[g]) < 0)
                rc = crt_scene_build_bvh(m->scene[g], nullptr, m->seed);
        }
        if (m->job_points) {
            rc = crt_scene_update_pose(m->scene[g], m->job_points, m->job_n_points, m->stream[g]);
            if (rc == CRT_OK && m->world == CRT_WORLD_BVH) rc = crt_scene_refit_bvh(m->scene[g], m->stream[g]);   // Update_BVH, render.h:215
        }
        if (rc == CRT_OK && m->job_camera && m->rng[g])
            rc = crt_render_strips_ex(m->scene[g], m->job_camera, m->rng[g], m->nx, m->ny, m->job_samples, m->job_max_depth,
                                      m->job_frame, g, kStripRows, m->n, m->world, m->shader, m->piece[g], m->stream[g]);
        m->job_rc[g] = rc;
        if (rc != CRT_OK) m->job_err[g] = crt_last_error();
        {
            std::lock_guard<std::mutex> lock(m->mu);
            if (--m->pending == 0) m->cv_done.notify_all();
        }
    }
}

int run_job(crt_multi* m) {
    {
        std::lock_guard<std::mutex> lock(m->mu);
        m->pending = m->n;
        ++m->generation;
    }
    m->cv_go.notify_all();
    {
        std::unique_lock<std::mutex> lock(m->mu);
        m->cv_done.wait(lock, [&] { return m->pending == 0; });
    }
    for (int g = 0; g < m->n; ++g)
        if (m->job_rc[g] != CRT_OK) return fail(m->job_rc[g], "device %d: %s", m->dev[g], m->job_err[g].c_str());
    return CRT_OK;
}

}  // namespace

extern "C" {

void crt_multi_free(crt_multi* m) {
    if (!m) return;
    {
        std::lock_guard<std::mutex> lock(m->mu);
        m->quit = true;
    }
    m->cv_go.notify_all();
    for (std::thread& t : m->worker)
        if (t.joinable()) t.join();
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (int g = 0; g < (int)m->dev.size(); ++g) {
        (void)hipSetDevice(m->dev[g]);
        if (g < (int)m->stream.size() && m->stream[g]) (void)hipStreamSynchronize(m->stream[g]);
        if (g < (int)m->comm.size() && m->comm[g] && rccl().ok) (void)rccl().CommDestroy(m->comm[g]);
        if (g < (int)m->scene.size()) crt_scene_free(m->scene[g]);
        if (g < (int)m->rng.size()) crt_rng_free(m->rng[g]);
        if (g < (int)m->piece.size() && m->piece[g]) (void)hipFree(m->piece[g]);
        if (g == 0) {
            if (m->parts) (void)hipFree(m->parts);
            if (m->full) (void)hipFree(m->full);
        }
        if (g < (int)m->stream.size() && m->stream[g]) {
            (void)crt_stream_release(m->stream[g]);   // the launch context of this stream (records, binning scratch, events)
            (void)hipStreamDestroy(m->stream[g]);
        }
    }
    (void)hipSetDevice(prev);
    delete m;
}

crt_multi* crt_multi_create(const float* points, int n_points, const float* idx3f, int n_tris,
                            const float* face_normals, int index_base, const float albedo[3], int nx, int ny,
                            uint64_t seed, int n_gpus) {
    const int have = crt_device_count();
    if (have < 1) { fail(CRT_ERR_NO_DEVICE, "crt_multi_create: no HIP device"); return nullptr; }
    if (n_gpus < 1 || n_gpus > have || nx <= 0 || ny <= 0) {
        fail(CRT_ERR_INVALID, "crt_multi_create: n_gpus = %d with %d visible devices, image %d x %d", n_gpus, have, nx, ny);
        return nullptr;
    }
    const Rccl& R = rccl();
    if (!R.ok) { fail(CRT_ERR_UNSUPPORTED, "crt_multi_create: RCCL (librccl.so.1) cannot be loaded"); return nullptr; }
    crt_multi* m = new (std::nothrow) crt_multi;
    if (!m) { fail(CRT_ERR_NOMEM, "crt_multi_create: out of memory"); return nullptr; }
    m->n = n_gpus; m->nx = nx; m->ny = ny; m->seed = seed;
    m->dev.resize(n_gpus);
    for (int g = 0; g < n_gpus; ++g) m->dev[g] = g;
    m->scene.assign(n_gpus, nullptr); m->rng.assign(n_gpus, nullptr); m->piece.assign(n_gpus, nullptr);
    m->stream.assign(n_gpus, nullptr); m->comm.assign(n_gpus, nullptr);
    m->job_rc.assign(n_gpus, CRT_OK); m->job_err.assign(n_gpus, std::string());
    m->part_stride = crt_strip_part_stride(nx, ny, kStripRows, n_gpus);
    int prev = 0;
    (void)hipGetDevice(&prev);
    bool ok = true;
    for (int g = 0; g < n_gpus && ok; ++g) {
        ok = hipSetDevice(m->dev[g]) == hipSuccess && hipStreamCreateWithFlags(&m->stream[g], hipStreamNonBlocking) == hipSuccess;
        if (!ok) { fail(CRT_ERR_HIP, "crt_multi_create: device %d: cannot create a stream", m->dev[g]); break; }
        m->scene[g] = crt_scene_from_mesh(points, n_points, idx3f, n_tris, face_normals, index_base, albedo);   // replicated
        ok = m->scene[g] != nullptr;
        const int rows = ((long long)g * kStripRows < ny) ? crt_strip_rows(ny, g, kStripRows, n_gpus) : 0;
        if (ok && rows > 0) {
            m->rng[g] = crt_rng_create_strips(nx, ny, g, kStripRows, n_gpus, seed);
            ok = m->rng[g] != nullptr && hipMalloc(&m->piece[g], (size_t)rows * nx * 3 * sizeof(float)) == hipSuccess;
        }
        if (ok && g == 0)
            ok = hipMalloc(&m->parts, (size_t)m->part_stride * n_gpus * sizeof(float)) == hipSuccess &&
                 hipMalloc(&m->full, (size_t)nx * ny * 3 * sizeof(float)) == hipSuccess;
    }
    if (ok && R.CommInitAll(m->comm.data(), n_gpus, m->dev.data()) != ncclSuccess) {
        fail(CRT_ERR_HIP, "crt_multi_create: ncclCommInitAll over %d devices failed", n_gpus);
        ok = false;
    }
    (void)hipSetDevice(prev);
    if (!ok) { crt_multi_free(m); return nullptr; }
    for (int g = 0; g < n_gpus; ++g) m->worker.emplace_back(worker_main, m, g);
    return m;
}

namespace {
// the CALLING thread's per-thread settings, handed to the workers with the job (the setters return the previous value)
void read_caller_settings(crt_multi* m) {
    m->job_variant = crt_set_kernel_variant(0);
    crt_set_kernel_variant(m->job_variant);
    m->job_timing = crt_set_kernel_timing(1);
    crt_set_kernel_timing(m->job_timing);
}
}  // namespace

int crt_multi_configure(crt_multi* m, int world, int shader, const crt_material* table, int n_materials, const int* triangle_material) {
    if (!m || (world != CRT_WORLD_LIST && world != CRT_WORLD_BVH) || shader < CRT_SHADER_LAMBERT || shader > CRT_SHADER_NORMAL ||
        (table && n_materials < 1))
        return fail(CRT_ERR_INVALID, "crt_multi_configure: bad argument (world %d, shader %d)", world, shader);
    read_caller_settings(m);
    const int prev_world = m->world, prev_shader = m->shader;
    m->world = world; m->shader = shader;
    m->job_camera = nullptr; m->job_points = nullptr;
    m->job_configure = true;
    m->job_table = table; m->job_n_materials = n_materials; m->job_triangle_material = triangle_material;
    const int rc = run_job(m);
    m->job_configure = false;
    m->job_table = nullptr; m->job_triangle_material = nullptr;
    if (rc != CRT_OK) { m->world = prev_world; m->shader = prev_shader; }
    return rc;
}

int crt_multi_update_pose(crt_multi* m, const float* new_points, int n_points) {
    if (!m || !new_points) return fail(CRT_ERR_INVALID, "crt_multi_update_pose: null argument");
    read_caller_settings(m);
    m->job_camera = nullptr;
    m->job_points = new_points;
    m->job_n_points = n_points;
    const int rc = run_job(m);
    m->job_points = nullptr;
    return rc;
}

namespace {
// every stream of the renderer idle, the caller's device current again: the state every exit of crt_multi_render leaves
hipError_t drain_and_restore(crt_multi* m, int prev) {
    hipError_t first = hipSuccess;
    for (int g = 0; g < m->n; ++g) {
        hipError_t e = hipSetDevice(m->dev[g]);
        if (e == hipSuccess) e = hipStreamSynchronize(m->stream[g]);
        if (e != hipSuccess && first == hipSuccess) first = e;
    }
    (void)hipSetDevice(prev);
    return first;
}
}  // namespace

int crt_multi_render(crt_multi* m, const crt_camera* camera, int samples, int max_depth, int frame_index, float* rgb_out) {
    if (!m || !camera) return fail(CRT_ERR_INVALID, "crt_multi_render: null argument");
    const Rccl& R = rccl();
    int prev = 0;
    (void)hipGetDevice(&prev);
    m->job_camera = camera;
    m->job_points = nullptr;
    m->job_samples = samples; m->job_max_depth = max_depth; m->job_frame = frame_index;
    read_caller_settings(m);
    int rc = run_job(m);                       // every device has its strips enqueued on its stream
    m->job_camera = nullptr;
    if (rc != CRT_OK) {                        // some device failed to enqueue: the others have work in flight — drain it
        const std::string text = crt_last_error();
        (void)drain_and_restore(m, prev);
        return fail(rc, "%s", text.c_str());
    }
    // one grouped send/recv over all communicators of this process, then the reassembly on device 0.  Errors inside
    // the group are recorded; the group is always closed, the streams drained and the caller's device restored.
    FirstError err;
    err.note(R.GroupStart(), "ncclGroupStart");
    if (err.ok()) {
        for (int g = 0; g < m->n && err.ok(); ++g) {
            const int rows = ((long long)g * kStripRows < m->ny) ? crt_strip_rows(m->ny, g, kStripRows, m->n) : 0;
            if (rows == 0) continue;
            err.note(send_or_injected_failure(R, m->piece[g], (size_t)rows * m->nx * 3, 0, m->comm[g], m->stream[g]), "ncclSend");
            if (err.ok())
                err.note(R.Recv(m->parts + (size_t)g * m->part_stride, (size_t)rows * m->nx * 3, ncclFloat, g, m->comm[0], m->stream[0]), "ncclRecv");
        }
        err.note(R.GroupEnd(), "ncclGroupEnd");
    }
    if (!err.ok()) {
        (void)drain_and_restore(m, prev);
        return fail(CRT_ERR_HIP, "crt_multi_render: %s failed: %s", err.what, R.GetErrorString(err.rc));
    }
    hipError_t e = hipSetDevice(m->dev[0]);
    if (e == hipSuccess) e = crt::launch_assemble_strips(m->parts, m->part_stride, m->nx, m->ny, kStripRows, m->n, m->full, m->stream[0]);
    if (e == hipSuccess && rgb_out)
        e = hipMemcpyAsync(rgb_out, m->full, (size_t)m->nx * m->ny * 3 * sizeof(float), hipMemcpyDeviceToHost, m->stream[0]);
    const hipError_t e2 = drain_and_restore(m, prev);   // the root first has everything, then the senders' streams are done: cheap
    if (e == hipSuccess) e = e2;
    if (e != hipSuccess) return fail(CRT_ERR_HIP, "crt_multi_render: %s", hipGetErrorString(e));
    return CRT_OK;
}

const float* crt_multi_frame(const crt_multi* m) { return m ? m->full : nullptr; }

int crt_multi_device_count(const crt_multi* m) { return m ? m->n : CRT_ERR_INVALID; }

}  // extern "C"
