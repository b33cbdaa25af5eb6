/*
 * crt.h — C ABI of libcrt_hip.so, the MI355X (gfx950) drop-in for the list-render path of
 * chocorone/CudaRaytracer.
 *
 * The reference has no plugin/FFI layer: the path is a __global__ function and a handful
 * of host functions in headers compiled into one executable.  Each entry point below
 * replaces one of those host functions / kernels; the citation after "replaces:" is the
 * reference interface (paths relative to CudaTest/src of the reference tree).
 *
 * Conventions
 *   - plain C types only; device buffers are passed as void* / float* device pointers
 *   - int return: 0 = CRT_OK, negative = error (never exit(), unlike the reference's
 *     checkCudaErrors / CHECK, core/deviceManage.h:3-17); crt_last_error() gives the text
 *   - the library owns device memory behind opaque handles; the caller owns every buffer
 *     it passes in (host or device)
 *   - devices: a handle (scene, RNG states) lives on the HIP device that was current when it
 *     was created; every call that uses it must be made with that device current, otherwise
 *     CRT_ERR_INVALID.
 *   - streams and threads: render entry points only ENQUEUE work on the stream they are given.
 *     Everything a launch writes besides its outputs (per-frame derived triangle records,
 *     binning scratch, timing events) belongs to a launch context keyed by (device, stream),
 *     so renders on different streams or devices — same scene or not, same host thread or not
 *     — are independent; renders on one stream are ordered by the stream.  One crt_rng must not
 *     be in flight on two streams at once (its states are advanced in place), and
 *     crt_scene_update_pose must be ordered by the caller against renders of that scene on
 *     OTHER streams (same stream: ordered by the stream).  crt_last_error (and the diagnostics of
 *     crt_debug.h: crt_last_kernel_*, crt_set_kernel_variant, crt_debug_tiled_stats) are per calling thread.
 *   - diagnostics, A/B switches, timing hooks and parity hooks are declared in crt_debug.h, not here.
 *   - "bit-exact" in this file means: identical to the in-repo CPU oracle (oracle/), which is
 *     a restatement of the reference and is PARITY UNPINNED (no reference fixtures exist and
 *     the reference cannot be built here; see oracle/crt_oracle.h).
 *   - there is NO CPU fallback: every compute entry point fails with CRT_ERR_NO_DEVICE
 *     when no HIP device is usable
 *   - ONE HIP runtime per process: libcrt_hip.so asks the loader for libamdhip64.so.7 (by SONAME).  A C/C++ host that
 *     links the ROCm runtime itself shares that copy and has nothing to do.  A host that carries its OWN copy of the
 *     runtime under another file name (PyTorch wheels bundle torch/lib/libamdhip64.so) must load that copy BEFORE
 *     libcrt_hip.so, so that the SONAME is already satisfied; loaded the other way round the process holds two HIP
 *     runtimes and the second one sees no device (torch.cuda.is_available() == False, or CRT_ERR_NO_DEVICE here).
 *     cudaraytracer_amd/_lib.py imports torch first for that reason; INTEGRATION.md "One HIP runtime per process".
 */
#ifndef CRT_H
#define CRT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRT_ABI_VERSION 4

enum {
    CRT_OK = 0,
    CRT_ERR_INVALID = -1,     /* bad argument (null pointer, non-positive size, index out of range) */
    CRT_ERR_NO_DEVICE = -2,   /* no usable HIP device / HIP runtime error at init */
    CRT_ERR_HIP = -3,         /* a HIP call or kernel launch failed */
    CRT_ERR_IO = -4,          /* file could not be read / written */
    CRT_ERR_UNSUPPORTED = -5, /* input whose reference behaviour is undefined (see crt_render_rows) */
    CRT_ERR_NOMEM = -6
};

/* replaces: class Camera's data members, core/camera.h:46-52 (88 bytes, same order). */
typedef struct crt_camera {
    float lower_left_corner[3];
    float horizontal[3];
    float vertical[3];
    float origin[3];
    float x[3], y[3], z[3];
    float lens_radius;
} crt_camera;

/* Opaque handles. */
typedef struct crt_scene crt_scene; /* replaces: HitableList** world (kernel.cu:78-90) */
typedef struct crt_rng crt_rng;     /* replaces: curandState* curand_state (kernel.cu:69-71) */

/* ---- library ------------------------------------------------------------------- */

int crt_abi_version(void);
/* Text of the last error raised on the calling thread ("" if none). */
const char *crt_last_error(void);
/* Number of visible HIP devices (0 when the runtime is unusable). Never fails. */
int crt_device_count(void);

/* ---- camera -------------------------------------------------------------------- */

/* replaces: Camera::Camera (core/camera.h:18-38) as called by create_camera
 * (createScene.h:142-154, init_camera :156-165).  Host-side; tan() is evaluated with the
 * host libm so the device never calls a transcendental. */
void crt_camera_make(crt_camera *out, const float lookfrom[3], const float lookat[3],
                     const float vup[3], float vfov, float aspect, float aperture,
                     float focus_dist);

/* ---- scene --------------------------------------------------------------------- */

/* replaces: create_FBXMesh + add_mesh_withNormal (createScene.h:175-210).
 * points:  n_points x 3 floats.  idx3f: n_tris x 3 floats — the reference stores vertex
 * indices as float triplets (vec3* idxVertex); triangle i uses
 * {points[int(idx[2])-base], points[int(idx[1])-base], points[int(idx[0])-base]}
 * (reversed winding, createScene.h:183).  face_normals: n_tris x 3 floats kept as the
 * stored normal (6-argument Triangle ctor, shapes/triangle.h:21-29), or NULL to apply the
 * 5-argument ctor rule unit_vector(cross(v1-v0, v2-v0)) (shapes/triangle.h:8-19).
 * albedo: the single Lambertian ConstantTexture colour (createScene.h:180).
 * Every triangle gets Transform(0,0,1) and EPSILON 1e-6 as in the reference.
 * All arrays are host pointers.  Returns NULL on error. */
crt_scene *crt_scene_from_mesh(const float *points, int n_points, const float *idx3f,
                               int n_tris, const float *face_normals, int index_base,
                               const float albedo[3]);

/* Replaces update_pose / updateFBXObj's device half (createScene.h:99-110,126-140): the vertices of
 * every triangle are re-read from `new_points` (n_points x 3 floats, host; n_points must equal the
 * count the scene was built from) through the scene's index triplets in the same reversed winding;
 * stored normals are kept (Triangle::SetVertices, shapes/triangle.h:40-45).  Enqueued on `stream`
 * (hipStream_t, NULL = default); the next crt_render_rows on that stream sees the new pose.
 * `new_points` is copied into a pinned staging buffer before the call returns: the caller may
 * reuse it at once. */
int crt_scene_update_pose(crt_scene *scene, const float *new_points, int n_points, void *stream);

/* replaces: parseObjByName (Loader/obj_loader.h:20-111) + computeMean/centering
 * (:120-133) + the historic `points[i] *= 100.0` scale (:164) + the scene pack above with
 * index_base 1 and computed normals.  `v x y z` and `f a b c` records are read as in the
 * reference.  Deliberate extension beyond it (the reference's `ss >> float`
 * stops at the first '/'): face tokens `a/b`, `a//c`, `a/b/c` use their leading vertex index,
 * negative indices count back from the latest vertex, polygons are fanned into triangles
 * (v0, vk, vk+1) — this is what makes CudaTest/objects/cbox_smallbox.obj loadable. */
crt_scene *crt_scene_from_obj(const char *path, float scale, const float albedo[3]);

int crt_scene_triangle_count(const crt_scene *scene);
void crt_scene_free(crt_scene *scene);

/* Host-side OBJ parse only (same rules as crt_scene_from_obj, before scaling; n_tris counts the
 * triangles after fanning): call with
 * NULL buffers to get the counts, then with buffers of n_points*3 / n_tris*3 floats. */
int crt_parse_obj(const char *path, float *points, float *idx3f, int *n_points, int *n_tris);

/* ---- RNG ----------------------------------------------------------------------- */

/* replaces: cudaMallocManaged(curandState[nx*ny]) + SetCurandState -> random_init
 * (kernel.cu:69-71, core/deviceManage.h:120-136).  Creates device-resident XORWOW states
 * for the `count` pixels [first_pixel, first_pixel+count) of an image and initialises
 * them like curand_init(seed, pixel_index, 0, &state[pixel_index]).  A rank that renders
 * a row band passes first_pixel = row_begin*nx, count = rows*nx. */
crt_rng *crt_rng_create(int64_t first_pixel, int64_t count, uint64_t seed);
/* Same for a rank that renders strips of `strip_rows` image rows dealt round-robin: strips
 * first_strip, first_strip + strip_step, ... (strip s covers rows [s*strip_rows,
 * (s+1)*strip_rows) clipped to ny).  States are stored strip after strip, row-major, like
 * the colour buffer of crt_render_strips.  strip_rows must be a multiple of 32. */
crt_rng *crt_rng_create_strips(int nx, int ny, int first_strip, int strip_rows, int strip_step,
                               uint64_t seed);
/* Number of image rows in that strip set (what the rank's buffers must hold). */
int crt_strip_rows(int ny, int first_strip, int strip_rows, int strip_step);
/* Re-run the initialisation (same as a fresh crt_rng_create). */
int crt_rng_reset(crt_rng *rng, uint64_t seed);
/* Copy states to the host as count x 6 uint32 {d, v0..v4} (test/diagnostic use). */
int crt_rng_export(const crt_rng *rng, uint32_t *out6);
void crt_rng_free(crt_rng *rng);

/* ---- render -------------------------------------------------------------------- */

/* replaces: the `render` kernel (core/render.h:105-129) and everything it calls —
 * Camera::get_ray, Hitable::hit / Transform::TransformRay, HitableList::
 * collision_detection, Triangle::collision_detection, Lambertian::scatter,
 * LambertShade, backgroundSky — for rows [row_begin, row_begin+row_count) of an
 * nx x ny image.
 *   d_rgb:  DEVICE pointer to row_count*nx*3 floats (AoS float3 like vec3* colorBuffer);
 *           element 0 is pixel (x=0, y=row_begin); row 0 of the image is its bottom row.
 *   rng:    states for exactly those pixels (first_pixel = row_begin*nx); advanced in
 *           place, so consecutive calls continue the streams like the reference's frame
 *           loop (core/render.h:174-186).
 *   max_depth, frame_index: accepted for signature parity; dead on this path
 *           (LambertShade never recurses, core/render.h:70-87,120).
 *   stream: hipStream_t (NULL = default stream).  The call only enqueues work.
 * CRT_ERR_UNSUPPORTED when camera->lens_radius != 0: the reference then reads lens samples
 * from state[0] in every thread concurrently (core/render.h:118), a data race with no
 * defined result. */
int crt_render_rows(const crt_scene *scene, const crt_camera *camera, crt_rng *rng, int nx,
                    int ny, int samples, int max_depth, int frame_index, int row_begin,
                    int row_count, float *d_rgb, void *stream);

/* The same kernel on a strip set (see crt_rng_create_strips): multi-GPU load balancing.
 * A contiguous band gives the ranks that own the object all the work; strips of 32 rows
 * dealt round-robin spread it.  d_rgb holds crt_strip_rows(...)*nx*3 floats, strip after
 * strip.  Pixels are identical to crt_render_rows on the same image rows. */
int crt_render_strips(const crt_scene *scene, const crt_camera *camera, crt_rng *rng, int nx,
                      int ny, int samples, int max_depth, int frame_index, int first_strip,
                      int strip_rows, int strip_step, float *d_rgb, void *stream);

/* replaces: SetCurandState + one frame of renderListAnimation (core/render.h:159-189, kernel.cu:69-71): RNG states
 * from `seed`, the full image on the current device's default stream, then nx*ny*3 floats copied to the HOST
 * buffer rgb_out.  The one-call surface `render(width, height, samples, hitable_list, camera) -> RGB buffer`.
 * The states (with a pristine copy that makes a repeated call with the same image size and seed a device-to-device
 * restore instead of a re-initialisation) and the device colour buffer are kept per device between calls and freed
 * by crt_stream_release(NULL).  Calls are serialised per process.  A frame LOOP should use crt_rng_create +
 * crt_render_rows on its own stream and buffers: the states then advance from frame to frame as the reference's
 * do (core/render.h:174-186) and nothing crosses PCIe unless the host asks.  rgb_out may be any host memory; a
 * page-locked buffer (hipHostMalloc / hipHostRegister) takes the frame at the link's rate, a pageable one goes through
 * the runtime's staging copy (bench.py: pcie_inclusive_one_call reports both). */
int crt_render(int nx, int ny, int samples, int max_depth, int frame_index,
               const crt_scene *scene, const crt_camera *camera, uint64_t seed,
               float *rgb_out);

/* ---- BVH world ------------------------------------------------------------------
 * The reference's shipped configuration (kernel.cu:97): the same `render` kernel and
 * LambertShade, with world = a BVHNode tree over the triangle list. */

/* replaces: create_BVHfromList -> create_BVH<<<1,1>>> -> BVHNode::BVHNode (createScene.h:42-44,
 * 212-219; hitable/bvh.h:76-125): median split of the list (n/2 | n - n/2, leaves of 1 or 2
 * triangles) after sorting every span by box.min() on an axis drawn per node as
 * int(3 * curand_uniform(state)) from STREAM 0 (create_BVH passes the base of the curandState
 * array).  rng: the states that hold pixel 0 — stream 0 continues from where it is and is left
 * one draw per node further, as in the reference; NULL (or states that do not hold pixel 0, e.g.
 * another rank's strips): a private curand_init(seed, 0, 0) stream is used, which is the same
 * sequence when the BVH is built right after SetCurandState like the reference does
 * (kernel.cu:69-71, :36).  The sort is Thrust's sequential merge sort with the reference's
 * comparator, restated (crt_bvh.cpp; PARITY UNPINNED: the comparator is not a strict weak
 * ordering, so the order is the algorithm's).  Host-built, uploaded flat; replaces any earlier
 * BVH of the scene.  CRT_ERR_UNSUPPORTED for an empty list (the reference does not terminate). */
int crt_scene_build_bvh(crt_scene *scene, crt_rng *rng, uint64_t seed);
/* replaces: Update_BVH -> UpdateBVH<<<1,1>>> -> BVHNode::UpdateBVH (createScene.h:46-58,
 * hitable/bvh.h:135-158): boxes recomputed bottom-up from the current triangle vertices (after
 * crt_scene_update_pose), topology kept.  Enqueued on `stream`. */
int crt_scene_refit_bvh(crt_scene *scene, void *stream);
int crt_scene_bvh_node_count(const crt_scene *scene);
/* replaces: render<<<>>> with world = BVHNode** (core/render.h:222; BVHNode::collision_detection
 * hitable/bvh.h:160-190, AABB::hit core/aabb.h:30-43).  Arguments as crt_render_rows /
 * crt_render_strips. */
int crt_render_rows_bvh(const crt_scene *scene, const crt_camera *camera, crt_rng *rng, int nx,
                        int ny, int samples, int max_depth, int frame_index, int row_begin,
                        int row_count, float *d_rgb, void *stream);
int crt_render_strips_bvh(const crt_scene *scene, const crt_camera *camera, crt_rng *rng, int nx,
                          int ny, int samples, int max_depth, int frame_index, int first_strip,
                          int strip_rows, int strip_step, float *d_rgb, void *stream);
/* replaces: writeCSV("output.csv", data) (Loader/CSVWriter.h:8-32) with the rows main() and
 * renderBVHAnimation collect (kernel.cu:61,18,37; core/render.h:232): header
 * frame,rendering,update,build; one row with the build time; one row per frame with the render
 * and BVH-update times (seconds, std::to_string format). */
int crt_write_timing_csv(const char *path, int n_frames, const int *frame, const double *rendering,
                         const double *update, double build);

/* ---- materials and shaders ---------------------------------------------------------
 * replaces: the Material* of every Triangle (shapes/triangle.h:54) and the classes of
 * material/material.h:57-161 over ConstantTexture (material/texture.h:12-22).  The scene pack
 * (createScene.h:180) gives all triangles ONE Lambertian(albedo): that is every scene's default
 * table {CRT_LAMBERTIAN, albedo}. */
enum { CRT_LAMBERTIAN = 0, CRT_METAL = 1, CRT_DIELECTRIC = 2, CRT_DIFFUSE_LIGHT = 3 };
typedef struct crt_material {
    int kind;
    float albedo[3]; /* Lambertian: albedo; Metal: albedo; DiffuseLight: emitted colour; Dielectric: unused */
    float param;     /* Metal: fuzz (clamped to 1 like the ctor, material.h:79-81); Dielectric: ref_idx */
} crt_material;
/* table: n_materials entries; triangle_material: one table index per triangle of the list (NULL =
 * all use entry 0).  Host pointers, copied. */
int crt_scene_set_materials(crt_scene *scene, const crt_material *table, int n_materials,
                            const int *triangle_material);

/* The two things the reference selects by (un)commenting source lines, as arguments:
 *   world   CRT_WORLD_LIST = the HitableList (renderListAnimation, core/render.h:179),
 *           CRT_WORLD_BVH  = the BVHNode tree (renderBVHAnimation, core/render.h:222)
 *   shader  the call at core/render.h:119-121: CRT_SHADER_LAMBERT = LambertShade (:70-87, the live
 *           one), CRT_SHADER_SHADE = the recursive path tracer `shade` (:48-67; max_depth bounces,
 *           0 <= max_depth <= 31), CRT_SHADER_NORMAL = shade_normal (:90-103).
 * LambertShade on a single-Lambertian scene runs the specialised kernels (identical to
 * crt_render_rows / crt_render_rows_bvh); everything else runs the generic kernel, which walks the
 * whole list per ray in the list world, like the reference. */
enum { CRT_WORLD_LIST = 0, CRT_WORLD_BVH = 1 };
enum { CRT_SHADER_LAMBERT = 0, CRT_SHADER_SHADE = 1, CRT_SHADER_NORMAL = 2 };
int crt_render_rows_ex(const crt_scene *scene, const crt_camera *camera, crt_rng *rng, int nx, int ny,
                       int samples, int max_depth, int frame_index, int row_begin, int row_count,
                       int world, int shader, float *d_rgb, void *stream);

/* Launch contexts.  Everything a render launch writes besides its outputs belongs to a context keyed by (current
 * device, stream), created at the first launch on that stream and grown with the frames it sees (at 1920x1080 and
 * 100 000 triangles: about 40 MB).  A host that creates streams as it goes calls this BEFORE hipStreamDestroy (no
 * launch of this library may be in progress on the stream from another thread): the stream is synchronised and its
 * context freed.  stream = NULL releases the default stream's context and what crt_render keeps.  CRT_OK also when
 * there is nothing to release.  A host that destroys a stream WITHOUT this call leaves its context behind; when the
 * runtime later hands the same handle to a new stream, the library notices only where the process's HIP runtime has
 * hipStreamGetId (HIP >= 7.1; looked up at run time — PyTorch 2.10's bundled 7.0 runtime does not have it, there the old
 * context is simply reused, which is safe as long as the old stream had drained) and then gives the handle a fresh
 * context state.  crt_multi_free releases the contexts of its own streams. */
int crt_stream_release(void *stream);

/* The same on a strip set (crt_render_strips' layout): what a rank of a multi-GPU split calls for the BVH world and the
 * other shaders. */
int crt_render_strips_ex(const crt_scene *scene, const crt_camera *camera, crt_rng *rng, int nx, int ny,
                         int samples, int max_depth, int frame_index, int first_strip, int strip_rows,
                         int strip_step, int world, int shader, float *d_rgb, void *stream);

/* ---- multi-GPU ------------------------------------------------------------------
 * The reference is single-GPU; this section is the north-star split: strips of
 * `strip_rows` (32) image rows dealt round-robin to `world` ranks, rendered with
 * crt_render_strips, then ONE exchange step — a grouped RCCL send/recv of the float3 pieces to
 * rank 0 (rccl.h ncclSend/ncclRecv in ncclGroupStart/End) — and a reassembly kernel there. */

/* Floats between the pieces of consecutive ranks in the root's receive buffer: the largest
 * piece (rank 0's) rounded up to 4 floats.  The buffer holds world * stride floats. */
int64_t crt_strip_part_stride(int nx, int ny, int strip_rows, int world);

/* Reassembly on the root (HIP kernel on `stream`): d_parts + r * part_stride holds rank r's
 * strips one after the other (layout of crt_render_strips); d_full receives the nx x ny float3
 * frame, row 0 = bottom.  Both are device pointers on the current device. */
int crt_assemble_strips(const float *d_parts, int64_t part_stride, int nx, int ny, int strip_rows,
                        int world, float *d_full, void *stream);

/* One process (or thread) per GPU: every rank calls this after crt_render_strips with the same
 * layout, its own rank, its ncclComm_t (`nccl_comm`, created by the caller with
 * ncclCommInitRank / ncclCommInitAll) and the stream the piece was rendered on.  Rank 0 also
 * passes the receive buffer (world * part_stride floats) and d_full (nx*ny*3 floats); other
 * ranks pass NULL for both.  Enqueues: ncclGroupStart; ncclSend(piece -> 0) on every rank,
 * the root included; on rank 0 one ncclRecv per rank; ncclGroupEnd; then on rank 0 the
 * reassembly kernel.  RCCL is bound at first use (dlopen librccl.so.1): CRT_ERR_UNSUPPORTED if
 * it cannot be loaded.  An RCCL error inside the group is reported AFTER the group has been closed and the
 * stream synchronised, so the calling thread's next RCCL call starts clean. */
int crt_gather_strips(const float *d_piece, int nx, int ny, int strip_rows, int rank, int world,
                      float *d_parts, int64_t part_stride, float *d_full, void *nccl_comm,
                      void *stream);

/* The communicator crt_gather_strips needs, for hosts that run one process (or thread) per GPU and have no RCCL code of
 * their own: rank 0 calls crt_comm_unique_id and hands the CRT_COMM_ID_BYTES bytes to every rank over whatever channel
 * the host already has (bench.py: a broadcast over the torch.distributed process group torchrun set up; an MPI_Bcast; a
 * file), then EVERY rank calls crt_comm_init_rank with the device it renders on current (rccl.h: ncclGetUniqueId /
 * ncclCommInitRank; collective: it returns when all `world` ranks have called it).  crt_comm_destroy before the process
 * ends.  CRT_ERR_UNSUPPORTED when RCCL cannot be loaded. */
#define CRT_COMM_ID_BYTES 128
int crt_comm_unique_id(void *id_bytes);
int crt_comm_init_rank(int world, const void *id_bytes, int rank, void **nccl_comm_out);
int crt_comm_destroy(void *nccl_comm);

/* One process, n_gpus devices (devices 0..n_gpus-1): the scene is replicated, every device gets
 * its strip set's RNG states (seed, stream = global pixel index), a stream and a host worker
 * thread; communicators come from ncclCommInitAll.  replaces: the driver shape of
 * renderListAnimation (core/render.h:159-189) for n_gpus devices.
 * crt_multi_render / crt_multi_update_pose hand the CALLING thread's crt_set_kernel_variant and
 * crt_set_kernel_timing settings (crt_debug.h) to the workers for that job (the workers' own crt_last_kernel_* are not
 * reachable: time a multi-device frame from outside).  Every exit of crt_multi_render — success, a device that
 * failed to enqueue, an RCCL error inside the grouped exchange — leaves the NCCL group closed, all of the
 * renderer's streams idle and the caller's device current.  crt_multi_free releases the launch contexts of its
 * streams.  UNTESTED ON MORE THAN ONE DEVICE until a multi-GPU run exists: the n_gpus > 1
 * exchange has only ever executed with n_gpus = 1 and under gloo on CPU. */
typedef struct crt_multi crt_multi;
crt_multi *crt_multi_create(const float *points, int n_points, const float *idx3f, int n_tris,
                            const float *face_normals, int index_base, const float albedo[3],
                            int nx, int ny, uint64_t seed, int n_gpus);
/* One frame: every device renders its strips, one grouped send/recv, reassembly on device 0,
 * and (rgb_out != NULL) a copy of the nx*ny*3 floats to the host.  Synchronous.  The RNG
 * states persist from call to call like the reference's frame loop. */
int crt_multi_render(crt_multi *m, const crt_camera *camera, int samples, int max_depth,
                     int frame_index, float *rgb_out);
/* The reference's shipped configuration (kernel.cu:97: renderBVH) and the other shaders behind the multi-device driver:
 * world = CRT_WORLD_LIST | CRT_WORLD_BVH, shader = CRT_SHADER_*; table / triangle_material as crt_scene_set_materials
 * (table == NULL keeps the single Lambertian).  CRT_WORLD_BVH builds the tree once per replica from the same private
 * stream-0 sequence (crt_scene_build_bvh(scene, NULL, seed)): the replicas hold identical trees.  Later
 * crt_multi_render calls use this world and shader (every device: crt_render_strips_ex), crt_multi_update_pose also
 * refits the trees.  Default after crt_multi_create: the list world with LambertShade. */
int crt_multi_configure(crt_multi *m, int world, int shader, const crt_material *table, int n_materials,
                        const int *triangle_material);
/* crt_scene_update_pose on every replica (and crt_scene_refit_bvh where the world is the BVH). */
int crt_multi_update_pose(crt_multi *m, const float *new_points, int n_points);
/* Device pointer (device 0) of the frame assembled by the last crt_multi_render. */
const float *crt_multi_frame(const crt_multi *m);
int crt_multi_device_count(const crt_multi *m);
void crt_multi_free(crt_multi *m);

/* ---- output -------------------------------------------------------------------- */

/* replaces: WritePng's conversion loop (core/render.h:137-146): vertical flip,
 * (unsigned char)(255.99 * c), alpha 255.  rgb: host nx*ny*3 floats; rgba: nx*ny*4 bytes. */
void crt_to_rgba8(const float *rgb, int nx, int ny, unsigned char *rgba);

/* replaces: WritePng (core/render.h:135-157): the conversion above, then an RGBA8 PNG at
 * `path`.  The reference calls stbi_write_png(path, nx, ny, 4, rgb, 0) (:153) of the
 * stb_image_write.h it vendors; this library restates that encoder's procedure (row filter
 * choice, its LZ77 matcher, fixed Huffman codes: csrc/crt_png.cpp) and writes the SAME FILE,
 * byte for byte — pinned against the reference's encoder itself (tests/test_png.py). */
int crt_write_png(const char *path, int nx, int ny, const float *rgb);
/* The encoder alone: rgba is nx*ny*4 bytes, top row first (what crt_to_rgba8 produces). */
int crt_write_png_rgba8(const char *path, int nx, int ny, const unsigned char *rgba);
/* The same file in memory: returns its length in bytes (negative: error) and copies it to `out`
 * when it fits `capacity` (out may be NULL to ask for the length). */
long long crt_encode_png_rgba8(const unsigned char *rgba, int nx, int ny, unsigned char *out, long long capacity);

#ifdef __cplusplus
}
#endif
#endif /* CRT_H */
