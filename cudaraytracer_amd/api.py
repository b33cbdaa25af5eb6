"""Host-side mirror of the reference's host interface for the list-render path, on top of
the C ABI (include/crt.h).  Function names, argument meaning and error behaviour follow the
reference (paths relative to CudaTest/src):

    create_camera / init_camera     createScene.h:142-165
    create_FBXMesh                  createScene.h:192-210  (mesh arrays instead of an FBXObject)
    SetCurandState                  core/deviceManage.h:130-136
    renderListAnimation             core/render.h:159-189
    create_BVHfromList / Update_BVH createScene.h:212-219, 46-58
    renderBVHAnimation / renderBVH  core/render.h:191-237, kernel.cu:24-39
    writeCSV                        Loader/CSVWriter.h:8-32
    WritePng                        core/render.h:135-157

torch is used only as plumbing: it owns the device colour buffer and the stream.
Errors raise CrtError (the reference prints and exit()s, core/deviceManage.h:3-17).
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import CrtCamera, CrtError, CrtMaterial, check, check_handle

LAMBERTIAN, METAL, DIELECTRIC, DIFFUSE_LIGHT = 0, 1, 2, 3       # crt_material.kind (material/material.h:57-161)
WORLD_LIST, WORLD_BVH = 0, 1                                    # core/render.h:179 / :222
SHADER_LAMBERT, SHADER_SHADE, SHADER_NORMAL = 0, 1, 2          # core/render.h:119-121

DEFAULT_ALBEDO = (0.65, 0.05, 0.05)     # createScene.h:180


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


class Camera:
    """Camera** camera of the reference; holds a crt_camera (core/camera.h:46-52)."""

    def __init__(self, lookfrom, lookat, vup, vfov, aspect, aperture, focus_dist):
        self.c = CrtCamera()
        _lib.lib().crt_camera_make(C.byref(self.c), _f3(lookfrom), _f3(lookat), _f3(vup),
                                   float(vfov), float(aspect), float(aperture), float(focus_dist))

    def as_array(self):
        c = self.c
        return np.array(list(c.lower_left_corner) + list(c.horizontal) + list(c.vertical) +
                        list(c.origin) + list(c.x) + list(c.y) + list(c.z) + [c.lens_radius],
                        dtype=np.float32)


def create_camera(nx, ny, lookfrom, lookat, dist_to_focus, aperture, vfov):
    """create_camera kernel, createScene.h:142-154: vup (0,1,0), aspect float(nx)/float(ny)."""
    aspect = np.float32(nx) / np.float32(ny)
    return Camera(lookfrom, lookat, (0.0, 1.0, 0.0), vfov, aspect, aperture, dist_to_focus)


def init_camera(nx, ny):
    """init_camera, createScene.h:156-165: the live call site's camera."""
    return create_camera(nx, ny, (0, 100, 1000), (0, 150, 0), 10.0, 0.0, 40)


class Scene:
    """HitableList** world of the reference: a device-resident triangle list."""

    def __init__(self, handle):
        self.handle = check_handle(handle)

    @classmethod
    def from_mesh(cls, points, idx3f, normals=None, index_base=0, albedo=DEFAULT_ALBEDO):
        pts = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
        idx = np.ascontiguousarray(idx3f, dtype=np.float32).reshape(-1, 3)
        nrm = None if normals is None else np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        if nrm is not None and len(nrm) != len(idx):
            raise CrtError(_lib.CRT_ERR_INVALID, "one face normal per triangle is required")
        alb = np.array(albedo, dtype=np.float32)
        h = _lib.lib().crt_scene_from_mesh(_fp(pts), len(pts), _fp(idx), len(idx),
                                           None if nrm is None else _fp(nrm), int(index_base), _fp(alb))
        return cls(h)

    @classmethod
    def from_obj(cls, path, scale=100.0, albedo=DEFAULT_ALBEDO):
        alb = np.array(albedo, dtype=np.float32)
        return cls(_lib.lib().crt_scene_from_obj(os.fsencode(path), float(scale), _fp(alb)))

    @property
    def n_triangles(self):
        return _lib.lib().crt_scene_triangle_count(self.handle)

    def update_pose(self, new_points, stream=None):
        """updateFBXObj's device half (update_pose, createScene.h:99-110): new vertex positions
        through the same indices; stored normals are kept."""
        pts = np.ascontiguousarray(new_points, dtype=np.float32).reshape(-1, 3)
        st = C.c_void_p(stream.cuda_stream) if stream is not None else None
        check(_lib.lib().crt_scene_update_pose(self.handle, _fp(pts), len(pts), st))

    def set_materials(self, table, triangle_material=None):
        """crt_scene_set_materials.  table: [(kind, (r, g, b), param), ...]; triangle_material: one
        table index per triangle (None = all 0)."""
        arr = (CrtMaterial * len(table))()
        for k, (kind, albedo, param) in enumerate(table):
            arr[k].kind = int(kind)
            arr[k].albedo = (C.c_float * 3)(*[float(x) for x in albedo])
            arr[k].param = float(param)
        of = None
        if triangle_material is not None:
            tm = np.ascontiguousarray(triangle_material, dtype=np.int32)
            if len(tm) != self.n_triangles:
                raise CrtError(_lib.CRT_ERR_INVALID, "one material index per triangle is required")
            of = tm.ctypes.data_as(C.POINTER(C.c_int))
        check(_lib.lib().crt_scene_set_materials(self.handle, arr, len(table), of))

    def build_bvh(self, curand_state=None, seed=0):
        """crt_scene_build_bvh: BVHNode over the list (hitable/bvh.h:76-125).  curand_state: the states
        that hold pixel 0 (stream 0 supplies the split axes and is advanced), or None."""
        check(_lib.lib().crt_scene_build_bvh(self.handle, curand_state.handle if curand_state is not None else None, int(seed)))

    def refit_bvh(self, stream=None):
        """crt_scene_refit_bvh: BVHNode::UpdateBVH after update_pose."""
        sptr = None
        if stream is not None:
            sptr = stream if isinstance(stream, int) else stream.cuda_stream
        check(_lib.lib().crt_scene_refit_bvh(self.handle, C.c_void_p(sptr)))

    def bvh_export(self):
        """(boxes[n,6], meta[n,4] = skip, first slot, count, depth; slot_triangle[n_tris]) from the device."""
        n = _lib.lib().crt_scene_bvh_node_count(self.handle)
        if n < 0:
            raise CrtError(_lib.CRT_ERR_INVALID, "the scene has no BVH")
        boxes = np.zeros((n, 6), dtype=np.float32)
        meta = np.zeros((n, 4), dtype=np.int32)
        slots = np.zeros(self.n_triangles, dtype=np.int32)
        check(_lib.lib().crt_debug_bvh_export(self.handle, _fp(boxes), meta.ctypes.data_as(C.POINTER(C.c_int)),
                                              slots.ctypes.data_as(C.POINTER(C.c_int))))
        return boxes, meta, slots

    def close(self):
        if getattr(self, "handle", None):
            _lib.lib().crt_scene_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def create_FBXMesh(points, idxVertex, normals, albedo=DEFAULT_ALBEDO):
    """create_FBXMesh, createScene.h:192-210: 0-based float-triplet indices + face normals."""
    return Scene.from_mesh(points, idxVertex, normals, 0, albedo)


def parse_obj(path):
    """parseObjByName, Loader/obj_loader.h:20-111 (+ mean centring): (points, idxVertex)."""
    L = _lib.lib()
    n_p, n_t = C.c_int(), C.c_int()
    check(L.crt_parse_obj(os.fsencode(path), None, None, C.byref(n_p), C.byref(n_t)))
    pts = np.zeros((n_p.value, 3), dtype=np.float32)
    idx = np.zeros((n_t.value, 3), dtype=np.float32)
    check(L.crt_parse_obj(os.fsencode(path), _fp(pts), _fp(idx), C.byref(n_p), C.byref(n_t)))
    return pts, idx


class CurandState:
    """curandState* curand_state of the reference for the pixels [first, first+count)."""

    def __init__(self, first_pixel, count, seed=0, _handle=None):
        self.first, self.count = int(first_pixel), int(count)
        self.handle = _handle if _handle is not None else \
            check_handle(_lib.lib().crt_rng_create(self.first, self.count, int(seed)))

    @classmethod
    def for_strips(cls, nx, ny, first_strip, strip_rows, strip_step, seed=0):
        """States of the strips first_strip, first_strip+strip_step, ... (crt_rng_create_strips)."""
        L = _lib.lib()
        rows = L.crt_strip_rows(ny, first_strip, strip_rows, strip_step)
        h = check_handle(L.crt_rng_create_strips(nx, ny, first_strip, strip_rows, strip_step, int(seed)))
        return cls(first_strip * strip_rows * nx, rows * nx, seed, _handle=h)

    def reset(self, seed=0):
        check(_lib.lib().crt_rng_reset(self.handle, int(seed)))

    def export(self):
        out = np.zeros((self.count, 6), dtype=np.uint32)
        check(_lib.lib().crt_rng_export(self.handle, out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out

    def close(self):
        if getattr(self, "handle", None):
            _lib.lib().crt_rng_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def SetCurandState(nx, ny, seed=0, row_begin=0, row_count=None):
    """SetCurandState, core/deviceManage.h:130-136 (curand_init(0, pixel_index, 0)); a rank
    that owns a row band asks only for its rows."""
    rows = ny - row_begin if row_count is None else row_count
    return CurandState(row_begin * nx, rows * nx, seed)


def render_rows(scene, camera, curand_state, nx, ny, samples, row_begin, row_count, d_rgb,
                max_depth=8, frame_index=0, stream=None, bvh=False):
    """One launch of the render kernel (core/render.h:105-129) on rows
    [row_begin, row_begin+row_count).  d_rgb: torch CUDA float32 tensor with
    row_count*nx*3 elements (or a raw device pointer int).  bvh: world = the scene's BVH
    (render.h:222) instead of its triangle list (render.h:179)."""
    ptr = d_rgb if isinstance(d_rgb, int) else d_rgb.data_ptr()
    sptr = None
    if stream is not None:
        sptr = stream if isinstance(stream, int) else stream.cuda_stream
    fn = _lib.lib().crt_render_rows_bvh if bvh else _lib.lib().crt_render_rows
    check(fn(scene.handle, C.byref(camera.c), curand_state.handle, nx, ny,
                                     samples, max_depth, frame_index, row_begin, row_count,
                                     C.c_void_p(ptr), C.c_void_p(sptr)))


def render_rows_ex(scene, camera, curand_state, nx, ny, samples, row_begin, row_count, d_rgb, world=WORLD_LIST,
                   shader=SHADER_LAMBERT, max_depth=8, frame_index=0, stream=None):
    """crt_render_rows_ex: the render kernel with an explicit world (list / BVH) and shader
    (LambertShade / shade / shade_normal: the call at core/render.h:119-121)."""
    ptr = d_rgb if isinstance(d_rgb, int) else d_rgb.data_ptr()
    sptr = None
    if stream is not None:
        sptr = stream if isinstance(stream, int) else stream.cuda_stream
    check(_lib.lib().crt_render_rows_ex(scene.handle, C.byref(camera.c), curand_state.handle, nx, ny, samples, max_depth,
                                        frame_index, row_begin, row_count, int(world), int(shader), C.c_void_p(ptr),
                                        C.c_void_p(sptr)))


def strip_rows(ny, first_strip, strip_rows_, strip_step):
    return _lib.lib().crt_strip_rows(ny, first_strip, strip_rows_, strip_step)


def render_strips(scene, camera, curand_state, nx, ny, samples, first_strip, strip_rows_, strip_step,
                  d_rgb, max_depth=8, frame_index=0, stream=None, bvh=False):
    """The render kernel on the strips first_strip, first_strip+strip_step, ... of strip_rows_
    image rows each (crt_render_strips); d_rgb holds them strip after strip."""
    ptr = d_rgb if isinstance(d_rgb, int) else d_rgb.data_ptr()
    sptr = None
    if stream is not None:
        sptr = stream if isinstance(stream, int) else stream.cuda_stream
    fn = _lib.lib().crt_render_strips_bvh if bvh else _lib.lib().crt_render_strips
    check(fn(scene.handle, C.byref(camera.c), curand_state.handle, nx, ny,
                                       samples, max_depth, frame_index, first_strip, strip_rows_, strip_step,
                                       C.c_void_p(ptr), C.c_void_p(sptr)))


def render_strips_ex(scene, camera, curand_state, nx, ny, samples, first_strip, strip_rows_, strip_step, d_rgb,
                     world=WORLD_LIST, shader=SHADER_LAMBERT, max_depth=8, frame_index=0, stream=None):
    """crt_render_strips_ex: a strip set with an explicit world and shader (what a rank of a split calls)."""
    ptr = d_rgb if isinstance(d_rgb, int) else d_rgb.data_ptr()
    sptr = None
    if stream is not None:
        sptr = stream if isinstance(stream, int) else stream.cuda_stream
    check(_lib.lib().crt_render_strips_ex(scene.handle, C.byref(camera.c), curand_state.handle, nx, ny, samples, max_depth,
                                          frame_index, first_strip, strip_rows_, strip_step, int(world), int(shader),
                                          C.c_void_p(ptr), C.c_void_p(sptr)))


COMM_ID_BYTES = 128


def comm_unique_id():
    """crt_comm_unique_id: the 128 bytes rank 0 hands to every rank (ncclGetUniqueId)."""
    buf = (C.c_ubyte * COMM_ID_BYTES)()
    check(_lib.lib().crt_comm_unique_id(buf))
    return bytes(buf)


def comm_init_rank(world, id_bytes, rank):
    """crt_comm_init_rank (ncclCommInitRank; collective over the `world` ranks): the ncclComm_t as an int."""
    if len(id_bytes) != COMM_ID_BYTES:
        raise CrtError(_lib.CRT_ERR_INVALID, "the communicator id is %d bytes" % COMM_ID_BYTES)
    buf = (C.c_ubyte * COMM_ID_BYTES).from_buffer_copy(id_bytes)
    comm = C.c_void_p()
    check(_lib.lib().crt_comm_init_rank(int(world), buf, int(rank), C.byref(comm)))
    return comm.value


def comm_destroy(comm):
    if comm:
        check(_lib.lib().crt_comm_destroy(C.c_void_p(comm)))


def strip_part_stride(nx, ny, strip_rows_, world):
    return _lib.lib().crt_strip_part_stride(nx, ny, strip_rows_, world)


def assemble_strips(d_parts, part_stride, nx, ny, strip_rows_, world, d_full, stream=None):
    """crt_assemble_strips: the root's reassembly kernel (rank r's strips start at
    d_parts + r*part_stride floats; d_full receives the nx x ny float3 frame)."""
    pp = d_parts if isinstance(d_parts, int) else d_parts.data_ptr()
    pf = d_full if isinstance(d_full, int) else d_full.data_ptr()
    sptr = None
    if stream is not None:
        sptr = stream if isinstance(stream, int) else stream.cuda_stream
    check(_lib.lib().crt_assemble_strips(C.c_void_p(pp), int(part_stride), nx, ny, strip_rows_, world,
                                         C.c_void_p(pf), C.c_void_p(sptr)))


def gather_strips(d_piece, nx, ny, strip_rows_, rank, world, d_parts, part_stride, d_full, nccl_comm, stream=None):
    """crt_gather_strips: grouped RCCL send/recv of the pieces to rank 0 + reassembly there.
    nccl_comm: the caller's ncclComm_t as an integer."""
    def ptr(t):
        return None if t is None else C.c_void_p(t if isinstance(t, int) else t.data_ptr())
    sptr = None
    if stream is not None:
        sptr = stream if isinstance(stream, int) else stream.cuda_stream
    check(_lib.lib().crt_gather_strips(ptr(d_piece), nx, ny, strip_rows_, rank, world, ptr(d_parts), int(part_stride),
                                       ptr(d_full), C.c_void_p(nccl_comm), C.c_void_p(sptr)))


class MultiRenderer:
    """crt_multi_*: one process driving n_gpus devices (ncclCommInitAll, one stream and one
    host worker per device): replicated scene, strips dealt round-robin, one grouped RCCL
    send/recv, reassembly on device 0."""

    def __init__(self, points, idx3f, normals, index_base, nx, ny, n_gpus, seed=0, albedo=DEFAULT_ALBEDO):
        pts = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
        idx = np.ascontiguousarray(idx3f, dtype=np.float32).reshape(-1, 3)
        nrm = None if normals is None else np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        alb = np.array(albedo, dtype=np.float32)
        self.nx, self.ny = nx, ny
        self.handle = check_handle(_lib.lib().crt_multi_create(
            _fp(pts), len(pts), _fp(idx), len(idx), None if nrm is None else _fp(nrm), int(index_base), _fp(alb),
            nx, ny, int(seed), int(n_gpus)))

    def render(self, camera, samples, max_depth=8, frame_index=0):
        out = np.zeros((self.ny, self.nx, 3), dtype=np.float32)
        check(_lib.lib().crt_multi_render(self.handle, C.byref(camera.c), samples, max_depth, frame_index, _fp(out)))
        return out

    def configure(self, world=WORLD_LIST, shader=SHADER_LAMBERT, table=None, triangle_material=None):
        """crt_multi_configure: the world (BVH: built once per replica, identical trees) and shader of later frames."""
        arr, of = None, None
        if table is not None:
            arr = (CrtMaterial * len(table))()
            for k, (kind, albedo, param) in enumerate(table):
                arr[k].kind = int(kind)
                arr[k].albedo = (C.c_float * 3)(*[float(x) for x in albedo])
                arr[k].param = float(param)
            if triangle_material is not None:
                tm = np.ascontiguousarray(triangle_material, dtype=np.int32)
                of = tm.ctypes.data_as(C.POINTER(C.c_int))
        check(_lib.lib().crt_multi_configure(self.handle, int(world), int(shader), arr, 0 if table is None else len(table), of))

    def update_pose(self, new_points):
        pts = np.ascontiguousarray(new_points, dtype=np.float32).reshape(-1, 3)
        check(_lib.lib().crt_multi_update_pose(self.handle, _fp(pts), len(pts)))

    def close(self):
        if getattr(self, "handle", None):
            _lib.lib().crt_multi_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def stream_release(stream=None):
    """crt_stream_release: frees the launch context of `stream` (a torch.cuda.Stream; None = the default stream and what
    crt_render keeps).  Call before the stream is destroyed."""
    check(_lib.lib().crt_stream_release(C.c_void_p(stream.cuda_stream) if stream is not None else None))


def last_kernel_ms():
    ms = _lib.lib().crt_last_kernel_ms()
    if ms < 0:
        raise CrtError(_lib.CRT_ERR_HIP, _lib.lib().crt_last_error().decode())
    return ms


def set_kernel_timing(enable):
    """crt_set_kernel_timing: the per-frame HIP events behind last_kernel_times (per thread); returns the previous value."""
    return bool(_lib.lib().crt_set_kernel_timing(1 if enable else 0))


def last_kernel_times():
    """(total_ms, bin_ms, shade_ms) of the last render_rows on this thread."""
    out = (C.c_float * 3)()
    check(_lib.lib().crt_last_kernel_times(out))
    return tuple(out)


def tiled_stats(enable):
    """crt_debug_tiled_stats: returns the 48 counters collected so far, then (re)starts (1: full counters,
    2: timeline only) or stops (0)."""
    out = (C.c_uint64 * 48)()
    check(_lib.lib().crt_debug_tiled_stats(int(enable), out))
    return list(out)


def wave_timeline(max_waves=8192):
    """crt_debug_wave_timeline: (n, 4) array per wavefront: 100 MHz time stamps (start, end of the shared-tile phase, end)
    and entries << 32 | tile of the block's last shared tile."""
    import numpy as np
    out = (C.c_uint64 * (4 * max_waves))()
    n = _lib.lib().crt_debug_wave_timeline(out, max_waves)
    if n < 0:
        check(n)
    return np.ctypeslib.as_array(out)[: 4 * n].reshape(n, 4).copy()


def sample_timeline():
    """crt_debug_wave_timeline(-1): (4, 256, 2) time stamps (master's draws done, slices done) per sample of blocks 0..3."""
    import numpy as np
    out = (C.c_uint64 * 2048)()
    n = _lib.lib().crt_debug_wave_timeline(out, -1)
    if n < 0:
        check(n)
    return np.ctypeslib.as_array(out).reshape(4, 256, 2).copy()


def light_timeline():
    """crt_debug_wave_timeline(-2): (8192, 2) per wavefront: start stamp of its last light tile with geometry;
    tile | entries << 32 | such tiles taken << 48."""
    import numpy as np
    out = (C.c_uint64 * (3 * 8192))()
    n = _lib.lib().crt_debug_wave_timeline(out, -2)
    if n < 0:
        check(n)
    a = np.ctypeslib.as_array(out)
    return np.concatenate([a[: 2 * 8192].reshape(8192, 2), a[2 * 8192:].reshape(8192, 1)], axis=1).copy()


def audit_tiles(scene, camera, nx, ny, samples, seed=0):
    """crt_debug_audit_tiles: (passing pairs, pairs missing from a tile list, cull-sure violations, offender)."""
    out = (C.c_uint64 * 4)()
    check(_lib.lib().crt_debug_audit_tiles(scene.handle, C.byref(camera.c), nx, ny, samples, int(seed), out))
    return tuple(out)


def trace_rays(scene, mode, origin, direction, arith=0):
    """crt_debug_trace_rays (include/crt_debug.h): hit, t, list index, normal, point of caller-supplied rays through the
    kernels' own device functions.  origin: 3 floats (modes 0, 2) or one per ray."""
    d = np.ascontiguousarray(direction, dtype=np.float32).reshape(-1, 3)
    o = np.ascontiguousarray(origin, dtype=np.float32).reshape(-1, 3)
    if mode in (0, 2):
        if len(o) != 1:
            raise CrtError(_lib.CRT_ERR_INVALID, "modes 0 and 2 take ONE origin (the camera's)")
    elif len(o) != len(d):
        o = np.ascontiguousarray(np.broadcast_to(o, d.shape), dtype=np.float32)
    n = len(d)
    hit, idx = np.zeros(n, np.int32), np.zeros(n, np.int32)
    t, nrm, pt = np.zeros(n, np.float32), np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    ip = C.POINTER(C.c_int)
    check(_lib.lib().crt_debug_trace_rays(scene.handle, int(mode), int(arith), n, _fp(o), _fp(d), hit.ctypes.data_as(ip), _fp(t),
                                          idx.ctypes.data_as(ip), _fp(nrm), _fp(pt)))
    return hit, t, idx, nrm, pt


def box_hit(form, box6, origin, direction):
    """crt_debug_box_hit: AABB::hit(r, 0.001, FLT_MAX) in the kernels' literal (0) or bounded-scene (1) form."""
    b = np.ascontiguousarray(box6, dtype=np.float32).reshape(-1, 6)
    o = np.ascontiguousarray(origin, dtype=np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(direction, dtype=np.float32).reshape(-1, 3)
    out = np.zeros(len(b), np.int32)
    check(_lib.lib().crt_debug_box_hit(int(form), len(b), _fp(b), _fp(o), _fp(d), out.ctypes.data_as(C.POINTER(C.c_int))))
    return out


def box_op(where, op, a6, b6):
    """crt_debug_box_op: surrounding_box (op 0) / TransformAABB with the identity Transform (op 1); where 0 = host build, 1 = refit kernel."""
    a = np.ascontiguousarray(a6, dtype=np.float32).reshape(-1, 6)
    b = np.ascontiguousarray(b6, dtype=np.float32).reshape(-1, 6)
    out = np.zeros_like(a)
    check(_lib.lib().crt_debug_box_op(int(where), int(op), len(a), _fp(a), _fp(b), _fp(out)))
    return out


def set_kernel_variant(v):
    return _lib.lib().crt_set_kernel_variant(int(v))


def render(nx, ny, samples, scene, camera, seed=0, max_depth=8, frame_index=0, out=None):
    """One frame through crt_render: float32 array [ny, nx, 3], row 0 = image bottom.
    out: the caller's host buffer of ny*nx*3 floats to render into — a numpy array or a CPU torch tensor; a PAGE-LOCKED one
    (torch.empty(..., pin_memory=True), hipHostMalloc) takes the frame at the link's rate, a pageable one through the
    runtime's staging."""
    if out is None:
        out = np.zeros((ny, nx, 3), dtype=np.float32)
        ptr, ret = _fp(out), out
    elif isinstance(out, np.ndarray):
        if out.dtype != np.float32 or out.size != ny * nx * 3 or not out.flags["C_CONTIGUOUS"]:
            raise ValueError("render: out must be a C-contiguous float32 array of ny*nx*3 elements")
        ptr, ret = _fp(out), out.reshape(ny, nx, 3)
    else:   # a CPU torch tensor
        import torch
        if out.device.type != "cpu" or out.dtype != torch.float32 or out.numel() != ny * nx * 3 or not out.is_contiguous():
            raise ValueError("render: out must be a contiguous float32 CPU tensor of ny*nx*3 elements")
        ptr, ret = C.cast(C.c_void_p(out.data_ptr()), C.POINTER(C.c_float)), out.numpy().reshape(ny, nx, 3)
    check(_lib.lib().crt_render(nx, ny, samples, max_depth, frame_index, scene.handle,
                                C.byref(camera.c), int(seed), ptr))
    return ret


def to_rgba8(rgb):
    ny, nx, _ = rgb.shape
    b = np.ascontiguousarray(rgb, dtype=np.float32)
    out = np.zeros((ny, nx, 4), dtype=np.uint8)
    _lib.lib().crt_to_rgba8(_fp(b), nx, ny, out.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return out


def WritePng(nx, ny, frameIndex, colorBuffer, folder="images/moveTest"):
    """WritePng, core/render.h:135-157: images/moveTest/picture_<frame>.png (the folder is
    created here; the reference requires it to exist)."""
    os.makedirs(folder, exist_ok=True)
    path = os.path.join(folder, "picture_%d.png" % frameIndex)
    b = np.ascontiguousarray(colorBuffer, dtype=np.float32).reshape(ny, nx, 3)
    check(_lib.lib().crt_write_png(os.fsencode(path), nx, ny, _fp(b)))
    return path


def _pinned_frame(nx, ny):
    """The frame loops' host buffer (the reference's `colorBuffer`, a malloc): page-locked, so the copy back runs at the
    link's rate behind the render kernel on its stream instead of through the runtime's staging of a pageable copy."""
    import torch
    return torch.empty(ny * nx * 3, dtype=torch.float32, pin_memory=True)


def _copy_back(h_colorBuffer, d_colorBuffer, nx, ny):
    """cudaDeviceSynchronize + cudaMemcpy(colorBuffer, d_colorBuffer) of the frame loops (render.h:180-183,222-225): the copy
    enqueued behind the frame's kernels, ONE synchronisation; returns the host buffer as a [ny, nx, 3] array (the next frame
    overwrites it, as the reference's does: keep a copy to keep the frame)."""
    import torch
    h_colorBuffer.copy_(d_colorBuffer, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    return h_colorBuffer.numpy().reshape(ny, nx, 3)


def renderListAnimation(nx, ny, samples, max_depth, beginFrame, endFrame, world, camera,
                        curand_state, write_png=True, folder="images/moveTest", poses=None):
    """renderListAnimation, core/render.h:159-189: per frame [update the mesh pose,] launch the
    render kernel on the whole image, synchronise, copy back, WritePng.  The RNG states persist
    across frames as in the reference.  `poses` (optional) maps a frame index to that frame's
    point positions — the output of the reference's calcPose, which needs the FBX SDK; the
    device half of updateFBXObj is Scene.update_pose.  Returns the list of frames."""
    import torch
    if not torch.cuda.is_available():
        raise CrtError(_lib.CRT_ERR_NO_DEVICE, "renderListAnimation: no HIP device (no CPU fallback)")
    d_colorBuffer = torch.zeros(ny * nx * 3, dtype=torch.float32, device="cuda")
    h_colorBuffer = _pinned_frame(nx, ny)
    frames = []
    for frameIndex in range(beginFrame, endFrame + 1):
        if poses is not None:                                   # updateFBXObj(frameIndex, obj, ...), render.h:177
            world.update_pose(poses[frameIndex], stream=torch.cuda.current_stream())
        render_rows(world, camera, curand_state, nx, ny, samples, 0, ny, d_colorBuffer,
                    max_depth=max_depth, frame_index=frameIndex,
                    stream=torch.cuda.current_stream())
        colorBuffer = _copy_back(h_colorBuffer, d_colorBuffer, nx, ny)
        if write_png:
            WritePng(nx, ny, frameIndex, colorBuffer, folder)
        frames.append(colorBuffer.copy())
    return frames


def create_BVHfromList(world, curand_state=None, seed=0):
    """create_BVHfromList, createScene.h:212-219: the BVH becomes part of `world` (the reference
    returns a BVHNode** that refers to the list's triangles)."""
    world.build_bvh(curand_state, seed)
    return world


def Update_BVH(world, stream=None):
    """Update_BVH(BVHNode**), createScene.h:52-58."""
    world.refit_bvh(stream)


def writeCSV(filename, data):
    """writeCSV, Loader/CSVWriter.h:8-32, for the rows main()/renderBVH/renderBVHAnimation collect:
    data[0] = the header, data[1] = ["", "", "", build], then [frame, rendering, update, ""]."""
    frames = [r for r in data[2:]]
    n = len(frames)
    fr = (C.c_int * max(n, 1))(*[int(r[0]) for r in frames])
    ren = (C.c_double * max(n, 1))(*[float(r[1]) for r in frames])
    upd = (C.c_double * max(n, 1))(*[float(r[2]) for r in frames])
    check(_lib.lib().crt_write_timing_csv(os.fsencode(filename), n, fr, ren, upd, float(data[1][3])))


def renderBVHAnimation(nx, ny, samples, max_depth, beginFrame, endFrame, world, camera, curand_state, data,
                       write_png=True, folder="images/moveTest", poses=None):
    """renderBVHAnimation, core/render.h:191-237: per frame [update the mesh pose,] refit the BVH
    (timed: "update"), launch the render kernel with world = the BVH, synchronise, copy back
    (timed: "rendering"), WritePng.  Appends [frame, rendering, update, ""] rows to `data`."""
    import time
    import torch
    if not torch.cuda.is_available():
        raise CrtError(_lib.CRT_ERR_NO_DEVICE, "renderBVHAnimation: no HIP device (no CPU fallback)")
    d_colorBuffer = torch.zeros(ny * nx * 3, dtype=torch.float32, device="cuda")
    h_colorBuffer = _pinned_frame(nx, ny)
    stream = torch.cuda.current_stream()
    frames = []
    for frameIndex in range(beginFrame, endFrame + 1):
        if poses is not None:                                   # updateFBXObj(frameIndex, obj, ...), render.h:210
            world.update_pose(poses[frameIndex], stream=stream)
        t0 = time.perf_counter()
        Update_BVH(world, stream)                               # render.h:215
        torch.cuda.synchronize()
        update_time = time.perf_counter() - t0
        t0 = time.perf_counter()
        render_rows(world, camera, curand_state, nx, ny, samples, 0, ny, d_colorBuffer, max_depth=max_depth,
                    frame_index=frameIndex, stream=stream, bvh=True)
        colorBuffer = _copy_back(h_colorBuffer, d_colorBuffer, nx, ny)
        render_time = time.perf_counter() - t0
        data.append([str(frameIndex), "%f" % render_time, "%f" % update_time, ""])
        if write_png:
            WritePng(nx, ny, frameIndex, colorBuffer, folder)
        frames.append(colorBuffer.copy())
    return frames


def renderBVH(nx, ny, samples, max_depth, beginFrame, endFrame, fbxList, camera, curand_state, data, **kw):
    """renderBVH, kernel.cu:24-39: timed BVH build, then renderBVHAnimation."""
    import time
    t0 = time.perf_counter()
    create_BVHfromList(fbxList, curand_state)
    data.append(["", "", "", "%f" % (time.perf_counter() - t0)])
    return renderBVHAnimation(nx, ny, samples, max_depth, beginFrame, endFrame, fbxList, camera, curand_state, data, **kw)
