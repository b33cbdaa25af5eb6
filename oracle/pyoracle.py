"""ctypes binding of the CPU oracle (oracle/libcrt_oracle.so).

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by the product package.  PARITY UNPINNED (see crt_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcrt_oracle.so")


def build(force=False):
    """Compile the oracle with the committed Makefile (gcc, -ffp-contract=off)."""
    src = [os.path.join(_HERE, f) for f in ("crt_oracle.c", "crt_oracle.h", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "libcrt_oracle.so"])
    return _LIB_PATH


class CurandState(C.Structure):
    _fields_ = [("d", C.c_uint), ("v", C.c_uint * 5)]


class Camera(C.Structure):
    _fields_ = [(n, C.c_float * 3) for n in
                ("lower_left_corner", "horizontal", "vertical", "origin", "x", "y", "z")] + \
               [("lens_radius", C.c_float)]


class Material(C.Structure):
    """o_material (crt_oracle.h)."""
    _fields_ = [("kind", C.c_int), ("albedo", C.c_float * 3), ("param", C.c_float)]


LAMBERTIAN, METAL, DIELECTRIC, DIFFUSE_LIGHT = 0, 1, 2, 3
SHADER_LAMBERT, SHADER_SHADE, SHADER_NORMAL = 0, 1, 2

STATE_DTYPE = np.dtype([("d", np.uint32), ("v", np.uint32, (5,))])

# o_bvh_node (crt_oracle.h): 2 transforms (9 floats each), box (6 floats), 8 ints
BVH_NODE_DTYPE = np.dtype([("transform", np.float32, (9,)), ("list_transform", np.float32, (9,)),
                           ("box", np.float32, (6,)), ("isLeaf", np.int32), ("left", np.int32), ("right", np.int32),
                           ("first", np.int32), ("count", np.int32), ("depth", np.int32), ("axis", np.int32),
                           ("skip", np.int32)])

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp = C.POINTER(C.c_float)
        L.o_curand_init.argtypes = [C.c_ulonglong, C.c_ulonglong, C.c_ulonglong, C.POINTER(CurandState)]
        L.o_curand.argtypes = [C.POINTER(CurandState)]
        L.o_curand.restype = C.c_uint
        L.o_curand_uniform.argtypes = [C.POINTER(CurandState)]
        L.o_curand_uniform.restype = C.c_float
        L.o_set_curand_state_range.argtypes = [C.c_void_p, C.c_longlong, C.c_longlong, C.c_ulonglong]
        L.o_camera_init.argtypes = [C.POINTER(Camera), fp, fp, fp, C.c_float, C.c_float, C.c_float, C.c_float]
        L.o_world_from_mesh.argtypes = [fp, C.c_int, fp, C.c_int, fp, C.c_int, fp]
        L.o_world_from_mesh.restype = C.c_void_p
        L.o_world_update_pose.argtypes = [C.c_void_p, fp, C.c_int, fp, C.c_int]
        L.o_world_update_pose.restype = C.c_int
        L.o_world_free.argtypes = [C.c_void_p]
        L.o_parse_obj.argtypes = [C.c_char_p, fp, fp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.o_render_window.argtypes = [fp, C.c_void_p, C.POINTER(Camera), C.c_void_p] + [C.c_int] * 9
        L.o_world_hit.argtypes = [C.c_void_p, fp, fp, fp, C.POINTER(C.c_int), fp]
        L.o_world_hit.restype = C.c_int
        L.o_to_rgba8.argtypes = [fp, C.c_int, C.c_int, C.POINTER(C.c_ubyte)]
        L.o_vec3_op.argtypes = [C.c_int, C.c_longlong, fp, fp, fp, fp]
        L.o_vec3_op.restype = C.c_int
        L.o_bvh_build.argtypes = [C.c_void_p, C.c_void_p]
        L.o_bvh_build.restype = C.c_void_p
        L.o_bvh_update.argtypes = [C.c_void_p]
        L.o_bvh_free.argtypes = [C.c_void_p]
        L.o_bvh_node_count.argtypes = [C.c_void_p]
        L.o_bvh_nodes.argtypes = [C.c_void_p]
        L.o_bvh_nodes.restype = C.c_void_p
        L.o_bvh_order.argtypes = [C.c_void_p]
        L.o_bvh_order.restype = C.c_void_p
        L.o_bvh_hit.argtypes = [C.c_void_p, fp, fp, fp, C.POINTER(C.c_int), fp]
        L.o_bvh_hit.restype = C.c_int
        L.o_world_set_materials.argtypes = [C.c_void_p, C.POINTER(Material), C.c_int, C.POINTER(C.c_int)]
        L.o_world_set_materials.restype = C.c_int
        L.o_render_window_ex.argtypes = [fp, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(Camera), C.c_void_p] + [C.c_int] * 9
        L.o_render_window_bvh.argtypes = [fp, C.c_void_p, C.POINTER(Camera), C.c_void_p] + [C.c_int] * 9
        ip = C.POINTER(C.c_int)
        L.o_thrust_sort.argtypes = [C.c_int, fp, ip]
        L.o_thrust_sort.restype = C.c_int
        L.o_world_hit_many.argtypes = [C.c_void_p, C.c_int, C.c_longlong, fp, fp, ip, fp, ip, fp, fp]
        L.o_bvh_hit_many.argtypes = [C.c_void_p, C.c_longlong, fp, fp, ip, fp, ip, fp, fp]
        L.o_aabb_hit_many.argtypes = [C.c_longlong, fp, fp, fp, fp, fp, fp, ip]
        L.o_aabb_op_many.argtypes = [C.c_int, C.c_longlong, fp, fp, fp, fp, fp, fp]
        L.o_aabb_op_many.restype = C.c_int
        L.o_world_bounding_boxes.argtypes = [C.c_void_p, C.c_int, fp, fp]
        L.o_world_bounding_boxes.restype = C.c_int
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def curand_stream(seed, subsequence, offset, n):
    """First n curand() u32 outputs and curand_uniform floats of (seed, subsequence, offset)."""
    L = lib()
    s = CurandState()
    L.o_curand_init(seed, subsequence, offset, C.byref(s))
    s2 = CurandState.from_buffer_copy(s)
    u32 = np.array([L.o_curand(C.byref(s)) for _ in range(n)], dtype=np.uint32)
    f32 = np.array([L.o_curand_uniform(C.byref(s2)) for _ in range(n)], dtype=np.float32)
    return u32, f32


def xorwow_skip_raw(x, d, subsequence, offset):
    """Raw states (x: n x 5 u32, d: n u32) advanced by subsequence * 2^67 + offset steps: o_xorwow_skip_raw per state."""
    L = lib()
    L.o_xorwow_skip_raw.argtypes = [C.POINTER(CurandState), C.c_ulonglong, C.c_ulonglong]
    L.o_xorwow_skip_raw.restype = None
    x = np.array(x, dtype=np.uint32).reshape(-1, 5)
    d = np.array(d, dtype=np.uint32).reshape(-1)
    for i in range(len(d)):
        s = CurandState()
        s.d = int(d[i])
        for k in range(5):
            s.v[k] = int(x[i, k])
        L.o_xorwow_skip_raw(C.byref(s), int(subsequence[i]), int(offset[i]))
        d[i] = s.d
        x[i] = [s.v[k] for k in range(5)]
    return x, d


def xorwow_next_raw(x, d, count):
    """One raw state stepped `count` times with o_curand: (outputs, x, d)."""
    L = lib()
    s = CurandState()
    s.d = int(d)
    for k in range(5):
        s.v[k] = int(x[k])
    out = np.array([L.o_curand(C.byref(s)) for _ in range(count)], dtype=np.uint32)
    return out, np.array([s.v[k] for k in range(5)], dtype=np.uint32), np.uint32(s.d)


def curand_states(first, count, seed=0):
    st = np.zeros(count, dtype=STATE_DTYPE)
    lib().o_set_curand_state_range(st.ctypes.data, first, count, seed)
    return st


def make_camera(lookfrom, lookat, vup, vfov, aspect, aperture, focus_dist):
    cam = Camera()
    lib().o_camera_init(C.byref(cam), _f3(lookfrom), _f3(lookat), _f3(vup),
                        vfov, aspect, aperture, focus_dist)
    return cam


def camera_as_array(cam):
    return np.array(list(cam.lower_left_corner) + list(cam.horizontal) + list(cam.vertical) +
                    list(cam.origin) + list(cam.x) + list(cam.y) + list(cam.z) + [cam.lens_radius],
                    dtype=np.float32)


class World:
    def __init__(self, points, idx3f, normals=None, index_base=0, albedo=(0.65, 0.05, 0.05)):
        self.points = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
        self.idx3f = np.ascontiguousarray(idx3f, dtype=np.float32).reshape(-1, 3)
        self.normals = None if normals is None else np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        alb = np.array(albedo, dtype=np.float32)
        self.handle = lib().o_world_from_mesh(
            _fp(self.points), len(self.points), _fp(self.idx3f), len(self.idx3f),
            None if self.normals is None else _fp(self.normals), index_base, _fp(alb))
        if not self.handle:
            raise ValueError("o_world_from_mesh: vertex index out of range")
        self.n_tris = len(self.idx3f)
        self.index_base = index_base

    def set_materials(self, table, triangle_material=None):
        """table: [(kind, (r, g, b), param), ...]; triangle_material: one index per triangle or None."""
        arr = (Material * len(table))()
        for k, (kind, albedo, param) in enumerate(table):
            arr[k].kind = int(kind)
            arr[k].albedo = (C.c_float * 3)(*[float(x) for x in albedo])
            arr[k].param = float(param)
        of = None
        if triangle_material is not None:
            tm = np.ascontiguousarray(triangle_material, dtype=np.int32)
            of = tm.ctypes.data_as(C.POINTER(C.c_int))
        if lib().o_world_set_materials(self.handle, arr, len(table), of) != 0:
            raise ValueError("o_world_set_materials: bad table or index")

    def update_pose(self, new_points):
        """update_pose (createScene.h:99-110): new vertex positions, stored normals kept."""
        pts = np.ascontiguousarray(new_points, dtype=np.float32).reshape(-1, 3)
        if lib().o_world_update_pose(self.handle, _fp(pts), len(pts), _fp(self.idx3f), self.index_base) != 0:
            raise ValueError("o_world_update_pose: vertex index out of range")
        self.points = pts

    def __del__(self):
        if getattr(self, "handle", None) and lib is not None:   # (module globals are gone at interpreter shutdown)
            lib().o_world_free(self.handle)
            self.handle = None

    def hit(self, origin, direction):
        t = C.c_float()
        idx = C.c_int()
        nrm = (C.c_float * 3)()
        flag = lib().o_world_hit(self.handle, _f3(origin), _f3(direction), C.byref(t), C.byref(idx), nrm)
        return bool(flag), np.float32(t.value), idx.value, np.array(list(nrm), dtype=np.float32)


class Bvh:
    """create_BVHfromList (createScene.h:212-219): BVHNode over the world's triangle list.
    `states` is the curandState array whose element 0 (stream 0) the build draws its split
    axes from; it is advanced in place."""

    def __init__(self, world, states):
        self.world = world          # keeps the list alive: the tree references its triangles
        self.handle = lib().o_bvh_build(world.handle, states.ctypes.data)
        if not self.handle:
            raise ValueError("o_bvh_build: empty list")

    def update(self):
        """Update_BVH (createScene.h:46-58) after world.update_pose."""
        lib().o_bvh_update(self.handle)

    def nodes(self):
        n = lib().o_bvh_node_count(self.handle)
        buf = (C.c_char * (n * BVH_NODE_DTYPE.itemsize)).from_address(lib().o_bvh_nodes(self.handle))
        return np.frombuffer(buf, dtype=BVH_NODE_DTYPE, count=n).copy()

    def order(self):
        n = self.world.n_tris
        buf = (C.c_int * n).from_address(lib().o_bvh_order(self.handle))
        return np.frombuffer(buf, dtype=np.int32, count=n).copy()

    def hit(self, origin, direction):
        t = C.c_float()
        idx = C.c_int()
        nrm = (C.c_float * 3)()
        flag = lib().o_bvh_hit(self.handle, _f3(origin), _f3(direction), C.byref(t), C.byref(idx), nrm)
        return bool(flag), np.float32(t.value), idx.value, np.array(list(nrm), dtype=np.float32)

    def __del__(self):
        if getattr(self, "handle", None) and lib is not None:
            lib().o_bvh_free(self.handle)
            self.handle = None


def fresh_states(nx, ny, window=None, seed=0):
    """curand states of the pixels a (windowed) render touches: stream 0 plus the window's rows."""
    L = lib()
    r0, r1, c0, c1 = window if window is not None else (0, ny, 0, nx)
    states = np.zeros(nx * ny, dtype=STATE_DTYPE)
    L.o_set_curand_state_range(states.ctypes.data, 0, 1, seed)
    for y in range(r0, r1):
        first = y * nx + c0
        L.o_set_curand_state_range(states[first:].ctypes.data, first, c1 - c0, seed)
    return states


def render_bvh(bvh, cam, nx, ny, samples, states, window=None, max_depth=8, frame=0):
    """The render kernel with world = the BVH (renderBVHAnimation, core/render.h:222).  `states`
    must be the array the BVH was built with (the build has advanced stream 0)."""
    if window is None:
        window = (0, ny, 0, nx)
    r0, r1, c0, c1 = window
    buf = np.zeros((ny, nx, 3), dtype=np.float32)
    lib().o_render_window_bvh(_fp(buf), bvh.handle, C.byref(cam), states.ctypes.data,
                              nx, ny, samples, max_depth, frame, r0, r1, c0, c1)
    return buf, states


def render_ex(world, bvh, shader, cam, nx, ny, samples, states, window=None, max_depth=8, frame=0):
    """The render kernel with an explicit world (bvh None = the list) and shader (render.h:119-121)."""
    if window is None:
        window = (0, ny, 0, nx)
    r0, r1, c0, c1 = window
    buf = np.zeros((ny, nx, 3), dtype=np.float32)
    lib().o_render_window_ex(_fp(buf), world.handle, bvh.handle if bvh is not None else None, int(shader), C.byref(cam),
                             states.ctypes.data, nx, ny, samples, max_depth, frame, r0, r1, c0, c1)
    return buf, states


def parse_obj(path):
    L = lib()
    npnt, ntri = C.c_int(), C.c_int()
    if L.o_parse_obj(path.encode(), None, None, C.byref(npnt), C.byref(ntri)) != 0:
        raise FileNotFoundError(path)
    pts = np.zeros((npnt.value, 3), dtype=np.float32)
    idx = np.zeros((ntri.value, 3), dtype=np.float32)
    L.o_parse_obj(path.encode(), _fp(pts), _fp(idx), C.byref(npnt), C.byref(ntri))
    return pts, idx


def render(world, cam, nx, ny, samples, seed=0, window=None, states=None, max_depth=8, frame=0):
    """Serial reference render. window=(row_begin,row_end,col_begin,col_end) renders a crop;
    RNG states are created for exactly the pixels needed (stream = global pixel index).
    Returns (colorBuffer[ny,nx,3] float32 with zeros outside the window, states)."""
    L = lib()
    if window is None:
        window = (0, ny, 0, nx)
    r0, r1, c0, c1 = window
    buf = np.zeros((ny, nx, 3), dtype=np.float32)
    if states is None:
        states = np.zeros(nx * ny, dtype=STATE_DTYPE)
        # stream 0 is always needed (get_ray draws from &state[0], core/render.h:118)
        L.o_set_curand_state_range(states.ctypes.data, 0, 1, seed)
        for y in range(r0, r1):
            first = y * nx + c0
            L.o_set_curand_state_range(states[first:].ctypes.data, first, c1 - c0, seed)
    L.o_render_window(_fp(buf), world.handle, C.byref(cam), states.ctypes.data,
                      nx, ny, samples, max_depth, frame, r0, r1, c0, c1)
    return buf, states


def to_rgba8(buf):
    ny, nx, _ = buf.shape
    out = np.zeros((ny, nx, 4), dtype=np.uint8)
    b = np.ascontiguousarray(buf, dtype=np.float32)
    lib().o_to_rgba8(_fp(b), nx, ny, out.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return out


def vec3_op(op, a, b, s):
    """o_vec3_op: one core/vec3.h operation on arrays (op codes of oracle/ref_vec3.cpp); None if the oracle does not
    restate that operation."""
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 3)
    b = np.ascontiguousarray(b, dtype=np.float32).reshape(-1, 3)
    s = np.ascontiguousarray(s, dtype=np.float32).reshape(-1)
    out = np.zeros_like(a)
    rc = lib().o_vec3_op(int(op), len(a), _fp(a), _fp(b), _fp(s), _fp(out))
    return out if rc == 0 else None


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _rays(origin, direction):
    d = np.ascontiguousarray(direction, dtype=np.float32).reshape(-1, 3)
    o = np.ascontiguousarray(np.broadcast_to(np.asarray(origin, dtype=np.float32).reshape(-1, 3), d.shape), dtype=np.float32)
    return o, d


def _hit_outputs(n):
    return (np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros(n, np.int32), np.zeros((n, 3), np.float32),
            np.zeros((n, 3), np.float32))


def world_hit_many(world, origin, direction, levels=0):
    """(*world)->hit(r, 0.001, FLT_MAX, rec) for many rays (core/render.h:75); `levels` extra Hitable::hit wrappers.
    Returns hit, t, index, normal, p."""
    o, d = _rays(origin, direction)
    h, t, i, nrm, p = _hit_outputs(len(d))
    lib().o_world_hit_many(world.handle, int(levels), len(d), _fp(o), _fp(d), _ip(h), _fp(t), _ip(i), _fp(nrm), _fp(p))
    return h, t, i, nrm, p


def bvh_hit_many(bvh, origin, direction):
    o, d = _rays(origin, direction)
    h, t, i, nrm, p = _hit_outputs(len(d))
    lib().o_bvh_hit_many(bvh.handle, len(d), _fp(o), _fp(d), _ip(h), _fp(t), _ip(i), _fp(nrm), _fp(p))
    return h, t, i, nrm, p


def aabb_hit_many(bmin, bmax, origin, direction, t_min, t_max):
    a = [np.ascontiguousarray(x, dtype=np.float32) for x in (bmin, bmax, origin, direction, t_min, t_max)]
    out = np.zeros(len(a[4]), np.int32)
    lib().o_aabb_hit_many(len(out), *[_fp(x) for x in a], _ip(out))
    return out


def aabb_op_many(op, amin, amax, bmin, bmax):
    a = [np.ascontiguousarray(x, dtype=np.float32).reshape(-1, 3) for x in (amin, amax, bmin, bmax)]
    omin, omax = np.zeros_like(a[0]), np.zeros_like(a[0])
    if lib().o_aabb_op_many(int(op), len(omin), *[_fp(x) for x in a], _fp(omin), _fp(omax)) != 0:
        raise ValueError("o_aabb_op_many: bad op")
    return omin, omax


def world_bounding_boxes(world, which):
    n = world.n_tris if which == 0 else 1
    omin, omax = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    flag = lib().o_world_bounding_boxes(world.handle, int(which), _fp(omin), _fp(omax))
    return flag, omin, omax


def thrust_sort(key, order=None):
    """The oracle's restatement of thrust::sort's sequential backend with BoxCompare's rule (hitable/bvh.h:9-45,86-92)."""
    key = np.ascontiguousarray(key, dtype=np.float32)
    order = np.arange(len(key), dtype=np.int32) if order is None else np.ascontiguousarray(order, dtype=np.int32).copy()
    if lib().o_thrust_sort(len(order), _fp(key), _ip(order)) != 0:
        raise MemoryError("o_thrust_sort")
    return order
