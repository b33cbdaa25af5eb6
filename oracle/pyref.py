"""ctypes bindings of oracle/_ref/libref_hit.so and libref_thrust_sort.so — pieces of the REFERENCE ITSELF (its own
hitable / triangle / aabb headers compiled for gfx950, oracle/ref_hit.hip) and of the Thrust it depends on (rocThrust's
thrust::sort, oracle/ref_thrust_sort.hip).

TEST INFRASTRUCTURE ONLY — imported by tests/ (and the fixture generators there), never by the product package.
Built by `make -C oracle ref` in the container that holds /root/reference; the GPU box uses the built files.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
HIT_PATH = os.path.join(_HERE, "_ref", "libref_hit.so")
THRUST_PATH = os.path.join(_HERE, "_ref", "libref_thrust_sort.so")
XORWOW_PATH = os.path.join(_HERE, "_ref", "libref_rocrand_xorwow.so")   # rocRAND's XORWOW engine on raw states (host code, no HIP runtime)

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)
_hit = None
_thrust = None
_xorwow = None


def _load(path):
    # one HIP runtime per process: torch's bundled copy first (see cudaraytracer_amd/_lib.py)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    return C.CDLL(path)


def have_hit():
    return os.path.exists(HIT_PATH)


def have_thrust():
    return os.path.exists(THRUST_PATH)


def hit_lib():
    global _hit
    if _hit is None:
        L = _load(HIT_PATH)
        L.ref_hit_scene_create.argtypes = [_fp, C.c_int, _fp, C.c_int, _fp, C.c_int, C.c_int]
        L.ref_hit_scene_create.restype = C.c_void_p
        L.ref_hit_scene_free.argtypes = [C.c_void_p]
        L.ref_hit_scene_set_vertices.argtypes = [C.c_void_p, _fp, C.c_int, _fp, C.c_int]
        L.ref_hit_rays.argtypes = [C.c_void_p, C.c_int, C.c_int, _fp, _fp, _ip, _fp, _ip, _fp, _fp]
        L.ref_aabb_hit.argtypes = [C.c_int, _fp, _fp, _fp, _fp, _fp, _fp, _ip]
        L.ref_aabb_op.argtypes = [C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp, _fp]
        L.ref_hit_bounding_boxes.argtypes = [C.c_void_p, C.c_int, _fp, _fp, _ip]
        _hit = L
    return _hit


def thrust_lib():
    global _thrust
    if _thrust is None:
        L = _load(THRUST_PATH)
        L.ref_thrust_sort_host.argtypes = [C.c_int, _fp, _ip]
        L.ref_thrust_sort_device.argtypes = [C.c_int, C.c_int, _fp, _ip]
        _thrust = L
    return _thrust


def have_xorwow():
    return os.path.exists(XORWOW_PATH)


def xorwow_lib():
    global _xorwow
    if _xorwow is None:
        L = C.CDLL(XORWOW_PATH)
        _up, _ullp = C.POINTER(C.c_uint), C.POINTER(C.c_ulonglong)
        L.ref_xorwow_skip.argtypes = [C.c_longlong, _up, _up, _ullp, _ullp]
        L.ref_xorwow_next.argtypes = [_up, _up, C.c_longlong, _up]
        _xorwow = L
    return _xorwow


def rocrand_xorwow_skip(x, d, subsequence, offset):
    """rocRAND's xorwow_engine on raw states: x (n x 5 u32), d (n u32) advanced by subsequence * 2^67, then offset steps."""
    x = np.array(x, dtype=np.uint32).reshape(-1, 5)
    d = np.array(d, dtype=np.uint32).reshape(-1)
    sub = np.ascontiguousarray(subsequence, dtype=np.uint64)
    off = np.ascontiguousarray(offset, dtype=np.uint64)
    assert len(sub) == len(off) == len(d) == len(x)
    up, ullp = C.POINTER(C.c_uint), C.POINTER(C.c_ulonglong)
    rc = xorwow_lib().ref_xorwow_skip(len(d), x.ctypes.data_as(up), d.ctypes.data_as(up), sub.ctypes.data_as(ullp), off.ctypes.data_as(ullp))
    assert rc == 0
    return x, d


def rocrand_xorwow_next(x, d, count):
    """rocRAND's xorwow_engine::next() `count` times from one raw state: (outputs, x, d)."""
    x = np.array(x, dtype=np.uint32).reshape(5)
    d = np.array([d], dtype=np.uint32)
    out = np.zeros(count, dtype=np.uint32)
    up = C.POINTER(C.c_uint)
    rc = xorwow_lib().ref_xorwow_next(x.ctypes.data_as(up), d.ctypes.data_as(up), count, out.ctypes.data_as(up))
    assert rc == 0
    return out, x, d[0]


def _f(a):
    return a.ctypes.data_as(_fp)


def _i(a):
    return a.ctypes.data_as(_ip)


class RefScene:
    """The reference's HitableList of Triangles, built on the device as add_mesh_withNormal builds it
    (createScene.h:175-190); `levels` extra one-element HitableList wrappers for the nested form."""

    def __init__(self, points, idx3f, normals=None, index_base=0, levels=0):
        self.points = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
        self.idx3f = np.ascontiguousarray(idx3f, dtype=np.float32).reshape(-1, 3)
        self.normals = None if normals is None else np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        self.index_base = index_base
        self.n_tris = len(self.idx3f)
        self.handle = hit_lib().ref_hit_scene_create(_f(self.points), len(self.points), _f(self.idx3f), self.n_tris,
                                                     None if self.normals is None else _f(self.normals), index_base, levels)
        if not self.handle:
            raise RuntimeError("ref_hit_scene_create failed (no GPU, device heap exhausted or bad indices)")

    def set_vertices(self, new_points):
        pts = np.ascontiguousarray(new_points, dtype=np.float32).reshape(-1, 3)
        rc = hit_lib().ref_hit_scene_set_vertices(self.handle, _f(pts), len(pts), _f(self.idx3f), self.index_base)
        if rc != 0:
            raise RuntimeError("ref_hit_scene_set_vertices: %d" % rc)

    def hit(self, origin, direction, nested=False):
        """hit, t, index, normal, p of (*world)->hit(r, 0.001, FLT_MAX, rec, 0) per ray."""
        d = np.ascontiguousarray(direction, dtype=np.float32).reshape(-1, 3)
        o = np.ascontiguousarray(np.broadcast_to(np.asarray(origin, dtype=np.float32).reshape(-1, 3), d.shape), dtype=np.float32)
        n = len(d)
        h, t, idx = np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros(n, np.int32)
        nrm, p = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
        rc = hit_lib().ref_hit_rays(self.handle, 1 if nested else 0, n, _f(o), _f(d), _i(h), _f(t), _i(idx), _f(nrm), _f(p))
        if rc != 0:
            raise RuntimeError("ref_hit_rays: %d" % rc)
        return h, t, idx, nrm, p

    def bounding_boxes(self, which):
        n = self.n_tris if which == 0 else 1
        omin, omax, flag = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), np.zeros(n, np.int32)
        rc = hit_lib().ref_hit_bounding_boxes(self.handle, which, _f(omin), _f(omax), _i(flag))
        if rc != 0:
            raise RuntimeError("ref_hit_bounding_boxes: %d" % rc)
        return flag, omin, omax

    def close(self):
        if self.handle:
            hit_lib().ref_hit_scene_free(self.handle)
            self.handle = None


def aabb_hit(bmin, bmax, origin, direction, t_min, t_max):
    a = [np.ascontiguousarray(x, dtype=np.float32) for x in (bmin, bmax, origin, direction, t_min, t_max)]
    out = np.zeros(len(a[4]), np.int32)
    rc = hit_lib().ref_aabb_hit(len(out), *[_f(x) for x in a], _i(out))
    if rc != 0:
        raise RuntimeError("ref_aabb_hit: %d" % rc)
    return out


def aabb_op(op, amin, amax, bmin, bmax):
    a = [np.ascontiguousarray(x, dtype=np.float32).reshape(-1, 3) for x in (amin, amax, bmin, bmax)]
    omin, omax = np.zeros_like(a[0]), np.zeros_like(a[0])
    rc = hit_lib().ref_aabb_op(int(op), len(omin), *[_f(x) for x in a], _f(omin), _f(omax))
    if rc != 0:
        raise RuntimeError("ref_aabb_op: %d" % rc)
    return omin, omax


def thrust_sort_host(key, order=None):
    """rocThrust's thrust::sort(thrust::seq, ...) with BoxCompare's rule (bvh.h:36-41) on the host."""
    key = np.ascontiguousarray(key, dtype=np.float32)
    order = np.arange(len(key), dtype=np.int32) if order is None else np.ascontiguousarray(order, dtype=np.int32).copy()
    rc = thrust_lib().ref_thrust_sort_host(len(order), _f(key), _i(order))
    if rc != 0:
        raise RuntimeError("ref_thrust_sort_host: %d" % rc)
    return order


def thrust_sort_device(key, order=None, mode=0):
    """The same call inside ONE device thread (what BVHNode's constructor does); mode 1: without an explicit policy."""
    key = np.ascontiguousarray(key, dtype=np.float32)
    order = np.arange(len(key), dtype=np.int32) if order is None else np.ascontiguousarray(order, dtype=np.int32).copy()
    rc = thrust_lib().ref_thrust_sort_device(int(mode), len(order), _f(key), _i(order))
    if rc != 0:
        raise RuntimeError("ref_thrust_sort_device: %d" % rc)
    return order
