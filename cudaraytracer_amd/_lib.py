"""ctypes binding of libcrt_hip.so — one prototype per symbol declared in include/crt.h and include/crt_debug.h."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libcrt_hip.so")
_CSRC = os.path.join(_HERE, "csrc")

CRT_OK = 0
CRT_ERR_INVALID = -1
CRT_ERR_NO_DEVICE = -2
CRT_ERR_HIP = -3
CRT_ERR_IO = -4
CRT_ERR_UNSUPPORTED = -5
CRT_ERR_NOMEM = -6


class CrtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("crt error %d: %s" % (code, message))
        self.code = code


class CrtMaterial(C.Structure):
    """struct crt_material (include/crt.h)."""
    _fields_ = [("kind", C.c_int), ("albedo", C.c_float * 3), ("param", C.c_float)]


class CrtCamera(C.Structure):
    """struct crt_camera (include/crt.h) == Camera's data members, core/camera.h:46-52."""
    _fields_ = [(n, C.c_float * 3) for n in
                ("lower_left_corner", "horizontal", "vertical", "origin", "x", "y", "z")] + \
               [("lens_radius", C.c_float)]


def library_path():
    return _LIB


def build_library(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 build of csrc/ into libcrt_hip.so (in-tree)."""
    args = ["make", "-C", _CSRC]
    if force:
        args.append("-B")
    if not verbose:
        args.append("-s")
    subprocess.check_call(args)
    return _LIB


# symbol -> (restype, argtypes); must list every function include/crt.h and include/crt_debug.h declare.
_fp = C.POINTER(C.c_float)
PROTOTYPES = {
    "crt_abi_version": (C.c_int, []),
    "crt_last_error": (C.c_char_p, []),
    "crt_device_count": (C.c_int, []),
    "crt_camera_make": (None, [C.POINTER(CrtCamera), _fp, _fp, _fp, C.c_float, C.c_float, C.c_float, C.c_float]),
    "crt_scene_from_mesh": (C.c_void_p, [_fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp]),
    "crt_scene_from_obj": (C.c_void_p, [C.c_char_p, C.c_float, _fp]),
    "crt_scene_triangle_count": (C.c_int, [C.c_void_p]),
    "crt_scene_update_pose": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_void_p]),
    "crt_scene_free": (None, [C.c_void_p]),
    "crt_parse_obj": (C.c_int, [C.c_char_p, _fp, _fp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "crt_rng_create": (C.c_void_p, [C.c_int64, C.c_int64, C.c_uint64]),
    "crt_rng_create_strips": (C.c_void_p, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64]),
    "crt_strip_rows": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "crt_rng_reset": (C.c_int, [C.c_void_p, C.c_uint64]),
    "crt_rng_export": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "crt_rng_free": (None, [C.c_void_p]),
    "crt_render_rows": (C.c_int, [C.c_void_p, C.POINTER(CrtCamera), C.c_void_p, C.c_int, C.c_int, C.c_int,
                                  C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "crt_render_strips": (C.c_int, [C.c_void_p, C.POINTER(CrtCamera), C.c_void_p, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "crt_scene_set_materials": (C.c_int, [C.c_void_p, C.POINTER(CrtMaterial), C.c_int, C.POINTER(C.c_int)]),
    "crt_render_rows_ex": (C.c_int, [C.c_void_p, C.POINTER(CrtCamera), C.c_void_p, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "crt_scene_build_bvh": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "crt_scene_refit_bvh": (C.c_int, [C.c_void_p, C.c_void_p]),
    "crt_scene_bvh_node_count": (C.c_int, [C.c_void_p]),
    "crt_render_rows_bvh": (C.c_int, [C.c_void_p, C.POINTER(CrtCamera), C.c_void_p, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "crt_render_strips_bvh": (C.c_int, [C.c_void_p, C.POINTER(CrtCamera), C.c_void_p, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "crt_debug_bvh_export": (C.c_int, [C.c_void_p, _fp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "crt_write_timing_csv": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double),
                                       C.POINTER(C.c_double), C.c_double]),
    "crt_render": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                             C.POINTER(CrtCamera), C.c_uint64, _fp]),
    "crt_stream_release": (C.c_int, [C.c_void_p]),
    "crt_set_kernel_variant": (C.c_int, [C.c_int]),
    "crt_set_kernel_timing": (C.c_int, [C.c_int]),
    "crt_debug_tiled_stats": (C.c_int, [C.c_int, C.POINTER(C.c_uint64)]),
    "crt_debug_wave_timeline": (C.c_int, [C.POINTER(C.c_uint64), C.c_int]),
    "crt_debug_check_reciprocal": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_uint64)]),
    "crt_debug_check_division": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint64)]),
    "crt_debug_audit_tiles": (C.c_int, [C.c_void_p, C.POINTER(CrtCamera), C.c_int, C.c_int, C.c_int, C.c_uint64,
                                        C.POINTER(C.c_uint64)]),
    "crt_debug_vec3": (C.c_int, [C.c_int, C.c_int, C.c_longlong, _fp, _fp, _fp, _fp]),
    "crt_last_kernel_ms": (C.c_float, []),
    "crt_last_kernel_times": (C.c_int, [C.POINTER(C.c_float)]),
    "crt_strip_part_stride": (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "crt_assemble_strips": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "crt_gather_strips": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "crt_debug_inject_send_failure": (C.c_int, [C.c_int]),
    "crt_render_strips_ex": (C.c_int, [C.c_void_p, C.POINTER(CrtCamera), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "crt_comm_unique_id": (C.c_int, [C.c_void_p]),
    "crt_comm_init_rank": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "crt_comm_destroy": (C.c_int, [C.c_void_p]),
    "crt_multi_configure": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(CrtMaterial), C.c_int, C.POINTER(C.c_int)]),
    "crt_multi_create": (C.c_void_p, [_fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int, C.c_int, C.c_uint64, C.c_int]),
    "crt_multi_render": (C.c_int, [C.c_void_p, C.POINTER(CrtCamera), C.c_int, C.c_int, C.c_int, _fp]),
    "crt_multi_update_pose": (C.c_int, [C.c_void_p, _fp, C.c_int]),
    "crt_multi_frame": (C.c_void_p, [C.c_void_p]),
    "crt_multi_device_count": (C.c_int, [C.c_void_p]),
    "crt_multi_free": (None, [C.c_void_p]),
    "crt_debug_trace_rays": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, _fp, _fp, C.POINTER(C.c_int), _fp,
                                       C.POINTER(C.c_int), _fp, _fp]),
    "crt_debug_box_hit": (C.c_int, [C.c_int, C.c_int, _fp, _fp, _fp, C.POINTER(C.c_int)]),
    "crt_debug_box_op": (C.c_int, [C.c_int, C.c_int, C.c_int, _fp, _fp, _fp]),
    "crt_debug_thrust_sort": (C.c_int, [C.c_int, _fp, C.POINTER(C.c_int)]),
    "crt_debug_xorwow_jump": (C.c_int, [C.c_int, C.POINTER(C.c_uint)]),
    "crt_debug_force_stream_recycle": (C.c_int, [C.c_int]),
    "crt_write_png_rgba8": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_ubyte)]),
    "crt_encode_png_rgba8": (C.c_longlong, [C.POINTER(C.c_ubyte), C.c_int, C.c_int, C.POINTER(C.c_ubyte), C.c_longlong]),
    "crt_to_rgba8": (None, [_fp, C.c_int, C.c_int, C.POINTER(C.c_ubyte)]),
    "crt_write_png": (C.c_int, [C.c_char_p, C.c_int, C.c_int, _fp]),
}

_handle = None


def lib():
    """The loaded library. Raises (never falls back) if it has not been built."""
    global _handle
    if _handle is None:
        if not os.path.exists(_LIB):
            raise CrtError(CRT_ERR_NO_DEVICE,
                           "libcrt_hip.so is not built (run __graft_entry__.build() or "
                           "`make -C cudaraytracer_amd/csrc`); there is no CPU fallback")
        # torch bundles its own libamdhip64.so / libhsa-runtime64.so and asks for them by file name,
        # libcrt_hip.so asks for libamdhip64.so.7 (SONAME).  If this library came first the loader
        # would give torch a SECOND copy of the HIP runtime, and two runtimes in one process cannot
        # both own the device (torch.cuda.is_available() turns False).  With torch's copy loaded
        # first, the SONAME matches and both share it.  (C/C++ callers have no torch and no issue.)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        h = C.CDLL(_LIB)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        _handle = h
    return _handle


def check(code):
    if code != CRT_OK:
        raise CrtError(code, lib().crt_last_error().decode("utf-8", "replace"))
    return code


def check_handle(ptr):
    if not ptr:
        raise CrtError(CRT_ERR_INVALID, lib().crt_last_error().decode("utf-8", "replace"))
    return ptr
