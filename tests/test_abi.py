"""CPU tests of the C-ABI library: it loads, exports every symbol include/crt.h and include/crt_debug.h declare,
its host-only entry points (camera, OBJ parse, RGBA8/PNG) agree with the oracle, and its
compute entry points fail loudly — not silently fall back — without a HIP device."""
import ctypes as C
import os
import re
import struct
import zlib

import numpy as np
import pytest

from cudaraytracer_amd import _lib, api, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(headers=("crt.h", "crt_debug.h")):
    names = set()
    for h in headers:
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(crt_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_drop_in_header_holds_no_diagnostics():
    """include/crt.h is the surface SURVEY §8b lists plus the f-row entry points; switches, counters, self-checks and
    timing hooks live in include/crt_debug.h."""
    drop_in = declared_functions(("crt.h",))
    for required in ("crt_scene_from_mesh", "crt_scene_from_obj", "crt_camera_make", "crt_render", "crt_write_png",
                     "crt_last_error", "crt_render_rows", "crt_gather_strips", "crt_scene_build_bvh", "crt_render_rows_ex"):
        assert required in drop_in
    assert not [n for n in drop_in if n.startswith("crt_debug_") or n.startswith("crt_set_kernel") or n.startswith("crt_last_kernel")]
    assert [n for n in declared_functions(("crt_debug.h",)) if n.startswith("crt_debug_")]


def test_library_exports_every_declared_symbol(crt_built):
    names = declared_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(crt_built, n), "libcrt_hip.so does not export %s" % n
        assert n in _lib.PROTOTYPES, "no ctypes prototype for %s" % n
    assert set(_lib.PROTOTYPES) == set(names)
    assert crt_built.crt_abi_version() == 4


def test_camera_matches_oracle(crt_built, oracle):
    cams = [((0, 100, 1000), (0, 150, 0), 40.0, 1920, 1080),
            ((278, 273, -800), (278, 273, 0), 40.0, 800, 600),
            ((0, 0, 0), (0, 0, -1), 40.0, 256, 256),
            ((3.5, -2.25, 7.0), (0.1, 0.2, 0.3), 63.7, 317, 211)]
    for lf, la, vfov, nx, ny in cams:
        a = api.create_camera(nx, ny, lf, la, 10.0, 0.0, vfov).as_array()
        o = oracle.camera_as_array(oracle.make_camera(lf, la, (0, 1, 0), vfov,
                                                      np.float32(nx) / np.float32(ny), 0.0, 10.0))
        assert np.array_equal(a.view(np.uint32), o.view(np.uint32))
    c = api.init_camera(1024, 512)      # the reference's live call site
    assert list(c.c.origin) == [0.0, 100.0, 1000.0] and c.c.lens_radius == 0.0


def test_parse_obj_matches_oracle(crt_built, oracle, tmp_path):
    sc = scenes.displaced_sphere(5, 7, index_base=1)
    path = scenes.write_obj(str(tmp_path / "m.obj"), sc)
    p1, i1 = api.parse_obj(path)
    p2, i2 = oracle.parse_obj(path)
    assert np.array_equal(p1.view(np.uint32), p2.view(np.uint32)) and np.array_equal(i1, i2)
    # `a//n` and `a/b/c` faces: the leading index of each token is what gets read
    with open(str(tmp_path / "n.obj"), "w") as f:
        f.write("vn 0 1 0\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1//1 2//1 3//1\nf 1/1/1 3/2/1 2/3/1\n")
    p, i = api.parse_obj(str(tmp_path / "n.obj"))
    assert i.tolist() == [[1, 2, 3], [1, 3, 2]] and p.shape == (3, 3)
    with pytest.raises(_lib.CrtError):
        api.parse_obj(str(tmp_path / "missing.obj"))


def test_parse_obj_quads_negative_indices_and_the_reference_cbox(crt_built, tmp_path):
    """Beyond the reference's reader: polygons are fanned, negative indices
    count back from the latest vertex, `a//n` tokens use their leading index — the format of
    the reference-held CudaTest/objects/cbox_smallbox.obj (reproduced here in its first two
    faces; the file itself is not available on the GPU box)."""
    with open(str(tmp_path / "q.obj"), "w") as f:
        f.write("v 0 0 0\nv 2 0 0\nv 2 2 0\nv 0 2 0\nf 1 2 3 4\n"         # quad -> (1,2,3), (1,3,4)
                "v 0 0 1\nf -1 -5 -4\n"                                       # relative: 5, 1, 2
                "f 1/9 2/9 3/9 4/9 5/9\n")                                    # pentagon -> 3 triangles
    p, i = api.parse_obj(str(tmp_path / "q.obj"))
    assert p.shape == (5, 3)
    assert i.tolist() == [[1, 2, 3], [1, 3, 4], [5, 1, 2], [1, 2, 3], [1, 3, 4], [1, 4, 5]]
    mean = np.array([[0, 0, 0], [2, 0, 0], [2, 2, 0], [0, 2, 0], [0, 0, 1]], dtype=np.float32).sum(axis=0) / np.float32(5)
    assert np.allclose(p[0], -mean)
    with open(str(tmp_path / "cbox.obj"), "w") as f:                          # Meshlab style, as in cbox_smallbox.obj:12-45
        f.write("vn 0.000000 1.000000 0.000000\nv 130.000000 165.000000 65.000000\nv 82.000000 165.000000 225.000000\n"
                "v 240.000000 165.000000 272.000000\nv 290.000000 165.000000 114.000000\n"
                "f 1//1 2//1 3//1\nf 1//1 3//1 4//1\n")
    p, i = api.parse_obj(str(tmp_path / "cbox.obj"))
    assert p.shape == (4, 3) and i.tolist() == [[1, 2, 3], [1, 3, 4]]
    ref = "/root/reference/CudaTest/objects/cbox_smallbox.obj"
    if os.path.exists(ref):                                                    # build container only
        p, i = api.parse_obj(ref)
        assert p.shape == (24, 3) and i.shape == (12, 3) and i.min() == 1 and i.max() == 24


def test_strip_part_stride_and_layout_checks(crt_built):
    L = crt_built
    for nx, ny, world in [(1920, 1080, 8), (3840, 2160, 8), (90, 150, 3), (7, 33, 2), (5, 40, 1)]:
        rows0 = L.crt_strip_rows(ny, 0, 32, world)
        stride = L.crt_strip_part_stride(nx, ny, 32, world)
        assert stride % 4 == 0 and rows0 * nx * 3 <= stride < rows0 * nx * 3 + 4
        for r in range(1, world):
            if r * 32 < ny:
                assert L.crt_strip_rows(ny, r, 32, world) <= rows0
    assert L.crt_strip_part_stride(0, 10, 32, 2) == _lib.CRT_ERR_INVALID
    assert L.crt_strip_part_stride(10, 10, 24, 2) == _lib.CRT_ERR_INVALID      # strips of a split are multiples of 32 rows


def decode_png(path):
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, hdr = 8, b"", None
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + body) & 0xFFFFFFFF
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    nx, ny, depth, ctype = hdr[:4]
    assert depth == 8 and ctype == 6
    raw = zlib.decompress(idat)                                    # an independent inflater (python's zlib)
    rows = np.frombuffer(raw, dtype=np.uint8).reshape(ny, nx * 4 + 1)
    out = np.zeros((ny, nx * 4), dtype=np.int64)
    for y in range(ny):                                            # undo the row filters (PNG spec 9.2)
        f, line = int(rows[y, 0]), rows[y, 1:].astype(np.int64)
        up = out[y - 1] if y else np.zeros(nx * 4, dtype=np.int64)
        assert 0 <= f <= 4
        if f == 0:
            out[y] = line
        elif f == 2:
            out[y] = (line + up) & 255
        else:
            for i in range(nx * 4):
                a = out[y, i - 4] if i >= 4 else 0
                b, c = up[i], (up[i - 4] if i >= 4 else 0)
                if f == 1:
                    pred = a
                elif f == 3:
                    pred = (a + b) >> 1
                else:
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                out[y, i] = (line[i] + pred) & 255
    return out.astype(np.uint8).reshape(ny, nx, 4)


def test_rgba8_and_png_match_oracle(crt_built, oracle, golden_dir, tmp_path):
    g = np.load(os.path.join(golden_dir, "c2_cornell_100x75x16.npz"))
    rgb = g["rgb"].copy()
    rgb[3, 4, 1] = np.nan
    assert np.array_equal(api.to_rgba8(rgb), oracle.to_rgba8(rgb))
    path = api.WritePng(100, 75, 7, rgb, folder=str(tmp_path / "images" / "moveTest"))
    assert path.endswith("picture_7.png")
    assert np.array_equal(decode_png(path), oracle.to_rgba8(rgb))


def test_strip_rows_matches_partition(crt_built):
    from cudaraytracer_amd.dist import partition_strips
    for ny, world in [(1080, 8), (2160, 8), (75, 2), (33, 3), (64, 2)]:
        for r in range(world):
            strips = partition_strips(ny, world, r)
            want = sum(c for _, c in strips)
            got = crt_built.crt_strip_rows(ny, r, 32, world)
            assert got == want, (ny, world, r)


def test_compute_entry_points_fail_loudly_without_device(crt_built):
    if crt_built.crt_device_count() > 0:
        pytest.skip("a HIP device is present")
    L = crt_built
    pts9 = np.zeros(9, dtype=np.float32).ctypes.data_as(C.POINTER(C.c_float))
    assert not L.crt_rng_create(0, 16, 0)
    assert b"no HIP device" in L.crt_last_error()
    sc = scenes.quad()
    with pytest.raises(_lib.CrtError) as e:
        api.Scene.from_mesh(sc["points"], sc["idx"], sc["normals"])
    assert "no HIP device" in str(e.value)
    out = np.zeros(12, dtype=np.float32)
    cam = api.init_camera(2, 2)
    rc = L.crt_render(2, 2, 1, 8, 0, None, C.byref(cam.c), 0, out.ctypes.data_as(C.POINTER(C.c_float)))
    assert rc == _lib.CRT_ERR_NO_DEVICE
    with pytest.raises(_lib.CrtError):
        api.renderListAnimation(2, 2, 1, 8, 0, 0, None, cam, None)
    assert L.crt_assemble_strips(C.c_void_p(16), 4096, 8, 8, 32, 1, C.c_void_p(16), None) == _lib.CRT_ERR_NO_DEVICE
    assert L.crt_gather_strips(C.c_void_p(16), 8, 8, 32, 0, 1, C.c_void_p(16), 4096, C.c_void_p(16), C.c_void_p(16), None) \
        == _lib.CRT_ERR_NO_DEVICE
    assert L.crt_gather_strips(C.c_void_p(16), 8, 8, 32, 0, 1, C.c_void_p(16), 4096, C.c_void_p(16), None, None) \
        == _lib.CRT_ERR_INVALID                                                 # no communicator
    a3 = (C.c_float * 3)(0.5, 0.5, 0.5)
    assert not L.crt_multi_create(pts9, 3, pts9, 1, None, 0, a3, 8, 8, 0, 1)
    assert b"no HIP device" in L.crt_last_error()
    pts = np.zeros(9, dtype=np.float32)
    assert L.crt_scene_update_pose(None, pts.ctypes.data_as(C.POINTER(C.c_float)), 3, None) == _lib.CRT_ERR_INVALID


def test_missing_library_is_an_error_not_a_fallback(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_handle", None)
    monkeypatch.setattr(_lib, "_LIB", str(tmp_path / "libcrt_hip.so"))
    with pytest.raises(_lib.CrtError) as e:
        _lib.lib()
    assert "no CPU fallback" in str(e.value)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "cudaraytracer_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".hpp", ".h", "Makefile")):
                text = open(os.path.join(dirpath, fn), errors="replace").read()
                assert "pyoracle" not in text and "crt_oracle" not in text and "libcrt_oracle" not in text, fn
