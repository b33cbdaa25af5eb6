"""WritePng's file (core/render.h:135-157): the reference calls stbi_write_png of the stb_image_write.h it vendors; the
product restates that encoder (csrc/crt_png.cpp) and must write the same bytes.  Pinned two ways:
  * against fixtures the reference's encoder produced (tests/golden/stb_png_fixtures.npz, tests/make_png_fixtures.py) —
    everywhere;
  * against the reference's encoder itself, built from the reference's own file (oracle/_ref/libref_stb.so, `make -C oracle
    ref`) — where that library exists (it is built in the development container and travels with the repository copy)."""
import ctypes as C
import os

import numpy as np
import pytest

from make_png_fixtures import images, reference_encoder


def product_encode(lib, img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w, _ = img.shape
    p = img.ctypes.data_as(C.POINTER(C.c_ubyte))
    n = lib.crt_encode_png_rgba8(p, w, h, None, 0)
    assert n > 0
    out = np.zeros(n, dtype=np.uint8)
    assert lib.crt_encode_png_rgba8(p, w, h, out.ctypes.data_as(C.POINTER(C.c_ubyte)), n) == n
    return out


def test_png_bytes_equal_the_reference_encoders_fixtures(crt_built, golden_dir):
    g = np.load(os.path.join(golden_dir, "stb_png_fixtures.npz"))
    names = sorted(k[5:] for k in g.files if k.startswith("rgba_"))
    assert len(names) >= 18
    for name in names:
        got = product_encode(crt_built, g["rgba_" + name])
        assert np.array_equal(got, g["png_" + name]), name
    for name, img in images().items():                      # the generator still describes the fixtures
        assert np.array_equal(img, g["rgba_" + name]), name


def test_png_bytes_equal_the_reference_encoder(crt_built):
    ref = reference_encoder()
    if ref is None:
        pytest.skip("oracle/_ref/libref_stb.so not built (no reference tree on this machine)")
    for seed in range(120):
        rng = np.random.RandomState(seed)
        h, w = int(rng.randint(1, 70)), int(rng.randint(1, 200))
        kind = seed % 5
        yy, xx = np.mgrid[0:h, 0:w]
        if kind == 0:
            pal = rng.randint(0, 256, (int(rng.randint(1, 5)), 4))
            img = pal[rng.randint(0, len(pal), (h, w))]
        elif kind == 1:
            t = rng.randint(0, 256, (int(rng.randint(1, 9)), int(rng.randint(1, 17)), 4))
            img = np.tile(t, (h // t.shape[0] + 1, w // t.shape[1] + 1, 1))[:h, :w]
        elif kind == 2:
            img = np.stack([xx + yy, xx * 2, yy * 3, xx * 0 + 255], -1) + rng.randint(0, int(rng.choice([1, 2, 4, 16])), (h, w, 4))
        elif kind == 3:
            t = yy / max(h - 1, 1)
            sky = np.stack([255 * (1 - 0.5 * t), 255 * (1 - 0.3 * t), 255 + 0 * t, 255 + 0 * t], -1)
            blob = ((xx - w / 2) ** 2 + (yy - h / 2) ** 2) < (min(h, w) / 3) ** 2
            img = np.where(blob[..., None], np.array([166, 12, 12, 255]), sky)
        else:
            img = rng.randint(0, 256, (h, w, 4))
        img = (np.asarray(img) % 256).astype(np.uint8)
        assert np.array_equal(product_encode(crt_built, img), ref(img)), (seed, kind, h, w)
    # one frame of the benchmark's size: sky gradient, a noisy red disc (8.3 MB of filtered rows, many window wrap-arounds)
    h, w = 1080, 1920
    yy, xx = np.mgrid[0:h, 0:w]
    t = yy / (h - 1)
    rng = np.random.RandomState(1)
    sky = np.stack([255 * (1 - 0.5 * t), 255 * (1 - 0.3 * t), 255 + 0 * t, 255 + 0 * t], -1)
    disc = ((xx - w / 2) ** 2 + (yy - h / 2) ** 2) < 300 ** 2
    red = np.stack([150 + rng.randint(0, 40, (h, w)), 10 + rng.randint(0, 8, (h, w)), 10 + rng.randint(0, 8, (h, w)), 255 + 0 * xx], -1)
    img = np.where(disc[..., None], red, sky).astype(np.uint8)
    assert np.array_equal(product_encode(crt_built, img), ref(img))


def test_write_png_is_the_reference_pipeline(crt_built, oracle, tmp_path):
    """crt_write_png = WritePng: the oracle's conversion (flip, (unsigned char)(255.99 * c), alpha 255) then the
    reference's encoder — the file on disk, byte for byte."""
    ref = reference_encoder()
    if ref is None:
        pytest.skip("oracle/_ref/libref_stb.so not built (no reference tree on this machine)")
    rng = np.random.RandomState(3)
    nx, ny = 61, 37
    rgb = rng.uniform(0.0, 1.0, (ny, nx, 3)).astype(np.float32)
    rgb[5:20, 10:40] = np.linspace(0.0, 1.0, 30, dtype=np.float32)[None, :, None]
    path = str(tmp_path / "picture_0.png")
    assert crt_built.crt_write_png(path.encode(), nx, ny, rgb.ctypes.data_as(C.POINTER(C.c_float))) == 0
    want = ref(oracle.to_rgba8(rgb))
    assert open(path, "rb").read() == want.tobytes()


def test_png_decodes_to_its_pixels(crt_built, tmp_path):
    """Independent of stb: python's zlib inflates the stream and the PNG filters undo to the input."""
    from test_abi import decode_png
    for name, img in images().items():
        path = str(tmp_path / (name + ".png"))
        h, w, _ = img.shape
        assert crt_built.crt_write_png_rgba8(path.encode(), w, h, np.ascontiguousarray(img).ctypes.data_as(C.POINTER(C.c_ubyte))) == 0
        assert np.array_equal(decode_png(path), img), name


def test_timing_csv_equals_the_reference_writer(crt_built, tmp_path):
    """output.csv (kernel.cu:61,101): crt_write_timing_csv against the reference's own writeCSV (Loader/CSVWriter.h, built from
    the reference's file into oracle/_ref/libref_csv.so) fed the rows its call sites build."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "oracle", "_ref", "libref_csv.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libref_csv.so not built (no reference tree on this machine)")
    ref = C.CDLL(path)
    ref.ref_write_csv.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double]
    rng = np.random.RandomState(11)
    for n in (0, 1, 7):
        frames = (C.c_int * max(n, 1))(*range(3, 3 + max(n, 1)))
        rend = (C.c_double * max(n, 1))(*[float(x) for x in rng.uniform(0.0, 40.0, max(n, 1))])
        upd = (C.c_double * max(n, 1))(*[float(x) for x in 10.0 ** rng.uniform(-7, 3, max(n, 1))])
        build = float(rng.uniform(0, 2))
        a, b = str(tmp_path / ("ref%d.csv" % n)), str(tmp_path / ("crt%d.csv" % n))
        assert ref.ref_write_csv(a.encode(), n, frames, rend, upd, build) == 0
        assert crt_built.crt_write_timing_csv(b.encode(), n, frames, rend, upd, C.c_double(build)) == 0
        assert open(a, "rb").read() == open(b, "rb").read(), n
