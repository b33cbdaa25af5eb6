#!/usr/bin/env python3
"""Generates tests/golden/stb_png_fixtures.npz: RGBA images and the PNG files the REFERENCE's encoder writes for them —
the stb_image_write.h the reference vendors and calls (core/render.h:6,153), built from the reference's own file by
`make -C oracle ref` (oracle/_ref/libref_stb.so; only where /root/reference exists).  The fixtures are data (inputs and
the reference's outputs); tests/test_png.py holds the product's encoder to them on machines without the reference.
usage: python tests/make_png_fixtures.py"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def reference_encoder():
    path = os.path.join(ROOT, "oracle", "_ref", "libref_stb.so")
    if not os.path.exists(path):
        return None
    lib = C.CDLL(path)
    lib.ref_stb_png.restype = C.c_longlong
    lib.ref_stb_png.argtypes = [C.POINTER(C.c_ubyte), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_ubyte), C.c_longlong]

    def encode(img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w, _ = img.shape
        cap = w * h * 8 + 4096
        out = np.zeros(cap, dtype=np.uint8)
        n = lib.ref_stb_png(img.ctypes.data_as(C.POINTER(C.c_ubyte)), w, h, 4, out.ctypes.data_as(C.POINTER(C.c_ubyte)), cap)
        assert 0 < n <= cap
        return out[:n].copy()
    return encode


def images():
    """Small images that walk the encoder's branches: every filter winning somewhere, matches of maximal length and
    far distances, full hash buckets, the stored-block fallback (noise), single pixels and single rows/columns."""
    rng = np.random.RandomState(7)
    out = {}
    for name, (h, w) in {"px1": (1, 1), "row": (1, 37), "col": (29, 1), "small": (17, 33), "frame": (54, 96)}.items():
        yy, xx = np.mgrid[0:h, 0:w]
        t = yy / max(h - 1, 1)
        sky = np.stack([255 * (1 - 0.5 * t), 255 * (1 - 0.3 * t), 255 + 0 * t, 255 + 0 * t], -1)
        blob = ((xx - w / 2) ** 2 + (yy - h / 2) ** 2) < (min(h, w) / 3) ** 2
        out[name + "_render"] = np.where(blob[..., None], np.array([166, 12, 12, 255]), sky).astype(np.uint8)
        out[name + "_noise"] = rng.randint(0, 256, (h, w, 4)).astype(np.uint8)
        out[name + "_grad"] = (np.stack([xx * 3, yy * 5 + xx, (xx + yy) // 2, xx * 0 + 255], -1) % 256).astype(np.uint8)
    tile = rng.randint(0, 256, (5, 11, 4))
    out["tiled"] = np.tile(tile, (13, 19, 1))[:60, :200].astype(np.uint8)          # repeats at many distances, > 32 KiB of rows
    out["flat"] = np.full((40, 300, 4), 200, dtype=np.uint8)                         # matches of length 258
    sparse = np.zeros((64, 64, 4), dtype=np.uint8)
    m = rng.rand(64, 64) < 0.03
    sparse[m] = rng.randint(0, 256, (int(m.sum()), 4))
    out["sparse"] = sparse
    return out


if __name__ == "__main__":
    enc = reference_encoder()
    if enc is None:
        raise SystemExit("oracle/_ref/libref_stb.so is missing: run `make -C oracle ref` where /root/reference exists")
    data = {}
    for name, img in images().items():
        data["rgba_" + name] = img
        data["png_" + name] = enc(img)
    path = os.path.join(ROOT, "tests", "golden", "stb_png_fixtures.npz")
    np.savez_compressed(path, **data)
    print("wrote %s: %d images, %d bytes" % (path, len(data) // 2, os.path.getsize(path)))
