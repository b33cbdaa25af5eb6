"""Golden vectors of the reference's own vec3 arithmetic.

`core/vec3.h` of the reference is `__host__ __device__` throughout and builds for the host from its own file
(oracle/ref_vec3.cpp -> oracle/_ref/libref_vec3.so, `make -C oracle ref`).  This script feeds it seeded inputs —
ordinary values over many magnitudes, the camera's ray directions, +-0, Inf, NaN, denormals, rotations != 0 — and
stores inputs and outputs in tests/golden/vec3_ref_vectors.npz, so that machines without the reference tree (the GPU
box) can hold the oracle, the product's host helpers and the kernels' device functions to the same vectors
(tests/test_vec3_ref.py).  Run here:  python tests/make_vec3_fixtures.py"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "libref_vec3.so")
GOLDEN = os.path.join(ROOT, "tests", "golden", "vec3_ref_vectors.npz")

OPS = {"add": 0, "sub": 1, "mulv": 2, "muls": 3, "divv": 4, "divs": 5, "dot": 6, "cross": 7, "length": 8,
       "squared_length": 9, "unit_vector": 10, "lerp": 11, "clip": 12, "div_assign": 13, "rotate": 14,
       "make_unit_vector": 15, "neg": 16, "sum": 17}


SCALAR_OPS = ("dot", "length", "squared_length", "sum")   # stored as n floats (the wrappers put them in column 0)


def reference_vec3():
    """op(code, a, b, s) -> n x 3 array from the reference's own vec3.h, or None where the library is not built."""
    if not os.path.exists(REF):
        return None
    L = C.CDLL(REF)
    fp = C.POINTER(C.c_float)
    L.ref_vec3_op.argtypes = [C.c_int, C.c_longlong, fp, fp, fp, fp]
    L.ref_vec3_op.restype = C.c_int

    def op(code, a, b, s):
        a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 3)
        b = np.ascontiguousarray(b, dtype=np.float32).reshape(-1, 3)
        s = np.ascontiguousarray(s, dtype=np.float32).reshape(-1)
        out = np.zeros_like(a)
        assert L.ref_vec3_op(int(code), len(a), a.ctypes.data_as(fp), b.ctypes.data_as(fp), s.ctypes.data_as(fp),
                             out.ctypes.data_as(fp)) == 0
        return out
    return op


SPECIALS = np.array([0.0, -0.0, 1.0, -1.0, 0.5, 2.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1.17549435e-38, 5.9e-39,
                     3.4028235e38, -3.4028235e38, 1e-20, 1e20, 9.094947e-13, 1.0995116e12, 0.001, 255.99], dtype=np.float32)


def inputs(seed=20261004, n_random=1500):
    """a, b (n x 3), s (n): the operands every operation is run on."""
    rng = np.random.RandomState(seed)
    parts_a, parts_b, parts_s = [], [], []
    # ordinary values over many magnitudes (log-uniform 2^-30..2^30, random signs)
    def ordinary(n):
        return (np.ldexp(rng.uniform(0.5, 1.0, (n, 3)), rng.randint(-30, 31, (n, 3))) * rng.choice([-1.0, 1.0], (n, 3))).astype(np.float32)
    parts_a.append(ordinary(n_random)); parts_b.append(ordinary(n_random))
    parts_s.append((np.ldexp(rng.uniform(0.5, 1.0, n_random), rng.randint(-20, 21, n_random)) * rng.choice([-1.0, 1.0], n_random)).astype(np.float32))
    # what the path feeds them: camera ray directions of the reference camera (|d0| ~ 10, llc cancellation), unit vectors,
    # colours in [0, 4], sample counts, jitter t in [0, 1]
    n = 800
    u, v = rng.uniform(0, 1, n).astype(np.float32), rng.uniform(0, 1, n).astype(np.float32)
    d0 = np.stack([np.float32(-6.47) + u * np.float32(12.94), np.float32(-3.1) + v * np.float32(7.28), np.full(n, -9.95, np.float32)], 1).astype(np.float32)
    parts_a.append(d0); parts_b.append(ordinary(n)); parts_s.append(rng.uniform(0, 1, n).astype(np.float32))
    col = rng.uniform(-0.5, 4.0, (n, 3)).astype(np.float32)
    parts_a.append(col); parts_b.append(rng.uniform(0, 1, (n, 3)).astype(np.float32))
    parts_s.append(rng.choice([1, 2, 3, 4, 5, 7, 16, 64, 100, 256, 1000], n).astype(np.float32))
    # rotations in degrees (b) applied to points (a): zero, axis-aligned, arbitrary, large
    rot = np.concatenate([np.zeros((100, 3)), rng.choice([0.0, 90.0, -90.0, 180.0, 45.0, 30.0], (300, 3)), rng.uniform(-720, 720, (400, 3))]).astype(np.float32)
    parts_a.append(ordinary(len(rot))); parts_b.append(rot); parts_s.append(rng.uniform(0, 1, len(rot)).astype(np.float32))
    # special values: every pair of specials in the first two components, a third cycling
    k = len(SPECIALS)
    ii, jj = np.meshgrid(np.arange(k), np.arange(k), indexing="ij")
    sa = np.stack([SPECIALS[ii.ravel()], SPECIALS[jj.ravel()], SPECIALS[(ii.ravel() + 2 * jj.ravel()) % k]], 1)
    sb = np.stack([SPECIALS[jj.ravel()], SPECIALS[(ii.ravel() + 1) % k], SPECIALS[(jj.ravel() + 3) % k]], 1)
    parts_a.append(sa); parts_b.append(sb); parts_s.append(SPECIALS[(ii.ravel() * 5 + jj.ravel()) % k])
    # specials against ordinary operands
    m = 20 * k
    oa = ordinary(m); oa[np.arange(m), rng.randint(0, 3, m)] = np.tile(SPECIALS, 20)
    parts_a.append(oa); parts_b.append(ordinary(m)); parts_s.append(np.tile(SPECIALS, 20)[::-1].copy())
    return (np.concatenate(parts_a).astype(np.float32), np.concatenate(parts_b).astype(np.float32),
            np.concatenate(parts_s).astype(np.float32))


def main():
    ref = reference_vec3()
    if ref is None:
        raise SystemExit("oracle/_ref/libref_vec3.so is not built: run `make -C oracle ref` where /root/reference exists")
    a, b, s = inputs()
    out = {"a": a, "b": b, "s": s}
    for name, code in OPS.items():
        r = ref(code, a, b, s)
        out["out_" + name] = r[:, 0].copy() if name in SCALAR_OPS else r
    # the sky lerp the path evaluates (render.h:45): lerp(t, vec3(1), vec3(0.5, 0.7, 1.0)) for t = s
    one = np.ones_like(a); to = np.tile(np.array([0.5, 0.7, 1.0], np.float32), (len(a), 1))
    out["out_sky_lerp"] = ref(OPS["lerp"], one, to, s)
    np.savez_compressed(GOLDEN, **out)
    print("wrote %s: %d operand rows x %d operations, %d bytes" % (GOLDEN, len(a), len(OPS) + 1, os.path.getsize(GOLDEN)))


if __name__ == "__main__":
    main()
