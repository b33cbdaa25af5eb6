"""SURVEY §8 row a6, PINNED: the vec3 arithmetic against the reference's own `core/vec3.h`.

That header is `__host__ __device__` throughout and builds for the host from its own file with the image's HIP front end
(oracle/ref_vec3.cpp -> oracle/_ref/libref_vec3.so; nothing copied, nothing stood in for).  Three things are held to it,
bit for bit (NaN == NaN whatever the payload):
  * the oracle's restatement (oracle/crt_oracle.c:26-84) — against the library where it exists, against the committed
    vectors it produced (tests/golden/vec3_ref_vectors.npz, tests/make_vec3_fixtures.py) everywhere;
  * the product's host helpers hsub / hmul / hcross / hunit / hrotate (csrc/crt_internal.hpp), through crt_debug_vec3(0, ..);
  * on the GPU, the kernels' device functions — unit_vector and its fast forms, the sky lerp, clip, `col /= ns`, the
    dot / cross spellings of the exact test — through crt_debug_vec3(1, ..) against the same committed vectors."""
import ctypes as C
import os

import numpy as np
import pytest

from make_vec3_fixtures import GOLDEN, OPS, SCALAR_OPS, inputs, reference_vec3


def same_bits(got, want):
    got = np.ascontiguousarray(got, dtype=np.float32)
    want = np.ascontiguousarray(want, dtype=np.float32)
    return (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))


def assert_same(got, want, what, rows=None):
    ok = same_bits(got, want)
    if rows is not None:
        ok = ok[rows]
    if not ok.all():
        bad = np.argwhere(~ok)[0]
        raise AssertionError("%s: %d of %d values differ, first at %s: got %r want %r"
                             % (what, int((~ok).sum()), ok.size, tuple(bad), got[rows][tuple(bad)] if rows is not None else got[tuple(bad)],
                                want[rows][tuple(bad)] if rows is not None else want[tuple(bad)]))


@pytest.fixture(scope="module")
def vectors():
    g = np.load(GOLDEN)
    return {k: g[k] for k in g.files}


def want_of(vectors, name):
    w = vectors["out_" + name]
    if name in SCALAR_OPS:
        w = np.stack([w, np.zeros_like(w), np.zeros_like(w)], 1)
    return w


def test_fixture_inputs_are_the_generators(vectors):
    a, b, s = inputs()
    for k, v in (("a", a), ("b", b), ("s", s)):
        assert same_bits(vectors[k], v).all(), k


def test_fixture_is_what_the_reference_header_computes(vectors):
    ref = reference_vec3()
    if ref is None:
        pytest.skip("oracle/_ref/libref_vec3.so not built (no reference tree on this machine)")
    for name, code in OPS.items():
        assert_same(ref(code, vectors["a"], vectors["b"], vectors["s"]), want_of(vectors, name), name)


def test_oracle_vec3_equals_the_reference_header(oracle, vectors):
    """crt_oracle.c's add / sub / ... / rotate on the committed operands == what vec3.h produced; and, where the
    library is here, on fresh random operands too."""
    a, b, s = vectors["a"], vectors["b"], vectors["s"]
    restated = 0
    for name, code in OPS.items():
        got = oracle.vec3_op(code, a, b, s)
        if got is None:
            assert name in ("make_unit_vector", "neg"), name      # nothing on the path calls these
            continue
        restated += 1
        assert_same(got, want_of(vectors, name), "oracle " + name)
    assert restated == len(OPS) - 2
    ref = reference_vec3()
    if ref is None:
        return
    for seed in range(5):
        a2, b2, s2 = inputs(seed=1000 + seed, n_random=4000)
        for name, code in OPS.items():
            got = oracle.vec3_op(code, a2, b2, s2)
            if got is not None:
                assert_same(got, ref(code, a2, b2, s2), "oracle %s (seed %d)" % (name, seed))


def product_vec3(lib, where, op, a, b, s):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    s = np.ascontiguousarray(s, dtype=np.float32)
    out = np.zeros_like(a)
    fp = C.POINTER(C.c_float)
    rc = lib.crt_debug_vec3(where, op, len(a), a.ctypes.data_as(fp), b.ctypes.data_as(fp), s.ctypes.data_as(fp), out.ctypes.data_as(fp))
    return rc, out


def test_product_host_helpers_equal_the_reference_header(crt_built, vectors):
    """hsub / hmul / hcross / hunit / hrotate (camera constructor, scene pack, shading normals, BVH boxes)."""
    a, b, s = vectors["a"], vectors["b"], vectors["s"]
    for name in ("sub", "muls", "cross", "unit_vector", "rotate"):
        rc, got = product_vec3(crt_built, 0, OPS[name], a, b, s)
        assert rc == 0, name
        assert_same(got, want_of(vectors, name), "host " + name)
    rc, _ = product_vec3(crt_built, 0, OPS["lerp"], a, b, s)
    assert rc == -5                                               # CRT_ERR_UNSUPPORTED: no such helper on the host side


@pytest.mark.gpu
def test_device_vec3_equals_the_reference_header(gpu, crt_built, vectors):
    crt = crt_built
    """The kernels' device functions on the GPU against the vectors the reference's vec3.h produced on the host."""
    a, b, s = vectors["a"], vectors["b"], vectors["s"]
    for name in ("sub", "muls", "dot", "cross", "length", "unit_vector", "clip"):
        rc, got = product_vec3(crt, 1, OPS[name], a, b, s)
        assert rc == 0, name
        assert_same(got, want_of(vectors, name), "device " + name)
    # col /= float(ns): the divisors the path uses are sample counts; the fixture's s also holds 0, Inf, NaN, denormals
    rc, got = product_vec3(crt, 1, OPS["div_assign"], a, b, s)
    assert rc == 0
    assert_same(got, vectors["out_div_assign"], "device div_assign")
    # the fast forms == the plain ones == vec3.h's unit_vector, for every operand row (the wave vote sends a wavefront
    # that holds a special value down the IEEE path, the others down the 3-instruction quotients)
    want = vectors["out_unit_vector"]
    rc, got = product_vec3(crt, 1, 100, a, b, s)
    assert rc == 0
    assert_same(got, want, "device unit_vector_fast")
    rc, got = product_vec3(crt, 1, 101, a, b, s)
    assert rc == 0
    assert_same(got[:, 1], want[:, 1], "device unit_vector_fast_y")
    # backgroundSky's lerp (render.h:45) for t in [0, 1]: rows whose s is such a t and for which y = 2t - 1 is exact
    t = s.astype(np.float32)
    with np.errstate(over="ignore", invalid="ignore"):   # s holds huge values and NaN on purpose: those rows are deselected just below
        y = (np.float32(2.0) * t - np.float32(1.0)).astype(np.float32)
        rows = np.isfinite(t) & (t >= 0) & (t <= 1) & ((np.float32(0.5) * (y + np.float32(1.0))).astype(np.float32) == t)
    assert rows.sum() > 1000
    rc, got = product_vec3(crt, 1, OPS["lerp"], a, b, s)
    assert rc == 0
    assert_same(got, vectors["out_sky_lerp"], "device sky lerp", rows)
    # sky_from_unit(v) = lerp(0.5f * (v.y + 1.0f), 1, (0.5, 0.7, 1.0)): the t it forms, then the same lerp
    unit = vectors["out_unit_vector"]
    rc, got = product_vec3(crt, 1, 102, unit, b, s)
    assert rc == 0
    ref = reference_vec3()
    t2 = (np.float32(0.5) * (unit[:, 1] + np.float32(1.0))).astype(np.float32)
    if ref is not None:
        one = np.ones_like(unit); to = np.tile(np.array([0.5, 0.7, 1.0], np.float32), (len(unit), 1))
        assert_same(got, ref(OPS["lerp"], one, to, t2), "device sky_from_unit")
    else:   # the GPU box: the same lerp in numpy float32 (one IEEE multiply and add per channel)
        with np.errstate(all="ignore"):
            to = np.array([0.5, 0.7, 1.0], np.float32) - np.float32(1.0)
            assert_same(got, (np.float32(1.0) + t2[:, None] * to[None, :]).astype(np.float32), "device sky_from_unit")
