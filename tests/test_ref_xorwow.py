"""Row a3 (cuRAND XORWOW), the part that CAN be pinned: the recurrence and the skips against an independent XORWOW.

cuRAND is not in the reference tree and not in the image; the oracle restates `curand_init` / `curand` from the published
algorithm (oracle/crt_oracle.c: o_curand, xw_skip, o_curand_init).  rocRAND's `xorwow_engine` (rocrand_xorwow.h, shipped
with ROCm) is the same generator family written by other people: the same recurrence and Weyl constant, the same
2^67-step subsequences (its jump matrices are precomputed tables in rocrand_xorwow_precomputed.h, the oracle builds its
own from the step by repeated squaring), other seed-scrambling constants and another uniform mapping.  On RAW states the
two must therefore agree word for word:
  * `next()` == o_curand (state and output), 257 steps from 8 states;
  * `discard_subsequence(k)` / `discard(n)` == the oracle's skips, 632 cases: the pixel indices SetCurandState uses
    (deviceManage.h:127: subsequence = pixel_index, up to 3840*2160), every base-4 digit of the count alone, 64-bit
    counts, offsets with the Weyl word, single-bit states (every row of the matrices);
  * a skip of n equals n steps where n is small enough to step.
The committed fixture tests/golden/rocrand_xorwow_vectors.npz is the library's output (tests/make_xorwow_fixtures.py);
where oracle/_ref/libref_rocrand_xorwow.so exists the fixture is re-derived from it.
STILL UNPINNED: cuRAND's scrambling constants (0xaad26b49, 0xf7dcefdd, 1099087573, 2591861531), its start words, and
curand_uniform's `x * 2^-32 + 2^-33` — only a CUDA toolkit could settle those."""
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from oracle import pyref

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rocrand_xorwow_vectors.npz")


@pytest.fixture(scope="module")
def g():
    return np.load(GOLDEN)


def test_fixture_is_what_rocrand_computes(g):
    if not pyref.have_xorwow():
        pytest.skip("oracle/_ref/libref_rocrand_xorwow.so not built on this machine")
    x, d = pyref.rocrand_xorwow_skip(g["x"], g["d"], g["subsequence"], g["offset"])
    assert np.array_equal(x, g["x_out"]) and np.array_equal(d, g["d_out"])
    for i in range(len(g["seq_out"])):
        out, xs, ds = pyref.rocrand_xorwow_next(g["x"][i], g["d"][i], g["seq_out"].shape[1])
        assert np.array_equal(out, g["seq_out"][i]) and np.array_equal(xs, g["seq_x"][i]) and ds == g["seq_d"][i]


def test_fixture_covers_the_pixel_indices_and_every_digit(g):
    sub, off = g["subsequence"].tolist(), g["offset"].tolist()
    for last in (256 * 256 - 1, 800 * 600 - 1, 1024 * 512 - 1, 1920 * 1080 - 1, 3840 * 2160 - 1):
        assert last in sub
    for p in range(32):
        for dgt in (1, 2, 3):
            assert (dgt << (2 * p)) in sub and (dgt << (2 * p)) in off
    assert 2 ** 64 - 1 in sub and 2 ** 64 - 1 in off


def test_oracle_step_equals_rocrand_next(g):
    for i in range(len(g["seq_out"])):
        out, xs, ds = po.xorwow_next_raw(g["x"][i], g["d"][i], g["seq_out"].shape[1])
        assert np.array_equal(out, g["seq_out"][i]), i
        assert np.array_equal(xs, g["seq_x"][i]) and ds == g["seq_d"][i], i


def test_oracle_skips_equal_rocrand_discard(g):
    """curand_init's two skips (subsequence * 2^67 steps with d unchanged, then offset steps with d += offset * 362437)."""
    x, d = po.xorwow_skip_raw(g["x"], g["d"], g["subsequence"], g["offset"])
    bad = np.nonzero((x != g["x_out"]).any(axis=1) | (d != g["d_out"]))[0]
    assert len(bad) == 0, (bad[:5], g["subsequence"][bad[:5]], g["offset"][bad[:5]])
    assert (g["subsequence"] != 0).sum() > 400 and (g["offset"] != 0).sum() > 200


def test_oracle_matrices_row_by_row(g):
    """Single-bit states: row b of T^(2^67), of T^(2^67 * 4^16) and of T itself, all 160 rows."""
    z = np.zeros(160, np.uint32)
    one, zero = np.ones(160, np.uint64), np.zeros(160, np.uint64)
    x, _ = po.xorwow_skip_raw(g["bits"], z, one, zero)
    assert np.array_equal(x, g["bits_seq1"])
    x, _ = po.xorwow_skip_raw(g["bits"], z, one << np.uint64(32), zero)
    assert np.array_equal(x, g["bits_seq_hi"])
    x, _ = po.xorwow_skip_raw(g["bits"], z, zero, one)
    assert np.array_equal(x, g["bits_off1"])


def test_small_offsets_equal_stepping(g):
    """Closes the loop inside the oracle: discard(n) == n calls of the step, for the offsets small enough to walk."""
    for i in np.nonzero((g["subsequence"] == 0) & (g["offset"] > 0) & (g["offset"] <= 1000))[0]:
        _, xs, ds = po.xorwow_next_raw(g["x"][i], g["d"][i], int(g["offset"][i]))
        assert np.array_equal(xs, g["x_out"][i]) and ds == g["d_out"][i], i


def test_curand_init_is_scramble_then_pinned_skip():
    """o_curand_init(seed, pixel, 0) == the published scrambling of the seed (UNPINNED constants) followed by the skip that
    the tests above pin: nothing else happens in it."""
    for seed in (0, 1, 1984, 2 ** 40 + 7, 2 ** 64 - 1):
        s0 = (seed & 0xffffffff) ^ 0xaad26b49
        s1 = (seed >> 32) ^ 0xf7dcefdd
        t0 = (1099087573 * s0) & 0xffffffff
        t1 = (2591861531 * s1) & 0xffffffff
        x0 = np.array([(123456789 + t0) & 0xffffffff, 362436069 ^ t0, (521288629 + t1) & 0xffffffff, 88675123 ^ t1,
                       (5783321 + t0) & 0xffffffff], dtype=np.uint32)
        d0 = np.uint32((6615241 + t1 + t0) & 0xffffffff)
        for pixel in (0, 1, 479999, 2073599, 8294399):
            x, d = po.xorwow_skip_raw(x0[None, :], [d0], [pixel], [0])
            st = po.curand_states(pixel, 1, seed)
            assert int(st["d"][0]) == int(d[0]) and st["v"][0].tolist() == x[0].tolist(), (seed, pixel)


def test_product_jump_matrices_equal_rocrand(g):
    """The table the product's rng_init_kernel walks (csrc/crt_api.cpp: jump_table, matrix k = T^(2^(67+k))), row by row:
    matrix 0 = one subsequence, matrix 32 = 2^32 subsequences, against rocRAND's precomputed matrices applied to single-bit
    states; and every matrix applied to the fixture's random states against rocRAND's discard_subsequence(2^k).  Host
    only (the device walk of the same table is held to the oracle by tests/test_gpu_parity.py)."""
    import ctypes as C
    from cudaraytracer_amd import _lib
    crt = _lib.lib()
    rows = np.zeros((160, 5), dtype=np.uint32)

    def matrix(k):
        assert crt.crt_debug_xorwow_jump(k, rows.ctypes.data_as(C.POINTER(C.c_uint))) == 0
        return rows.copy()

    assert crt.crt_debug_xorwow_jump(64, rows.ctypes.data_as(C.POINTER(C.c_uint))) != 0
    assert np.array_equal(matrix(0), g["bits_seq1"])
    assert np.array_equal(matrix(32), g["bits_seq_hi"])
    # subsequence counts with a single binary digit: d << 2p with d = 1 (k = 2p) and d = 2 (k = 2p + 1)
    sub = g["subsequence"]
    seen = set()
    for i in np.nonzero((g["offset"] == 0) & (sub > 0) & ((sub & (sub - np.uint64(1))) == 0))[0]:
        k = int(sub[i]).bit_length() - 1
        m = matrix(k)
        v = np.zeros(5, dtype=np.uint32)
        for w in range(5):
            for b in range(32):
                if (int(g["x"][i, w]) >> b) & 1:
                    v ^= m[w * 32 + b]
        assert np.array_equal(v, g["x_out"][i]), k
        seen.add(k)
    assert seen == set(range(64))
