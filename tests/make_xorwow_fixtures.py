"""Generates tests/golden/rocrand_xorwow_vectors.npz: what rocRAND's OWN XORWOW engine (rocrand_device::xorwow_engine,
/opt/rocm/include/rocrand/rocrand_xorwow.h, jump matrices precomputed by its authors) does to RAW states — `next()`
sequences, `discard_subsequence` (k * 2^67 steps) and `discard` (offsets).  Run where oracle/_ref/libref_rocrand_xorwow.so
exists (make -C oracle ref); the fixture lets any machine hold the oracle's restated step and skips to the library.
Data only: input states, skip counts, output states and outputs."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyref  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden", "rocrand_xorwow_vectors.npz")


def inputs(seed=20261005):
    """Raw states and skip counts.  Subsequences: what SetCurandState uses (the pixel index: first pixels, the last pixel
    of every BASELINE frame size, every base-4 digit position of cuRAND's two-bits-at-a-time walk alone and with all
    digits set) and full 64-bit values; offsets likewise."""
    rng = np.random.default_rng(seed)
    # the state curand_init(0, ., .) starts from (seed 0 through the published scrambling) and cuRAND's unscrambled constants
    seeded0 = [123456789 + 1099087573 * 0xaad26b49, 362436069 ^ (1099087573 * 0xaad26b49), 521288629 + 2591861531 * 0xf7dcefdd,
               88675123 ^ (2591861531 * 0xf7dcefdd), 5783321 + 1099087573 * 0xaad26b49]
    seeded0 = [v & 0xffffffff for v in seeded0]
    plain = [123456789, 362436069, 521288629, 88675123, 5783321]
    subs = [0, 1, 2, 3, 4, 5, 63, 64, 65, 255, 256, 65535, 65536, 256 * 256 - 1, 800 * 600 - 1, 1024 * 512 - 1, 1920 * 1080 - 1,
            1920 * 1080, 3840 * 2160 - 1, 3840 * 2160, 2 ** 31 - 1, 2 ** 32 - 1, 2 ** 32, 2 ** 63, 2 ** 64 - 1]
    subs += [d << (2 * p) for p in range(0, 32) for d in (1, 2, 3)]
    subs += [int(v) for v in rng.integers(0, 2 ** 63, 60, dtype=np.uint64)] + [int(v) for v in rng.integers(0, 3840 * 2160, 200)]
    offs = [0] * len(subs)
    # offsets alone, and both together
    o_only = [1, 2, 3, 4, 5, 7, 64, 1000, 2 ** 20 + 3, 2 ** 32 - 1, 2 ** 32, 2 ** 32 + 1, 2 ** 40 + 12345, 2 ** 63 + 1, 2 ** 64 - 1]
    o_only += [d << (2 * p) for p in range(0, 32) for d in (1, 2, 3)]
    subs += [0] * len(o_only)
    offs += o_only
    both_s = [int(v) for v in rng.integers(0, 3840 * 2160, 100)] + [int(v) for v in rng.integers(0, 2 ** 63, 40, dtype=np.uint64)]
    both_o = [int(v) for v in rng.integers(0, 2 ** 20, 100)] + [int(v) for v in rng.integers(0, 2 ** 63, 40, dtype=np.uint64)]
    subs += both_s
    offs += both_o
    n = len(subs)
    x = rng.integers(0, 2 ** 32, (n, 5), dtype=np.uint64).astype(np.uint32)
    d = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    x[0::3] = np.array(seeded0, dtype=np.uint32)          # a third from the state seed 0 scrambles to
    d[0::3] = np.uint32((6615241 + 2591861531 * 0xf7dcefdd + 1099087573 * 0xaad26b49) & 0xffffffff)
    x[1::7] = np.array(plain, dtype=np.uint32)
    # single-bit states: every row of the one-step matrix is read
    bits = np.zeros((160, 5), dtype=np.uint32)
    for b in range(160):
        bits[b, b // 32] = np.uint32(1) << np.uint32(b % 32)
    return (x, d, np.array(subs, dtype=np.uint64), np.array(offs, dtype=np.uint64)), bits


def main():
    if not pyref.have_xorwow():
        raise SystemExit("oracle/_ref/libref_rocrand_xorwow.so is not built: make -C oracle ref")
    (x, d, sub, off), bits = inputs()
    x_out, d_out = pyref.rocrand_xorwow_skip(x, d, sub, off)
    # single-bit states through one subsequence and through the digit-16 subsequence matrix (4^16 subsequences)
    one = np.ones(160, dtype=np.uint64)
    zero = np.zeros(160, dtype=np.uint64)
    bits_seq1, _ = pyref.rocrand_xorwow_skip(bits, np.zeros(160, np.uint32), one, zero)
    bits_seq_hi, _ = pyref.rocrand_xorwow_skip(bits, np.zeros(160, np.uint32), one << np.uint64(32), zero)
    bits_off1, _ = pyref.rocrand_xorwow_skip(bits, np.zeros(160, np.uint32), zero, one)
    # step sequences from 8 of the states
    seq_out, seq_x, seq_d = [], [], []
    for i in range(8):
        o, xs, ds = pyref.rocrand_xorwow_next(x[i], d[i], 257)
        seq_out.append(o)
        seq_x.append(xs)
        seq_d.append(ds)
    np.savez_compressed(GOLDEN, source=np.array("rocRAND xorwow_engine, host call, %s" % "rocrand_xorwow.h"),
                        x=x, d=d, subsequence=sub, offset=off, x_out=x_out, d_out=d_out,
                        bits=bits, bits_seq1=bits_seq1, bits_seq_hi=bits_seq_hi, bits_off1=bits_off1,
                        seq_out=np.array(seq_out), seq_x=np.array(seq_x), seq_d=np.array(seq_d, dtype=np.uint32))
    print("wrote %s: %d skip cases, %d bytes" % (GOLDEN, len(d), os.path.getsize(GOLDEN)))


if __name__ == "__main__":
    main()
