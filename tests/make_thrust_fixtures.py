"""Generates tests/golden/thrust_sort_orders.npz: the orders rocThrust's REAL thrust::sort(thrust::seq, ...) produces
with BoxCompare's rule (hitable/bvh.h:36-41: true on equal keys, not a strict weak ordering) on key arrays full of
ties.  Run where oracle/_ref/libref_thrust_sort.so exists (make -C oracle ref); the fixture lets machines without
rocThrust hold the two restatements (oracle/crt_oracle.c: thrust_seq_sort, csrc/crt_bvh.cpp: thrust_sequential_sort)
to the real library.  Data only: keys (inputs) and orders (outputs)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyref  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "thrust_sort_orders.npz")
# --device [out.npz]: sort inside ONE device thread on the GPU box (what BVHNode's constructor does).  Thrust's sequential
# backend is the ITERATIVE merge sort in device code and the RECURSIVE one (halves, inplace_merge) in host code
# (stable_merge_sort.inl: NV_IF_TARGET): with BoxCompare's rule they agree on ordinary keys and differ where the rule is
# not even transitive (NaN keys, inf - inf).  The committed fixture is the device run.
DEVICE = "--device" in sys.argv
if DEVICE and len(sys.argv) > sys.argv.index("--device") + 1:
    OUT = sys.argv[sys.argv.index("--device") + 1]


def cases(seed=20261005):
    """(name, keys, start order) — sizes around the 32-element insertion-sort partition and its merge passes."""
    rng = np.random.default_rng(seed)
    sizes = list(range(1, 70)) + [95, 96, 97, 127, 128, 129, 130, 160, 191, 192, 193, 200, 255, 256, 257, 513]
    out = []
    for n in sizes:
        ident = np.arange(n, dtype=np.int32)
        out.append(("equal_%d" % n, np.full(n, 0.5, np.float32), ident))
        out.append(("two_%d" % n, rng.integers(0, 2, n).astype(np.float32), ident))
        out.append(("few_%d" % n, rng.integers(-3, 4, n).astype(np.float32) * np.float32(0.25), ident))
        out.append(("rand_%d" % n, rng.standard_normal(n).astype(np.float32), ident))
        out.append(("asc_%d" % n, np.arange(n, dtype=np.float32), ident))
        out.append(("desc_%d" % n, np.arange(n, dtype=np.float32)[::-1].copy(), ident))
        k = rng.integers(0, 5, n).astype(np.float32)
        out.append(("perm_%d" % n, k, rng.permutation(n).astype(np.int32)))     # a span that earlier sorts already permuted
        z = rng.integers(0, 3, n).astype(np.float32) - np.float32(1.0)
        z[z == 0] = np.where(rng.integers(0, 2, int((z == 0).sum())) == 0, np.float32(0.0), np.float32(-0.0))
        out.append(("zeros_%d" % n, z, ident))                                   # +0 / -0 are equal keys
        if n in (5, 33, 64, 65, 200):
            kn = rng.integers(0, 4, n).astype(np.float32)
            kn[rng.integers(0, n, max(1, n // 8))] = np.nan                      # NaN: `val1 - val2 < 0.0` is false -> true
            out.append(("nan_%d" % n, kn, ident))
            ki = rng.integers(0, 3, n).astype(np.float32)
            ki[rng.integers(0, n, max(1, n // 8))] = np.inf                      # inf - inf = NaN
            out.append(("inf_%d" % n, ki, ident))
    return out


def main():
    names, keys, starts, orders, offsets = [], [], [], [], [0]
    for name, key, start in cases():
        order = pyref.thrust_sort_device(key, start, 0) if DEVICE else pyref.thrust_sort_host(key, start)
        assert sorted(order.tolist()) == list(range(len(key)))
        names.append(name)
        keys.append(key)
        starts.append(start.astype(np.int16))
        orders.append(order.astype(np.int16))
        offsets.append(offsets[-1] + len(key))
    # one flat array per field: case k is [offsets[k], offsets[k + 1])
    np.savez_compressed(OUT, source=np.array("device thread" if DEVICE else "host call"), names=np.array(names), offsets=np.array(offsets, np.int32), keys=np.concatenate(keys),
                        starts=np.concatenate(starts), orders=np.concatenate(orders))
    print("wrote %s: %d cases, %d bytes" % (OUT, len(names), os.path.getsize(OUT)))


if __name__ == "__main__":
    main()
