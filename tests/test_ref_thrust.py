"""f1's tree shape, PINNED: the sort rule against a real Thrust.

BVHNode's constructor sorts every span with `thrust::sort(l, l + n, BoxCompare(axis))` inside one device thread
(hitable/bvh.h:84-92).  BoxCompare answers true on equal keys (bvh.h:36-41), so the resulting order is a property of the
algorithm — Thrust's sequential stable merge sort (no relocatable device code in CudaTest.vcxproj, so no CDP dispatch).
Thrust is not in the reference tree; the image ships the same library as rocThrust, and oracle/ref_thrust_sort.hip calls
its thrust::sort itself.  Held to it, order for order:
  * the committed fixture tests/golden/thrust_sort_orders.npz — produced by that call inside ONE device thread on an
    MI355X (tests/make_thrust_fixtures.py --device), re-checked on the GPU box with the sequential policy and with the
    reference's policy-free spelling, and against the host call wherever the library exists;
  * the oracle's restatement (oracle/crt_oracle.c: thrust_seq_sort) and the product's (csrc/crt_bvh.cpp:
    thrust_sequential_sort, through crt_debug_thrust_sort) — against the fixture, everywhere;
  * whole trees: leaf orders of keyed scenes full of ties, oracle and product, against the recursion
    `sort the span, split n/2 | n - n/2` (bvh.h:86-112) driven by the pinned sort."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import pyref

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "thrust_sort_orders.npz")


@pytest.fixture(scope="module")
def cases():
    g = np.load(GOLDEN)
    off = g["offsets"]
    return [(str(name), g["keys"][off[k]:off[k + 1]], g["starts"][off[k]:off[k + 1]].astype(np.int32),
             g["orders"][off[k]:off[k + 1]].astype(np.int32)) for k, name in enumerate(g["names"])]


def product_sort(crt, key, start):
    order = np.ascontiguousarray(start, dtype=np.int32).copy()
    key = np.ascontiguousarray(key, dtype=np.float32)
    rc = crt.crt_debug_thrust_sort(len(order), key.ctypes.data_as(C.POINTER(C.c_float)), order.ctypes.data_as(C.POINTER(C.c_int)))
    assert rc == 0
    return order


def test_fixture_covers_the_partition_boundaries(cases):
    sizes = {len(k) for _, k, _, _ in cases}
    assert {1, 2, 31, 32, 33, 64, 65, 96, 97, 128, 129, 200, 257, 513} <= sizes
    assert any(n.startswith("nan_") for n, *_ in cases) and any(n.startswith("zeros_") for n, *_ in cases)
    for name, key, start, order in cases:
        assert sorted(order.tolist()) == list(range(len(key))) == sorted(start.tolist()), name


def test_fixture_is_the_device_run():
    assert str(np.load(GOLDEN)["source"]) == "device thread"


def test_host_call_of_rocthrust_gives_the_fixture_orders(cases):
    """The fixture was sorted inside ONE device thread on an MI355X (make_thrust_fixtures.py --device).  The same library
    called on the HOST takes Thrust's recursive merge sort (halves + inplace_merge; the iterative one is for device
    code: stable_merge_sort.inl, NV_IF_TARGET).  With BoxCompare's rule the two agree on every key array of ordinary
    numbers — ties included — and differ only where the rule is not even transitive: NaN keys in spans longer than one
    32-element partition (measured: nan_33, nan_65, nan_200).  So the host call is a second witness for everything a
    mesh can hold, and the device run is the authority."""
    if not pyref.have_thrust():
        pytest.skip("oracle/_ref/libref_thrust_sort.so not built (make -C oracle ref)")
    differ = []
    for name, key, start, order in cases:
        if not np.array_equal(pyref.thrust_sort_host(key, start), order):
            differ.append(name)
    assert all(n.startswith("nan_") for n in differ), differ
    # equal keys come out reversed, whatever the size (what the hand-derived trees of tests/test_bvh.py assume)
    for n in (2, 5, 32, 33, 70, 200):
        assert pyref.thrust_sort_host(np.full(n, 3.0, np.float32)).tolist() == list(range(n))[::-1]


def test_oracle_sort_equals_rocthrust(oracle, cases):
    for name, key, start, order in cases:
        assert np.array_equal(oracle.thrust_sort(key, start), order), name


def test_product_sort_equals_rocthrust(crt_built, cases):
    for name, key, start, order in cases:
        assert np.array_equal(product_sort(crt_built, key, start), order), name


def test_restatements_equal_rocthrust_on_fresh_keys(oracle, crt_built):
    if not pyref.have_thrust():
        pytest.skip("oracle/_ref/libref_thrust_sort.so not built")
    rng = np.random.default_rng(7)
    for trial in range(400):
        n = int(rng.integers(1, 700))
        key = (rng.integers(0, int(rng.integers(1, 9)), n).astype(np.float32) if trial % 2 else rng.standard_normal(n).astype(np.float32))
        start = rng.permutation(n).astype(np.int32)
        want = pyref.thrust_sort_host(key, start)
        assert np.array_equal(oracle.thrust_sort(key, start), want), (trial, n)
        assert np.array_equal(product_sort(crt_built, key, start), want), (trial, n)


# ---- whole trees ---------------------------------------------------------------------------------------------------
def keyed_triangles(keys):
    """Triangle i has box minimum (k_i, k_i, k_i) — the same key on whichever axis a node draws — and a thick box."""
    pts, idx = [], []
    for i, k in enumerate(keys):
        pts += [(k, k, k), (k + 1.0, k + 2.0, k), (k, k + 1.0, k + 3.0)]
        idx.append((3 * i, 3 * i + 1, 3 * i + 2))
    return np.array(pts, dtype=np.float32), np.array(idx, dtype=np.float32)


def tree_order(sort, key, ids):
    """BVHNode's constructor (bvh.h:84-112): sort the span, leaves of 1 or 2, else n/2 | n - n/2."""
    ids = sort(key, np.asarray(ids, dtype=np.int32))
    n = len(ids)
    if n <= 2:
        return ids.tolist()
    return tree_order(sort, key, ids[:n // 2]) + tree_order(sort, key, ids[n // 2:])


def tree_keys():
    rng = np.random.default_rng(11)
    out = [("ties3_200", rng.integers(0, 3, 200).astype(np.float32)), ("ties7_131", rng.integers(0, 7, 131).astype(np.float32)),
           ("equal_97", np.full(97, 2.0, np.float32)), ("distinct_64", rng.permutation(64).astype(np.float32)),
           ("ties2_33", rng.integers(0, 2, 33).astype(np.float32))]
    return out


def pinned_sort(oracle):
    # the real library where it exists; elsewhere the oracle's restatement, which the fixture pins
    return pyref.thrust_sort_host if pyref.have_thrust() else oracle.thrust_sort


@pytest.mark.parametrize("name,keys", tree_keys(), ids=[t[0] for t in tree_keys()])
def test_oracle_tree_leaf_order_follows_the_pinned_sort(oracle, name, keys):
    pts, idx = keyed_triangles(keys)
    w = oracle.World(pts, idx, None, 0)
    bvh = oracle.Bvh(w, oracle.fresh_states(8, 8))
    want = tree_order(pinned_sort(oracle), keys + np.float32(0.0), np.arange(len(keys)))
    assert bvh.order().tolist() == want


@pytest.mark.gpu
@pytest.mark.parametrize("name,keys", tree_keys(), ids=[t[0] for t in tree_keys()])
def test_gpu_product_tree_leaf_order_follows_the_pinned_sort(gpu, oracle, name, keys):
    from cudaraytracer_amd import api
    pts, idx = keyed_triangles(keys)
    scene = api.Scene.from_mesh(pts, idx, None, 0)
    scene.build_bvh(None, 0)
    _, _, slot_triangle = scene.bvh_export()
    want = tree_order(pinned_sort(oracle), keys, np.arange(len(keys)))
    assert np.asarray(slot_triangle).tolist() == want


@pytest.mark.gpu
def test_gpu_thrust_sort_inside_one_device_thread_gives_the_fixture_orders(gpu, cases):
    """What the reference does: thrust::sort called from ONE device thread (create_BVH<<<1,1>>>, createScene.h:42-44).
    mode 0 names the sequential policy, mode 1 is the reference's policy-free call (bvh.h:86)."""
    if not pyref.have_thrust():
        pytest.fail("oracle/_ref/libref_thrust_sort.so did not travel to the GPU box")
    for name, key, start, order in cases:
        if len(key) > 260 and not name.startswith(("few_", "perm_")):
            continue                                  # (one thread sorts: keep the long cases to two patterns)
        for mode in (0, 1):
            got = pyref.thrust_sort_device(key, start, mode)
            assert np.array_equal(got, order), (name, mode)
