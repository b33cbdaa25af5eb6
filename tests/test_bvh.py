"""BVH world (hitable/bvh.h, core/aabb.h): CPU tests of the oracle restatement and of the host
logic, GPU tests (marked) of the product against the oracle through the C ABI.

PARITY UNPINNED at the thrust::sort boundary: the reference's comparator (BoxCompare, bvh.h:9-45)
returns true on equal keys, so the order of a span is whatever Thrust's sequential merge sort does
with it; the oracle and the product both restate that algorithm, independently (C and C++)."""
import os

import numpy as np
import pytest

from cudaraytracer_amd import _lib, api, scenes

BVH_GOLDEN = {
    "bvh_sphere1000_96x54x4": lambda: scenes.displaced_sphere(20, 25),
    "bvh_soup96_48x48x4": lambda: scenes.random_soup(96, 7),
    "bvh_cornell_100x75x8": scenes.cornell32,
}


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def oracle_scene(oracle, sc, nx, ny):
    w = oracle.World(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
    c = sc["camera"]
    cam = oracle.make_camera(c["lookfrom"], c["lookat"], (0, 1, 0), c["vfov"], np.float32(nx) / np.float32(ny),
                             c["aperture"], c["focus"])
    return w, cam


# ------------------------------------------------------------------ oracle (CPU)
@pytest.mark.parametrize("name", sorted(BVH_GOLDEN))
def test_oracle_bvh_matches_golden(oracle, golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    nx, ny, spp = int(g["nx"]), int(g["ny"]), int(g["spp"])
    w, cam = oracle_scene(oracle, BVH_GOLDEN[name](), nx, ny)
    states = oracle.fresh_states(nx, ny)
    bvh = oracle.Bvh(w, states)
    nodes = bvh.nodes()
    assert np.array_equal(bits(nodes["box"]), bits(g["boxes"]))
    for k, f in [("skip", "skip"), ("first", "first"), ("count", "count"), ("is_leaf", "isLeaf"), ("depth", "depth"), ("axis", "axis")]:
        assert np.array_equal(nodes[f], g[k]), k
    assert np.array_equal(bvh.order(), g["order"])
    buf, states = oracle.render_bvh(bvh, cam, nx, ny, spp, states)
    assert np.array_equal(bits(buf), bits(g["rgb"]))


def test_oracle_tree_shape(oracle):
    """Median split n/2 | n - n/2, leaves of 1 or 2 triangles, every triangle in exactly one leaf,
    one axis draw per node from stream 0, boxes contain their children."""
    sc = scenes.displaced_sphere(12, 9)          # 216 triangles
    w, _ = oracle_scene(oracle, sc, 8, 8)
    states = oracle.fresh_states(8, 8)
    s0 = states[0].copy()
    bvh = oracle.Bvh(w, states)
    nodes, order = bvh.nodes(), bvh.order()
    assert sorted(order.tolist()) == list(range(216))
    leaves = nodes[nodes["isLeaf"] == 1]
    assert set(leaves["count"].tolist()) <= {1, 2} and leaves["count"].sum() == 216
    assert nodes[0]["count"] == 216 and nodes[0]["skip"] == len(nodes)
    for i, nd in enumerate(nodes):
        if not nd["isLeaf"]:
            l, r = nodes[nd["left"]], nodes[nd["right"]]
            assert nd["left"] == i + 1 and nd["right"] == l["skip"]
            assert l["count"] == nd["count"] // 2 and r["count"] == nd["count"] - nd["count"] // 2
            assert (nd["box"][:3] <= np.minimum(l["box"][:3], r["box"][:3])).all()
            assert (nd["box"][3:] >= np.maximum(l["box"][3:], r["box"][3:])).all()
    # stream 0 advanced by exactly one draw per node
    probe = np.zeros(1, dtype=oracle.STATE_DTYPE)
    probe[0] = s0
    import ctypes as C
    st = oracle.CurandState.from_buffer(probe)
    for _ in range(len(nodes)):
        oracle.lib().o_curand(C.byref(st))
    assert probe[0] == states[0]


# ---- hand-derived trees: expected leaf orders that come from READING Thrust's sequential sort, not from running any
# restatement of it.  thrust::sort inside a device thread = sequential iterative stable merge sort
# (thrust/system/detail/sequential/stable_merge_sort.inl: insertion_sort per 32 elements, then merge of adjacent runs);
# the comparator BoxCompare (bvh.h:9-45) is comp(a, b) = !(key(a) - key(b) < 0) = key(a) >= key(b): "a goes before b when
# its key is not smaller" — descending, and TRUE on equal keys.
#   insertion_sort (insertion_sort.h): for i = 1..n-1: tmp = x[i]; if comp(tmp, x[0]) move x[0..i) up one and put tmp first;
#     else shift x[k] up while comp(tmp, x[k]) from k = i-1 downwards, drop tmp behind the first k that fails.
#   merge (merge.inl): while both runs hold elements: take from run 2 if comp(*run2, *run1) else from run 1; append rests.
# Every triangle below has the same box minimum on all three axes, so the drawn axes (stream 0) do not matter.
#
# (A) keys 1, 1, 2, 1, 0 (triangles 0..4), worked by hand:
#   root sort:  [0 1 2 3 4] -> i=1: comp(1;0) 1>=1 true  -> [1 0 2 3 4]
#                           -> i=2: comp(2;1) 2>=1 true  -> [2 1 0 3 4]
#                           -> i=3: comp(3;2) 1>=2 false; comp(3;0) 1>=1 true, comp(3;1) 1>=1 true, comp(3;2) false -> [2 3 1 0 4]
#                           -> i=4: comp(4;2) 0>=2 false; comp(4;0) 0>=1 false                                       -> [2 3 1 0 4]
#   left  = first 5/2 = 2: [2 3], keys 2, 1: i=1: comp(3;2) false, comp(3;2) false -> [2 3]: leaf (2, 3)
#   right = [1 0 4], keys 1, 1, 0: i=1: comp(0;1) 1>=1 true -> [0 1 4]; i=2: comp(4;0) false, comp(4;1) false -> [0 1 4]
#     left = [0]: leaf (0);  right = [1 4], keys 1, 0: stays: leaf (1, 4)
#   leaves in preorder: (2, 3) (0) (1, 4); 5 nodes.
# (B) n equal keys: insertion sort REVERSES a run (every tmp satisfies comp(tmp, first)); merge takes all of run 2, then
#   run 1 — so two reversed runs [31..0], [n-1..32] merge to [n-1..0], and so on up: every node reverses its span, then
#   splits n/2 | n - n/2.
HAND_TREE_A_KEYS = [1.0, 1.0, 2.0, 1.0, 0.0]
HAND_TREE_A_ORDER = [2, 3, 0, 1, 4]


def equal_key_order(ids):
    ids = ids[::-1]
    if len(ids) <= 2:
        return ids
    return equal_key_order(ids[:len(ids) // 2]) + equal_key_order(ids[len(ids) // 2:])


def keyed_triangles(keys):
    """Triangle i has box minimum (k_i, k_i, k_i) and a box that is thick on every axis."""
    pts, idx = [], []
    for i, k in enumerate(keys):
        pts += [(k, k, k), (k + 1.0, k + 2.0, k), (k, k + 1.0, k + 3.0)]
        idx.append((3 * i, 3 * i + 1, 3 * i + 2))
    return dict(points=np.array(pts, dtype=np.float32), idx=np.array(idx, dtype=np.float32), normals=None, index_base=0,
                camera=dict(scenes.REFERENCE_CAMERA))


HAND_TREES = [("A", HAND_TREE_A_KEYS, HAND_TREE_A_ORDER),
              ("B5", [3.0] * 5, equal_key_order(list(range(5)))),
              ("B70", [0.5] * 70, equal_key_order(list(range(70)))),       # three runs of 32, 32, 6: two merge passes
              ("B33", [7.0] * 33, equal_key_order(list(range(33))))]


@pytest.mark.parametrize("name,keys,order", HAND_TREES, ids=[t[0] for t in HAND_TREES])
def test_oracle_leaf_order_equals_the_hand_derived_tree(oracle, name, keys, order):
    # (B) for n = 5 by hand: root reverses to [4 3 2 1 0]; left [4 3] reverses to the leaf (3, 4); right [2 1 0] reverses to [0 1 2]:
    # leaf (0), then [1 2] reverses to the leaf (2, 1)
    assert equal_key_order(list(range(5))) == [3, 4, 0, 2, 1]
    w, _ = oracle_scene(oracle, keyed_triangles(keys), 8, 8)
    bvh = oracle.Bvh(w, oracle.fresh_states(8, 8))
    assert bvh.order().tolist() == order, name
    if name == "A":
        nodes = bvh.nodes()
        assert len(nodes) == 5 and nodes["isLeaf"].tolist() == [0, 1, 0, 1, 1] and nodes["count"][nodes["isLeaf"] == 1].tolist() == [2, 1, 2]


def test_oracle_bvh_semantics_differ_from_the_list_where_the_reference_does(oracle):
    """(a) a triangle behind the eye: the list accepts its negative t (no range test,
    triangle.h:92-99), the BVH never reaches it — AABB::hit starts from t_min = 0.001 (aabb.h:30-43).
    (b) coincident duplicates in different leaves: the list keeps the lowest index (strict <), the
    tree's merge `left.t < right.t ? left : right` keeps the right-hand leaf (bvh.h:172-179)."""
    # tilted: an axis-aligned triangle has a flat box, which AABB::hit never reports (t_max <= t_min)
    tri = np.array([[-1, -1, 0.25], [1, -1, 0], [0, 1, -0.25]], dtype=np.float32)
    behind = tri + np.array([0, 0, 5], dtype=np.float32)
    front = tri + np.array([0, 0, -5], dtype=np.float32)
    far = tri * 3 + np.array([40, 0, -5], dtype=np.float32)
    far2 = tri * 3 + np.array([-40, 0, -5], dtype=np.float32)

    def world(tris):
        pts = np.concatenate(tris).astype(np.float32)
        idx = np.arange(len(pts), dtype=np.float32).reshape(-1, 3)
        nrm = np.tile(np.array([[0, 0, -1]], dtype=np.float32), (len(idx), 1))
        return oracle.World(pts, idx, nrm, 0)

    flat = world([np.array([[-1, -1, -5], [1, -1, -5], [0, 1, -5]], dtype=np.float32)])
    assert flat.hit((0.0, 0.0, 0.0), (0.0, 0.0, -1.0))[0]
    assert not oracle.Bvh(flat, oracle.fresh_states(4, 4)).hit((0.0, 0.0, 0.0), (0.0, 0.0, -1.0))[0]   # flat box: never hit

    org, d = (0.0, 0.0, 0.0), (0.0, 0.0, -1.0)
    w = world([behind, far, far2])
    f_list, t_list, i_list, _ = w.hit(org, d)
    assert f_list and t_list < 0 and i_list == 0
    states = oracle.fresh_states(4, 4)
    f_bvh, _, _, _ = oracle.Bvh(w, states).hit(org, d)
    assert not f_bvh
    w = world([front, front, front, front, front])
    assert w.hit(org, d)[2] == 0
    bvh = oracle.Bvh(w, oracle.fresh_states(4, 4))
    f, t, i, _ = bvh.hit(org, d)
    order, nodes = bvh.order(), bvh.nodes()
    last_leaf = nodes[nodes["isLeaf"] == 1][-1]
    assert f and abs(t - 5.0) < 0.2 and i == order[last_leaf["first"]]      # last leaf, its first triangle


# ------------------------------------------------------------------ host logic / ABI (CPU)
def test_timing_csv_format(crt_built, tmp_path):
    p = str(tmp_path / "output.csv")
    api.writeCSV(p, [["frame", "rendering", "update", "build"], ["", "", "", "0.125"],
                     ["0", "1.5", "0.25", ""], ["1", "2", "0.5", ""]])
    assert open(p).read() == "frame,rendering,update,build\n,,,0.125000\n0,1.500000,0.250000,\n1,2.000000,0.500000,\n"
    with pytest.raises(_lib.CrtError):
        api.writeCSV(str(tmp_path / "no_such_dir" / "x.csv"), [["frame"], ["", "", "", "0"]])


def test_bvh_entry_points_fail_loudly_without_device(crt_built):
    if crt_built.crt_device_count() > 0:
        pytest.skip("a HIP device is present")
    assert crt_built.crt_scene_build_bvh(None, None, 0) == _lib.CRT_ERR_INVALID
    assert crt_built.crt_scene_refit_bvh(None, None) == _lib.CRT_ERR_INVALID
    assert crt_built.crt_scene_bvh_node_count(None) == _lib.CRT_ERR_INVALID


# ------------------------------------------------------------------ product vs oracle (GPU)
def make(sc, nx, ny):
    scene = api.Scene.from_mesh(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
    c = sc["camera"]
    return scene, api.create_camera(nx, ny, c["lookfrom"], c["lookat"], c["focus"], c["aperture"], c["vfov"])


def hip_bvh_frames(gpu, scene, cam, nx, ny, spp, frames=1, variant=0, build=True, poses=None):
    st = api.SetCurandState(nx, ny, 0)
    if build:
        api.create_BVHfromList(scene, st)
    d = gpu.zeros(ny * nx * 3, dtype=gpu.float32, device="cuda")
    prev = _lib.lib().crt_set_kernel_variant(variant)
    outs = []
    try:
        for f in range(frames):
            if poses is not None and f > 0:
                scene.update_pose(poses[f], stream=gpu.cuda.current_stream())
                api.Update_BVH(scene, gpu.cuda.current_stream())
            api.render_rows(scene, cam, st, nx, ny, spp, 0, ny, d, frame_index=f, stream=gpu.cuda.current_stream(), bvh=True)
            gpu.cuda.synchronize()
            outs.append(d.cpu().numpy().reshape(ny, nx, 3).copy())
    finally:
        _lib.lib().crt_set_kernel_variant(prev)
    return outs, st.export()


def assert_same_tree(scene, bvh):
    boxes, meta, slots = scene.bvh_export()
    nodes = bvh.nodes()
    assert len(boxes) == len(nodes)
    assert np.array_equal(bits(boxes), bits(nodes["box"]))
    assert np.array_equal(meta[:, 0], nodes["skip"]) and np.array_equal(meta[:, 3], nodes["depth"])
    leaf = nodes["isLeaf"] == 1
    assert np.array_equal(meta[:, 2], np.where(leaf, nodes["count"], 0))
    assert np.array_equal(meta[leaf, 1], nodes["first"][leaf])
    assert np.array_equal(slots, bvh.order())


def assert_nested(scene):
    """Every inner node's box contains its children's boxes EXACTLY (compared as floats, bounds equal or outside): what lets the
    tiled kernels' BVH mode test a winner's leaf box instead of its chain (crt_bvh_kernels.hpp, "the chain of boxes of a winner")."""
    boxes, meta, _ = scene.bvh_export()
    inner = np.where(meta[:, 2] == 0)[0]
    assert len(inner) > 0
    left = inner + 1
    right = meta[left, 0]
    for child in (left, right):
        assert (boxes[inner, :3] <= boxes[child, :3]).all() and (boxes[inner, 3:] >= boxes[child, 3:]).all()
    # ... and tightly: every bound of a parent is the bound of one of its children
    assert np.array_equal(boxes[inner, :3], np.minimum(boxes[left, :3], boxes[right, :3]))
    assert np.array_equal(boxes[inner, 3:], np.maximum(boxes[left, 3:], boxes[right, 3:]))


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 1, 5, 6, 7, 9, 10])
@pytest.mark.parametrize("name", sorted(BVH_GOLDEN))
def test_gpu_bvh_matches_golden(gpu, golden_dir, name, variant):
    """Tree (boxes, structure, leaf order) and frame equal the oracle's: primary rays through the
    tiled kernels (variant 0: tile lists + the chain of boxes for every pixel's winner), the per-ray
    tree walk (1) and its two forced general modes (5, 6); stream 0 is left where the reference leaves it."""
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    nx, ny, spp = int(g["nx"]), int(g["ny"]), int(g["spp"])
    scene, cam = make(BVH_GOLDEN[name](), nx, ny)
    (rgb,), states = hip_bvh_frames(gpu, scene, cam, nx, ny, spp, variant=variant)
    boxes, meta, slots = scene.bvh_export()
    if variant == 0 and len(boxes) > 1:
        assert_nested(scene)
    assert np.array_equal(bits(boxes), bits(g["boxes"]))
    assert np.array_equal(meta[:, 0], g["skip"]) and np.array_equal(meta[:, 3], g["depth"])
    assert np.array_equal(slots, g["order"])
    assert np.array_equal(bits(rgb), bits(g["rgb"])), "max |diff| %g" % np.nanmax(np.abs(rgb - g["rgb"]))
    assert np.array_equal(states[1:2048], g["states_after"][1:])


@pytest.mark.gpu
@pytest.mark.parametrize("seed,ntri,nx,ny,spp", [(1, 64, 40, 24, 3), (2, 200, 33, 17, 2), (3, 17, 16, 16, 8), (4, 500, 64, 40, 2),
                                                   (5, 1, 16, 16, 2), (6, 2, 16, 16, 2), (7, 3, 16, 16, 2)])
def test_gpu_bvh_random_soup_matches_oracle(gpu, oracle, seed, ntri, nx, ny, spp):
    """Adversarial soups: triangles behind and through the eye, duplicates, degenerate triangles
    (NaN normals), 1-, 2- and 3-triangle lists."""
    sc = scenes.random_soup(ntri, seed)
    scene, cam = make(sc, nx, ny)
    w, ocam = oracle_scene(oracle, sc, nx, ny)
    states = oracle.fresh_states(nx, ny)
    bvh = oracle.Bvh(w, states)
    ref, states = oracle.render_bvh(bvh, ocam, nx, ny, spp, states)
    for variant in (0, 1, 5, 6, 7, 9):
        (rgb,), st = hip_bvh_frames(gpu, scene, cam, nx, ny, spp, variant=variant)
        assert_same_tree(scene, bvh)
        assert np.array_equal(bits(rgb), bits(ref)), variant
        after = np.concatenate([states["d"][:, None], states["v"]], axis=1)
        assert np.array_equal(st[1:], after[1:])


@pytest.mark.gpu
def test_gpu_bvh_unbounded_scene_takes_the_ieee_loop(gpu, oracle):
    """Coordinates beyond 2^38 and tiny nonzero coordinates: the host selects the literal-division
    kernel; same pixels as the oracle."""
    for scale, shift in [(1.0e13, 0.0), (1.0e-19, 0.0)]:
        sc = scenes.random_soup(40, 21)
        sc["points"] = (sc["points"] * np.float32(scale)).astype(np.float32)
        nx, ny, spp = 24, 16, 2
        scene, cam = make(sc, nx, ny)
        w, ocam = oracle_scene(oracle, sc, nx, ny)
        states = oracle.fresh_states(nx, ny)
        bvh = oracle.Bvh(w, states)
        ref, _ = oracle.render_bvh(bvh, ocam, nx, ny, spp, states)
        (rgb,), _ = hip_bvh_frames(gpu, scene, cam, nx, ny, spp)
        assert np.array_equal(bits(rgb), bits(ref)), scale


@pytest.mark.gpu
def test_gpu_bvh_refit_and_frame_loop(gpu, oracle):
    """renderBVHAnimation (core/render.h:191-237): per frame update_pose, Update_BVH (refit), render;
    the RNG streams continue.  Boxes after the refit equal BVHNode::UpdateBVH's, frames equal the oracle's."""
    sc = scenes.displaced_sphere(16, 20)
    nx, ny, spp = 64, 40, 2
    scene, cam = make(sc, nx, ny)
    w, ocam = oracle_scene(oracle, sc, nx, ny)
    pts0 = sc["points"]
    poses = [pts0, (pts0 * np.float32(1.15) + np.float32(3.0)).astype(np.float32),
             (pts0 * np.array([0.8, 1.2, 1.0], dtype=np.float32)).astype(np.float32)]
    states = oracle.fresh_states(nx, ny)
    bvh = oracle.Bvh(w, states)
    refs = []
    for f in range(3):
        if f > 0:
            w.update_pose(poses[f])
            bvh.update()
        r, states = oracle.render_bvh(bvh, ocam, nx, ny, spp, states, frame=f)
        refs.append(r)
    outs, _ = hip_bvh_frames(gpu, scene, cam, nx, ny, spp, frames=3, poses=poses)
    assert_same_tree(scene, bvh)                                  # after two refits
    assert_nested(scene)
    for f in range(3):
        skip = 0 if f == 0 else 1                                 # pixel 0 of later frames: state[0] race
        assert np.array_equal(bits(outs[f]).reshape(-1, 3)[skip:], bits(refs[f]).reshape(-1, 3)[skip:]), f
    # the Python mirror of the reference's driver, with its timing rows
    scene2, _ = make(sc, nx, ny)
    st = api.SetCurandState(nx, ny, 0)
    data = [["frame", "rendering", "update", "build"]]
    frames = api.renderBVH(nx, ny, spp, 8, 0, 2, scene2, cam, st, data, write_png=False, poses=poses)
    assert len(data) == 5 and data[1][:3] == ["", "", ""] and data[2][0] == "0" and data[4][3] == ""
    assert np.array_equal(bits(frames[0]), bits(refs[0]))


@pytest.mark.gpu
def test_gpu_bvh_strips_and_list_agreement_on_the_bench_mesh(gpu, oracle, tmp_path):
    """The C3 mesh (100 000 triangles) at 1920x1080 x 2 spp through the BVH world: close to the list
    world's frame but NOT identical to it (a ray that grazes an edge shared by two triangles ties:
    the list keeps the lower index, the tree the later leaf; the triangles see the direction
    normalised depth + 3 times instead of twice) — the oracle is the judge: a 64x16 crop on the
    silhouette matches bit for bit, and the 8 strip sets equal the full frame."""
    path, nx, ny, _ = scenes.baseline_obj("C3", str(tmp_path))
    spp = 2
    scene = api.Scene.from_obj(path, scale=1.0)
    cam = api.init_camera(nx, ny)
    (rgb,), st_bvh = hip_bvh_frames(gpu, scene, cam, nx, ny, spp)                     # tiled kernels + chain of boxes
    (walk,), st_walk = hip_bvh_frames(gpu, scene, cam, nx, ny, spp, variant=1)        # per-ray tree walk
    assert np.array_equal(bits(rgb), bits(walk)) and np.array_equal(st_bvh, st_walk)
    st = api.SetCurandState(nx, ny, 0)
    d = gpu.zeros(ny * nx * 3, dtype=gpu.float32, device="cuda")
    api.render_rows(scene, cam, st, nx, ny, spp, 0, ny, d, stream=gpu.cuda.current_stream())
    gpu.cuda.synchronize()
    lst = d.cpu().numpy().reshape(ny, nx, 3)
    differ = (bits(rgb) != bits(lst)).any(axis=2)
    assert differ.mean() < 0.02 and np.abs(rgb - lst).max() < 0.5, (differ.sum(), float(np.abs(rgb - lst).max()))
    from cudaraytracer_amd.dist import partition_strips
    for r in (0, 5):
        strips = partition_strips(ny, 8, r)
        rows = sum(c for _, c in strips)
        s = api.CurandState.for_strips(nx, ny, r, 32, 8, 0)
        p = gpu.zeros(rows * nx * 3, dtype=gpu.float32, device="cuda")
        api.render_strips(scene, cam, s, nx, ny, spp, r, 32, 8, p, stream=gpu.cuda.current_stream(), bvh=True)
        gpu.cuda.synchronize()
        got = p.cpu().numpy().reshape(rows, nx, 3)
        want = np.concatenate([rgb[y0:y0 + c] for y0, c in strips], axis=0)
        skip = 1 if r == 0 else 0            # rank 0's strips start at pixel 0, whose fresh stream was not advanced by a build
        assert np.array_equal(bits(got).reshape(-1, 3)[skip:], bits(want).reshape(-1, 3)[skip:]), r
    pts, idx = oracle.parse_obj(path)
    w = oracle.World(pts, idx, None, 1)
    ocam = oracle.make_camera((0, 100, 1000), (0, 150, 0), (0, 1, 0), 40.0, np.float32(nx) / np.float32(ny), 0.0, 10.0)
    hit = (rgb[..., 0] - rgb[..., 2]) > 0.05
    rows_hit = np.where(hit.any(axis=1))[0]
    r_mid = int((rows_hit[0] + rows_hit[-1]) // 2)
    c_edge = int(np.where(hit[r_mid])[0][0])
    win = (r_mid, r_mid + 16, c_edge - 32, c_edge + 32)
    states = oracle.fresh_states(nx, ny, window=win)
    bvh = oracle.Bvh(w, states)
    ref, _ = oracle.render_bvh(bvh, ocam, nx, ny, spp, states, window=win)
    r0, r1, c0, c1 = win
    assert np.array_equal(bits(rgb[r0:r1, c0:c1]), bits(ref[r0:r1, c0:c1]))


@pytest.mark.gpu
@pytest.mark.parametrize("kind,seed", [("flat", 31), ("dense", 32), ("sliver", 33), ("pile", 34), ("stack", 35)])
def test_gpu_bvh_tiled_equals_tree_walk_on_adversarial_scenes(gpu, oracle, kind, seed):
    """The tiled BVH mode against the per-ray tree walk and the oracle where the two conditions of a
    hit come apart: axis-aligned triangles (flat boxes: passes the triangle test, never the box —
    every winner is excluded and the next one tried), dense overlapping soups (ties, long lists,
    tree-walk tiles), slivers (thin boxes at the limit of the slab test)."""
    rng = np.random.RandomState(seed)
    if kind == "flat":
        n = 120
        c = np.stack([rng.uniform(-3, 3, n), rng.uniform(-3, 3, n), rng.uniform(-12, -4, n)], axis=-1)
        tri = c[:, None, :] + rng.uniform(-1.2, 1.2, (n, 3, 3))
        tri[::2, :, 2] = tri[::2, :1, 2]                      # every other triangle lies in a plane z = const
        tri[1::4, :, 0] = tri[1::4, :1, 0]                    # some in x = const
        tri[5::16, :, 0] = 0.0                                # ... and some in the plane x = 0 THROUGH the eye (0 / d quotients: not dropped at staging)
    elif kind == "dense":
        n = 600
        c = np.stack([rng.uniform(-1.5, 1.5, n), rng.uniform(-1.5, 1.5, n), rng.uniform(-9, -5, n)], axis=-1)
        tri = c[:, None, :] + rng.uniform(-1.0, 1.0, (n, 3, 3))
        tri[50:100] = tri[0:50]                               # coincident duplicates in other leaves
    elif kind == "pile":                                       # > 256 entries in the central tiles: the restaging paths
        n = 2500
        c = np.stack([rng.uniform(-0.6, 0.6, n), rng.uniform(-0.6, 0.6, n), rng.uniform(-9, -5, n)], axis=-1)
        tri = c[:, None, :] + rng.uniform(-0.5, 0.5, (n, 3, 3))
    elif kind == "stack":
        # 2050 large triangles behind one another, each covering the middle of the image: the central tiles hold all of them,
        # and 2050 entries over 8 waves are slices of 257 (five staging chunks) and a last one of 251 (four) — the mode of
        # a shared tile (staged chunks or tree walk) must not depend on the wave that looks
        n = 2050
        base = np.array([[-4.0, -3.0, 0.0], [4.0, -3.0, 0.0], [0.0, 5.0, 0.0]])
        tri = base[None] + rng.uniform(-0.3, 0.3, (n, 3, 3))
        tri[:, :, 2] += np.linspace(-30.0, -6.0, n)[:, None]
    else:
        n = 200
        c = np.stack([rng.uniform(-3, 3, n), rng.uniform(-3, 3, n), rng.uniform(-10, -5, n)], axis=-1)
        tri = c[:, None, :] + rng.uniform(-2.0, 2.0, (n, 3, 3)) * np.array([1.0, 1.0e-4, 1.0])
    pts = tri.reshape(-1, 3).astype(np.float32)
    sc = dict(points=pts, idx=np.arange(n * 3, dtype=np.float32).reshape(n, 3), normals=None, index_base=0,
              camera=dict(lookfrom=(0.0, 0.0, 0.0), lookat=(0.0, 0.0, -1.0), vfov=60.0, aperture=0.0, focus=10.0))
    nx, ny, spp = 96, 64, 3
    scene, cam = make(sc, nx, ny)
    w, ocam = oracle_scene(oracle, sc, nx, ny)
    states = oracle.fresh_states(nx, ny)
    bvh = oracle.Bvh(w, states)
    ref, states = oracle.render_bvh(bvh, ocam, nx, ny, spp, states)
    after = np.concatenate([states["d"][:, None], states["v"]], axis=1)
    for variant in (0, 1, 7, 9, 10):                            # 10: shared tiles from 4 entries on, dealt out as row slabs
        if kind == "stack" and variant == 0:
            api.tiled_stats(1)
        (rgb,), st = hip_bvh_frames(gpu, scene, cam, nx, ny, spp, variant=variant)
        if kind == "stack" and variant == 0:
            assert api.tiled_stats(0)[6] == n                  # the scene does what it was built for: a tile with all 2050 entries
        assert np.array_equal(bits(rgb), bits(ref)), (kind, variant, int((bits(rgb) != bits(ref)).any(axis=2).sum()))
        assert np.array_equal(st[1:], after[1:]), (kind, variant)


@pytest.mark.gpu
@pytest.mark.parametrize("n_lat,n_lon,nx,ny", [(8, 3000, 320, 180), (6, 12000, 256, 144)])
def test_gpu_bvh_pole_tiles_longer_than_the_staging_areas(gpu, n_lat, n_lon, nx, ny):
    """The BVH world over a lat-long sphere with many meridians: the pole tiles' slices are longer than the 32-entry staging
    areas and go as far pairs with the tree's tie order in the keys (trace_far_pairs<KEYED>), the heaviest as eight row slabs;
    the second mesh overflows the pair areas (staged + keyed wave-uniform walk).  Tiled == per-ray tree walk, pixels and states,
    also with every shared tile dealt out as row slabs (variant 10)."""
    sc = scenes.displaced_sphere(n_lat, n_lon, amp=0.02)
    scene, cam = make(sc, nx, ny)
    spp = 5
    api.tiled_stats(1)
    (tiled,), st_t = hip_bvh_frames(gpu, scene, cam, nx, ny, spp, variant=0)
    stats = api.tiled_stats(0)
    assert stats[43] > 0 and stats[44] > 0, stats[43:47]
    (walk,), st_w = hip_bvh_frames(gpu, scene, cam, nx, ny, spp, variant=1, build=False)
    (slabs,), st_s = hip_bvh_frames(gpu, scene, cam, nx, ny, spp, variant=10, build=False)
    flat = lambda a: bits(a).reshape(-1, 3)[1:]                # pixel 0 reads stream 0, which the build (first call only) advanced
    assert np.array_equal(flat(tiled), flat(walk)) and np.array_equal(flat(slabs), flat(walk))
    assert np.array_equal(st_t[1:], st_w[1:]) and np.array_equal(st_s[1:], st_w[1:])


@pytest.mark.gpu
@pytest.mark.parametrize("name,keys,order", HAND_TREES, ids=[t[0] for t in HAND_TREES])
def test_gpu_leaf_order_equals_the_hand_derived_tree(gpu, name, keys, order):
    """The product's restatement of the sort (csrc/crt_bvh.cpp) against the same hand-derived orders."""
    sc = keyed_triangles(keys)
    scene = api.Scene.from_mesh(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
    scene.build_bvh(None, 0)
    _, _, slot_triangle = scene.bvh_export()
    assert np.asarray(slot_triangle).tolist() == order, name
