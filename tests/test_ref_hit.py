"""SURVEY §8 rows a5, a7-a10 and f1's box arithmetic, PINNED against the reference's own device code.

hitable/hitable.h, hitable/transform.h, hitable/hitable_list.h, shapes/triangle.h, core/aabb.h and core/ray.h include
nothing but <float.h>, core/vec3.h and each other: the image's HIP compiler builds them for gfx950 as they are
(oracle/ref_hit.hip includes them by path from /root/reference; nothing copied, nothing stood in for; the built
oracle/_ref/libref_hit.so travels to the GPU box).  Its kernels build the reference's object graph with device `new`
the way add_mesh_withNormal does (createScene.h:175-190) and call the reference's member functions.

Held to it on the GPU box, bit for bit (hit flag, rec.t, winning triangle, rec.normal, rec.p; NaN == NaN):
  * the oracle's restatement: o_world_hit (world list), the per-level Hitable::hit wrapper (nested lists), AABB::hit,
    surrounding_box, TransformAABB, Triangle::bounding_box / GetBV, HitableList::bounding_box;
  * the product's device functions through crt_debug_trace_rays / crt_debug_box_hit / crt_debug_box_op: the exact_test
    walk over per-frame records (literal and fast arithmetic), the any-origin triangle test, both forms of the slab test,
    the build's and the refit kernel's box helpers — and, composed from reference operations only, every node box of the
    product's and the oracle's trees after the build and after a refit.
What the reference's headers cannot answer here (they need cuRAND / Thrust): BVHNode's constructor and merge rule —
the tree walks of oracle and product are compared with each other on the same rays (test_bvh.py has the frames)."""
import numpy as np
import pytest

from cudaraytracer_amd import api, scenes
from oracle import pyref

pytestmark = pytest.mark.gpu

FLT_MAX = np.float32(3.4028234663852886e38)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def same(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    return (bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))


def assert_hits_equal(got, want, what, with_point=True):
    gh, gt, gi, gn, gp = got
    wh, wt, wi, wn, wp = want
    assert np.array_equal(gh, wh), "%s: hit flags differ at %s" % (what, np.argwhere(gh != wh)[:5].ravel().tolist())
    m = wh != 0
    assert same(gt[m], wt[m]).all(), "%s: rec.t differs (%d rays)" % (what, int((~same(gt[m], wt[m])).sum()))
    assert np.array_equal(gi[m], wi[m]), "%s: winning triangle differs at %s" % (what, np.argwhere(gi[m] != wi[m])[:5].ravel().tolist())
    assert same(gn[m], wn[m]).all(), "%s: rec.normal differs" % what
    if with_point:
        assert same(gp[m], wp[m]).all(), "%s: rec.p differs" % what
    return int(m.sum())


@pytest.fixture(scope="module")
def ref(gpu):
    if not pyref.have_hit():
        pytest.fail("oracle/_ref/libref_hit.so did not travel to the GPU box (make -C oracle ref in the build container)")
    return pyref


def rays_for(sc, n, seed, origin=None):
    """Un-normalised directions as `render` hands them to world->hit: towards the soup from its camera, lengths around the
    reference's focus distance, plus axis-parallel and tiny / huge ones."""
    rng = np.random.RandomState(seed)
    pts = sc["points"].astype(np.float64)
    o = np.array(sc["camera"]["lookfrom"] if origin is None else origin, np.float64)
    target = pts[rng.randint(0, len(pts), n)] + rng.normal(size=(n, 3)) * 0.05 * (np.abs(pts).max() + 1.0)
    d = target - o
    d *= (10.0 / np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-30)) * rng.uniform(0.5, 2.0, (n, 1))
    d[::17] = -d[::17]                                    # behind the origin: the list accepts negative t
    d[5::41] *= 1e-12
    d[7::43] *= 1e12
    k = np.arange(3, n, 59)
    d[k, rng.randint(0, 3, len(k))] = 0.0                 # a zero component (slab tests divide by it)
    return np.float32(o), d.astype(np.float32)


def soups():
    out = [("soup200", scenes.random_soup(200, 5)),
           ("around_eye", scenes.random_soup(300, 11, extent=3.0, size=2.5, z_range=(-4.0, 4.0))),
           ("tiny_far", scenes.random_soup(400, 12, extent=30.0, size=0.02, z_range=(-80.0, -60.0))),
           ("huge", scenes.random_soup(40, 13, extent=50.0, size=400.0, z_range=(-300.0, 300.0)))]
    rn = scenes.random_soup(200, 14)
    rn["normals"] = np.random.RandomState(14).normal(size=(200, 3)).astype(np.float32)     # stored normals unrelated to geometry
    out.append(("random_normals", rn))
    nn = scenes.random_soup(64, 15)
    nrm = np.random.RandomState(15).normal(size=(64, 3)).astype(np.float32)
    nrm[::7] = np.nan; nrm[3::11] = 0.0; nrm[5::13] *= np.float32(1e-42); nrm[6::17] = np.inf   # NaN / zero / subnormal / inf normals
    nn["normals"] = nrm
    out.append(("odd_normals", nn))
    # coincident duplicates (index ties), a parallel sheet (|a| < EPSILON for rays in its plane), one triangle through the origin
    tri = np.array([[-1, -1, -5], [1, -1, -5], [0, 1, -5]], np.float32)
    pts = np.concatenate([tri, tri, tri + np.float32([0, 0, 10]), np.float32([[-1, 0, -1], [1, 0, -3], [0, 0, 2]]),
                          np.float32([[-0.5, -0.5, 0], [0.5, -0.5, 0], [0, 0.7, 0]])])
    idx = np.arange(len(pts), dtype=np.float32).reshape(-1, 3)
    out.append(("ties", dict(points=pts, idx=idx, normals=np.tile(np.float32([[0, 0, -1]]), (len(idx), 1)), index_base=0,
                             camera=dict(lookfrom=(0.0, 0.0, 0.0), lookat=(0.0, 0.0, -1.0), vfov=60.0, aperture=0.0, focus=10.0))))
    mesh = scenes.displaced_sphere(10, 14)
    out.append(("sphere_mesh", mesh))
    return out


SOUPS = soups()


@pytest.mark.parametrize("name,sc", SOUPS, ids=[s[0] for s in SOUPS])
def test_world_list_hit_equals_the_reference(ref, oracle, name, sc):
    """(*world)->hit(r, 0.001, FLT_MAX, rec): reference == oracle == product (camera-origin records, literal and fast
    arithmetic; any-origin base records)."""
    rs = ref.RefScene(sc["points"], sc["idx"], sc["normals"], sc["index_base"], levels=2)
    w = oracle.World(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
    scene = api.Scene.from_mesh(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
    o, d = rays_for(sc, 4096, 3)
    want = rs.hit(o, d)
    n_hit = assert_hits_equal(oracle.world_hit_many(w, o, d), want, name + ": oracle")
    assert n_hit > 50 or name == "tiny_far", (name, n_hit)
    assert_hits_equal(api.trace_rays(scene, 0, o, d, arith=0), want, name + ": product, literal", with_point=False)
    assert_hits_equal(api.trace_rays(scene, 0, o, d, arith=1), want, name + ": product, fast forms", with_point=False)
    # rays from anywhere (what scattered rays are): one origin per ray
    rng = np.random.RandomState(9)
    oo = (np.float32(o) + rng.normal(size=d.shape).astype(np.float32) * np.float32(2.0)).astype(np.float32)
    want_any = rs.hit(oo, d)
    assert_hits_equal(oracle.world_hit_many(w, oo, d), want_any, name + ": oracle, any origin")
    assert_hits_equal(api.trace_rays(scene, 1, oo, d, arith=0), want_any, name + ": product any origin, literal")
    assert_hits_equal(api.trace_rays(scene, 1, oo, d, arith=1), want_any, name + ": product any origin, fast forms")
    # two more Hitable::hit levels around the list (what every BVH level adds: one more ScaleRay normalisation)
    assert_hits_equal(oracle.world_hit_many(w, o, d, levels=2), rs.hit(o, d, nested=True), name + ": oracle, nested")
    rs.close()


def test_hundred_thousand_random_ray_list_cases(ref, oracle):
    """10^5 (ray, 64-triangle list) cases: 200 lists x 512 rays, fresh geometry each."""
    total = hits = 0
    for k in range(200):
        sc = scenes.random_soup(64, 1000 + k, extent=3.0 + (k % 5), size=0.5 + 0.5 * (k % 7), z_range=(-10.0, 4.0))
        if k % 3 == 0:
            sc["normals"] = np.random.RandomState(k).normal(size=(64, 3)).astype(np.float32)
        rs = ref.RefScene(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
        w = oracle.World(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
        scene = api.Scene.from_mesh(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
        o, d = rays_for(sc, 512, k)
        want = rs.hit(o, d)
        hits += assert_hits_equal(oracle.world_hit_many(w, o, d), want, "list %d: oracle" % k)
        assert_hits_equal(api.trace_rays(scene, 0, o, d, arith=k % 2), want, "list %d: product" % k, with_point=False)
        total += len(d)
        rs.close()
        scene.close()
    assert total == 102400 and hits > 20000, (total, hits)


def test_pose_update_rewrites_the_vertices_like_setvertices(ref, oracle):
    """update_pose (createScene.h:99-110) = Triangle::SetVertices (triangle.h:40-45): normals kept."""
    sc = scenes.displaced_sphere(8, 12)
    rs = ref.RefScene(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
    w = oracle.World(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
    scene = api.Scene.from_mesh(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
    new_pts = (sc["points"] * np.float32(1.07) + np.float32([3.0, -2.0, 1.0])).astype(np.float32)
    rs.set_vertices(new_pts)
    w.update_pose(new_pts)
    scene.update_pose(new_pts)
    o, d = rays_for(dict(points=new_pts, camera=sc["camera"]), 2048, 4)
    want = rs.hit(o, d)
    assert assert_hits_equal(oracle.world_hit_many(w, o, d), want, "pose: oracle") > 100
    assert_hits_equal(api.trace_rays(scene, 0, o, d, arith=1), want, "pose: product", with_point=False)
    rs.close()


# ---- boxes ------------------------------------------------------------------------------------------------------------
def box_cases(n, seed):
    rng = np.random.RandomState(seed)
    c = rng.uniform(-20, 20, (n, 3))
    h = np.abs(rng.normal(size=(n, 3))) * rng.choice([0.01, 1.0, 30.0], (n, 1))
    bmin, bmax = (c - h).astype(np.float32), (c + h).astype(np.float32)
    flat = np.arange(0, n, 13)
    bmax[flat, rng.randint(0, 3, len(flat))] = bmin[flat, rng.randint(0, 3, len(flat))]    # some flat (or crossed) boxes
    o = rng.uniform(-25, 25, (n, 3)).astype(np.float32)
    inside = np.arange(1, n, 7)
    o[inside] = ((bmin[inside] + bmax[inside]) * np.float32(0.5)).astype(np.float32)        # origins inside the box
    d = rng.normal(size=(n, 3)).astype(np.float32)
    towards = np.arange(2, n, 3)
    d[towards] = (c[towards] - o[towards]).astype(np.float32)
    return bmin, bmax, o, d


def test_aabb_hit_equals_the_reference(ref, oracle):
    """AABB::hit (core/aabb.h:30-43): reference == oracle == the kernels' literal form; the bounded-scene form on the
    inputs it is used for (ordinary directions); zero components, origins on a face, flat boxes for the literal one."""
    n = 20000
    bmin, bmax, o, d = box_cases(n, 1)
    dz = d.copy()
    k = np.arange(0, n, 5)
    dz[k, np.random.RandomState(2).randint(0, 3, len(k))] = 0.0      # division by zero: +-inf or NaN quotients
    onface = np.arange(3, n, 11)
    o2 = o.copy()
    o2[onface, 0] = bmin[onface, 0]                                  # 0 / 0 where the direction is zero too
    t_min = np.full(n, 0.001, np.float32)
    t_max = np.full(n, FLT_MAX, np.float32)
    for name, oo, dd in (("ordinary", o, d), ("zero components", o2, dz)):
        want = ref.aabb_hit(bmin, bmax, oo, dd, t_min, t_max)
        assert 0.05 < want.mean() < 0.95, (name, want.mean())
        assert np.array_equal(oracle.aabb_hit_many(bmin, bmax, oo, dd, t_min, t_max), want), name + ": oracle"
        box6 = np.concatenate([bmin, bmax], 1)
        assert np.array_equal(api.box_hit(0, box6, oo, dd), want), name + ": box_hit_ieee"
    # other t ranges (the tree always passes 0.001 / FLT_MAX, aabb.h is general)
    rng = np.random.RandomState(3)
    t0 = rng.uniform(-5, 5, n).astype(np.float32)
    t1 = (t0 + np.abs(rng.normal(size=n)).astype(np.float32) * np.float32(30.0)).astype(np.float32)
    assert np.array_equal(oracle.aabb_hit_many(bmin, bmax, o, d, t0, t1), ref.aabb_hit(bmin, bmax, o, d, t0, t1))
    # the fast form: every |direction component| >= 2^-40, coordinates bounded, differences 0 or >= 2^-80
    big = np.abs(d).min(1) >= 1e-6
    want = ref.aabb_hit(bmin[big], bmax[big], o[big], d[big], t_min[big], t_max[big])
    got = api.box_hit(1, np.concatenate([bmin, bmax], 1)[big], o[big], d[big])
    assert np.array_equal(got, want), "box_hit_fast differs on %d of %d" % (int((got != want).sum()), len(want))


def test_box_arithmetic_equals_the_reference(ref, oracle):
    """surrounding_box (aabb.h:58-66) and TransformAABB (transform.h:22-31; rotation 0, any position for the oracle,
    position 0 for the product, which only ever applies the identity Transform)."""
    n = 8000
    amin, amax, _, _ = box_cases(n, 4)
    bmin, bmax, _, _ = box_cases(n, 5)
    want = ref.aabb_op(0, amin, amax, bmin, bmax)
    got = oracle.aabb_op_many(0, amin, amax, bmin, bmax)
    assert same(got[0], want[0]).all() and same(got[1], want[1]).all()
    a6, b6 = np.concatenate([amin, amax], 1), np.concatenate([bmin, bmax], 1)
    for where in (0, 1):
        out = api.box_op(where, 0, a6, b6)
        assert same(out[:, :3], want[0]).all() and same(out[:, 3:], want[1]).all(), "surrounding, where %d" % where
    pos = np.random.RandomState(6).uniform(-100, 100, (n, 3)).astype(np.float32)
    zero = np.zeros((n, 3), np.float32)
    want = ref.aabb_op(1, amin, amax, pos, zero)
    got = oracle.aabb_op_many(1, amin, amax, pos, zero)
    assert same(got[0], want[0]).all() and same(got[1], want[1]).all()
    want0 = ref.aabb_op(1, amin, amax, zero, zero)
    for where in (0, 1):
        out = api.box_op(where, 1, a6, a6)
        assert same(out[:, :3], want0[0]).all() and same(out[:, 3:], want0[1]).all(), "TransformAABB, where %d" % where


def compose_tree_boxes(ref, tri_min, tri_max, meta, slot_triangle, refit):
    """Every node box of a tree (nodes in preorder; meta = skip, first slot, count, depth) from REFERENCE operations only.
    Build (BVHNode::BVHNode, bvh.h:94-122): leaf = childList->GetBV = HitableList::bounding_box (surrounding_box over the
    triangles' GetBV, hitable_list.h:94-118) + TransformAABB; inner = surrounding_box(left->GetBV, right->GetBV), GetBV of a
    node = its box + TransformAABB.  Refit (UpdateBVH, bvh.h:135-158): leaf = childList->bounding_box WITHOUT the list's
    TransformAABB; inner = surrounding_box(right->GetBV, left->GetBV)."""
    n = len(meta)
    zero = np.zeros((1, 3), np.float32)
    box = np.zeros((n, 6), np.float32)

    def tf(b):
        lo, hi = ref.aabb_op(1, b[None, :3], b[None, 3:], zero, zero)
        return np.concatenate([lo[0], hi[0]])

    def sur(a, b):
        lo, hi = ref.aabb_op(0, a[None, :3], a[None, 3:], b[None, :3], b[None, 3:])
        return np.concatenate([lo[0], hi[0]])

    for i in range(n - 1, -1, -1):
        skip, first, count, _ = meta[i]
        if count:
            b = None
            for s in range(first, first + count):
                tb = np.concatenate([tri_min[slot_triangle[s]], tri_max[slot_triangle[s]]])
                b = tb if b is None else sur(b, tb)
            box[i] = b if refit else tf(b)
        else:
            l, r = i + 1, meta[i + 1][0]
            box[i] = sur(tf(box[r]), tf(box[l])) if refit else sur(tf(box[l]), tf(box[r]))
    return box


def test_tree_boxes_follow_from_reference_operations(ref, oracle):
    """Per-triangle GetBV, the list's GetBV and every node box of the product's and the oracle's trees — after the build and
    after a pose update + refit — equal what the reference's own box functions give for the same tree shape."""
    sc = scenes.random_soup(150, 21, extent=6.0, size=1.5)
    rs = ref.RefScene(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
    w = oracle.World(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
    flag, tmin, tmax = rs.bounding_boxes(0)
    assert flag.all()
    oflag, omin, omax = oracle.world_bounding_boxes(w, 0)
    assert same(omin, tmin).all() and same(omax, tmax).all()
    lflag, lmin, lmax = rs.bounding_boxes(1)
    oflag, olmin, olmax = oracle.world_bounding_boxes(w, 1)
    assert lflag[0] == 1 and oflag == 1 and same(olmin, lmin).all() and same(olmax, lmax).all()

    scene = api.Scene.from_mesh(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
    scene.build_bvh(None, 0)
    boxes, meta, slots = scene.bvh_export()
    want = compose_tree_boxes(ref, tmin, tmax, meta.tolist(), slots.tolist(), refit=False)
    assert same(boxes, want).all(), "product tree after the build: %d boxes differ" % int((~same(boxes, want).all(1)).sum())
    bvh = oracle.Bvh(w, oracle.fresh_states(8, 8))
    nodes = bvh.nodes()
    for nd, m in zip(nodes, meta.tolist()):                                          # same shape (tests/test_bvh.py holds that too)
        assert (int(nd["skip"]), int(nd["depth"])) == (m[0], m[3])
        assert (int(nd["first"]), int(nd["count"])) == (m[1], m[2]) if nd["isLeaf"] else m[2] == 0
    assert len(nodes) == len(meta) and bvh.order().tolist() == slots.tolist()
    assert same(nodes["box"], want).all(), "oracle tree after the build"

    new_pts = (sc["points"] * np.float32(0.93) + np.float32([1.5, 0.25, -2.0])).astype(np.float32)
    rs.set_vertices(new_pts)
    w.update_pose(new_pts)
    scene.update_pose(new_pts)
    scene.refit_bvh()
    bvh.update()
    _, tmin2, tmax2 = rs.bounding_boxes(0)
    want2 = compose_tree_boxes(ref, tmin2, tmax2, meta.tolist(), slots.tolist(), refit=True)
    boxes2, _, _ = scene.bvh_export()
    assert same(boxes2, want2).all(), "product tree after the refit"
    assert same(bvh.nodes()["box"], want2).all(), "oracle tree after the refit"
    rs.close()


@pytest.mark.parametrize("name,sc", SOUPS[:3] + SOUPS[-1:], ids=[s[0] for s in SOUPS[:3] + SOUPS[-1:]])
def test_tree_walks_of_product_and_oracle_agree_on_single_rays(gpu, oracle, name, sc):
    """BVHNode::collision_detection needs Thrust / cuRAND headers and cannot run here; its pieces are pinned above (triangle
    test, per-level Hitable::hit, AABB::hit, boxes) and the sort rule in test_ref_thrust.py.  The composition — oracle's
    recursive walk against the product's three walks (skip links, any origin, pair records) — on the same rays."""
    w = oracle.World(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
    bvh = oracle.Bvh(w, oracle.fresh_states(8, 8))
    scene = api.Scene.from_mesh(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
    scene.build_bvh(None, 0)
    o, d = rays_for(sc, 4096, 8)
    ok = np.abs(d).min(1) > 0                      # (a zero component makes 0 * inf NaNs in the tree's slab tests: literal arithmetic only)
    want = oracle.bvh_hit_many(bvh, o, d)
    assert want[0].sum() > 50 or name == "tiny_far"
    assert_hits_equal(api.trace_rays(scene, 2, o, d, arith=0), want, name + ": skip links, literal", with_point=False)
    got = api.trace_rays(scene, 2, o, d[ok], arith=1)
    assert_hits_equal(got, tuple(x[ok] for x in want), name + ": skip links, fast forms", with_point=False)
    rng = np.random.RandomState(10)
    oo = (np.float32(o) + rng.normal(size=d.shape).astype(np.float32) * np.float32(1.5)).astype(np.float32)
    want_any = oracle.bvh_hit_many(bvh, oo, d)
    assert_hits_equal(api.trace_rays(scene, 3, oo, d, arith=0), want_any, name + ": any origin, literal")
    for mode in (3, 4):
        got = api.trace_rays(scene, mode, oo[ok], d[ok], arith=1)
        assert_hits_equal(got, tuple(x[ok] for x in want_any), name + ": any origin, fast forms, mode %d" % mode)
