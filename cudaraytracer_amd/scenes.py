"""Deterministic synthetic scenes for the BASELINE.json configurations.

Every generator returns a dict
    points   float32 [P,3]
    idx      float32 [T,3]   vertex indices as float triplets, like the reference's idxVertex
    normals  float32 [T,3] or None (None = Triangle's 5-argument ctor rule)
    index_base  0 (FBX-style) or 1 (OBJ-style, indices stored undecremented)
    camera   dict(lookfrom, lookat, vfov, aperture, focus)  — vup is (0,1,0), aspect nx/ny
`bunny.obj` / `dragon.obj` of the reference are absent (.MISSING_LARGE_BLOBS), so the
100 k / 1 M-triangle meshes are displaced lat-long spheres.  Faces are emitted
counter-clockwise seen from outside; the scene pack reverses the winding
(createScene.h:183), which makes the stored normal point inward, and the reference's
cull `dot(d, n) < 0 -> miss` (shapes/triangle.h:61) then keeps the camera-facing side.
"""
import numpy as np

REFERENCE_CAMERA = dict(lookfrom=(0.0, 100.0, 1000.0), lookat=(0.0, 150.0, 0.0), vfov=40.0,
                        aperture=0.0, focus=10.0)      # createScene.h:160


def quad():
    """C1: the two-triangle quad at z=-5 with stored normal (0,0,-1)."""
    pts = np.array([[-1, -1, -5], [1, -1, -5], [1, 1, -5], [-1, 1, -5]], dtype=np.float32)
    idx = np.array([[0, 1, 2], [0, 2, 3]], dtype=np.float32)
    nrm = np.array([[0, 0, -1], [0, 0, -1]], dtype=np.float32)
    cam = dict(lookfrom=(0.0, 0.0, 0.0), lookat=(0.0, 0.0, -1.0), vfov=40.0, aperture=0.0, focus=10.0)
    return dict(points=pts, idx=idx, normals=nrm, index_base=0, camera=cam)


def _quad_faces(pts, faces, a, b, c, d, front):
    """Append quad a-b-c-d (4 points) as two triangles, oriented CCW seen from `front`."""
    base = len(pts)
    q = [np.asarray(p, dtype=np.float64) for p in (a, b, c, d)]
    n = np.cross(q[1] - q[0], q[2] - q[0])
    if np.dot(n, np.asarray(front, dtype=np.float64)) < 0:
        q = [q[0], q[3], q[2], q[1]]
    pts.extend(q)
    faces.append((base, base + 1, base + 2))
    faces.append((base, base + 2, base + 3))


def _box(pts, faces, corners_xz, y0, y1):
    """Five faces (4 sides + top) of a box whose footprint is the quad corners_xz."""
    c = [np.array([x, 0.0, z]) for x, z in corners_xz]
    centre = sum(c) / 4.0
    top = [p + np.array([0, y1, 0]) for p in c]
    bot = [p + np.array([0, y0, 0]) for p in c]
    _quad_faces(pts, faces, top[0], top[1], top[2], top[3], (0, 1, 0))
    for i in range(4):
        j = (i + 1) % 4
        outward = (bot[i] + bot[j]) / 2.0 - centre
        outward[1] = 0.0
        _quad_faces(pts, faces, bot[i], bot[j], top[j], top[i], outward)


def cornell32():
    """C2: Cornell box, 5 walls + light + short box + tall box = 32 triangles."""
    pts, faces = [], []
    X, Y, Z = 555.0, 555.0, 555.0
    _quad_faces(pts, faces, (0, 0, 0), (X, 0, 0), (X, 0, Z), (0, 0, Z), (0, 1, 0))        # floor
    _quad_faces(pts, faces, (0, Y, 0), (X, Y, 0), (X, Y, Z), (0, Y, Z), (0, -1, 0))       # ceiling
    _quad_faces(pts, faces, (0, 0, Z), (X, 0, Z), (X, Y, Z), (0, Y, Z), (0, 0, -1))       # back wall
    _quad_faces(pts, faces, (0, 0, 0), (0, 0, Z), (0, Y, Z), (0, Y, 0), (1, 0, 0))        # right wall (x=0)
    _quad_faces(pts, faces, (X, 0, 0), (X, 0, Z), (X, Y, Z), (X, Y, 0), (-1, 0, 0))       # left wall (x=555)
    _quad_faces(pts, faces, (213, Y - 1, 227), (343, Y - 1, 227), (343, Y - 1, 332), (213, Y - 1, 332), (0, -1, 0))  # light
    _box(pts, faces, [(130, 65), (82, 225), (240, 272), (290, 114)], 0.0, 165.0)          # short box
    _box(pts, faces, [(423, 247), (265, 296), (314, 456), (472, 406)], 0.0, 330.0)        # tall box
    cam = dict(lookfrom=(278.0, 273.0, -800.0), lookat=(278.0, 273.0, 0.0), vfov=40.0, aperture=0.0, focus=10.0)
    return dict(points=np.array(pts, dtype=np.float32), idx=np.array(faces, dtype=np.float32),
                normals=None, index_base=0, camera=cam)


def displaced_sphere(n_lat, n_lon, radius=250.0, centre=(0.0, 150.0, 0.0), amp=0.1, index_base=0):
    """Closed-free lat-long sphere with r = R*(1 + amp*sin(7*theta)*sin(5*phi)):
    (n_lat+1) rings x n_lon points, n_lat*n_lon*2 triangles (200x250 -> 100 000,
    500x1000 -> 1 000 000).  The polar caps are left open so no triangle is degenerate."""
    i = np.arange(n_lat + 1, dtype=np.float64)
    j = np.arange(n_lon, dtype=np.float64)
    theta = np.pi * (i + 0.5) / (n_lat + 1)
    phi = 2.0 * np.pi * j / n_lon
    th, ph = np.meshgrid(theta, phi, indexing="ij")
    r = radius * (1.0 + amp * np.sin(7.0 * th) * np.sin(5.0 * ph))
    x = r * np.sin(th) * np.cos(ph) + centre[0]
    y = r * np.cos(th) + centre[1]
    z = r * np.sin(th) * np.sin(ph) + centre[2]
    pts = np.stack([x, y, z], axis=-1).reshape(-1, 3).astype(np.float32)
    ii, jj = np.meshgrid(np.arange(n_lat), np.arange(n_lon), indexing="ij")
    a = ii * n_lon + jj
    b = ii * n_lon + (jj + 1) % n_lon
    c = (ii + 1) * n_lon + (jj + 1) % n_lon
    d = (ii + 1) * n_lon + jj
    # outward CCW: ring i is above ring i+1 (theta grows downwards), phi grows CCW seen from +y
    t1 = np.stack([a, b, c], axis=-1).reshape(-1, 3)
    t2 = np.stack([a, c, d], axis=-1).reshape(-1, 3)
    idx = np.empty((t1.shape[0] * 2, 3), dtype=np.float32)
    idx[0::2] = t1 + index_base
    idx[1::2] = t2 + index_base
    return dict(points=pts, idx=idx, normals=None, index_base=index_base, camera=dict(REFERENCE_CAMERA))


def bunny100k():
    """C3/C4 stand-in: 200 x 250 x 2 = 100 000 triangles."""
    return displaced_sphere(200, 250)


def dragon1m():
    """C5 stand-in: 500 x 1000 x 2 = 1 000 000 triangles."""
    return displaced_sphere(500, 1000)


def midpoint_subdivide(points, tris):
    """One level of midpoint subdivision: every triangle (a, b, c) becomes (a, ab, ca), (ab, b, bc), (ca, bc, c),
    (ab, bc, ca) — same orientation, the four children adjacent in the list — with one float32 midpoint
    0.5 * (p_a + p_b) per undirected edge.  points [n, 3] float32, tris [m, 3] int (0-based)."""
    tris = np.asarray(tris, dtype=np.int64)
    e = np.concatenate([tris[:, [0, 1]], tris[:, [1, 2]], tris[:, [2, 0]]], axis=0)
    e.sort(axis=1)
    key = e[:, 0] * (len(points) + 1) + e[:, 1]
    uniq, first, inverse = np.unique(key, return_index=True, return_inverse=True)
    ends = e[first]
    mid = (np.float32(0.5) * (points[ends[:, 0]] + points[ends[:, 1]])).astype(np.float32)
    m = len(tris)
    ab, bc, ca = (inverse[0:m] + len(points)), (inverse[m:2 * m] + len(points)), (inverse[2 * m:3 * m] + len(points))
    a, b, c = tris[:, 0], tris[:, 1], tris[:, 2]
    out = np.stack([np.stack([a, ab, ca], 1), np.stack([ab, b, bc], 1), np.stack([ca, bc, c], 1), np.stack([ab, bc, ca], 1)], axis=1)
    return np.concatenate([points, mid], axis=0).astype(np.float32), out.reshape(-1, 3)


def small_bunny_arrays():
    """The reference-held mesh CudaTest/objects/small_bunny.obj as arrays (2503 points, already x100 and mean-centred by
    the loader rule; 4968 faces, 1-based float indices): the data fixture tests/golden/small_bunny_64x64x4.npz
    (tools/make_golden.py wrote it from the OBJ file; the GPU box has no reference tree)."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = np.load(os.path.join(root, "tests", "golden", "small_bunny_64x64x4.npz"))
    return g["points"].astype(np.float32), (g["idx"].astype(np.int64) - 1)


def bunny_ground_100k(index_base=1):
    """SURVEY §8d's other C3 mesh, without a pole: small_bunny midpoint-subdivided twice (4968 x 16 = 79 488 triangles)
    standing on a ground grid of 641 x 16 cells (20 512 triangles) = 100 000.  The bunny is scaled x25 (about 390 units
    tall, the size of the lat-long mesh); the ground lies at the bunny's lowest point, x in [-801.25, 801.25] (cells of
    2.5), z in [-400, 400] (cells of 50), counter-clockwise seen from above like the bunny's faces seen from outside."""
    pts, tris = small_bunny_arrays()
    pts = (pts * np.float32(25.0)).astype(np.float32)
    for _ in range(2):
        pts, tris = midpoint_subdivide(pts, tris)
    assert len(tris) == 79488
    nxc, nzc = 641, 16
    y0 = np.float32(pts[:, 1].min())
    gx = (np.float32(-801.25) + np.float32(2.5) * np.arange(nxc + 1, dtype=np.float32)).astype(np.float32)
    gz = (np.float32(-400.0) + np.float32(50.0) * np.arange(nzc + 1, dtype=np.float32)).astype(np.float32)
    zz, xx = np.meshgrid(gz, gx, indexing="ij")                       # point (k, i) = row k (z), column i (x)
    ground = np.stack([xx, np.full_like(xx, y0), zz], axis=-1).reshape(-1, 3).astype(np.float32)
    kk, ii = np.meshgrid(np.arange(nzc), np.arange(nxc), indexing="ij")
    p00 = kk * (nxc + 1) + ii
    p01, p10, p11 = p00 + 1, p00 + (nxc + 1), p00 + (nxc + 1) + 1     # +x, +z, both
    base = len(pts)
    # seen from +y with +x to the right and +z towards the viewer, (p00, p10, p11) and (p00, p11, p01) run counter-clockwise
    g1 = np.stack([p00, p10, p11], axis=-1).reshape(-1, 3) + base
    g2 = np.stack([p00, p11, p01], axis=-1).reshape(-1, 3) + base
    gt = np.empty((len(g1) * 2, 3), dtype=np.int64)
    gt[0::2], gt[1::2] = g1, g2
    allp = np.concatenate([pts, ground], axis=0).astype(np.float32)
    allt = np.concatenate([tris, gt], axis=0)
    assert len(allt) == 100000
    return dict(points=allp, idx=(allt + index_base).astype(np.float32), normals=None, index_base=index_base,
                camera=dict(REFERENCE_CAMERA))


# ---- the BASELINE.json frames, ONE definition for bench.py and the tests -------------------
# name -> (nx, ny, spp, n_lat, n_lon): the mesh is written as an OBJ file (1-based `f a b c`
# records) and read back through the OBJ loader (mean centring, Loader/obj_loader.h:120-133),
# scale 1, reference camera (createScene.h:160) — exactly what bench.py times.
BASELINE_FRAMES = {
    "C3": (1920, 1080, 64, 200, 250),      # configs[2]: 100 000 triangles, 1 GPU
    "C4": (1920, 1080, 256, 200, 250),     # configs[3]: same mesh, 256 spp, 8-way split
    "C5": (3840, 2160, 64, 500, 1000),     # configs[4]: 1 000 000 triangles
    "C3B": (1920, 1080, 64, "bunny_ground", 0),   # configs[2] on SURVEY §8d's other mesh: subdivided small_bunny + ground grid, no pole
}


def baseline_mesh(name):
    """The mesh of a BASELINE frame before the OBJ round trip (index_base 1)."""
    n_lat, n_lon = BASELINE_FRAMES[name][3:5]
    if n_lat == "bunny_ground":
        return bunny_ground_100k(index_base=1)
    return displaced_sphere(n_lat, n_lon, radius=200.0, centre=(0.0, 0.0, 0.0), index_base=1)


def baseline_obj(name, folder):
    """Write the frame's OBJ file into `folder` (once) and return (path, nx, ny, spp)."""
    import os
    nx, ny, spp = BASELINE_FRAMES[name][:3]
    n_lat, n_lon = BASELINE_FRAMES[name][3:5]
    path = os.path.join(folder, ("synthetic_%dx%d.obj" % (n_lat, n_lon)) if n_lat != "bunny_ground" else "synthetic_bunny_ground.obj")
    if not os.path.exists(path):
        write_obj(path, baseline_mesh(name))
    return path, nx, ny, spp


def random_soup(n_tris, seed, extent=4.0, size=1.5, z_range=(-12.0, 6.0)):
    """Adversarial triangle soup around a camera at the origin looking down -z: triangles in
    front of AND behind the camera (the reference accepts negative t), random stored
    normals' orientation (via winding), some exact duplicates (index ties) and some
    degenerate triangles (NaN normals under the 5-argument ctor rule)."""
    rng = np.random.RandomState(seed)
    c = np.stack([rng.uniform(-extent, extent, n_tris), rng.uniform(-extent, extent, n_tris),
                  rng.uniform(z_range[0], z_range[1], n_tris)], axis=-1)
    tri = c[:, None, :] + rng.uniform(-size, size, (n_tris, 3, 3))
    n_dup = max(1, n_tris // 16)
    src = rng.randint(0, n_tris, n_dup)
    dst = rng.randint(0, n_tris, n_dup)
    tri[dst] = tri[src]                                  # coplanar duplicates -> t ties
    n_deg = max(1, n_tris // 32)
    deg = rng.randint(0, n_tris, n_deg)
    tri[deg, 2] = tri[deg, 1]                            # degenerate (zero area)
    pts = tri.reshape(-1, 3).astype(np.float32)
    idx = np.arange(n_tris * 3, dtype=np.float32).reshape(n_tris, 3)
    cam = dict(lookfrom=(0.0, 0.0, 0.0), lookat=(0.0, 0.0, -1.0), vfov=60.0, aperture=0.0, focus=10.0)
    return dict(points=pts, idx=idx, normals=None, index_base=0, camera=cam)


def write_obj(path, scene):
    """Plain `v x y z` / `f a b c` OBJ (the only records the reference parser reads,
    Loader/obj_loader.h:44-64); indices are written 1-based."""
    pts, idx = scene["points"], scene["idx"]
    off = 1 - scene["index_base"]
    tmp = path + ".tmp%d" % __import__("os").getpid()
    with open(tmp, "w") as f:
        f.write("# synthetic mesh: %d vertices, %d faces\n" % (len(pts), len(idx)))
        f.write("".join(["v %.9g %.9g %.9g\n" % (p[0], p[1], p[2]) for p in pts.astype(np.float64).tolist()]))
        f.write("".join(["f %d %d %d\n" % (t[0], t[1], t[2]) for t in (idx.astype(np.int64) + off).tolist()]))
    __import__("os").replace(tmp, path)
    return path
