"""The reference's other two shaders and its material classes: recursive `shade`
(core/render.h:48-67), shade_normal (:90-103), Metal / Dielectric / DiffuseLight
(material/material.h:77-161), on both worlds.  CPU: the oracle's restatement against closed forms;
GPU (marked): the generic kernel against the oracle through crt_render_rows_ex.

PARITY UNPINNED additionally at pow(1 - cosine, 5.0f) (schlick, material.h:25): CUDA's powf is not
reproducible off-device; oracle and kernels form the fifth power in double and round once."""
import os

import numpy as np
import pytest

from cudaraytracer_amd import _lib, api, scenes

MATS = [(api.LAMBERTIAN, (0.73, 0.73, 0.73), 0.0), (api.LAMBERTIAN, (0.65, 0.05, 0.05), 0.0),
        (api.LAMBERTIAN, (0.12, 0.45, 0.15), 0.0), (api.DIFFUSE_LIGHT, (15.0, 15.0, 15.0), 0.0),
        (api.METAL, (0.8, 0.85, 0.88), 0.1), (api.DIELECTRIC, (1.0, 1.0, 1.0), 1.5), (api.METAL, (0.7, 0.6, 0.5), 3.0)]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def same(a, b):
    """Bit equality; NaNs match NaNs (the payload of sqrt(negative) differs between x86 and gfx950)."""
    a, b = np.ascontiguousarray(a, dtype=np.float32), np.ascontiguousarray(b, dtype=np.float32)
    return bool(((bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))).all())


def material_scene(kind):
    """(scene dict, per-triangle material indices): meshes whose boxes are not flat, so that both
    worlds see them."""
    if kind == "cornell":
        sc = scenes.cornell32()
        of = np.zeros(32, np.int32)
        of[6:8], of[8:10], of[10:12], of[12:22], of[22:32] = 1, 2, 3, 4, 5
    elif kind == "spheres":
        a = scenes.displaced_sphere(10, 14, radius=120.0, centre=(-140.0, 150.0, 0.0))
        b = scenes.displaced_sphere(10, 14, radius=120.0, centre=(140.0, 150.0, 0.0))
        g = scenes.displaced_sphere(6, 8, radius=900.0, centre=(0.0, -880.0, 0.0), amp=0.0)
        pts = np.concatenate([a["points"], b["points"], g["points"]]).astype(np.float32)
        idx = np.concatenate([a["idx"], b["idx"] + len(a["points"]), g["idx"] + len(a["points"]) + len(b["points"])]).astype(np.float32)
        sc = dict(points=pts, idx=idx, normals=None, index_base=0, camera=dict(scenes.REFERENCE_CAMERA))
        of = np.concatenate([np.full(len(a["idx"]), 4), np.full(len(b["idx"]), 5), np.full(len(g["idx"]), 2)]).astype(np.int32)
        of[::7] = 3                                   # some lights
        of[3::11] = 6                                 # fuzz > 1 (clamped by the ctor)
    else:
        sc = scenes.random_soup(60, 9)
        of = (np.arange(60) % len(MATS)).astype(np.int32)
    return sc, of


def oracle_world(oracle, sc, of, nx, ny):
    w = oracle.World(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
    w.set_materials(MATS, of)
    c = sc["camera"]
    cam = oracle.make_camera(c["lookfrom"], c["lookat"], (0, 1, 0), c["vfov"], np.float32(nx) / np.float32(ny), c["aperture"], c["focus"])
    return w, cam


SHADE_GOLDEN = ["shade_spheres_64x36x3_d8", "shade_soup60_40x30x2_d8"]   # tools/make_golden.py: shade_goldens()


def golden_shade_case(g):
    sc, _ = material_scene(str(g["kind"]))
    mats = [(int(m[0]), (float(m[1]), float(m[2]), float(m[3])), float(m[4])) for m in g["materials"]]
    return sc, mats, g["material_of"].astype(np.int32), int(g["nx"]), int(g["ny"]), int(g["spp"]), int(g["depth"])


# ------------------------------------------------------------------ oracle (CPU)
@pytest.mark.parametrize("name", SHADE_GOLDEN)
def test_oracle_shade_matches_golden(oracle, golden_dir, name):
    """Regression pin of the oracle's recursive shade over the BVH world (pixels and RNG states of the committed fixture)."""
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    sc, mats, of, nx, ny, spp, depth = golden_shade_case(g)
    w = oracle.World(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
    w.set_materials(mats, of)
    c = sc["camera"]
    cam = oracle.make_camera(c["lookfrom"], c["lookat"], (0, 1, 0), c["vfov"], np.float32(nx) / np.float32(ny), c["aperture"], c["focus"])
    states = oracle.fresh_states(nx, ny)
    bvh = oracle.Bvh(w, states)
    buf, states = oracle.render_ex(w, bvh, oracle.SHADER_SHADE, cam, nx, ny, spp, states, max_depth=depth)
    assert same(buf, g["rgb"])
    assert np.array_equal(np.concatenate([states["d"][:, None], states["v"]], axis=1), g["states_after"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", SHADE_GOLDEN)
def test_gpu_shade_matches_golden(gpu, golden_dir, name):
    """The generic kernel against the committed fixture, every walk: shared by the wavefront (0), the same with a short list
    (12), one stack per lane (11), skip links (9), and the general loops (5, 6)."""
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    sc, mats, of, nx, ny, spp, depth = golden_shade_case(g)
    c = sc["camera"]
    for variant in (0, 12, 11, 9, 5, 6):
        scene = api.Scene.from_mesh(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
        cam = api.create_camera(nx, ny, c["lookfrom"], c["lookat"], c["focus"], c["aperture"], c["vfov"])
        st = api.SetCurandState(nx, ny, 0)
        api.create_BVHfromList(scene, st)
        scene.set_materials(mats, of)
        got = hip_render(gpu, scene, cam, st, nx, ny, spp, api.WORLD_BVH, api.SHADER_SHADE, depth, variant)
        assert same(got, g["rgb"]), (name, variant)
        assert np.array_equal(st.export()[1:], g["states_after"][1:]), (name, variant)   # state[0]: DESIGN.md §1


def test_oracle_shade_closed_forms(oracle):
    """One big triangle in front of the camera covering the image centre."""
    pts = np.array([[-50, -50, -5.5], [50, -50, -5], [0, 50, -4.5]], dtype=np.float32)
    idx = np.array([[0, 1, 2]], dtype=np.float32)
    nrm = np.array([[0, 0, -1]], dtype=np.float32)
    nx = ny = 8
    cam = oracle.make_camera((0, 0, 0), (0, 0, -1), (0, 1, 0), 40.0, np.float32(1), 0.0, 10.0)

    def centre(kind, albedo, shader, depth):
        w = oracle.World(pts, idx, nrm, 0)
        w.set_materials([(kind, albedo, 1.5)])
        img, _ = oracle.render_ex(w, None, shader, cam, nx, ny, 2, oracle.fresh_states(nx, ny), max_depth=depth)
        return img[4, 4]

    light = centre(oracle.DIFFUSE_LIGHT, (0.25, 0.5, 0.04), oracle.SHADER_SHADE, 8)
    # DiffuseLight never scatters: emitted + vec3(0.1), then sqrt, clip
    want = np.sqrt(np.array([0.25, 0.5, 0.04], dtype=np.float32) + np.float32(0.1)).astype(np.float32)
    assert np.allclose(light, want, atol=1e-6)
    # depth 0: scatter is not even called -> 0 + 0.1 for any material that does not emit
    for kind in (oracle.LAMBERTIAN, oracle.METAL, oracle.DIELECTRIC):
        assert np.allclose(centre(kind, (0.9, 0.9, 0.9), oracle.SHADER_SHADE, 0), np.sqrt(np.float32(0.1)), atol=1e-6)
    # shade_normal returns rec.normal: (0, 0, -1) -> sqrt(0), sqrt(0), sqrt(-1) = NaN, which clip() passes through
    n = centre(oracle.LAMBERTIAN, (0.5, 0.5, 0.5), oracle.SHADER_NORMAL, 8)
    assert n[0] == 0 and n[1] == 0 and np.isnan(n[2])
    # a dielectric slab seen head-on reflects a little and transmits the sky
    glass = centre(oracle.DIELECTRIC, (1, 1, 1), oracle.SHADER_SHADE, 8)
    assert np.isfinite(glass).all() and glass.min() > 0.3


def test_oracle_lambert_shade_is_unchanged_by_the_default_material_table(oracle, golden_dir):
    import os
    g = np.load(os.path.join(golden_dir, "sphere1000_96x54x4.npz"))
    sc = scenes.displaced_sphere(20, 25)
    nx, ny, spp = int(g["nx"]), int(g["ny"]), int(g["spp"])
    w = oracle.World(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
    c = sc["camera"]
    cam = oracle.make_camera(c["lookfrom"], c["lookat"], (0, 1, 0), c["vfov"], np.float32(nx) / np.float32(ny), 0.0, 10.0)
    img, _ = oracle.render_ex(w, None, oracle.SHADER_LAMBERT, cam, nx, ny, spp, oracle.fresh_states(nx, ny))
    assert np.array_equal(bits(img), bits(g["rgb"]))
    w.set_materials([(oracle.LAMBERTIAN, (0.65, 0.05, 0.05), 0.0)])
    img, _ = oracle.render_ex(w, None, oracle.SHADER_LAMBERT, cam, nx, ny, spp, oracle.fresh_states(nx, ny))
    assert np.array_equal(bits(img), bits(g["rgb"]))


def test_material_table_validation(crt_built):
    if crt_built.crt_device_count() > 0:
        pytest.skip("a HIP device is present")
    assert crt_built.crt_scene_set_materials(None, None, 0, None) == _lib.CRT_ERR_INVALID
    assert crt_built.crt_render_rows_ex(None, None, None, 8, 8, 1, 8, 0, 0, 8, 7, 0, None, None) == _lib.CRT_ERR_INVALID


# ------------------------------------------------------------------ product vs oracle (GPU)
def hip_render(gpu, scene, cam, st, nx, ny, spp, world, shader, depth, variant=0):
    d = gpu.zeros(ny * nx * 3, dtype=gpu.float32, device="cuda")
    prev = _lib.lib().crt_set_kernel_variant(variant)
    try:
        api.render_rows_ex(scene, cam, st, nx, ny, spp, 0, ny, d, world=world, shader=shader, max_depth=depth,
                           stream=gpu.cuda.current_stream())
        gpu.cuda.synchronize()
    finally:
        _lib.lib().crt_set_kernel_variant(prev)
    return d.cpu().numpy().reshape(ny, nx, 3)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["cornell", "spheres", "soup"])
@pytest.mark.parametrize("world", ["list", "bvh"])
def test_gpu_shaders_and_materials_match_oracle(gpu, oracle, kind, world):
    """All three shaders, all four material classes, max_depth 0 / 1 / 8, both worlds; RNG states
    after the frame included (the number of draws depends on every bounce)."""
    sc, of = material_scene(kind)
    nx, ny, spp = (48, 36, 3) if kind != "soup" else (32, 24, 2)
    w, ocam = oracle_world(oracle, sc, of, nx, ny)
    scene = api.Scene.from_mesh(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
    c = sc["camera"]
    cam = api.create_camera(nx, ny, c["lookfrom"], c["lookat"], c["focus"], c["aperture"], c["vfov"])
    if world == "bvh":                                 # built before the materials are set: the slot order must follow
        api.create_BVHfromList(scene, None, seed=0)
    scene.set_materials(MATS, of)
    for shader, depth in [(api.SHADER_SHADE, 8), (api.SHADER_SHADE, 1), (api.SHADER_SHADE, 0), (api.SHADER_LAMBERT, 8),
                          (api.SHADER_NORMAL, 8)]:
        states = oracle.fresh_states(nx, ny)
        bvh = oracle.Bvh(w, states) if world == "bvh" else None
        ref, states = oracle.render_ex(w, bvh, shader, ocam, nx, ny, spp, states, max_depth=depth)
        after = np.concatenate([states["d"][:, None], states["v"]], axis=1)
        # shade: the walk's forced general modes and the skip links; LambertShade with materials: the tiled kernels (0; list world also
        # with tiny capacities = shared tiles and the pool fallback, 3, and with a full binning log, 8) and the generic kernel (1)
        variants = (0,)
        if world == "bvh" and shader == api.SHADER_SHADE and depth == 8:
            variants = (0, 5, 6, 9, 11, 12)   # 9 skip links, 11 one stack per lane, 12 the shared walk with a short list (abandoned trips)
        elif shader in (api.SHADER_LAMBERT, api.SHADER_NORMAL):
            variants = (0, 1, 3, 8) if world == "list" else (0, 1)
        for variant in variants:
            st = api.SetCurandState(nx, ny, 0)
            if world == "bvh":
                api.create_BVHfromList(scene, st)      # the reference's order: SetCurandState, build, render
            got = hip_render(gpu, scene, cam, st, nx, ny, spp, api.WORLD_BVH if world == "bvh" else api.WORLD_LIST, shader, depth, variant)
            assert same(got, ref), (shader, depth, variant, float(np.nanmax(np.abs(got - ref))))
            assert np.array_equal(st.export()[1:], after[1:]), (shader, depth, variant)


@pytest.mark.gpu
def test_gpu_lambert_entry_points_follow_the_material_table(gpu, oracle):
    """crt_render_rows / crt_render_rows_bvh on a scene whose table is not one Lambertian shade every hit with its own
    material (the tiled kernels' material mode, or the generic kernel where they are not eligible: same pixels as the
    oracle's LambertShade with those materials); setting the table back to one Lambertian restores the specialised
    kernels and their pixels."""
    sc, of = material_scene("spheres")
    nx, ny, spp = 64, 40, 2
    w, ocam = oracle_world(oracle, sc, of, nx, ny)
    scene = api.Scene.from_mesh(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
    c = sc["camera"]
    cam = api.create_camera(nx, ny, c["lookfrom"], c["lookat"], c["focus"], c["aperture"], c["vfov"])
    scene.set_materials(MATS, of)
    ref, _ = oracle.render_ex(w, None, oracle.SHADER_LAMBERT, ocam, nx, ny, spp, oracle.fresh_states(nx, ny))
    st = api.SetCurandState(nx, ny, 0)
    d = gpu.zeros(ny * nx * 3, dtype=gpu.float32, device="cuda")
    api.render_rows(scene, cam, st, nx, ny, spp, 0, ny, d, stream=gpu.cuda.current_stream())
    gpu.cuda.synchronize()
    assert same(d.cpu().numpy().reshape(ny, nx, 3), ref)
    one = [(api.LAMBERTIAN, (0.2, 0.4, 0.6), 0.0)]
    scene.set_materials(one)
    w.set_materials(one)
    ref, _ = oracle.render_ex(w, None, oracle.SHADER_LAMBERT, ocam, nx, ny, spp, oracle.fresh_states(nx, ny))
    st = api.SetCurandState(nx, ny, 0)
    api.render_rows(scene, cam, st, nx, ny, spp, 0, ny, d, stream=gpu.cuda.current_stream())
    gpu.cuda.synchronize()
    assert same(d.cpu().numpy().reshape(ny, nx, 3), ref)
    with pytest.raises(_lib.CrtError):
        api.render_rows_ex(scene, cam, st, nx, ny, spp, 0, ny, d, shader=api.SHADER_SHADE, max_depth=40)
    with pytest.raises(_lib.CrtError):
        scene.set_materials([(9, (0, 0, 0), 0.0)])


@pytest.mark.gpu
@pytest.mark.parametrize("world", ["list", "bvh"])
def test_gpu_lambert_materials_through_the_tiled_kernels_on_a_dense_pile(gpu, oracle, world):
    """LambertShade with a Material* per triangle rides the tiled kernels (render_pixel<.., MAT>): what a hit draws depends
    on the hit triangle's material class, so the shared tiles' master wave cannot run the hit branch's draws ahead.  A pile
    of 2500 overlapping triangles puts hundreds of entries into every central tile (shared tiles, chunked staging in the BVH
    world) with all four classes mixed: pixels and RNG states against the oracle, and against the generic kernel."""
    rng = np.random.RandomState(77)
    n = 2500
    c = np.stack([rng.uniform(-0.6, 0.6, n), rng.uniform(-0.6, 0.6, n), rng.uniform(-9, -5, n)], axis=-1)
    tri = c[:, None, :] + rng.uniform(-0.5, 0.5, (n, 3, 3))
    sc = dict(points=tri.reshape(-1, 3).astype(np.float32), idx=np.arange(n * 3, dtype=np.float32).reshape(n, 3), normals=None, index_base=0,
              camera=dict(lookfrom=(0.0, 0.0, 0.0), lookat=(0.0, 0.0, -1.0), vfov=60.0, aperture=0.0, focus=10.0))
    of = rng.randint(0, len(MATS), n).astype(np.int32)
    nx, ny, spp = 96, 64, 3
    w, ocam = oracle_world(oracle, sc, of, nx, ny)
    scene = api.Scene.from_mesh(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
    cam = api.create_camera(nx, ny, (0.0, 0.0, 0.0), (0.0, 0.0, -1.0), 10.0, 0.0, 60.0)
    states = oracle.fresh_states(nx, ny)
    bvh = oracle.Bvh(w, states) if world == "bvh" else None
    ref, states = oracle.render_ex(w, bvh, oracle.SHADER_LAMBERT, ocam, nx, ny, spp, states)
    after = np.concatenate([states["d"][:, None], states["v"]], axis=1)
    for variant in (0, 1):
        st = api.SetCurandState(nx, ny, 0)
        if world == "bvh":
            api.create_BVHfromList(scene, st)
        scene.set_materials(MATS, of)
        if variant == 0:
            api.tiled_stats(1)
        got = hip_render(gpu, scene, cam, st, nx, ny, spp, api.WORLD_BVH if world == "bvh" else api.WORLD_LIST, api.SHADER_LAMBERT, 8, variant)
        if variant == 0:
            stats = api.tiled_stats(0)
            assert stats[4] > 0 and stats[6] > 64, stats[:8]          # the tiled kernels ran, and some tile holds more than one staging chunk
            assert stats[43] > 0, stats[43:47]                        # ... with slices that go as far pairs (MAT instantiations)
        assert same(got, ref), (world, variant, int((bits(got) != bits(ref)).any(axis=2).sum()))
        assert np.array_equal(st.export()[1:], after[1:]), (world, variant)


@pytest.mark.gpu
@pytest.mark.parametrize("world", ["list", "bvh"])
def test_gpu_generic_kernel_edge_frames(gpu, oracle, world):
    """The pixel queue at its edges: frames smaller than a tile, one pixel, zero samples (0 / 0 = NaN pixels, states untouched),
    against the oracle."""
    sc, of = material_scene("soup")
    for nx, ny, spp in ((1, 1, 2), (3, 2, 1), (9, 7, 0), (17, 5, 3)):
        w, ocam = oracle_world(oracle, sc, of, nx, ny)
        c = sc["camera"]
        scene = api.Scene.from_mesh(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
        cam = api.create_camera(nx, ny, c["lookfrom"], c["lookat"], c["focus"], c["aperture"], c["vfov"])
        st = api.SetCurandState(nx, ny, 0)
        states = oracle.fresh_states(nx, ny)
        bvh = None
        if world == "bvh":
            api.create_BVHfromList(scene, st)
            bvh = oracle.Bvh(w, states)
        scene.set_materials(MATS, of)
        ref, states = oracle.render_ex(w, bvh, oracle.SHADER_SHADE, ocam, nx, ny, spp, states, max_depth=4)
        after = np.concatenate([states["d"][:, None], states["v"]], axis=1)
        got = hip_render(gpu, scene, cam, st, nx, ny, spp, api.WORLD_BVH if world == "bvh" else api.WORLD_LIST, api.SHADER_SHADE, 4)
        assert same(got, ref), (world, nx, ny, spp)
        assert np.array_equal(st.export()[1:], after[1:]), (world, nx, ny, spp)


@pytest.mark.gpu
@pytest.mark.parametrize("world", ["list", "bvh"])
def test_gpu_generic_kernel_on_strip_sets_equals_the_full_frame(gpu, world):
    """The generic kernel's pixel queue runs over the rows a launch owns: every rank's strip set (32-row strips dealt
    round-robin, crt_render_strips / _bvh; a scene with materials, variant 1 = LambertShade through the generic kernel) is the
    full frame's rows, pixels and RNG states — frames whose height is not a multiple of the strip or the tile height."""
    from cudaraytracer_amd.dist import partition_strips
    rng = np.random.RandomState(5)
    n = 300
    sc = scenes.random_soup(n, 5, extent=3.0, size=1.5)
    of = rng.randint(0, len(MATS), n).astype(np.int32)
    nx, ny, spp, ranks = 75, 213, 2, 3
    c = sc["camera"]
    scene = api.Scene.from_mesh(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
    cam = api.create_camera(nx, ny, c["lookfrom"], c["lookat"], c["focus"], c["aperture"], c["vfov"])
    st = api.SetCurandState(nx, ny, 0)
    if world == "bvh":
        api.create_BVHfromList(scene, st)
    scene.set_materials(MATS, of)
    full = hip_render(gpu, scene, cam, st, nx, ny, spp, api.WORLD_BVH if world == "bvh" else api.WORLD_LIST, api.SHADER_LAMBERT, 8, variant=1)
    full_states = st.export()
    prev = _lib.lib().crt_set_kernel_variant(1)
    try:
        for r in range(ranks):
            strips = partition_strips(ny, ranks, r)
            rows = sum(cnt for _, cnt in strips)
            s = api.CurandState.for_strips(nx, ny, r, 32, ranks, 0)
            piece = gpu.zeros(rows * nx * 3, dtype=gpu.float32, device="cuda")
            api.render_strips(scene, cam, s, nx, ny, spp, r, 32, ranks, piece, stream=gpu.cuda.current_stream(), bvh=world == "bvh")
            gpu.cuda.synchronize()
            got = piece.cpu().numpy().reshape(rows, nx, 3)
            want = np.concatenate([full[y0:y0 + cnt] for y0, cnt in strips], axis=0)
            want_states = np.concatenate([full_states[y0 * nx:(y0 + cnt) * nx] for y0, cnt in strips], axis=0)
            skip = 1 if (r == 0 and world == "bvh") else 0   # pixel 0's stream: the BVH build drew from the full frame's state[0]
            assert same(got.reshape(-1, 3)[skip:], want.reshape(-1, 3)[skip:]), (world, r)
            assert np.array_equal(s.export()[skip:], want_states[skip:]), (world, r)
    finally:
        _lib.lib().crt_set_kernel_variant(prev)


@pytest.mark.gpu
def test_gpu_shade_on_the_c3_mesh_matches_oracle_crops(gpu, oracle, tmp_path):
    """The generic kernel at the size the benchmarks use: BASELINE configs[2]'s mesh (100 000 triangles, OBJ round
    trip, reference camera), BVH world, all four material classes dealt over the triangles, recursive shade with
    depth 8 at 1920x1080 — crops on the silhouette and in the interior against the oracle, bit for bit."""
    path, nx, ny, _ = scenes.baseline_obj("C3", str(tmp_path))
    spp = 2
    mats = [(api.LAMBERTIAN, (0.65, 0.05, 0.05), 0.0), (api.METAL, (0.8, 0.85, 0.88), 0.1), (api.DIELECTRIC, (1.0, 1.0, 1.0), 1.5),
            (api.DIFFUSE_LIGHT, (4.0, 4.0, 4.0), 0.0)]
    scene = api.Scene.from_obj(path, scale=1.0)
    of = (np.arange(scene.n_triangles) % 4).astype(np.int32)
    cam = api.init_camera(nx, ny)
    st = api.SetCurandState(nx, ny, 0)
    api.create_BVHfromList(scene, st)
    scene.set_materials(mats, of)
    got = hip_render(gpu, scene, cam, st, nx, ny, spp, api.WORLD_BVH, api.SHADER_SHADE, 8)
    after = st.export()
    for variant in (9, 11, 12):   # the whole frame, every walk: skip links, one stack per lane, the shared walk with abandoned trips
        st2 = api.SetCurandState(nx, ny, 0)   # state[0] (pixel 0's stream) differs: the BVH build drew from st's
        got2 = hip_render(gpu, scene, cam, st2, nx, ny, spp, api.WORLD_BVH, api.SHADER_SHADE, 8, variant)
        assert same(got2.reshape(-1, 3)[1:], got.reshape(-1, 3)[1:]), variant
        assert np.array_equal(st2.export()[1:], after[1:]), variant
    lam = (got[..., 0] - got[..., 2]) > 0.05                                   # something that is not sky
    rows_hit = np.where(lam.any(axis=1))[0]
    r_mid = int((rows_hit[0] + rows_hit[-1]) // 2)
    c_edge = int(np.where(lam[r_mid])[0][0])
    pts, idx = oracle.parse_obj(path)
    w = oracle.World(pts, idx, None, 1)
    w.set_materials(mats, of)
    ocam = oracle.make_camera((0, 100, 1000), (0, 150, 0), (0, 1, 0), 40.0, np.float32(nx) / np.float32(ny), 0.0, 10.0)
    for win in [(r_mid, r_mid + 12, c_edge - 24, c_edge + 24), (r_mid + 40, r_mid + 48, nx // 2 - 16, nx // 2 + 16)]:
        states = oracle.fresh_states(nx, ny, window=win)
        bvh = oracle.Bvh(w, states)
        ref, _ = oracle.render_ex(w, bvh, oracle.SHADER_SHADE, ocam, nx, ny, spp, states, window=win, max_depth=8)
        r0, r1, c0, c1 = win
        assert same(got[r0:r1, c0:c1], ref[r0:r1, c0:c1]), win


@pytest.mark.gpu
@pytest.mark.parametrize("block", range(4))
def test_gpu_shade_random_soups_match_oracle(gpu, oracle, block):
    """Randomised: small soups (triangles in front of, around and behind the eye, coincident duplicates, degenerate
    ones), random material tables and assignments, both worlds, random depth and shader — pixels (NaNs included) and
    RNG states against the oracle."""
    for seed in range(1000 + 12 * block, 1000 + 12 * (block + 1)):
        rng = np.random.RandomState(seed)
        n = int(rng.choice([3, 17, 64, 150]))
        sc = scenes.random_soup(n, seed, extent=float(rng.choice([2.0, 4.0])), size=float(rng.choice([0.7, 1.5, 3.0])))
        kinds = [api.LAMBERTIAN, api.METAL, api.DIELECTRIC, api.DIFFUSE_LIGHT]
        mats = [(int(rng.choice(kinds)), tuple(float(x) for x in rng.uniform(0.05, 1.0, 3)), float(rng.choice([0.0, 0.3, 1.5, 2.4])))
                for _ in range(int(rng.randint(1, 7)))]
        of = rng.randint(0, len(mats), n).astype(np.int32)
        nx, ny, spp = int(rng.choice([16, 33])), int(rng.choice([8, 20])), int(rng.choice([1, 2, 3]))
        world = str(rng.choice(["list", "bvh"]))
        shader = int(rng.choice([api.SHADER_SHADE, api.SHADER_SHADE, api.SHADER_LAMBERT, api.SHADER_NORMAL]))
        depth = int(rng.choice([0, 1, 2, 8, 31]))
        w = oracle.World(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
        w.set_materials(mats, of)
        c = sc["camera"]
        ocam = oracle.make_camera(c["lookfrom"], c["lookat"], (0, 1, 0), c["vfov"], np.float32(nx) / np.float32(ny), c["aperture"], c["focus"])
        scene = api.Scene.from_mesh(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
        cam = api.create_camera(nx, ny, c["lookfrom"], c["lookat"], c["focus"], c["aperture"], c["vfov"])
        st = api.SetCurandState(nx, ny, 0)
        states = oracle.fresh_states(nx, ny)
        bvh = None
        if world == "bvh":
            api.create_BVHfromList(scene, st)
            bvh = oracle.Bvh(w, states)
        scene.set_materials(mats, of)
        ref, states = oracle.render_ex(w, bvh, shader, ocam, nx, ny, spp, states, max_depth=depth)
        after = np.concatenate([states["d"][:, None], states["v"]], axis=1)
        got = hip_render(gpu, scene, cam, st, nx, ny, spp, api.WORLD_BVH if world == "bvh" else api.WORLD_LIST, shader, depth)
        assert same(got, ref), (seed, world, shader, depth, n)
        assert np.array_equal(st.export()[1:], after[1:]), (seed, world, shader, depth, n)


@pytest.mark.gpu
def test_gpu_shade_with_tiny_and_huge_stored_normals_matches_oracle(gpu, oracle):
    """Stored normals of subnormal size: the reference's float dot(direction, normal) rounds to +-0, `< 0` is false and the
    triangle IS tested (triangle.h:61) whatever the exact sign says — so the pair records' normal cones (built in double on
    the host) must not cover such a triangle's subtree.  Same for normals near FLT_MAX (inf - inf = NaN in the float dot).
    Every other triangle of the pile keeps an honest normal, so cones exist around the odd ones."""
    rng = np.random.RandomState(77)
    sc = scenes.random_soup(240, 77, extent=3.0, size=1.2, z_range=(-9.0, -3.0))
    nrm = rng.normal(size=(240, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True).astype(np.float32)
    nrm[::3] *= np.float32(-1.0)
    tiny, huge = np.arange(5, 240, 16), np.arange(11, 240, 32)
    nrm[tiny] = (nrm[tiny].astype(np.float64) * 1e-42).astype(np.float32)          # subnormal components
    nrm[huge] = (nrm[huge].astype(np.float64) * 3e38).astype(np.float32)
    sc["normals"] = nrm
    mats = [(api.LAMBERTIAN, (0.65, 0.05, 0.05), 0.0), (api.METAL, (0.8, 0.85, 0.88), 0.0), (api.DIELECTRIC, (1, 1, 1), 1.5)]
    of = (np.arange(240) % 3).astype(np.int32)
    nx, ny, spp, depth = 48, 32, 3, 6
    c = sc["camera"]
    w = oracle.World(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
    w.set_materials(mats, of)
    ocam = oracle.make_camera(c["lookfrom"], c["lookat"], (0, 1, 0), c["vfov"], np.float32(nx) / np.float32(ny), c["aperture"], c["focus"])
    states = oracle.fresh_states(nx, ny)
    bvh = oracle.Bvh(w, states)
    ref, states = oracle.render_ex(w, bvh, oracle.SHADER_SHADE, ocam, nx, ny, spp, states, max_depth=depth)
    after = np.concatenate([states["d"][:, None], states["v"]], axis=1)
    for variant in (0, 11, 9):
        scene = api.Scene.from_mesh(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
        cam = api.create_camera(nx, ny, c["lookfrom"], c["lookat"], c["focus"], c["aperture"], c["vfov"])
        st = api.SetCurandState(nx, ny, 0)
        api.create_BVHfromList(scene, st)
        scene.set_materials(mats, of)
        got = hip_render(gpu, scene, cam, st, nx, ny, spp, api.WORLD_BVH, api.SHADER_SHADE, depth, variant)
        assert same(got, ref), variant
        assert np.array_equal(st.export()[1:], after[1:]), variant
