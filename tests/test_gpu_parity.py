"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI,
against the CPU oracle and the committed golden fixtures.  Bar: bit-exact float buffers
(BASELINE.json allows +-1 ULP; we hold 0 ULP) and identical RNG consumption."""
import ctypes as C
import os

import numpy as np
import pytest

from cudaraytracer_amd import _lib, api, scenes

pytestmark = pytest.mark.gpu


VARIANTS = {"brute": 1, "tiled": 2, "tiled_tinycaps": 3, "tiled_scalar_light": 4, "tiled_tinylog": 8, "tiled_row_slabs": 10}


@pytest.fixture(params=sorted(VARIANTS))
def variant(request, gpu, crt_built):
    """Every kernel variant must produce the same pixels (crt_set_kernel_variant)."""
    prev = crt_built.crt_set_kernel_variant(VARIANTS[request.param])
    yield request.param
    crt_built.crt_set_kernel_variant(prev)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def make(sc, nx, ny):
    scene = api.Scene.from_mesh(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
    c = sc["camera"]
    cam = api.create_camera(nx, ny, c["lookfrom"], c["lookat"], c["focus"], c["aperture"], c["vfov"])
    return scene, cam


def oracle_pair(oracle, sc, nx, ny):
    w = oracle.World(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
    c = sc["camera"]
    cam = oracle.make_camera(c["lookfrom"], c["lookat"], (0, 1, 0), c["vfov"],
                             np.float32(nx) / np.float32(ny), c["aperture"], c["focus"])
    return w, cam


def hip_render_with_states(gpu, scene, cam, nx, ny, spp, frames=1, row_begin=0, row_count=None):
    rows = ny - row_begin if row_count is None else row_count
    st = api.SetCurandState(nx, ny, 0, row_begin, rows)
    d = gpu.zeros(rows * nx * 3, dtype=gpu.float32, device="cuda")
    outs = []
    for f in range(frames):
        api.render_rows(scene, cam, st, nx, ny, spp, row_begin, rows, d, frame_index=f,
                        stream=gpu.cuda.current_stream())
        gpu.cuda.synchronize()
        outs.append(d.cpu().numpy().reshape(rows, nx, 3).copy())
    return outs, st.export()


# ---------------------------------------------------------------- arithmetic self-checks
def test_fast_reciprocal_is_exact(gpu, crt_built):
    """f = 1.0/a is computed as v_rcp_f32 + one Newton step on bounded scenes: it must equal the
    IEEE quotient for EVERY float in 2^-100 .. 2^100 (3.4e9 values, both signs)."""
    bad = C.c_uint64(123)
    assert crt_built.crt_debug_check_reciprocal(27, 227, C.byref(bad)) == 0
    assert bad.value == 0


def test_three_instruction_quotients_are_exact(gpu, crt_built):
    """x/len and (x+U)/nx are computed as q0 = x*y, r = fma(-q0, b, x), q = fma(r, y, q0) with
    y = RN(1/b) in the shading kernel: compared with the IEEE quotient for EVERY pair of
    significands (2^23 x 2^23 = 7.04e13 pairs, ~30 s) and, for the scaling and sign paths, for
    every float divisor in 2^-20..2^20 against 64 numerators of smaller exponent."""
    bad = C.c_uint64(123)
    assert crt_built.crt_debug_check_division(1, 0, 63, C.byref(bad)) == 0
    assert bad.value == 0
    assert crt_built.crt_debug_check_division(2, 107, 147, C.byref(bad)) == 0
    assert bad.value == 0


def test_fast_sqrt_is_exact(gpu, crt_built):
    """len = sqrt(x*x + y*y + z*z) uses v_sqrt_f32 + the residuals of its two neighbours on the
    guarded range: equal to sqrtf for EVERY float in 2^-100 .. 2^100."""
    bad = C.c_uint64(123)
    assert crt_built.crt_debug_check_division(3, 27, 227, C.byref(bad)) == 0
    assert bad.value == 0


def test_fused_uniform_forms_are_exact(gpu, crt_built):
    """curand_uniform = x * 2^-32 + 2^-33 (two operations; the product is a power-of-two scaling of a float, hence exact) and
    2.0f * u - 1.0f are computed with ONE fma for the sum in the kernels: the same bits for EVERY one of the 2^32 draws."""
    bad = C.c_uint64(123)
    assert crt_built.crt_debug_check_division(4, 0, 0, C.byref(bad)) == 0
    assert bad.value == 0


def test_unbounded_scene_uses_ieee_division_and_brute_force(gpu, oracle, crt_built):
    """Coordinates beyond 2^38: no tiling, no fast reciprocal — still the reference's pixels."""
    sc = scenes.random_soup(40, 5)
    sc["points"] = (sc["points"] * np.float32(1.0e12)).astype(np.float32)
    nx, ny, spp = 24, 16, 2
    scene, cam = make(sc, nx, ny)
    (rgb,), _ = hip_render_with_states(gpu, scene, cam, nx, ny, spp)
    w, ocam = oracle_pair(oracle, sc, nx, ny)
    ref, _ = oracle.render(w, ocam, nx, ny, spp)
    assert np.array_equal(bits(rgb), bits(ref))


# ---------------------------------------------------------------- RNG
def test_rng_init_matches_oracle(gpu, oracle, golden_dir):
    kat = np.load(os.path.join(golden_dir, "xorwow_kat.npz"))
    st = api.CurandState(0, 300, 0)
    assert np.array_equal(st.export(), kat["states_0_300"])
    # a band far inside a 4K image, and 64-bit seeds
    for first, count, seed in [(2073599 - 100, 333, 0), (8294399 - 64, 65, 0), (12345, 257, 0x123456789ABCDEF)]:
        got = api.CurandState(first, count, seed).export()
        ref = oracle.curand_states(first, count, seed)
        ref = np.concatenate([ref["d"][:, None], ref["v"]], axis=1)
        assert np.array_equal(got, ref), (first, count, seed)


# ---------------------------------------------------------------- golden framebuffers
GOLDEN = {
    "c1_quad_256x256x1": scenes.quad,
    "c2_cornell_100x75x16": scenes.cornell32,
    "sphere1000_96x54x4": lambda: scenes.displaced_sphere(20, 25),
    "soup96_48x48x4": lambda: scenes.random_soup(96, 7),
}


@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_render_matches_golden(gpu, golden_dir, name, variant):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    nx, ny, spp = int(g["nx"]), int(g["ny"]), int(g["spp"])
    scene, cam = make(GOLDEN[name](), nx, ny)
    assert np.array_equal(bits(cam.as_array()), bits(g["camera"]))
    (rgb,), states = hip_render_with_states(gpu, scene, cam, nx, ny, spp)
    assert np.array_equal(bits(rgb), bits(g["rgb"])), "max |diff| %g" % np.nanmax(np.abs(rgb - g["rgb"]))
    # stream 0 is excluded: in the oracle it also absorbs every pixel's lens-sample draws
    # (the reference passes &state[0] to get_ray in every thread, core/render.h:118)
    assert np.array_equal(states[1:2048], g["states_after"][1:])
    assert np.array_equal(api.to_rgba8(rgb), g["rgba"])
    # host convenience entry point (crt_render: alloc + rng init + render + copy)
    assert np.array_equal(bits(api.render(nx, ny, spp, scene, cam)), bits(g["rgb"]))


def test_small_bunny_fixture(gpu, golden_dir, variant):
    """The reference-held mesh (CudaTest/objects/small_bunny.obj, arrays in the fixture), 64x64x4."""
    g = np.load(os.path.join(golden_dir, "small_bunny_64x64x4.npz"))
    nx, ny, spp = int(g["nx"]), int(g["ny"]), int(g["spp"])
    scene = api.Scene.from_mesh(g["points"], g["idx"], None, 1)
    assert scene.n_triangles == 4968
    cam = api.create_camera(nx, ny, tuple(g["lookfrom"]), tuple(g["lookat"]), 10.0, 0.0, 40.0)
    assert np.array_equal(bits(cam.as_array()), bits(g["camera"]))
    (rgb,), states = hip_render_with_states(gpu, scene, cam, nx, ny, spp)
    assert np.array_equal(bits(rgb), bits(g["rgb"]))
    assert np.array_equal(states[1:], g["states_after"][1:])
    assert np.array_equal(api.to_rgba8(rgb), g["rgba"])


# ---------------------------------------------------------------- live oracle comparisons
@pytest.mark.parametrize("seed,ntri,nx,ny,spp", [(1, 64, 40, 24, 3), (2, 200, 33, 17, 2), (3, 17, 16, 16, 8),
                                                 (4, 1, 9, 9, 2), (5, 500, 24, 40, 1)])
def test_random_soup_matches_oracle(gpu, oracle, variant, seed, ntri, nx, ny, spp):
    sc = scenes.random_soup(ntri, seed)
    scene, cam = make(sc, nx, ny)
    (rgb,), states = hip_render_with_states(gpu, scene, cam, nx, ny, spp)
    w, ocam = oracle_pair(oracle, sc, nx, ny)
    ref, ost = oracle.render(w, ocam, nx, ny, spp)
    assert np.array_equal(bits(rgb), bits(ref))
    assert np.array_equal(states[1:], np.concatenate([ost["d"][:, None], ost["v"]], axis=1)[1:])


def test_ragged_sizes_and_empty_scene(gpu, oracle, variant):
    # image sizes that are not multiples of the 16x16 block / 8x8 wave tile, and N = 0
    sc = scenes.cornell32()
    for nx, ny in [(1, 1), (7, 3), (17, 9), (31, 33)]:
        scene, cam = make(sc, nx, ny)
        (rgb,), _ = hip_render_with_states(gpu, scene, cam, nx, ny, 2)
        w, ocam = oracle_pair(oracle, sc, nx, ny)
        ref, _ = oracle.render(w, ocam, nx, ny, 2)
        assert np.array_equal(bits(rgb), bits(ref)), (nx, ny)
    empty = dict(points=np.zeros((3, 3), np.float32), idx=np.zeros((0, 3), np.float32), normals=None,
                 index_base=0, camera=sc["camera"])
    scene, cam = make(empty, 20, 10)
    (rgb,), _ = hip_render_with_states(gpu, scene, cam, 20, 10, 2)
    w, ocam = oracle_pair(oracle, empty, 20, 10)
    ref, _ = oracle.render(w, ocam, 20, 10, 2)
    assert np.array_equal(bits(rgb), bits(ref))


def test_frames_continue_the_rng_streams(gpu, oracle, variant):
    """renderListAnimation keeps curand_state across frames (core/render.h:174-186)."""
    sc = scenes.displaced_sphere(6, 8)
    nx, ny, spp = 32, 24, 2
    scene, cam = make(sc, nx, ny)
    outs, _ = hip_render_with_states(gpu, scene, cam, nx, ny, spp, frames=2)
    w, ocam = oracle_pair(oracle, sc, nx, ny)
    f0, st = oracle.render(w, ocam, nx, ny, spp)
    f1, st = oracle.render(w, ocam, nx, ny, spp, states=st)
    assert np.array_equal(bits(outs[0]), bits(f0))
    # pixel 0 of later frames depends on the reference's state[0] race (core/render.h:118):
    # excluded by definition (DESIGN.md "serial-order semantics")
    a, b = bits(outs[1]).reshape(-1, 3)[1:], bits(f1).reshape(-1, 3)[1:]
    assert np.array_equal(a, b)


def test_pose_updates_between_frames_match_oracle(gpu, oracle, variant):
    """The animation loop (core/render.h:174-186): updateFBXObj's update_pose rewrites every
    triangle's vertices from the frame's point positions (normals kept, createScene.h:99-110),
    then the frame is rendered with the RNG states of the previous one."""
    sc = scenes.displaced_sphere(10, 12)
    rng = np.random.RandomState(3)
    nx, ny, spp = 48, 32, 3
    pts0 = np.asarray(sc["points"], np.float32).reshape(-1, 3)
    idx = np.asarray(sc["idx"], np.float32).reshape(-1, 3)
    normals = rng.normal(size=(len(idx), 3)).astype(np.float32)          # create_FBXMesh always passes face normals
    normals /= np.linalg.norm(normals, axis=1, keepdims=True).astype(np.float32)
    sc = dict(sc, normals=normals)
    poses = [pts0,
             (pts0 * np.float32(1.1) + np.array([5.0, -3.0, 2.0], np.float32)).astype(np.float32),
             (pts0 + rng.normal(scale=4.0, size=pts0.shape)).astype(np.float32)]
    scene, cam = make(sc, nx, ny)
    w, ocam = oracle_pair(oracle, sc, nx, ny)
    st = api.SetCurandState(nx, ny, 0)
    d = gpu.zeros(ny * nx * 3, dtype=gpu.float32, device="cuda")
    ost = None
    for f, pose in enumerate(poses):
        scene.update_pose(pose, stream=gpu.cuda.current_stream())
        api.render_rows(scene, cam, st, nx, ny, spp, 0, ny, d, frame_index=f, stream=gpu.cuda.current_stream())
        gpu.cuda.synchronize()
        w.update_pose(pose)
        ref, ost = oracle.render(w, ocam, nx, ny, spp, states=ost)
        got = bits(d.cpu().numpy().reshape(ny, nx, 3)).reshape(-1, 3)
        skip = 0 if f == 0 else 1                                          # pixel 0 of later frames: state[0] race
        assert np.array_equal(got[skip:], bits(ref).reshape(-1, 3)[skip:]), f
    with pytest.raises(api.CrtError):
        scene.update_pose(pts0[:-1])                                       # wrong point count


def test_row_bands_equal_full_image(gpu, variant):
    sc = scenes.displaced_sphere(10, 12)
    nx, ny, spp = 100, 75, 2
    scene, cam = make(sc, nx, ny)
    (full,), _ = hip_render_with_states(gpu, scene, cam, nx, ny, spp)
    from cudaraytracer_amd.dist import partition_rows
    for world in (2, 3, 8):
        parts = []
        for r in range(world):
            b, c = partition_rows(ny, world, r)
            (band,), _ = hip_render_with_states(gpu, scene, cam, nx, ny, spp, row_begin=b, row_count=c)
            parts.append(band)
        assert np.array_equal(bits(np.concatenate(parts, axis=0)), bits(full)), world


def test_round_robin_strips_equal_full_image(gpu, variant):
    """crt_render_strips (multi-GPU load balancing): every rank's strip set, rendered on its
    own, reproduces the full frame bit for bit, and so do the RNG states."""
    from cudaraytracer_amd.dist import partition_strips
    sc = scenes.displaced_sphere(10, 12)
    nx, ny, spp = 90, 150, 2
    scene, cam = make(sc, nx, ny)
    (full,), full_states = hip_render_with_states(gpu, scene, cam, nx, ny, spp)
    full_states = full_states.reshape(ny, nx, 6)
    for world in (2, 3, 8):
        for r in range(world):
            strips = partition_strips(ny, world, r)
            if not strips:
                continue
            rows = sum(c for _, c in strips)
            assert rows == api.strip_rows(ny, r, 32, world)
            st = api.CurandState.for_strips(nx, ny, r, 32, world, 0)
            d = gpu.zeros(rows * nx * 3, dtype=gpu.float32, device="cuda")
            api.render_strips(scene, cam, st, nx, ny, spp, r, 32, world, d, stream=gpu.cuda.current_stream())
            gpu.cuda.synchronize()
            got = d.cpu().numpy().reshape(rows, nx, 3)
            ref = np.concatenate([full[y0:y0 + c] for y0, c in strips], axis=0)
            assert np.array_equal(bits(got), bits(ref)), (world, r)
            ref_st = np.concatenate([full_states[y0:y0 + c] for y0, c in strips], axis=0).reshape(-1, 6)
            got_st = st.export()
            keep = np.ones(len(ref_st), bool)
            if r == 0:
                keep[0] = False            # stream 0 (pixel 0): see DESIGN.md §1
            assert np.array_equal(got_st[keep], ref_st[keep]), (world, r)


def test_obj_path_and_png(gpu, oracle, tmp_path):
    sc = scenes.displaced_sphere(8, 10, index_base=1)
    path = scenes.write_obj(str(tmp_path / "mesh.obj"), sc)
    scene = api.Scene.from_obj(path, scale=1.5)
    assert scene.n_triangles == 160
    nx, ny, spp = 48, 27, 2
    cam = api.init_camera(nx, ny)
    rgb = api.render(nx, ny, spp, scene, cam)
    pts, idx = oracle.parse_obj(path)
    w = oracle.World((pts * np.float32(1.5)).astype(np.float32), idx, None, 1)
    ocam = oracle.make_camera((0, 100, 1000), (0, 150, 0), (0, 1, 0), 40.0, np.float32(nx) / np.float32(ny), 0.0, 10.0)
    ref, _ = oracle.render(w, ocam, nx, ny, spp)
    assert np.array_equal(bits(rgb), bits(ref))
    from test_abi import decode_png
    p = api.WritePng(nx, ny, 0, rgb, folder=str(tmp_path / "images" / "moveTest"))
    assert np.array_equal(decode_png(p), oracle.to_rgba8(ref))


def test_aperture_is_rejected(gpu):
    sc = scenes.quad()
    scene = api.Scene.from_mesh(sc["points"], sc["idx"], sc["normals"])
    cam = api.create_camera(8, 8, (0, 0, 0), (0, 0, -1), 10.0, 0.5, 40)
    with pytest.raises(_lib.CrtError) as e:
        api.render(8, 8, 1, scene, cam)
    assert e.value.code == _lib.CRT_ERR_UNSUPPORTED


# ---------------------------------------------------------------- BASELINE sizes
def test_c2_full_size_matches_oracle(gpu, c2_oracle_frame, variant):
    """configs[1]: 800x600, 16 spp, Cornell-32 — the oracle's full frame (ten seconds of CPU) is rendered once per session."""
    sc, nx, ny, spp, ref, _ = c2_oracle_frame
    scene, cam = make(sc, nx, ny)
    (rgb,), _ = hip_render_with_states(gpu, scene, cam, nx, ny, spp)
    assert np.array_equal(bits(rgb), bits(ref))


def _soup_scene(kind, seed):
    """Adversarial geometry for the tiled classification: the camera sits inside the soup,
    triangles pass through / behind / next to the eye, sizes span 6 orders of magnitude."""
    rng = np.random.RandomState(seed)
    if kind == "around_eye":
        sc = scenes.random_soup(300, seed, extent=3.0, size=2.5, z_range=(-4.0, 4.0))
    elif kind == "tiny_far":
        sc = scenes.random_soup(400, seed, extent=30.0, size=0.02, z_range=(-80.0, -60.0))
    elif kind == "huge":
        sc = scenes.random_soup(40, seed, extent=50.0, size=400.0, z_range=(-300.0, 300.0))
    elif kind == "mixed_scales":
        a = scenes.random_soup(150, seed, extent=2.0, size=0.001, z_range=(-3.0, -1.0))
        b = scenes.random_soup(150, seed + 1, extent=200.0, size=150.0, z_range=(-500.0, 100.0))
        pts = np.concatenate([a["points"], b["points"]])
        idx = np.concatenate([a["idx"], b["idx"] + len(a["points"])])
        perm = rng.permutation(len(idx))
        sc = dict(points=pts, idx=idx[perm], normals=None, index_base=0, camera=a["camera"])
    elif kind == "random_normals":
        sc = scenes.random_soup(200, seed)
        sc["normals"] = rng.normal(size=(200, 3)).astype(np.float32)      # stored normals unrelated to geometry
    elif kind == "offset_camera":
        sc = scenes.random_soup(250, seed, extent=6.0, size=2.0, z_range=(-15.0, 5.0))
        sc["points"] = (sc["points"] + np.array([1000.0, -2000.0, 500.0], np.float32)).astype(np.float32)
        sc["camera"] = dict(lookfrom=(1000.0, -2000.0, 500.0), lookat=(1000.3, -2000.2, 499.0), vfov=75.0,
                            aperture=0.0, focus=10.0)
    else:
        raise ValueError(kind)
    return sc


@pytest.mark.parametrize("kind", ["around_eye", "tiny_far", "huge", "mixed_scales", "random_normals", "offset_camera"])
@pytest.mark.parametrize("seed", [11, 12])
def test_tiled_equals_brute_force_and_oracle_on_adversarial_soups(gpu, oracle, crt_built, kind, seed):
    sc = _soup_scene(kind, seed)
    nx, ny, spp = 72, 56, 3
    scene, cam = make(sc, nx, ny)
    outs = {}
    for name, v in VARIANTS.items():
        prev = crt_built.crt_set_kernel_variant(v)
        try:
            (outs[name],), _ = hip_render_with_states(gpu, scene, cam, nx, ny, spp)
        finally:
            crt_built.crt_set_kernel_variant(prev)
    w, ocam = oracle_pair(oracle, sc, nx, ny)
    ref, _ = oracle.render(w, ocam, nx, ny, spp)
    for name in outs:
        assert np.array_equal(bits(outs[name]), bits(ref)), (kind, seed, name)


@pytest.mark.parametrize("kind", ["around_eye", "tiny_far", "huge", "mixed_scales", "random_normals", "offset_camera"])
def test_binning_is_conservative_for_every_passing_pair(gpu, kind):
    """Stronger than pixel equality: EVERY (ray, triangle) pair that passes the reference's
    triangle test — closest hit or occluded — must have its triangle in the tile's list, and a
    cull-sure entry must never see dot(d, n) < 0.  Brute-force audit on the GPU."""
    for seed in (21, 22, 23):
        sc = _soup_scene(kind, seed)
        nx, ny, spp = 88, 72, 8
        scene, cam = make(sc, nx, ny)
        n_pass, n_missing, n_cull, offender = api.audit_tiles(scene, cam, nx, ny, spp, seed)
        assert n_pass > 0 or kind == "tiny_far", (kind, seed)   # sub-pixel triangles may catch no ray at all
        assert n_missing == 0 and n_cull == 0, (kind, seed, n_pass, n_missing, n_cull, offender >> 32, offender & 0xffffffff)


@pytest.mark.parametrize("cam", [
    dict(lookfrom=(3.0e4, -2.0e4, 5.0e4), lookat=(3.0e4 + 0.2, -2.0e4 + 0.1, 5.0e4 - 1.0), vfov=40.0, focus=0.05),   # |llc| / |d0| ~ 1e6
    dict(lookfrom=(0.0, 0.0, 4.0), lookat=(0.3, -0.2, 0.0), vfov=150.0, focus=10.0),                                 # very wide
    dict(lookfrom=(7.0, 9.0, 4.0), lookat=(6.0, 2.0, 1.0), vfov=3.0, focus=1000.0),                                  # very narrow, long focus
    dict(lookfrom=(0.0, 50.0, 0.001), lookat=(0.0, 0.0, 0.0), vfov=60.0, focus=2.0),                                 # looking almost along vup
])
def test_position_slack_on_extreme_cameras(gpu, crt_built, cam):
    """The binning prices the float evaluation of the ray direction as a shift of the screen
    position (camera_slacks): cameras with heavy cancellation, extreme fields of view and a view
    direction next to vup must still be conservative (audit) and equal to the brute force."""
    rng = np.random.RandomState(21)
    lf, la = np.array(cam["lookfrom"], np.float64), np.array(cam["lookat"], np.float64)
    w = (la - lf) / np.linalg.norm(la - lf)
    n = 300
    depth = rng.uniform(2.0, 40.0, size=(n, 1))
    centre = lf + w * depth + rng.normal(size=(n, 3)) * depth * np.tan(np.radians(min(cam["vfov"], 120.0)) / 2) * 0.8
    size = depth * rng.uniform(0.002, 0.3, size=(n, 1))
    pts = (centre[:, None, :] + rng.normal(size=(n, 3, 3)) * size[:, None, :]).reshape(-1, 3).astype(np.float32)
    idx = np.arange(3 * n, dtype=np.float32).reshape(n, 3)
    sc = dict(points=pts, idx=idx, normals=None, index_base=0,
              camera=dict(lookfrom=cam["lookfrom"], lookat=cam["lookat"], vfov=cam["vfov"], aperture=0.0, focus=cam["focus"]))
    nx, ny, spp = 160, 96, 4
    scene, camera = make(sc, nx, ny)
    outs = []
    for v in (1, 2):
        prev = crt_built.crt_set_kernel_variant(v)
        try:
            (o,), st = hip_render_with_states(gpu, scene, camera, nx, ny, spp)
        finally:
            crt_built.crt_set_kernel_variant(prev)
        outs.append((o, st))
    assert np.array_equal(bits(outs[0][0]), bits(outs[1][0]))
    assert np.array_equal(outs[0][1], outs[1][1])
    try:
        n_pass, n_missing, n_cull, _ = api.audit_tiles(scene, camera, nx, ny, 4)
    except api.CrtError:
        return                      # ill-conditioned camera: the library renders it with the brute-force kernel
    assert n_missing == 0 and n_cull == 0, (n_pass, n_missing, n_cull)


def test_binning_is_conservative_on_meshes(gpu):
    for sc, nx, ny, spp in [(scenes.cornell32(), 200, 150, 8), (scenes.displaced_sphere(60, 80), 320, 180, 4),
                            (scenes.quad(), 64, 64, 8)]:
        scene, cam = make(sc, nx, ny)
        n_pass, n_missing, n_cull, _ = api.audit_tiles(scene, cam, nx, ny, spp)
        assert n_pass > 0 and n_missing == 0 and n_cull == 0, (nx, ny, n_pass, n_missing, n_cull)


def test_tiled_equals_brute_force_with_shuffled_triangle_order(gpu, crt_built):
    """The binning walks 64 consecutive triangles per wavefront: a shuffled mesh makes every
    wavefront's candidates incoherent (all regions entered, lane = tile classification, unordered
    lists with ties between coincident edges) — same pixels and RNG states as the brute force."""
    sc = scenes.displaced_sphere(90, 120)
    perm = np.random.RandomState(5).permutation(len(sc["idx"]))
    sc["idx"] = sc["idx"][perm]
    if sc.get("normals") is not None and len(sc["normals"]) == len(perm):
        sc["normals"] = sc["normals"][perm]
    nx, ny, spp = 640, 360, 3
    scene, cam = make(sc, nx, ny)
    outs = []
    for v in (1, 2):
        prev = crt_built.crt_set_kernel_variant(v)
        try:
            (o,), st = hip_render_with_states(gpu, scene, cam, nx, ny, spp)
        finally:
            crt_built.crt_set_kernel_variant(prev)
        outs.append((o, st))
    assert np.array_equal(bits(outs[0][0]), bits(outs[1][0]))
    assert np.array_equal(outs[0][1], outs[1][1])
    n_pass, n_missing, n_cull, _ = api.audit_tiles(scene, cam, nx, ny, 2)
    assert n_pass > 0 and n_missing == 0 and n_cull == 0


@pytest.mark.parametrize("n_lat,n_lon,nx,ny,fits", [(8, 3000, 320, 180, True), (6, 12000, 256, 144, False)])
def test_pole_tiles_longer_than_the_staging_areas(gpu, crt_built, n_lat, n_lon, nx, ny, fits):
    """A lat-long sphere with many meridians: the tiles at its poles hold thousands of entries, so the slices of the
    waves that share them are longer than their 64-entry staging areas and go as far pairs (build_far_pairs: every
    entry classified per pixel, records from global memory) — or, when the pairs do not fit a wave's area, staged +
    wave-uniform walk as before.  Same pixels and RNG states as the brute force, also with the tiles dealt as row slabs."""
    sc = scenes.displaced_sphere(n_lat, n_lon, amp=0.02)
    spp = 6
    scene, cam = make(sc, nx, ny)
    outs = {}
    for v in (1, 2, 10):
        prev = crt_built.crt_set_kernel_variant(v)
        try:
            if v == 2:
                api.tiled_stats(1)
            (o,), st = hip_render_with_states(gpu, scene, cam, nx, ny, spp)
            if v == 2:
                stats = api.tiled_stats(0)
        finally:
            crt_built.crt_set_kernel_variant(prev)
        outs[v] = (o, st)
    for v in (2, 10):
        assert np.array_equal(bits(outs[v][0]), bits(outs[1][0])), v
        assert np.array_equal(outs[v][1], outs[1][1]), v
    far_slices, far_pairs, overflowed = stats[43], stats[44], stats[45]
    assert far_slices > 0 and far_pairs > 0 and (overflowed == 0) == fits, stats[43:46]
    n_pass, n_missing, n_cull, _ = api.audit_tiles(scene, cam, nx, ny, 2)
    assert n_pass > 0 and n_missing == 0 and n_cull == 0


@pytest.mark.gpu
def test_diagnostics_do_not_change_pixels(gpu, crt_built):
    """crt_debug_tiled_stats: neither the counters (mode 1) nor the wave timeline (mode 2) may touch
    pixels or RNG states; the timeline has one record per wavefront with start <= end of the
    shared-tile phase <= end."""
    sc = scenes.displaced_sphere(60, 80)
    nx, ny, spp = 320, 200, 3
    scene, cam = make(sc, nx, ny)
    (ref,), st_ref = hip_render_with_states(gpu, scene, cam, nx, ny, spp)
    for mode in (1, 2):
        api.tiled_stats(mode)
        try:
            (o,), st = hip_render_with_states(gpu, scene, cam, nx, ny, spp)
            tl = api.wave_timeline() if mode == 2 else None
        finally:
            api.tiled_stats(0)
        assert np.array_equal(bits(o), bits(ref)) and np.array_equal(st, st_ref), mode
        if tl is not None:
            tl = tl[tl[:, 2] > 0].astype(np.int64)
            assert len(tl) > 0 and len(tl) % 8 == 0
            assert np.all(tl[:, 0] <= tl[:, 1]) and np.all(tl[:, 1] <= tl[:, 2])


@pytest.mark.gpu
def test_kernel_timing_can_be_switched_off(gpu, crt_built):
    """crt_set_kernel_timing(0): no event markers between a frame's kernels, same pixels, and
    crt_last_kernel_times says so instead of returning stale numbers."""
    sc = scenes.displaced_sphere(40, 50)
    nx, ny, spp = 160, 96, 2
    scene, cam = make(sc, nx, ny)
    (ref,), st_ref = hip_render_with_states(gpu, scene, cam, nx, ny, spp)
    assert api.last_kernel_times()[0] > 0.0
    prev = api.set_kernel_timing(False)
    try:
        (o,), st = hip_render_with_states(gpu, scene, cam, nx, ny, spp)
        with pytest.raises(api.CrtError):
            api.last_kernel_times()
    finally:
        api.set_kernel_timing(prev)
    assert np.array_equal(bits(o), bits(ref)) and np.array_equal(st, st_ref)
    hip_render_with_states(gpu, scene, cam, nx, ny, spp)
    assert api.last_kernel_times()[0] > 0.0
