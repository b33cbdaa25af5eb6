"""GPU parity at the BASELINE.json configurations themselves (run with -m gpu on an MI355X).

The frames come from cudaraytracer_amd.scenes.baseline_obj — the SAME definition bench.py
times (OBJ round trip, loader centring, reference camera):
  C3  1920x1080 x 64 spp, 100 000 triangles   (the bench frame)
  C3B the same configuration on the pole-free mesh (subdivided small_bunny + ground grid), bench.py's second workload
  C4  1920x1080 x 256 spp, 100 000 triangles, the 8 strip sets of an 8-GPU split
  C5  3840x2160 x 64 spp, 1 000 000 triangles
Bar: the tiled kernels equal the brute-force list walk bit for bit (pixels and RNG states), and
both equal the CPU oracle on crops placed on the silhouette and in the interior; the binning audit
(every passing (ray, triangle) pair is in its tile's list) is clean.  The oracle needs ~40 ns per
(ray, triangle) pair, which bounds the crops at these triangle counts.
"""
import numpy as np
import pytest

from cudaraytracer_amd import api, scenes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def obj_dir(tmp_path_factory):
    return str(tmp_path_factory.mktemp("baseline_obj"))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def load(name, obj_dir):
    path, nx, ny, spp = scenes.baseline_obj(name, obj_dir)
    scene = api.Scene.from_obj(path, scale=1.0)
    return path, scene, api.init_camera(nx, ny), nx, ny, spp


def oracle_world(oracle, path, nx, ny):
    pts, idx = oracle.parse_obj(path)
    w = oracle.World(pts, idx, None, 1)
    cam = oracle.make_camera((0, 100, 1000), (0, 150, 0), (0, 1, 0), 40.0, np.float32(nx) / np.float32(ny), 0.0, 10.0)
    return w, cam


def render(gpu, scene, cam, nx, ny, spp, row_begin=0, row_count=None, variant=0):
    from cudaraytracer_amd import _lib
    rows = ny - row_begin if row_count is None else row_count
    st = api.SetCurandState(nx, ny, 0, row_begin, rows)
    d = gpu.zeros(rows * nx * 3, dtype=gpu.float32, device="cuda")
    prev = _lib.lib().crt_set_kernel_variant(variant)
    try:
        api.render_rows(scene, cam, st, nx, ny, spp, row_begin, rows, d, stream=gpu.cuda.current_stream())
        gpu.cuda.synchronize()
    finally:
        _lib.lib().crt_set_kernel_variant(prev)
    return d.cpu().numpy().reshape(rows, nx, 3), st.export()


def hit_mask(rgb):
    """Lambert-shaded pixels are red (albedo .65,.05,.05), the sky is blue-white."""
    return (rgb[..., 0] - rgb[..., 2]) > 0.05


def silhouette_and_interior(rgb, w, h):
    """Two (r0, r1, c0, c1) windows of w x h pixels: one straddling the object's left silhouette
    on the row through its middle, one well inside."""
    m = hit_mask(rgb)
    rows = np.where(m.any(axis=1))[0]
    assert len(rows) > 8 * h, "the mesh must be visible"
    r_mid = int((rows[0] + rows[-1]) // 2)
    cols = np.where(m[r_mid])[0]
    c_edge, c_mid = int(cols[0]), int((cols[0] + cols[-1]) // 2)
    sil = (r_mid, r_mid + h, c_edge - w // 2, c_edge - w // 2 + w)
    assert m[sil[0]:sil[1], sil[2]:sil[3]].any() and not m[sil[0]:sil[1], sil[2]:sil[3]].all()
    return sil, (r_mid, r_mid + h, c_mid, c_mid + w)


def check_crop(oracle, w, ocam, full, nx, ny, spp, win, row0=0):
    ref, _ = oracle.render(w, ocam, nx, ny, spp, window=win)
    r0, r1, c0, c1 = win
    assert np.array_equal(bits(full[r0 - row0:r1 - row0, c0:c1]), bits(ref[r0:r1, c0:c1])), win


def test_c3_full_frame_64spp_tiled_equals_brute_force_and_oracle(gpu, oracle, obj_dir):
    """BASELINE configs[2] exactly as bench.py times it: 1920x1080, 64 spp, 100 000 triangles."""
    path, scene, cam, nx, ny, spp = load("C3", obj_dir)
    assert (nx, ny, spp, scene.n_triangles) == (1920, 1080, 64, 100000)
    tiled, st_t = render(gpu, scene, cam, nx, ny, spp, variant=2)
    brute, st_b = render(gpu, scene, cam, nx, ny, spp, variant=1)       # 1.3e13 pair tests, ~6.4 s
    assert np.array_equal(bits(tiled), bits(brute))
    assert np.array_equal(st_t, st_b)
    assert np.isfinite(tiled).all() and tiled.min() >= 0.0 and tiled.max() <= 1.0
    cover = hit_mask(tiled).mean()
    assert 0.05 < cover < 0.5, cover
    w, ocam = oracle_world(oracle, path, nx, ny)
    for win in silhouette_and_interior(tiled, 8, 4):                     # 2 x 2048 rays x 100 k triangles
        check_crop(oracle, w, ocam, tiled, nx, ny, spp, win)
    n_pass, n_missing, n_cull, _ = api.audit_tiles(scene, cam, nx, ny, 4)
    assert n_pass > 0 and n_missing == 0 and n_cull == 0


def test_c3b_bunny_on_ground_64spp_tiled_equals_brute_force_and_oracle(gpu, oracle, obj_dir):
    """BASELINE configs[2] on SURVEY §8d's OTHER 100 000-triangle mesh — small_bunny midpoint-subdivided twice on a
    ground grid, no pole fan (scenes.bunny_ground_100k; bench.py reports it as its second workload): the same bar as
    C3.  Crops: the bunny's silhouette against the sky, its interior, the ground's horizon, the ground in front."""
    path, scene, cam, nx, ny, spp = load("C3B", obj_dir)
    assert (nx, ny, spp, scene.n_triangles) == (1920, 1080, 64, 100000)
    tiled, st_t = render(gpu, scene, cam, nx, ny, spp, variant=2)
    brute, st_b = render(gpu, scene, cam, nx, ny, spp, variant=1)       # 1.3e13 pair tests
    assert np.array_equal(bits(tiled), bits(brute))
    assert np.array_equal(st_t, st_b)
    assert np.isfinite(tiled).all() and tiled.min() >= 0.0 and tiled.max() <= 1.0
    m = hit_mask(tiled)
    assert 0.1 < m.mean() < 0.8, m.mean()
    r_top = int(np.where(m.any(axis=1))[0][-1])                          # row 0 = image bottom: the tip of an ear
    r_sil, r_in = r_top - 60, r_top - 200
    c_sil = int(np.where(m[r_sil])[0][0])
    cols_in = np.where(m[r_in])[0]
    col_g = 200                                                          # a column left of the bunny: ground below, sky above
    r_hor = int(np.where(m[:, col_g])[0][-1])
    wins = [(r_sil, r_sil + 4, c_sil - 4, c_sil + 4), (r_in, r_in + 4, int(cols_in[len(cols_in) // 2]), int(cols_in[len(cols_in) // 2]) + 8),
            (r_hor - 1, r_hor + 3, col_g, col_g + 8), (40, 44, 960, 968)]
    for k, win in enumerate(wins):
        sub = m[win[0]:win[1], win[2]:win[3]]
        assert sub.any() and (k not in (0, 2) or not sub.all()), (k, win)   # silhouette and horizon windows hold both
    w, ocam = oracle_world(oracle, path, nx, ny)
    for win in wins:                                                     # 4 x 2048 rays x 100 k triangles
        check_crop(oracle, w, ocam, tiled, nx, ny, spp, win)
    n_pass, n_missing, n_cull, _ = api.audit_tiles(scene, cam, nx, ny, 4)
    assert n_pass > 0 and n_missing == 0 and n_cull == 0


def test_c4_256spp_eight_strip_sets_reassemble_to_the_full_frame(gpu, oracle, obj_dir):
    """BASELINE configs[3]: 1920x1080 x 256 spp split 8 ways (32-row strips dealt round-robin,
    crt_render_strips).  The 8 ranks' pieces, rendered one after the other on this GPU and put
    in place like the gather does, equal the single-launch frame (pixels and RNG states) and the
    oracle on a silhouette and an interior crop."""
    from cudaraytracer_amd.dist import partition_strips
    path, scene, cam, nx, ny, spp = load("C4", obj_dir)
    assert (nx, ny, spp) == (1920, 1080, 256)
    full, full_states = render(gpu, scene, cam, nx, ny, spp)
    full_states = full_states.reshape(ny, nx, 6)
    world = 8
    frame = np.zeros_like(full)
    for r in range(world):
        strips = partition_strips(ny, world, r)
        rows = sum(c for _, c in strips)
        st = api.CurandState.for_strips(nx, ny, r, 32, world, 0)
        d = gpu.zeros(rows * nx * 3, dtype=gpu.float32, device="cuda")
        api.render_strips(scene, cam, st, nx, ny, spp, r, 32, world, d, stream=gpu.cuda.current_stream())
        gpu.cuda.synchronize()
        piece = d.cpu().numpy().reshape(rows, nx, 3)
        got_st = st.export().reshape(rows, nx, 6)
        at = 0
        for y0, c in strips:
            frame[y0:y0 + c] = piece[at:at + c]
            ref_st, g = full_states[y0:y0 + c].reshape(-1, 6), got_st[at:at + c].reshape(-1, 6)
            skip = 1 if y0 == 0 else 0                                   # stream 0 (pixel 0): DESIGN.md §1
            assert np.array_equal(g[skip:], ref_st[skip:]), (r, y0)
            at += c
    assert np.array_equal(bits(frame), bits(full))
    w, ocam = oracle_world(oracle, path, nx, ny)
    for win in silhouette_and_interior(full, 4, 2):                      # 2 x 2048 rays x 100 k triangles
        check_crop(oracle, w, ocam, frame, nx, ny, spp, win)


def test_c5_full_frame_64spp_audit_bands_and_crops(gpu, oracle, obj_dir):
    """BASELINE configs[4]: 3840x2160 x 64 spp, 1 000 000 triangles.  The whole tiled frame; the
    binning audit over the whole frame; two 16-row bands (through the silhouette top and the
    middle) against the brute-force walk at the full 64 spp; two oracle crops at 64 spp."""
    path, scene, cam, nx, ny, spp = load("C5", obj_dir)
    assert (nx, ny, spp, scene.n_triangles) == (3840, 2160, 64, 1000000)
    full, full_states = render(gpu, scene, cam, nx, ny, spp, variant=2)
    assert np.isfinite(full).all() and full.min() >= 0.0 and full.max() <= 1.0
    full_states = full_states.reshape(ny, nx, 6)
    m = hit_mask(full)
    rows = np.where(m.any(axis=1))[0]
    top, mid = int(rows[-1]), int((rows[0] + rows[-1]) // 2)
    for rb in (top - 8, mid):                                            # 2 x 3.9e12 pair tests
        rb -= rb % 8
        band, st = render(gpu, scene, cam, nx, ny, spp, row_begin=rb, row_count=16, variant=1)
        assert np.array_equal(bits(band), bits(full[rb:rb + 16])), rb
        assert np.array_equal(st, full_states[rb:rb + 16].reshape(-1, 6)), rb
    n_pass, n_missing, n_cull, _ = api.audit_tiles(scene, cam, nx, ny, 4)
    assert n_pass > 0 and n_missing == 0 and n_cull == 0
    w, ocam = oracle_world(oracle, path, nx, ny)
    for win in silhouette_and_interior(full, 2, 2):                      # 2 x 256 rays x 1 M triangles
        check_crop(oracle, w, ocam, full, nx, ny, spp, win)


def test_c2_cornell_800x600x16_full_frame_against_the_oracle(gpu, c2_oracle_frame):
    """BASELINE configs[1] at its own size: 800x600, 16 spp, the 32-triangle Cornell box.  The whole frame — pixels and
    RNG states — against the CPU oracle (7.7 M rays x 32 triangles), tiled and brute force."""
    sc, nx, ny, spp, ref, want_states = c2_oracle_frame
    scene = api.Scene.from_mesh(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
    c = sc["camera"]
    cam = api.create_camera(nx, ny, c["lookfrom"], c["lookat"], c["focus"], c["aperture"], c["vfov"])
    for variant in (2, 1):
        rgb, st = render(gpu, scene, cam, nx, ny, spp, variant=variant)
        assert np.array_equal(bits(rgb), bits(ref)), (variant, int((bits(rgb) != bits(ref)).any(axis=2).sum()))
        assert np.array_equal(st[1:], want_states[1:]), variant   # stream 0: every pixel's lens draws in the serial oracle
