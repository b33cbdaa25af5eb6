#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the CPU oracle (oracle/crt_oracle.c).

PARITY UNPINNED: the reference ships no golden vectors and cannot be built in this image
(CUDA + cuRAND + FBX SDK), so these files pin the ORACLE (regression) and carry its outputs
to the GPU box; they are not outputs of the reference program.  Re-run after any
intentional oracle change:   python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cudaraytracer_amd import scenes  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

# (name, scene factory, nx, ny, spp)
RENDERS = [
    ("c1_quad_256x256x1", scenes.quad, 256, 256, 1),                       # BASELINE configs[0], full
    ("c2_cornell_100x75x16", scenes.cornell32, 100, 75, 16),               # configs[1] at 1/8 size
    ("sphere1000_96x54x4", lambda: scenes.displaced_sphere(20, 25), 96, 54, 4),
    ("soup96_48x48x4", lambda: scenes.random_soup(96, 7), 48, 48, 4),
]

KAT_SUBSEQ = [0, 1, 2, 3, 4, 5, 63, 64, 65535, 2073599, 8294399, (1 << 31) - 1]


def camera_of(sc, nx, ny):
    c = sc["camera"]
    return po.make_camera(c["lookfrom"], c["lookat"], (0, 1, 0), c["vfov"],
                          np.float32(nx) / np.float32(ny), c["aperture"], c["focus"])


def main():
    os.makedirs(OUT, exist_ok=True)
    # 1. XORWOW known answers (seed 0 and one 64-bit seed, offset 0 and 5)
    kat = {}
    for seed in (0, 0x123456789ABCDEF):
        for sub in KAT_SUBSEQ:
            for off in (0, 5):
                u32, f32 = po.curand_stream(seed, sub, off, 8)
                kat["u32_%d_%d_%d" % (seed, sub, off)] = u32
                kat["f32_%d_%d_%d" % (seed, sub, off)] = f32
    st = po.curand_states(0, 300, 0)
    kat["states_0_300"] = np.concatenate([st["d"][:, None], st["v"]], axis=1)
    np.savez_compressed(os.path.join(OUT, "xorwow_kat.npz"), **kat)

    # 2. framebuffers + per-ray closest hits
    for name, factory, nx, ny, spp in RENDERS:
        sc = factory()
        w = po.World(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
        cam = camera_of(sc, nx, ny)
        buf, states = po.render(w, cam, nx, ny, spp)
        # closest hit of the pixel-centre rays of a coarse grid (pins tie-breaking, negative t)
        org = np.array(list(cam.origin), dtype=np.float32)
        llc = np.array(list(cam.lower_left_corner), dtype=np.float32)
        hor = np.array(list(cam.horizontal), dtype=np.float32)
        ver = np.array(list(cam.vertical), dtype=np.float32)
        rays, hits = [], []
        for gy in range(0, ny, max(1, ny // 12)):
            for gx in range(0, nx, max(1, nx // 12)):
                u = np.float32(gx + 0.5) / np.float32(nx)
                v = np.float32(gy + 0.5) / np.float32(ny)
                d = ((llc + u * hor) + v * ver) - org
                flag, t, idx, nrm = w.hit(org, d)
                rays.append(d)
                hits.append((float(flag), float(t) if flag else 0.0, float(idx)))
        after = np.concatenate([states["d"][:, None], states["v"]], axis=1)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), rgb=buf, camera=po.camera_as_array(cam),
                            rays=np.array(rays, dtype=np.float32), hits=np.array(hits, dtype=np.float64),
                            rgba=po.to_rgba8(buf), states_after=after[:2048],
                            states_after_colsum=after.astype(np.uint64).sum(axis=0),
                            nx=nx, ny=ny, spp=spp)
        print(name, buf.shape, "hit-rays %d/%d" % (int(sum(h[0] for h in hits)), len(hits)))


BVH_RENDERS = [
    ("bvh_sphere1000_96x54x4", lambda: scenes.displaced_sphere(20, 25), 96, 54, 4),
    ("bvh_soup96_48x48x4", lambda: scenes.random_soup(96, 7), 48, 48, 4),
    ("bvh_cornell_100x75x8", scenes.cornell32, 100, 75, 8),
]


def bvh_goldens():
    """The BVH world (hitable/bvh.h): tree (boxes, structure, leaf order) + frame, built right after
    SetCurandState like the reference's main() (kernel.cu:69-71,97): the build advances stream 0."""
    for name, factory, nx, ny, spp in BVH_RENDERS:
        sc = factory()
        w = po.World(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
        cam = camera_of(sc, nx, ny)
        states = po.fresh_states(nx, ny)
        bvh = po.Bvh(w, states)
        nodes = bvh.nodes()
        buf, states = po.render_bvh(bvh, cam, nx, ny, spp, states)
        after = np.concatenate([states["d"][:, None], states["v"]], axis=1)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), rgb=buf, camera=po.camera_as_array(cam),
                            boxes=nodes["box"], skip=nodes["skip"], first=nodes["first"], count=nodes["count"],
                            is_leaf=nodes["isLeaf"], depth=nodes["depth"], axis=nodes["axis"], order=bvh.order(),
                            states_after=after[:2048], nx=nx, ny=ny, spp=spp)
        print(name, buf.shape, "nodes %d depth %d" % (len(nodes), nodes["depth"].max()))


def small_bunny():
    """SURVEY §8c golden (4): the only real mesh the reference ships for this path,
    CudaTest/objects/small_bunny.obj (2503 v / 4968 f, plain `f a b c`), read by the oracle's
    restatement of parseObjByName (mean-centred), scaled x100 like the historic BuildObjMesh
    (Loader/obj_loader.h:164), 64x64x4 spp.  The fixture holds the mesh arrays (data, so the
    GPU box needs no /root/reference) and the oracle's frame.  The historic camera
    (create_camera_origin) no longer exists in the reference; this one looks at the bunny from
    (0, 5, 35)."""
    ref = "/root/reference/CudaTest/objects/small_bunny.obj"
    if not os.path.exists(ref):
        print("small_bunny: reference mesh not available here, fixture left as is")
        return
    pts, idx = po.parse_obj(ref)
    assert pts.shape == (2503, 3) and idx.shape == (4968, 3)
    pts = (pts * np.float32(100.0)).astype(np.float32)
    nx = ny = 64
    spp = 4
    w = po.World(pts, idx, None, 1)
    look = dict(lookfrom=(0.0, 5.0, 35.0), lookat=(0.0, 0.0, 0.0), vfov=40.0, aperture=0.0, focus=10.0)
    cam = po.make_camera(look["lookfrom"], look["lookat"], (0, 1, 0), look["vfov"], np.float32(nx) / np.float32(ny), 0.0, 10.0)
    buf, states = po.render(w, cam, nx, ny, spp)
    after = np.concatenate([states["d"][:, None], states["v"]], axis=1)
    np.savez_compressed(os.path.join(OUT, "small_bunny_64x64x4.npz"), rgb=buf, points=pts, idx=idx,
                        camera=po.camera_as_array(cam), lookfrom=np.array(look["lookfrom"], dtype=np.float32),
                        lookat=np.array(look["lookat"], dtype=np.float32), states_after=after,
                        rgba=po.to_rgba8(buf), nx=nx, ny=ny, spp=spp)
    hit = (buf[..., 0] - buf[..., 2]) > 0.05
    print("small_bunny", buf.shape, "hit pixels %d / %d" % (int(hit.sum()), nx * ny))


SHADE_MATS = [(0, (0.73, 0.73, 0.73), 0.0), (0, (0.65, 0.05, 0.05), 0.0), (1, (0.8, 0.85, 0.88), 0.1), (2, (1.0, 1.0, 1.0), 1.5),
              (3, (4.0, 4.0, 4.0), 0.0), (1, (0.7, 0.6, 0.5), 0.6)]   # (class: 0 Lambertian 1 Metal 2 Dielectric 3 DiffuseLight, albedo, fuzz / ref_idx)


def shade_goldens():
    """Row f2: the recursive shade (core/render.h:48-67) with the four material classes (material/material.h:57-161) over a
    BVH world: two glass / metal / light spheres on a Lambertian ground (many bounces, total internal reflection) and the
    96-triangle soup (triangles behind the eye, coincident ones); depth 8; pixels and RNG states of the oracle."""
    from tests.test_shade import material_scene
    for name, kind, nx, ny, spp, depth in (("shade_spheres_64x36x3_d8", "spheres", 64, 36, 3, 8), ("shade_soup60_40x30x2_d8", "soup", 40, 30, 2, 8)):
        sc, of = material_scene(kind)
        of = (of % len(SHADE_MATS)).astype(np.int32)
        w = po.World(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
        w.set_materials(SHADE_MATS, of)
        cam = camera_of(sc, nx, ny)
        states = po.fresh_states(nx, ny)
        bvh = po.Bvh(w, states)
        buf, states = po.render_ex(w, bvh, po.SHADER_SHADE, cam, nx, ny, spp, states, max_depth=depth)
        after = np.concatenate([states["d"][:, None], states["v"]], axis=1)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), rgb=buf, states_after=after, material_of=of,
                            materials=np.array([(k, a[0], a[1], a[2], f) for k, a, f in SHADE_MATS], dtype=np.float64),
                            nx=nx, ny=ny, spp=spp, depth=depth, kind=kind)
        print(name, buf.shape, "NaN %d, mean %.3f" % (int(np.isnan(buf).sum()), float(np.nanmean(buf))))


if __name__ == "__main__":
    shade_goldens()
    small_bunny()
    bvh_goldens()
    main()
