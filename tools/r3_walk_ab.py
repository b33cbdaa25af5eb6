#!/usr/bin/env python3
"""A/B of the tree walks on the C3 mesh: the per-ray walk of LambertShade on one Lambertian (variant 1; render_bvh_kernel's general loop: variant 5) and the generic
kernel's walks for `shade` — shared by the wavefront (default), one stack per lane (variant 11), skip links (9) — and for
LambertShade / shade_normal through the generic kernel (variant 1: the shared walk).   python tools/r3_walk_ab.py"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cudaraytracer_amd import api, scenes

path, nx, ny, _ = scenes.baseline_obj("C3", tempfile.mkdtemp())
d = torch.zeros(ny * nx * 3, dtype=torch.float32, device="cuda")
s = torch.cuda.current_stream()


def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for materials in (False, True):
    scene = api.Scene.from_obj(path, scale=1.0)
    cam = api.init_camera(nx, ny)
    st = api.SetCurandState(nx, ny, 0)
    api.create_BVHfromList(scene, st)
    if materials:
        mats = [(api.LAMBERTIAN, (0.65, 0.05, 0.05), 0.0), (api.METAL, (0.8, 0.85, 0.88), 0.1), (api.DIELECTRIC, (1, 1, 1), 1.5),
                (api.DIFFUSE_LIGHT, (4.0, 4.0, 4.0), 0.0)]
        scene.set_materials(mats, (np.arange(scene.n_triangles) // 2500 % 4).astype(np.int32))
    if materials:
        for variant, name in ((0, "pair records, walk shared by the wavefront"), (11, "pair records, one stack per lane"), (9, "skip links")):
            api.set_kernel_variant(variant)
            ms = timed(lambda: api.render_rows_ex(scene, cam, st, nx, ny, 4, 0, ny, d, world=api.WORLD_BVH, shader=api.SHADER_SHADE, max_depth=8, stream=s))
            print("generic kernel, shade depth 8, 4 materials, 4 spp, %s: %.2f ms" % (name, ms), flush=True)
        api.set_kernel_variant(1)
        for shader, sn in ((api.SHADER_LAMBERT, "LambertShade, 4 materials"), (api.SHADER_NORMAL, "shade_normal")):
            ms = timed(lambda: api.render_rows_ex(scene, cam, st, nx, ny, 4, 0, ny, d, world=api.WORLD_BVH, shader=shader, max_depth=8, stream=s))
            print("generic kernel (variant 1), %s, 4 spp, pair records, walk shared by the wavefront: %.2f ms" % (sn, ms), flush=True)
        api.set_kernel_variant(0)
        for shader, sn in ((api.SHADER_LAMBERT, "LambertShade, 4 materials"), (api.SHADER_NORMAL, "shade_normal")):
            ms = timed(lambda: api.render_rows_ex(scene, cam, st, nx, ny, 4, 0, ny, d, world=api.WORLD_BVH, shader=shader, max_depth=8, stream=s))
            print("tiled kernels, material mode (default), %s, 4 spp: %.2f ms" % (sn, ms), flush=True)
    else:
        api.set_kernel_variant(1)
        ms = timed(lambda: api.render_rows(scene, cam, st, nx, ny, 4, 0, ny, d, stream=s, bvh=True))
        print("per-ray walk (variant 1: the generic kernel, walk shared by the wavefront), one Lambertian, 4 spp: %.2f ms" % ms, flush=True)
        api.set_kernel_variant(5)
        ms = timed(lambda: api.render_rows(scene, cam, st, nx, ny, 4, 0, ny, d, stream=s, bvh=True))
        print("render_bvh_kernel, its general loop for every wavefront (variant 5: IEEE divisions over the skip links), one Lambertian, 4 spp: %.2f ms" % ms, flush=True)
        api.set_kernel_variant(0)
        ms = timed(lambda: api.render_rows(scene, cam, st, nx, ny, 4, 0, ny, d, stream=s, bvh=True))
        print("tiled kernels, BVH mode (default), one Lambertian, 4 spp: %.2f ms" % ms, flush=True)
