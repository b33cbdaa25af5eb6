#!/usr/bin/env python3
"""A/B of the per-ray tree walk: pair records (default) against skip links (variant 9), on the C3 mesh.
   python tools/r3_walk_ab.py"""
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
    for variant, name in ((1 if not materials else 0, "pair records"), (9, "skip links")):
        api.set_kernel_variant(variant)
        if materials:
            for shader, sn in ((api.SHADER_SHADE, "shade depth 8"), (api.SHADER_LAMBERT, "LambertShade, 4 materials"), (api.SHADER_NORMAL, "shade_normal")):
                ms = timed(lambda: api.render_rows_ex(scene, cam, st, nx, ny, 4, 0, ny, d, world=api.WORLD_BVH, shader=shader, max_depth=8, stream=s))
                print("generic kernel, %s, 4 spp, %s: %.2f ms" % (sn, name, ms), flush=True)
        else:
            ms = timed(lambda: api.render_rows(scene, cam, st, nx, ny, 4, 0, ny, d, stream=s, bvh=True))
            print("render_bvh_kernel (per-ray walk, one Lambertian), 4 spp, %s: %.2f ms" % (name, ms), flush=True)
        api.set_kernel_variant(0)
