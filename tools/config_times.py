#!/usr/bin/env python3
"""Kernel times of the BASELINE configurations that are not the bench line (one GPU, HIP events): C1/C2 as the
parity tests render them, C4 and C5 as scenes.BASELINE_FRAMES defines them (the frames tests/test_gpu_configs.py
checks), list world and BVHNode world."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cudaraytracer_amd import api, scenes


def timed(scene, cam, nx, ny, spp, bvh=False, reps=4):
    st = api.SetCurandState(nx, ny)
    if bvh:
        api.create_BVHfromList(scene, st)
    d = torch.zeros(nx * ny * 3, device="cuda")
    ts = []
    for _ in range(reps):
        api.render_rows(scene, cam, st, nx, ny, spp, 0, ny, d, bvh=bvh)
        torch.cuda.synchronize()
        ts.append(api.last_kernel_times())
    return min(ts[1:])


for name, sc, nx, ny, spp in [("C1 quad 200x100x4", scenes.quad(), 200, 100, 4), ("C2 cornell32 800x600x16", scenes.cornell32(), 800, 600, 16)]:
    scene = api.Scene.from_mesh(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
    c = sc["camera"]
    cam = api.create_camera(nx, ny, c["lookfrom"], c["lookat"], c["focus"], c["aperture"], c["vfov"])
    t = timed(scene, cam, nx, ny, spp)
    print("%-34s list world: total/bin/shade ms %.3f %.3f %.3f  -> %.0f Mrays/s" % (name, t[0], t[1], t[2], nx * ny * spp / t[0] / 1e3), flush=True)
tmp = tempfile.mkdtemp()
for frame in ("C4", "C5"):
    path, nx, ny, spp = scenes.baseline_obj(frame, tmp)
    scene = api.Scene.from_obj(path, scale=1.0)
    cam = api.init_camera(nx, ny)
    for bvh in (False, True):
        t = timed(scene, cam, nx, ny, spp, bvh=bvh, reps=3)
        print("%-34s %s world: total/bin/shade ms %.3f %.3f %.3f  -> %.0f Mrays/s"
              % ("%s %d triangles %dx%dx%d" % (frame, scene.n_triangles, nx, ny, spp), "BVH " if bvh else "list", t[0], t[1], t[2], nx * ny * spp / t[0] / 1e3), flush=True)
