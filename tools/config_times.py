#!/usr/bin/env python3
"""Kernel times of the BASELINE configs that are not the bench line (one GPU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cudaraytracer_amd import api, scenes
for name, sc, nx, ny, spp in [("C1 quad 200x100x4", scenes.quad(), 200, 100, 4),
                              ("C2 cornell32 800x600x16", scenes.cornell32(), 800, 600, 16),
                              ("C4 bunny100k 1920x1080x256", scenes.bunny100k(), 1920, 1080, 256)]:
    scene = api.Scene.from_mesh(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
    c = sc["camera"]
    cam = api.create_camera(nx, ny, c["lookfrom"], c["lookat"], c["focus"], c["aperture"], c["vfov"])
    st = api.SetCurandState(nx, ny)
    d = torch.zeros(nx * ny * 3, device="cuda")
    ts = []
    for _ in range(4):
        api.render_rows(scene, cam, st, nx, ny, spp, 0, ny, d); torch.cuda.synchronize()
        ts.append(api.last_kernel_times())
    t = min(ts[1:])
    print("%-28s total/bin/shade ms %.3f %.3f %.3f  -> %.0f Mrays/s" % (name, t[0], t[1], t[2], nx * ny * spp / t[0] / 1e3), flush=True)
