#!/usr/bin/env python3
"""BASELINE configs[4] geometry on one GPU: 3840x2160, 1 M-triangle synthetic mesh.
Times the tiled kernels and checks a 64-row band against the brute-force kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cudaraytracer_amd import api, scenes
nx, ny = 3840, 2160
sc = scenes.dragon1m()
t0 = time.time()
scene = api.Scene.from_mesh(sc["points"], sc["idx"], None, 0)
cam = api.init_camera(nx, ny)
print("scene %d triangles (%.1fs)" % (scene.n_triangles, time.time() - t0), flush=True)
st = api.SetCurandState(nx, ny)
d = torch.zeros(nx * ny * 3, device="cuda")
for spp in (1, 4, 16):
    for _ in range(2):
        api.render_rows(scene, cam, st, nx, ny, spp, 0, ny, d); torch.cuda.synchronize()
    print("tiled full frame spp %d: total/bin/shade ms" % spp, api.last_kernel_times(), flush=True)
api.tiled_stats(True)
api.render_rows(scene, cam, st, nx, ny, 1, 0, ny, d); torch.cuda.synchronize()
o = api.tiled_stats(False)
print("regions %d sum1 %d max1 %d | tiles %d sum2 %d max2 %d pool overflows %d" % (o[0], o[1], o[2], o[4], o[5], o[6], o[7]), flush=True)
# correctness: rows [1040, 1104) tiled vs brute force, 1 spp
rb, rc = 1040, 64
outs = []
for v in (2, 1):
    api.set_kernel_variant(v)
    s2 = api.SetCurandState(nx, ny, 0, rb, rc)
    b = torch.zeros(rc * nx * 3, device="cuda")
    api.render_rows(scene, cam, s2, nx, ny, 1, rb, rc, b); torch.cuda.synchronize()
    print("variant %d band ms" % v, api.last_kernel_times()[0], flush=True)
    outs.append(b.cpu().numpy())
api.set_kernel_variant(0)
print("band tiled == brute force:", np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32)))
