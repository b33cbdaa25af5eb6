#!/usr/bin/env python3
"""Frame time of the BVH world (crt_render_rows_bvh) on the BASELINE frames, next to the list world's
tiled kernels: python tools/bvh_time.py [C3|C5] [spp]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from cudaraytracer_amd import api, scenes  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
path, nx, ny, spp = scenes.baseline_obj(name, tempfile.mkdtemp())
if len(sys.argv) > 2:
    spp = int(sys.argv[2])
scene = api.Scene.from_obj(path, scale=1.0)
cam = api.init_camera(nx, ny)
st = api.SetCurandState(nx, ny, 0)
t0 = time.perf_counter()
api.create_BVHfromList(scene, st)
build_s = time.perf_counter() - t0
d = torch.zeros(ny * nx * 3, dtype=torch.float32, device="cuda")
s = torch.cuda.current_stream()
for bvh in (True, False):
    for _ in range(2):
        api.render_rows(scene, cam, st, nx, ny, spp, 0, ny, d, stream=s, bvh=bvh)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        api.render_rows(scene, cam, st, nx, ny, spp, 0, ny, d, stream=s, bvh=bvh)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print("%s %dx%dx%d %s world: %.3f ms/frame, %.0f Mrays/s" % (name, nx, ny, spp, "BVH " if bvh else "list", ms, nx * ny * spp / ms / 1e3))
api.tiled_stats(True)
api.render_rows(scene, cam, st, nx, ny, 1, 0, ny, d, stream=s, bvh=True)
torch.cuda.synchronize()
o = api.tiled_stats(False)
print("%s tile lists (slot order): sum %d, max %d, histogram (bucket b: 2^(b-1) <= n < 2^b) %s" % (name, o[5], o[6], o[8:20]))
t0 = time.perf_counter()
for _ in range(10):
    api.Update_BVH(scene, s)
torch.cuda.synchronize()
print("%s: %d triangles, %d nodes, host build %.3f s, refit %.3f ms" % (name, scene.n_triangles, len(scene.bvh_export()[0]), build_s,
                                                                          (time.perf_counter() - t0) / 10 * 1e3))
