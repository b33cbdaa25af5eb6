#!/usr/bin/env python3
"""C5 geometry (3840x2160, 1 M triangles), tiled kernels only, a few frames at 4 spp (for rocprofv3 --kernel-trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cudaraytracer_amd import api, scenes
nx, ny = 3840, 2160
sc = scenes.dragon1m()
scene = api.Scene.from_mesh(sc["points"], sc["idx"], None, 0)
cam = api.init_camera(nx, ny)
st = api.SetCurandState(nx, ny)
d = torch.zeros(nx * ny * 3, device="cuda")
for _ in range(4):
    api.render_rows(scene, cam, st, nx, ny, 4, 0, ny, d); torch.cuda.synchronize()
print(api.last_kernel_times())
