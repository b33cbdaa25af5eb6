#!/usr/bin/env python3
"""Frame time of the generic kernel (crt_render_rows_ex): recursive `shade` with mixed materials on the C3 mesh,
BVH world (the list world walks all 100 000 triangles per ray, like the reference): python tools/shade_time.py [spp] [depth]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from cudaraytracer_amd import api, scenes  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 4
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 8
path, nx, ny, _ = scenes.baseline_obj("C3", tempfile.mkdtemp())
scene = api.Scene.from_obj(path, scale=1.0)
cam = api.init_camera(nx, ny)
st = api.SetCurandState(nx, ny, 0)
api.create_BVHfromList(scene, st)
n = scene.n_triangles
mats = [(api.LAMBERTIAN, (0.65, 0.05, 0.05), 0.0), (api.METAL, (0.8, 0.85, 0.88), 0.1), (api.DIELECTRIC, (1, 1, 1), 1.5),
        (api.DIFFUSE_LIGHT, (4.0, 4.0, 4.0), 0.0)]
of = (np.arange(n) // 2500 % 4).astype(np.int32)          # bands of materials around the mesh
scene.set_materials(mats, of)
d = torch.zeros(ny * nx * 3, dtype=torch.float32, device="cuda")
s = torch.cuda.current_stream()
for shader, name in ((api.SHADER_SHADE, "shade depth %d" % depth), (api.SHADER_LAMBERT, "LambertShade (4 materials: tiled kernels, material mode)"), (api.SHADER_NORMAL, "shade_normal (tiled kernels)")):
    api.render_rows_ex(scene, cam, st, nx, ny, spp, 0, ny, d, world=api.WORLD_BVH, shader=shader, max_depth=depth, stream=s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        api.render_rows_ex(scene, cam, st, nx, ny, spp, 0, ny, d, world=api.WORLD_BVH, shader=shader, max_depth=depth, stream=s)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    img = d.cpu().numpy()
    print("C3 mesh, 4 materials, BVH world, %dx%dx%d, %s: %.2f ms/frame, %.0f M primary rays/s, mean colour %.3f, NaN %d"
          % (nx, ny, spp, name, ms, nx * ny * spp / ms / 1e3, float(np.nanmean(img)), int(np.isnan(img).sum())))
