#!/usr/bin/env python3
"""ONE of the f1/f2 workloads on the C3 frame (1920x1080, the 100 000-triangle bench mesh), for rocprofv3:
    python3 tools/f_workload.py <which> [spp] [frames] [mesh]
which:  shade        recursive `shade`, depth 8, 4 materials, BVH world     (render.h:48-67; spp default 4)
        lambert4     LambertShade with the same 4 materials, BVH world       (render.h:70-87; spp default 4)
        lambert4_list  the same in the list world
        lambert4_generic  LambertShade with the 4 materials, BVH world, through the generic kernel (variant 1)
        normal       shade_normal, BVH world                                 (render.h:90-103)
        bvh_walk     LambertShade, one Lambertian, BVH world, per-ray tree walk only (variant 1: the generic kernel's shared walk; spp default 4)
        bvh_tiled    LambertShade, one Lambertian, BVH world, tiled kernels  (the default path; spp default 64)
        list         LambertShade, one Lambertian, list world (what bench.py times; spp default 64)
mesh:   C3 (default; the lat-long mesh) or C3B (subdivided bunny + ground grid)
Prints one line: ms per frame from wall clock around `frames` launches."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from cudaraytracer_amd import api, scenes  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "shade"
spp = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) > 0 else (64 if which in ("bvh_tiled", "list") else 4)
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 5
mesh = sys.argv[4] if len(sys.argv) > 4 else "C3"
path, nx, ny, _ = scenes.baseline_obj(mesh, tempfile.mkdtemp())
scene = api.Scene.from_obj(path, scale=1.0)
cam = api.init_camera(nx, ny)
st = api.SetCurandState(nx, ny, 0)
if which not in ("list", "lambert4_list"):
    api.create_BVHfromList(scene, st)
n = scene.n_triangles
if which in ("shade", "lambert4", "lambert4_list", "lambert4_generic", "normal"):
    mats = [(api.LAMBERTIAN, (0.65, 0.05, 0.05), 0.0), (api.METAL, (0.8, 0.85, 0.88), 0.1), (api.DIELECTRIC, (1, 1, 1), 1.5),
            (api.DIFFUSE_LIGHT, (4.0, 4.0, 4.0), 0.0)]
    scene.set_materials(mats, (np.arange(n) // 2500 % 4).astype(np.int32))
d = torch.zeros(ny * nx * 3, dtype=torch.float32, device="cuda")
s = torch.cuda.current_stream()


def frame():
    if which == "shade":
        api.render_rows_ex(scene, cam, st, nx, ny, spp, 0, ny, d, world=api.WORLD_BVH, shader=api.SHADER_SHADE, max_depth=8, stream=s)
    elif which == "lambert4":
        api.render_rows_ex(scene, cam, st, nx, ny, spp, 0, ny, d, world=api.WORLD_BVH, shader=api.SHADER_LAMBERT, max_depth=8, stream=s)
    elif which == "lambert4_generic":
        api.set_kernel_variant(1)
        api.render_rows_ex(scene, cam, st, nx, ny, spp, 0, ny, d, world=api.WORLD_BVH, shader=api.SHADER_LAMBERT, max_depth=8, stream=s)
        api.set_kernel_variant(0)
    elif which == "lambert4_list":
        api.render_rows_ex(scene, cam, st, nx, ny, spp, 0, ny, d, world=api.WORLD_LIST, shader=api.SHADER_LAMBERT, max_depth=8, stream=s)
    elif which == "normal":
        api.render_rows_ex(scene, cam, st, nx, ny, spp, 0, ny, d, world=api.WORLD_BVH, shader=api.SHADER_NORMAL, max_depth=8, stream=s)
    elif which == "bvh_walk":
        api.set_kernel_variant(1)
        api.render_rows(scene, cam, st, nx, ny, spp, 0, ny, d, stream=s, bvh=True)
        api.set_kernel_variant(0)
    elif which == "bvh_tiled":
        api.render_rows(scene, cam, st, nx, ny, spp, 0, ny, d, stream=s, bvh=True)
    elif which == "list":
        api.render_rows(scene, cam, st, nx, ny, spp, 0, ny, d, stream=s)
    else:
        sys.exit("unknown workload " + which)


frame()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(frames):
    frame()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / frames * 1e3
print("%s on %s %dx%dx%d, %d triangles: %.3f ms/frame, %.0f M primary rays/s" % (which, mesh, nx, ny, spp, n, ms, nx * ny * spp / ms / 1e3))
