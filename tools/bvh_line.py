#!/usr/bin/env python3
"""One line for A/B scripts: BVH-world and list-world frame times (HIP events, best of 8) of a BASELINE frame:
    python tools/bvh_line.py [C3|C4|C5]"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from cudaraytracer_amd import api, scenes  # noqa: E402

out = []
for name in (sys.argv[1:] or ["C3"]):
    path, nx, ny, spp = scenes.baseline_obj(name, tempfile.mkdtemp())
    scene = api.Scene.from_obj(path, scale=1.0)
    cam = api.init_camera(nx, ny)
    st = api.SetCurandState(nx, ny, 0)
    api.create_BVHfromList(scene, st)
    d = torch.zeros(ny * nx * 3, dtype=torch.float32, device="cuda")
    for bvh in (True, False):
        ts = []
        for _ in range(10):
            api.render_rows(scene, cam, st, nx, ny, spp, 0, ny, d, bvh=bvh)
            torch.cuda.synchronize()
            ts.append(api.last_kernel_times())
        best = min(ts[2:])
        out.append("%s %s total/bin/shade %.3f %.3f %.3f" % (name, "BVH" if bvh else "list", best[0], best[1], best[2]))
print(" | ".join(out))
