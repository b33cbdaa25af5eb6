#!/usr/bin/env python3
"""Slowest rank of a W-way strip split of a BASELINE frame, rendered rank by rank on ONE GPU (HIP events per launch):
    python tools/strip_time.py [frame=C3] [worlds=1,2,4,8] [spp] [list|bvh]
frame: C3 (the lat-long mesh bench.py's headline uses), C3B (subdivided small_bunny + ground grid, no pole), C4 (C3's mesh
at 256 spp) or C5 (3840x2160, 1 M triangles) — the two configurations BASELINE assigns to 8 GPUs; world = the triangle
list or the BVHNode tree (the reference's shipped configuration).
Prints per W: the slowest rank's total / binning / shading ms (best of 5 launches per rank), the ideal T(1)/W, and the
speed-up before the gather that the slowest rank allows."""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from cudaraytracer_amd import api, scenes  # noqa: E402
from cudaraytracer_amd.dist import BandRenderer  # noqa: E402

frame = sys.argv[1] if len(sys.argv) > 1 else "C3"
worlds = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8").split(",")]
obj, nx, ny, spp = scenes.baseline_obj(frame, tempfile.mkdtemp())
if len(sys.argv) > 3 and int(sys.argv[3]) > 0:
    spp = int(sys.argv[3])
bvh = len(sys.argv) > 4 and sys.argv[4] == "bvh"
scene = api.Scene.from_obj(obj, scale=1.0)
cam = api.init_camera(nx, ny)
if bvh:
    api.create_BVHfromList(scene, None)
t1 = None
for world in worlds:
    worst, per_rank = (0.0, 0.0, 0.0), []
    for r in range(world):
        br = BandRenderer(scene, cam, nx, ny, r, world, mode="strips", bvh=bvh)
        ts = []
        for _ in range(6):
            br.render(spp)
            torch.cuda.synchronize()
            ts.append(tuple(api.last_kernel_times()))
        best = min(ts[1:])
        per_rank.append(best[0])
        worst = max(worst, best)
    if world == 1:
        t1 = worst[0]
    scaling = ("ideal %.3f | speed-up before the gather %.2fx" % (t1 / world, t1 / worst[0])) if t1 else "(no W=1 time in this run)"
    print("%s %s world %dx%dx%d W=%d: slowest rank total/bin/shade ms %.3f %.3f %.3f | %s | ranks: %s"
          % (frame, "BVH" if bvh else "list", nx, ny, spp, world, worst[0], worst[1], worst[2], scaling, " ".join("%.3f" % x for x in per_rank)), flush=True)
