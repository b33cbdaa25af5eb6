#!/usr/bin/env python3
"""One rank's strips of a W-way split, rendered N times (for rocprofv3 --kernel-trace --stats): 
    python tools/strip_one.py [frame=C5] [world=8] [rank=0] [frames=20] [list|bvh]"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from cudaraytracer_amd import api, scenes  # noqa: E402
from cudaraytracer_amd.dist import BandRenderer  # noqa: E402

frame = sys.argv[1] if len(sys.argv) > 1 else "C5"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rank = int(sys.argv[3]) if len(sys.argv) > 3 else 0
n = int(sys.argv[4]) if len(sys.argv) > 4 else 20
bvh = len(sys.argv) > 5 and sys.argv[5] == "bvh"
obj, nx, ny, spp = scenes.baseline_obj(frame, tempfile.mkdtemp())
scene = api.Scene.from_obj(obj, scale=1.0)
cam = api.init_camera(nx, ny)
if bvh:
    api.create_BVHfromList(scene, None)
br = BandRenderer(scene, cam, nx, ny, rank, world, mode="strips" if world > 1 else "bands", bvh=bvh)
for _ in range(n):
    br.render(spp)
torch.cuda.synchronize()
print("%s W=%d rank %d x %d frames: last kernel times total/bin/shade %s" % (frame, world, rank, n, api.last_kernel_times()))
