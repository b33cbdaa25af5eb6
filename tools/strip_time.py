#!/usr/bin/env python3
"""Kernel time of ONE rank's strip set on the bench scene: strip_time.py [world] [rank]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cudaraytracer_amd import api
from cudaraytracer_amd.dist import BandRenderer
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ranks = [int(sys.argv[2])] if len(sys.argv) > 2 else range(world)
obj = bench.build_scene_obj(tempfile.mkdtemp())
scene = api.Scene.from_obj(obj, scale=1.0)
cam = api.init_camera(bench.NX, bench.NY)
worst = (0, 0, 0)
for r in ranks:
    br = BandRenderer(scene, cam, bench.NX, bench.NY, r, world, mode="strips")
    ts = []
    for _ in range(5):
        br.render(64); torch.cuda.synchronize(); ts.append(api.last_kernel_times())
    worst = max(worst, min(ts))
print("world %d env %s: slowest rank total/bin/shade ms %.3f %.3f %.3f" % ((world, {k: v for k, v in os.environ.items() if k.startswith("CRT_")}) + worst))
