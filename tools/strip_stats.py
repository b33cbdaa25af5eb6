#!/usr/bin/env python3
"""Survivor / cycle statistics of ONE rank's strip set (world 8 by default) on the bench scene."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cudaraytracer_amd import api
from cudaraytracer_amd.dist import BandRenderer
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
obj = bench.build_scene_obj(tempfile.mkdtemp())
scene = api.Scene.from_obj(obj, scale=1.0)
cam = api.init_camera(bench.NX, bench.NY)
for r in range(min(world, 3)):
    br = BandRenderer(scene, cam, bench.NX, bench.NY, r, world, mode="strips")
    br.render(64); torch.cuda.synchronize()
    api.tiled_stats(True)
    br.render(64); torch.cuda.synchronize()
    t = api.last_kernel_times()
    o = api.tiled_stats(False)
    print("rank %d: total %.3f bin %.3f shade %.3f ms | tiles %d sum2 %d max2 %d | hist %s" % (r, t[0], t[1], t[2], o[4], o[5], o[6], o[8:20]))
    print("   shade cycles: heavy sum/blocks %d max %d | light sum/waves %d max %d | max wave total %d" % tuple(o[24:29]))
    print("   heavy tiles %d (sum survivors %d): master cycles per tile: setup+merge+shade %.0f, own trace %.0f, wait B2 %.0f" % (o[35], o[36], o[32]/max(o[35],1), o[33]/max(o[35],1), o[34]/max(o[35],1)))
    br2 = BandRenderer(scene, cam, bench.NX, bench.NY, r, world, mode="strips")
    ts = []
    for _ in range(5):
        br2.render(64); torch.cuda.synchronize(); ts.append(api.last_kernel_times())
    print("   without stats: total/bin/shade ms", min(ts))
