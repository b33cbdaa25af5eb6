#!/usr/bin/env python3
"""Does the timeline mode of crt_debug_tiled_stats leave the shading kernel's time alone? (run on the GPU box)"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cudaraytracer_amd import _lib, api
obj = bench.build_scene_obj(tempfile.mkdtemp())
scene = api.Scene.from_obj(obj, scale=1.0)
cam = api.init_camera(bench.NX, bench.NY)
st = api.SetCurandState(bench.NX, bench.NY)
d = torch.zeros(bench.NX * bench.NY * 3, device="cuda")
L = _lib.lib()
def frames(n):
    for _ in range(n):
        api.render_rows(scene, cam, st, bench.NX, bench.NY, 64, 0, bench.NY, d)
    torch.cuda.synchronize()
    return api.last_kernel_times()
print("plain    ", frames(30))
L.crt_debug_tiled_stats(2, None)
print("timeline ", frames(30))
L.crt_debug_tiled_stats(0, None)
print("plain    ", frames(30))
L.crt_debug_tiled_stats(1, None)
print("full     ", frames(30))
L.crt_debug_tiled_stats(0, None)
