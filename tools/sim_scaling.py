#!/usr/bin/env python3
"""Estimate multi-GPU strong scaling on ONE GPU: render each rank's rows separately and take
the slowest rank's kernel time (the gather is not included).  usage: sim_scaling.py [spp]"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from cudaraytracer_amd import api
from cudaraytracer_amd.dist import BandRenderer

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
obj = bench.build_scene_obj(tempfile.mkdtemp())
scene = api.Scene.from_obj(obj, scale=1.0)
cam = api.init_camera(bench.NX, bench.NY)
for mode in ("bands", "strips"):
    for world in (1, 2, 4, 8):
        times = []
        for r in range(world):
            br = BandRenderer(scene, cam, bench.NX, bench.NY, r, world, mode=mode)
            ms = []
            for _ in range(4):
                br.render(spp)
                torch.cuda.synchronize()
                ms.append(api.last_kernel_times()[0])
            times.append(min(ms[1:]))
        print("%-6s world %d: per-rank kernel ms min %.3f max %.3f  -> speedup vs sum %.2f (ideal %d)"
              % (mode, world, min(times), max(times), sum(times) / max(times) if world > 1 else 1.0, world))
