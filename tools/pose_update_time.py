#!/usr/bin/env python3
"""Time of one animation step on the bench scene: crt_scene_update_pose (H2D copy of the points +
update_pose kernel) and the following frame."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from cudaraytracer_amd import api, scenes
sc = scenes.bunny100k()
pts = np.asarray(sc["points"], np.float32).reshape(-1, 3)
scene = api.Scene.from_mesh(pts, sc["idx"], sc["normals"], sc["index_base"])
cam = api.init_camera(bench.NX, bench.NY)
st = api.SetCurandState(bench.NX, bench.NY)
d = torch.zeros(bench.NX * bench.NY * 3, device="cuda")
s = torch.cuda.current_stream()
for f in range(6):
    pose = (pts * np.float32(1.0 + 0.01 * f)).astype(np.float32)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    scene.update_pose(pose, stream=s)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    api.render_rows(scene, cam, st, bench.NX, bench.NY, 64, 0, bench.NY, d, frame_index=f, stream=s)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("frame %d: update_pose (%d points, host wall incl. H2D) %.3f ms, render wall %.3f ms, kernels %s" % (f, len(pts), (t1 - t0) * 1e3, (t2 - t1) * 1e3, api.last_kernel_times()))
