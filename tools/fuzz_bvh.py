#!/usr/bin/env python3
"""Randomised soak of the BVHNode world on the GPU: the tiled mode (variant 0), the same with every lane on the general
path (variant 7) and the per-ray tree walk (variant 1) must agree bit for bit, pixels and RNG states, on random soups /
cameras / image sizes — long lists (shared tiles, staged chunks, tree-walk tiles) included.
usage: fuzz_bvh.py [seconds] [first_seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cudaraytracer_amd import api, _lib

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
L = _lib.lib()
t0 = time.time()
last_print = t0
n_cases = 0
while time.time() - t0 < budget:
    rng = np.random.RandomState(seed)
    n = int(rng.choice([2, 7, 63, 64, 65, 300, 2000, 6000]))
    lf = rng.uniform(-50, 50, 3) * rng.choice([1.0, 1.0, 100.0])
    w = rng.normal(size=3); w /= np.linalg.norm(w)
    if abs(w[1]) > 0.98:
        w = np.array([0.3, 0.2, -0.93]); w /= np.linalg.norm(w)
    vfov = float(rng.choice([5.0, 40.0, 90.0, 140.0]))
    focus = float(rng.choice([0.1, 10.0, 300.0]))
    depth = rng.uniform(-5.0, 60.0, size=(n, 1))                     # some triangles behind the eye
    spread = np.abs(depth) * np.tan(np.radians(min(vfov, 120.0)) / 2) * float(rng.choice([1.2, 0.2, 0.02])) + 0.1   # also piles
    centre = lf + w * depth + rng.normal(size=(n, 3)) * spread
    size = (np.abs(depth) + 0.5) * 10.0 ** rng.uniform(-3.5, 0.3, size=(n, 1))
    pts = (centre[:, None, :] + rng.normal(size=(n, 3, 3)) * size[:, None, :]).reshape(-1, 3).astype(np.float32)
    idx = np.arange(3 * n, dtype=np.float32).reshape(n, 3)
    if rng.rand() < 0.3:                                              # shared vertices / coincident edges: ties
        idx = rng.randint(0, max(3, n), size=(n, 3)).astype(np.float32)
    if rng.rand() < 0.2:                                              # axis-aligned triangles: flat boxes are never hit
        pts.reshape(n, 3, 3)[::3, :, 2] = pts.reshape(n, 3, 3)[::3, :1, 2]
    nx, ny = int(rng.choice([8, 33, 160, 257])), int(rng.choice([8, 17, 96, 130]))
    spp = int(rng.choice([1, 2, 5]))
    scene = api.Scene.from_mesh(pts, idx, None, 0)
    cam = api.create_camera(nx, ny, tuple(lf), tuple(lf + w), focus, 0.0, vfov)
    outs = []
    for v in (0, 1, 7):
        prev = L.crt_set_kernel_variant(v)
        st = api.SetCurandState(nx, ny, seed)
        api.create_BVHfromList(scene, st)            # consumes draws of stream 0 (axis choices): the same tree every time
        d = torch.zeros(nx * ny * 3, device="cuda")
        api.render_rows(scene, cam, st, nx, ny, spp, 0, ny, d, bvh=True)
        torch.cuda.synchronize()
        L.crt_set_kernel_variant(prev)
        outs.append((d.cpu().numpy().view(np.uint32), st.export()))
    ok = all(np.array_equal(outs[0][0], o[0]) and np.array_equal(outs[0][1][1:], o[1][1:]) for o in outs[1:])
    if not ok:
        bad = [int((outs[0][0] != o[0]).sum()) for o in outs[1:]]
        print("MISMATCH seed %d: n=%d %dx%d spp %d vfov %g focus %g, differing words vs walk / general: %s" % (seed, n, nx, ny, spp, vfov, focus, bad), flush=True)
        sys.exit(1)
    n_cases += 1
    seed += 1
    if time.time() - last_print > 60:
        print("  ... %d cases, %.0f s" % (n_cases, time.time() - t0), flush=True)
        last_print = time.time()
print("bvh fuzz ok: %d cases (seeds up to %d): tiled == general == tree walk, pixels and states, %.0f s" % (n_cases, seed - 1, time.time() - t0))
