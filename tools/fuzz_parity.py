#!/usr/bin/env python3
"""Randomised parity soak on the GPU: tiled == brute force (pixels and RNG states, bit for bit)
and a clean binning audit on random soups / cameras / image sizes.
usage: fuzz_parity.py [seconds] [first_seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cudaraytracer_amd import api, _lib

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
L = _lib.lib()
t0 = time.time()
last_print = t0
n_cases = n_audit_pairs = n_strip_cases = 0
while time.time() - t0 < budget:
    rng = np.random.RandomState(seed)
    n = int(rng.choice([1, 7, 63, 64, 65, 300, 2000, 20000]))
    lf = rng.uniform(-50, 50, 3) * rng.choice([1.0, 1.0, 100.0])
    w = rng.normal(size=3); w /= np.linalg.norm(w)
    if abs(w[1]) > 0.98:
        w = np.array([0.3, 0.2, -0.93]); w /= np.linalg.norm(w)
    vfov = float(rng.choice([5.0, 40.0, 90.0, 140.0]))
    focus = float(rng.choice([0.1, 10.0, 300.0]))
    depth = rng.uniform(-5.0, 60.0, size=(n, 1))                     # some triangles behind the eye
    spread = np.abs(depth) * np.tan(np.radians(min(vfov, 120.0)) / 2) * 1.2 + 0.1
    centre = lf + w * depth + rng.normal(size=(n, 3)) * spread
    size = (np.abs(depth) + 0.5) * 10.0 ** rng.uniform(-3.5, 0.3, size=(n, 1))
    pts = (centre[:, None, :] + rng.normal(size=(n, 3, 3)) * size[:, None, :]).reshape(-1, 3).astype(np.float32)
    idx = np.arange(3 * n, dtype=np.float32).reshape(n, 3)
    if rng.rand() < 0.3:                                              # shared vertices / coincident edges: ties
        idx = rng.randint(0, max(3, n), size=(n, 3)).astype(np.float32)
    normals = rng.normal(size=(n, 3)).astype(np.float32) if rng.rand() < 0.3 else None
    nx, ny = int(rng.choice([8, 33, 160, 257, 640])), int(rng.choice([8, 17, 96, 130, 360]))
    spp = int(rng.choice([1, 2, 5]))
    scene = api.Scene.from_mesh(pts, idx, normals, 0)
    cam = api.create_camera(nx, ny, tuple(lf), tuple(lf + w), focus, 0.0, vfov)
    outs = []
    for v in (1, 2, 10):   # brute force, tiled, tiled with shared tiles from 4 entries on dealt out as row slabs
        prev = L.crt_set_kernel_variant(v)
        st = api.SetCurandState(nx, ny, seed)
        d = torch.zeros(nx * ny * 3, device="cuda")
        api.render_rows(scene, cam, st, nx, ny, spp, 0, ny, d)
        torch.cuda.synchronize()
        L.crt_set_kernel_variant(prev)
        outs.append((d.cpu().numpy().view(np.uint32), st.export()))
    ok = all(np.array_equal(outs[0][0], o[0]) and np.array_equal(outs[0][1], o[1]) for o in outs[1:])
    audit = None
    try:
        audit = api.audit_tiles(scene, cam, nx, ny, 2)
        ok = ok and audit[1] == 0 and audit[2] == 0
        n_audit_pairs += audit[0]
    except api.CrtError:
        pass
    if ok and ny >= 64 and rng.rand() < 0.5:   # one rank's round-robin strips of a W-way split == those rows of the full frame
        from cudaraytracer_amd.dist import partition_strips
        W = int(rng.choice([2, 3, 8])); r = int(rng.randint(0, W))
        strips = partition_strips(ny, W, r)
        if strips:
            rows = sum(c for _, c in strips)
            stp = api.CurandState.for_strips(nx, ny, r, 32, W, seed)
            piece = torch.zeros(rows * nx * 3, device="cuda")
            api.render_strips(scene, cam, stp, nx, ny, spp, r, 32, W, piece)
            torch.cuda.synchronize()
            full = outs[1][0].reshape(ny, nx, 3)
            want = np.concatenate([full[y0:y0 + c] for y0, c in strips])
            ok = np.array_equal(piece.cpu().numpy().view(np.uint32).reshape(rows, nx, 3), want)
            n_strip_cases += 1
    if not ok:
        print("MISMATCH seed %d: n=%d %dx%d spp %d vfov %g focus %g audit %s" % (seed, n, nx, ny, spp, vfov, focus, audit), flush=True)
        sys.exit(1)
    n_cases += 1
    seed += 1
    if time.time() - last_print > 60:
        print("  ... %d cases, %.0f s" % (n_cases, time.time() - t0), flush=True)
        last_print = time.time()
print("fuzz ok: %d cases (seeds up to %d), %d of them also as one rank's strips, %d audited passing pairs, %.0f s" % (n_cases, seed - 1, n_strip_cases, n_audit_pairs, time.time() - t0))
