#!/usr/bin/env python3
"""Randomised soak of crt_render_rows_ex (recursive shade over the pair records; LambertShade with materials and shade_normal
through the tiled kernels' material mode; the four material classes, both
worlds) against the CPU oracle — the loop of tests/test_shade.py::test_gpu_shade_random_soups_match_oracle on a time budget.
Lives under tests/ because it uses the oracle.   usage: python tests/fuzz_shade.py [seconds] [first_seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from cudaraytracer_amd import _lib, api, scenes  # noqa: E402
from oracle import pyoracle as oracle  # noqa: E402
from test_shade import same  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
t0 = time.time()
last = t0
n_cases = 0
while time.time() - t0 < budget:
    rng = np.random.RandomState(seed)
    n = int(rng.choice([3, 17, 64, 150, 400]))
    sc = scenes.random_soup(n, seed, extent=float(rng.choice([2.0, 4.0])), size=float(rng.choice([0.7, 1.5, 3.0])))
    kinds = [api.LAMBERTIAN, api.METAL, api.DIELECTRIC, api.DIFFUSE_LIGHT]
    mats = [(int(rng.choice(kinds)), tuple(float(x) for x in rng.uniform(0.05, 1.0, 3)), float(rng.choice([0.0, 0.3, 1.5, 2.4])))
            for _ in range(int(rng.randint(1, 7)))]
    of = rng.randint(0, len(mats), n).astype(np.int32)
    nx, ny, spp = int(rng.choice([16, 33, 70])), int(rng.choice([8, 20, 40])), int(rng.choice([1, 2, 3]))
    world = str(rng.choice(["list", "bvh"]))
    shader = int(rng.choice([api.SHADER_SHADE, api.SHADER_SHADE, api.SHADER_LAMBERT, api.SHADER_NORMAL]))
    depth = int(rng.choice([0, 1, 2, 8, 31]))
    w = oracle.World(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
    w.set_materials(mats, of)
    c = sc["camera"]
    ocam = oracle.make_camera(c["lookfrom"], c["lookat"], (0, 1, 0), c["vfov"], np.float32(nx) / np.float32(ny), c["aperture"], c["focus"])
    scene = api.Scene.from_mesh(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
    cam = api.create_camera(nx, ny, c["lookfrom"], c["lookat"], c["focus"], c["aperture"], c["vfov"])
    st = api.SetCurandState(nx, ny, 0)
    states = oracle.fresh_states(nx, ny)
    bvh = None
    if world == "bvh":
        api.create_BVHfromList(scene, st)
        bvh = oracle.Bvh(w, states)
    scene.set_materials(mats, of)
    ref, states = oracle.render_ex(w, bvh, shader, ocam, nx, ny, spp, states, max_depth=depth)
    after = np.concatenate([states["d"][:, None], states["v"]], axis=1)
    d = torch.zeros(ny * nx * 3, dtype=torch.float32, device="cuda")
    # the walks of the generic kernel: 0 shared by the wavefront (default), 12 the same with a short list (abandoned trips),
    # 11 one stack per lane, 9 skip links; 1: LambertShade / shade_normal through the generic kernel instead of the tiled ones
    variant = (0, 0, 12, 11, 9)[seed % 5] if shader == api.SHADER_SHADE else (0, 1)[seed % 2]
    prev = _lib.lib().crt_set_kernel_variant(variant)
    api.render_rows_ex(scene, cam, st, nx, ny, spp, 0, ny, d, world=api.WORLD_BVH if world == "bvh" else api.WORLD_LIST, shader=shader,
                       max_depth=depth)
    torch.cuda.synchronize()
    _lib.lib().crt_set_kernel_variant(prev)
    got = d.cpu().numpy().reshape(ny, nx, 3)
    if not (same(got, ref) and np.array_equal(st.export()[1:], after[1:])):
        print("MISMATCH seed %d: n=%d %dx%dx%d %s shader %d depth %d variant %d" % (seed, n, nx, ny, spp, world, shader, depth, variant), flush=True)
        sys.exit(1)
    n_cases += 1
    seed += 1
    if time.time() - last > 60:
        print("  ... %d cases, %.0f s" % (n_cases, time.time() - t0), flush=True)
        last = time.time()
print("shade fuzz ok: %d cases (seeds up to %d): generic kernel == oracle, pixels and RNG states, %.0f s" % (n_cases, seed - 1, time.time() - t0))
