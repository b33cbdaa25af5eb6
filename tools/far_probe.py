#!/usr/bin/env python3
"""Which lat-long spheres put slices of shared tiles on the far-pair path (stats 43..45) — sizing of the parity test."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from cudaraytracer_amd import api, scenes, _lib  # noqa: E402

for n_lat, n_lon, nx, ny in [(8, 700, 320, 180), (8, 1500, 320, 180), (8, 3000, 320, 180), (6, 6000, 256, 144), (8, 1500, 640, 360),
                             (12, 3000, 640, 360), (6, 12000, 256, 144)]:
    sc = scenes.displaced_sphere(n_lat, n_lon, amp=0.02)
    scene = api.Scene.from_mesh(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
    c = sc["camera"]
    cam = api.create_camera(nx, ny, c["lookfrom"], c["lookat"], c["focus"], c["aperture"], c["vfov"])
    st = api.SetCurandState(nx, ny, 0, 0, ny)
    d = torch.zeros(ny * nx * 3, dtype=torch.float32, device="cuda")
    _lib.lib().crt_set_kernel_variant(2)
    api.tiled_stats(1)
    api.render_rows(scene, cam, st, nx, ny, 4, 0, ny, d, stream=torch.cuda.current_stream())
    torch.cuda.synchronize()
    s = api.tiled_stats(0)
    print(n_lat, n_lon, nx, ny, "longest list", s[6], "shared tiles", s[35], "far slices / pairs / overflowed", s[43:47], flush=True)
