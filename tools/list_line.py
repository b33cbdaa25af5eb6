#!/usr/bin/env python3
"""One line for A/B scripts: list-world frame time of a BASELINE frame (HIP events; median and best of 30 after a warm-up):
    python tools/list_line.py [C3|C3B|C4|C5]"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from cudaraytracer_amd import api, scenes  # noqa: E402

out = []
for name in (sys.argv[1:] or ["C3"]):
    path, nx, ny, spp = scenes.baseline_obj(name, tempfile.mkdtemp())
    scene = api.Scene.from_obj(path, scale=1.0)
    cam = api.init_camera(nx, ny)
    st = api.SetCurandState(nx, ny, 0)
    d = torch.zeros(ny * nx * 3, dtype=torch.float32, device="cuda")
    ts = []
    for _ in range(60):
        api.render_rows(scene, cam, st, nx, ny, spp, 0, ny, d)
        torch.cuda.synchronize()
        ts.append(api.last_kernel_times())
    a = np.array(ts[30:])
    out.append("%s list total/bin/shade median %.4f %.4f %.4f best %.4f" % (name, *np.median(a, axis=0), a[:, 0].min()))
print(" | ".join(out))
