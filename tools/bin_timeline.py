#!/usr/bin/env python3
"""bin_tri_kernel on the device clock (timeline mode of crt_debug_tiled_stats): when the last wavefront ended and which
wavefront took longest.   python tools/bin_timeline.py [C3|C5] [list|bvh]
(Three same-address atomics per wavefront: they queue up behind one another, 20 ns each — with 15 000 wavefronts the times
printed are several times those of an undisturbed launch; what the tool is good for is WHICH wavefront is the last.)"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from cudaraytracer_amd import api, scenes  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C5"
path, nx, ny, spp = scenes.baseline_obj(name, tempfile.mkdtemp())
scene = api.Scene.from_obj(path, scale=1.0)
cam = api.init_camera(nx, ny)
st = api.SetCurandState(nx, ny, 0)
d = torch.zeros(ny * nx * 3, dtype=torch.float32, device="cuda")
for _ in range(3):
    api.render_rows(scene, cam, st, nx, ny, 1, 0, ny, d)
torch.cuda.synchronize()
for rep in range(3):
    api.tiled_stats(2)
    api.render_rows(scene, cam, st, nx, ny, 1, 0, ny, d)
    torch.cuda.synchronize()
    o = api.tiled_stats(0)
    end = (o[30] - (1 << 32)) / 100.0
    longest, start, block = (o[31] >> 40) / 100.0, (((o[31] >> 24) & 0xffff) - (1 << 15)) / 100.0, o[31] & 0xffffff
    print("%s: bin_tri_kernel: last wavefront ends %.1f us after the first began; the longest wavefront took %.1f us (began at %.1f us): triangles %d .. ; kernel ms %s"
          % (name, end, longest, start, block * 64, api.last_kernel_times()), flush=True)
