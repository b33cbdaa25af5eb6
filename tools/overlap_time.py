#!/usr/bin/env python3
"""What the overlapped exchange hides, measured at world 1 on ONE GPU (the only RCCL world a one-GPU box has):
    python tools/overlap_time.py [frame=C3] [steps=50]
A step = this rank's strips (here: the whole frame) + crt_gather_strips (the root's send to itself goes through RCCL: the
whole 24.9 MB frame, against 1/W of it per peer in a W-way split) + the reassembly kernel.  Three timings of the same K
steps between two device synchronisations: the render alone, render + exchange on one stream (BandRenderer(overlap=False)),
render + exchange on a stream of its own with two pieces in turn (overlap=True, what bench.py --gpus N runs)."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from cudaraytracer_amd import _lib, api, scenes  # noqa: E402
from cudaraytracer_amd.dist import BandRenderer  # noqa: E402

frame = sys.argv[1] if len(sys.argv) > 1 else "C3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
obj, nx, ny, spp = scenes.baseline_obj(frame, tempfile.mkdtemp())
scene = api.Scene.from_obj(obj, scale=1.0)
cam = api.init_camera(nx, ny)
_lib.lib().crt_set_kernel_timing(0)


def timed(br, exchange):
    def step():
        br.render(spp)
        if exchange:
            br.gather()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:      # clock ramp
        for _ in range(10):
            step()
        torch.cuda.synchronize()
    out = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        br.wait()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / steps * 1e3)
    return sorted(out)[1]


for overlap in (False, True):
    br = BandRenderer(scene, cam, nx, ny, 0, 1, mode="strips", gather="c", overlap=overlap)
    alone = timed(br, False)
    both = timed(br, True)
    print("%s %dx%dx%d world 1, exchange %s: render alone %.3f ms/step, render + exchange %.3f ms/step (median of 3 x %d steps)"
          % (frame, nx, ny, spp, "on a stream of its own, two pieces in turn" if overlap else "on the render's stream", alone, both, steps), flush=True)
    br.close()
