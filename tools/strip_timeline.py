#!/usr/bin/env python3
"""Wave timeline of ONE rank's strips of a W-way split (shade_tiles_kernel, timeline mode: full speed):
    python tools/strip_timeline.py [frame=C3] [world=8] [rank=0]
When do the waves start, when does each block leave its shared tiles, when do the waves end — and which tiles do the
last ones hold?"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from cudaraytracer_amd import _lib, api, scenes  # noqa: E402
from cudaraytracer_amd.dist import BandRenderer  # noqa: E402

frame = sys.argv[1] if len(sys.argv) > 1 else "C3"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rank = int(sys.argv[3]) if len(sys.argv) > 3 else 0
obj, nx, ny, spp = scenes.baseline_obj(frame, tempfile.mkdtemp())
scene = api.Scene.from_obj(obj, scale=1.0)
cam = api.init_camera(nx, ny)
br = BandRenderer(scene, cam, nx, ny, rank, world, mode="strips")
L = _lib.lib()
for _ in range(5):
    br.render(spp)
torch.cuda.synchronize()
L.crt_debug_tiled_stats(1, None)
br.render(spp)
torch.cuda.synchronize()
import ctypes as C  # noqa: E402
out = (C.c_uint64 * 48)()
L.crt_debug_tiled_stats(0, out)
o = list(out)
print("%s W=%d rank %d: tiles %d, sum of lists %d, max %d; histogram (bucket b: 2^(b-1) <= n < 2^b) %s" % (frame, world, rank, o[4], o[5], o[6], o[8:20]))
L.crt_debug_tiled_stats(2, None)
tx = (nx + 7) // 8
for rep in range(3):
    for _ in range(5):
        br.render(spp)
    torch.cuda.synchronize()
    ms = api.last_kernel_times()
    tl = api.wave_timeline()
    tl = tl[tl[:, 2] > 0].astype(np.int64)
    t0 = tl[:, 0].min()
    start, mid, end = (tl[:, 0] - t0) / 100.0, (tl[:, 1] - t0) / 100.0, (tl[:, 2] - t0) / 100.0
    heavy = (tl[:, 3] >> 20) != 0                       # the block held a shared tile
    print("frame %d: kernel ms total/bin/shade %.3f %.3f %.3f | waves %d | start max %.1f us | shared phase ends (blocks that had one: %d waves): median %.1f p90 %.1f max %.1f"
          " | wave ends: p10 %.1f median %.1f p90 %.1f max %.1f"
          % (rep, ms[0], ms[1], ms[2], len(tl), start.max(), int(heavy.sum()), np.median(mid[heavy]) if heavy.any() else 0, np.percentile(mid[heavy], 90) if heavy.any() else 0,
             mid[heavy].max() if heavy.any() else 0, np.percentile(end, 10), np.median(end), np.percentile(end, 90), end.max()))
    last = np.argsort(-end)[:6]
    print("   last waves: " + "; ".join("end %.0f us (shared phase until %.0f, last shared tile (%d, %d) %d entries)"
                                        % (end[i], mid[i], ((tl[i, 3] >> 20) & 0xffffff) % tx, ((tl[i, 3] >> 20) & 0xffffff) // tx, tl[i, 3] >> 44) for i in last))
    longest = np.argsort(-mid)[::8][:4]
    print("   longest shared phases: " + "; ".join("%.0f us tile (%d, %d) %d entries" % (mid[i], ((tl[i, 3] >> 20) & 0xffffff) % tx, ((tl[i, 3] >> 20) & 0xffffff) // tx, tl[i, 3] >> 44) for i in longest))
    after = end - np.maximum(mid, start)
    print("   time after the shared phase (light tiles): median %.1f p90 %.1f max %.1f us" % (np.median(after), np.percentile(after, 90), after.max()))
edges = np.arange(0, end.max() + 20.0, 20.0)
print("last frame, wave ends per 20 us:", np.histogram(end, edges)[0].tolist())
L.crt_debug_tiled_stats(0, None)
