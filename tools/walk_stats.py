#!/usr/bin/env python3
"""Lane occupancy of the generic kernel's loops on the shade workload (TUNING build: make -C cudaraytracer_amd/csrc tuning).
    python3 tools/walk_stats.py [spp] [mesh] [variant]      variant: 0 the shared walk (default), 11 one stack per lane, 9 skip links"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cudaraytracer_amd import api, scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 4
mesh = sys.argv[2] if len(sys.argv) > 2 else "C3"
variant = int(sys.argv[3]) if len(sys.argv) > 3 else 0
path, nx, ny, _ = scenes.baseline_obj(mesh, tempfile.mkdtemp())
scene = api.Scene.from_obj(path, scale=1.0)
cam = api.init_camera(nx, ny); st = api.SetCurandState(nx, ny, 0)
api.create_BVHfromList(scene, st)
n = scene.n_triangles
mats = [(api.LAMBERTIAN, (0.65, 0.05, 0.05), 0.0), (api.METAL, (0.8, 0.85, 0.88), 0.1), (api.DIELECTRIC, (1, 1, 1), 1.5), (api.DIFFUSE_LIGHT, (4.0, 4.0, 4.0), 0.0)]
scene.set_materials(mats, (np.arange(n) // 2500 % 4).astype(np.int32))
d = torch.zeros(ny * nx * 3, dtype=torch.float32, device="cuda")
api.set_kernel_variant(variant)
print("shade, depth 8, 4 materials, BVH world, %s, %d spp, kernel variant %d" % (mesh, spp, variant))
api.tiled_stats(1)
api.render_rows_ex(scene, cam, st, nx, ny, spp, 0, ny, d, world=api.WORLD_BVH, shader=api.SHADER_SHADE, max_depth=8, stream=torch.cuda.current_stream())
torch.cuda.synchronize()
s = api.tiled_stats(0)
for name, k in (("trips (lanes in flight)", 0), ("walk steps (lanes walking)", 2), ("triangle rounds (lanes testing)", 4)):
    print("%-34s %12d   lanes %.1f of 64" % (name, s[k], s[k + 1] / max(1, s[k])))
print("abandoned trips: %d" % s[6])
print("rays: %d (%.2f per primary ray); walk steps per ray %.1f, triangle tests per ray %.1f" % (s[1], s[1] / (nx * ny * spp), s[3] / max(1, s[1]), s[5] / max(1, s[1])))
api.tiled_stats(2)   # time stamps only, at full speed
api.render_rows_ex(scene, cam, st, nx, ny, spp, 0, ny, d, world=api.WORLD_BVH, shader=api.SHADER_SHADE, max_depth=8, stream=torch.cuda.current_stream())
torch.cuda.synchronize()
tl = api.wave_timeline(4096)
sl = api.sample_timeline().reshape(-1, 2)
s = api.tiled_stats(0)
t0 = (~s[10]) & 0xffffffffffffffff
w = max(1, s[13])
print("waves %d: queue empty for the last wave after %.2f ms, last wave ends after %.2f ms; mean over waves: queue empty after %.2f ms, end after %.2f ms"
      % (s[13], (s[8] - t0) / 1e5, (s[9] - t0) / 1e5, s[11] / w / 1e5, s[12] / w / 1e5))
print("time inside the walks (trace_any): %.1f %% of the wavefronts' lifetimes" % (100.0 * s[14] / max(1, s[12])))
tl = tl[tl[:, 2] != 0]
t0w = tl[:, 0].min()
end = (tl[:, 2] - t0w) / 1e5
empty = (tl[:, 1] - t0w) / 1e5
steps = tl[:, 3] >> 40
trips = (tl[:, 3] >> 20) & 0xfffff
pix0 = tl[:, 3] & 0xfffff
order = np.argsort(end)
print("per wavefront, by end time: percentile  end ms  queue-empty ms  walk steps  trips  pixels of lane 0")
for q in (0, 10, 25, 50, 75, 90, 95, 98, 99, 99.5, 99.9, 100):
    i = order[min(len(order) - 1, int(q / 100 * len(order)))]
    print("  %5.1f %8.2f %8.2f %8d %6d %5d" % (q, end[i], empty[i], steps[i], trips[i], pix0[i]))
print("%d wavefronts; sum of steps %d; ending after 5 ms: %d, after 8 ms: %d, after 10 ms: %d" % (len(tl), steps.sum(), (end > 5).sum(), (end > 8).sum(), (end > 10).sum()))
# the drain of the first 1024 wavefronts: when fewer than 16 pixels were left, when one, how many trips after "fewer than 16"
n1 = min(1024, len(tl))
few = (sl[:n1, 0].astype(np.int64) - int(t0w)) / 1e5
one = ((sl[:n1, 1] >> 16).astype(np.int64) - int(t0w)) / 1e5
trips_after = sl[:n1, 1] & 0xffff
e1 = end[:n1]
o1 = np.argsort(e1)
print("drain of wavefronts 0..%d, by end time: percentile  end ms  <16 pixels at ms  1 pixel at ms  trips after <16" % (n1 - 1))
for q in (50, 90, 99, 100):
    i = o1[min(n1 - 1, int(q / 100 * n1))]
    print("  %5.1f %8.2f %8.2f %8.2f %6d" % (q, e1[i], few[i], one[i], trips_after[i]))
