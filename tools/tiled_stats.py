#!/usr/bin/env python3
"""Print the tiled kernels' survivor statistics, cycle accounting and wave timeline (run on the GPU box).
usage: tiled_stats.py [spp] [list|bvh] [C3|C4|C5]   (default: the bench frame C3, list world)"""
import ctypes as C
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from cudaraytracer_amd import _lib, api

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 1
BVH = len(sys.argv) > 2 and sys.argv[2] == "bvh"   # the BVHNode world through the tiled kernels
from cudaraytracer_amd import scenes
FRAME = sys.argv[3] if len(sys.argv) > 3 else bench.FRAME
obj, NX, NY, _ = scenes.baseline_obj(FRAME, tempfile.mkdtemp())
scene = api.Scene.from_obj(obj, scale=1.0)
cam = api.init_camera(NX, NY)
st = api.SetCurandState(NX, NY)
if BVH:
    api.create_BVHfromList(scene, st)
d = torch.zeros(NX * NY * 3, device="cuda")
L = _lib.lib()
L.crt_debug_tiled_stats(1, None)
api.render_rows(scene, cam, st, NX, NY, spp, 0, NY, d, bvh=BVH)
torch.cuda.synchronize()
out = (C.c_uint64 * 48)()
L.crt_debug_tiled_stats(0, out)
o = list(out)
print("bin wavefronts %d: regions entered %d (%.1f per wave), tiles box-tested %d (%.1f), tiles with survivors %d (%.1f)" % (o[0], o[1], o[1] / max(o[0], 1), o[2], o[2] / max(o[0], 1), o[3], o[3] / max(o[0], 1)))
print("tiles %d  sum2 %d (avg %.1f) max2 %d overflow2 %d" % (o[4], o[5], o[5] / max(o[4], 1), o[6], o[7]))
print("hist (bucket b: 2^(b-1) <= n < 2^b):", o[8:24])
print("shade kernel cycles: heavy phase sum over blocks %d max %d | light phase sum over waves %d max %d | max wave total %d" % tuple(o[24:29]))
print("bin busiest wavefront: regions %d, tiles box-tested %d (maxima over the wavefronts); the slowest took %d cycles: triangles %d .." % (o[29], o[30], o[31] >> 24, (o[31] & 0xffffff) * 64))
print("heavy tiles %d (sum survivors %d): master wave cycles per tile: setup+merge+shade %.0f, own trace %.0f, wait B2 %.0f" % (o[35], o[36], o[32]/max(o[35],1), o[33]/max(o[35],1), o[34]/max(o[35],1)))
print("shared tiles' slices longer than a staging area: %d as far pairs (%d pairs), %d did not fit; most pairs of one slice %d" % tuple(o[43:47]))
print("kernel ms", api.last_kernel_times(), flush=True)
print("light tiles with geometry %d: list entries %.1f per tile; per-pixel candidates: longest of the wavefront %.1f, mean %.2f; pair buffer overflowed in %d tiles, %.0f pairs per tile otherwise"
      % (o[37], o[38] / max(o[37], 1), o[39] / max(o[37], 1), o[40] / max(o[37], 1) / 64.0, o[41], o[42] / max(o[37] - o[41], 1)))

# timeline mode: wave start / end times on the 100 MHz device clock, with the kernel's timing intact
L.crt_debug_tiled_stats(2, None)
tx = (NX + 7) // 8
for rep in range(6):
    for _ in range((40 if rep == 0 else 7) if NX * NY < 4000000 else 3):   # warm clocks; every frame continues the RNG streams, so the frames differ
        api.render_rows(scene, cam, st, NX, NY, spp, 0, NY, d, bvh=BVH)
    torch.cuda.synchronize()
    tl = api.wave_timeline()
    tl = tl[tl[:, 2] > 0].astype(np.int64)
    t0 = tl[:, 0].min()
    start, mid, end = (tl[:, 0] - t0) / 100.0, (tl[:, 1] - t0) / 100.0, (tl[:, 2] - t0) / 100.0
    order = np.argsort(-mid)[::8][:3]   # one wave per block
    print("frame %d: kernel ms %.3f; heavy phase ends: median %.1f p90 %.1f max %.1f us; wave ends: p10 %.1f median %.1f max %.1f; mean lifetime %.1f; longest heavy phases: %s"
          % (rep, api.last_kernel_times()[2], np.median(mid), np.percentile(mid, 90), mid.max(), np.percentile(end, 10), np.median(end), end.max(), (end - start).mean(),
             "; ".join("%.0f us tile (%d, %d) %d entries" % (mid[i], ((tl[i, 3] >> 20) & 0xffffff) % tx, ((tl[i, 3] >> 20) & 0xffffff) // tx, tl[i, 3] >> 44) for i in order)))
    hw = tl[:, 3] & 0xfffff                      # HW_REG_HW_ID: wave [3:0], SIMD [5:4], pipe [7:6], CU [11:8], SH [12], SE [15:13]
    simd = ((hw >> 4) & 3).reshape(-1, 8)        # per block
    per_simd = np.stack([(simd == k).sum(axis=1) for k in range(4)], axis=1)
    print("   waves of a block per SIMD: blocks by their most crowded SIMD: %s; block 0: %s, block 1: %s"
          % (np.bincount(per_simd.max(axis=1), minlength=9).tolist(), per_simd[0].tolist(), per_simd[1].tolist()))
    sl = api.sample_timeline().astype(np.int64)
    for b in range(2):
        t_b2 = sl[b, :spp, 1]
        per = np.diff(t_b2) / 100.0
        wait = (sl[b, :spp, 1] - sl[b, :spp, 0]) / 100.0
        if BVH:   # stamps: slices done (after B2), winners' chains checked
            per = np.diff(sl[b, :spp, 0]) / 100.0
            print("   block %d per-sample us: median %.2f max %.2f; of which the master's chain checks after the slices: median %.2f max %.2f"
                  % (b, np.median(per), per.max(), np.median(wait), wait.max()))
            continue
        print("   block %d per-sample us: median %.2f max %.2f (sample %d), samples over 3x the median: %d; master's wait for the slices after its draws: median %.2f max %.2f"
              % (b, np.median(per), per.max(), int(per.argmax()) + 1, int((per > 3 * np.median(per)).sum()), np.median(wait), wait.max()))
ll = api.light_timeline()[: len(api.wave_timeline())].astype(np.int64)
ll = ll[api.wave_timeline()[:, 2] > 0]
late = np.argsort(-end)[:12]
dur = (ll[:, 2] & 0xfffff) / 100.0
cnts = ll[:, 2] >> 20
print("last frame, the waves that end last: " + "; ".join(
    "end %.0f: last geometry tile (%d, %d) %d entries started %.0f took %.0f us, %d such tiles, heavy phase until %.0f [wave-samples with chain loads %d, lanes %d, of which x face %d, other %d]" %
    (end[i], (ll[i, 1] & 0xffffffff) % tx, (ll[i, 1] & 0xffffffff) // tx, (ll[i, 1] >> 32) & 0xffff, (ll[i, 0] - t0) / 100.0, dur[i], ll[i, 1] >> 48, mid[i],
     cnts[i] & 0x7ff, (cnts[i] >> 11) & 0x7ff, (cnts[i] >> 22) & 0x7ff, (cnts[i] >> 33) & 0x7ff) for i in late))
has = (ll[:, 1] >> 48) > 0
print("duration of the waves' last geometry tile, us: p10 %.0f median %.0f p90 %.0f p99 %.0f max %.0f; by entries:" % tuple(np.percentile(dur[has], [10, 50, 90, 99, 100])),
      "; ".join("%d-%d: median %.0f max %.0f (%d)" % (a, b, np.median(dur[has & (e >= a) & (e <= b)]), dur[has & (e >= a) & (e <= b)].max(), (has & (e >= a) & (e <= b)).sum())
                for e in [(ll[:, 1] >> 32) & 0xffff] for a, b in [(1, 4), (5, 8), (9, 16), (17, 32), (33, 64)] if (has & (e >= a) & (e <= b)).any()))
slow = np.argsort(-np.where(has, dur, 0))[:10]
print("slowest: " + "; ".join("(%d, %d) %d entries %.0f us" % ((ll[i, 1] & 0xffffffff) % tx, (ll[i, 1] & 0xffffffff) // tx, (ll[i, 1] >> 32) & 0xffff, dur[i]) for i in slow))
ng = ll[:, 1] >> 48
print("geometry tiles per wave: histogram", np.bincount(ng)[:12].tolist(), "; duration of a wave's last geometry tile until its end (upper bound) us: median %.0f p90 %.0f max %.0f"
      % tuple(np.percentile((end - (ll[:, 0] - t0) / 100.0)[ng > 0], [50, 90, 100])))
edges = np.arange(0, end.max() + 25.0, 25.0)
print("last frame, wave ends per 25 us:", np.histogram(end, edges)[0].tolist())
print("last frame, heavy phase ends per 25 us (waves):", np.histogram(mid, edges)[0].tolist())
L.crt_debug_tiled_stats(0, None)
print("raw light log rows:", ll[:4].tolist())
