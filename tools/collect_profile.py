#!/usr/bin/env python3
"""After tools/profile_bench.sh <tag>: copy the kernel-trace stats and condense the PMC passes of the two dominant
kernels into profiles/<dest>/ (run on the GPU box; the files come back through gpurun_out/).
usage: tools/collect_profile.py <tag> <dest-dir-under-gpurun_out> [algorithmic bytes of the shading kernel]"""
import csv
import glob
import os
import shutil
import subprocess
import sys

tag, dest = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "gpurun_out", "prof_" + tag)
out = os.path.join(root, "gpurun_out", dest)
os.makedirs(out, exist_ok=True)
stats = glob.glob(os.path.join(prof, "trace", "*", "*kernel_stats.csv"))
if not stats:
    sys.exit("no kernel_stats.csv under %s/trace" % prof)
shutil.copy(stats[0], os.path.join(out, "kernel_stats.csv"))
avg = {}
for r in csv.DictReader(open(stats[0])):
    avg[r["Name"]] = float(r["AverageNs"]) * 1e-9
for key, name in (("shade_tiles_kernel", "shade"), ("bin_tri_kernel", "bin_walk"), ("cls_prepass_kernel", "prepass")):
    match = [k for k in avg if key in k]
    if match:
        cmd = [sys.executable, os.path.join(root, "tools", "summarise_pmc.py"), prof, key, os.path.join(out, "pmc_summary_%s.json" % name)]
        if name == "shade" and len(sys.argv) > 3:
            cmd.append(sys.argv[3])                      # algorithmic bytes per launch of the shading kernel (bench.py's figure)
        subprocess.call(cmd)
for f in ("bench_trace.json",):
    if os.path.exists(os.path.join(prof, f)):
        shutil.copy(os.path.join(prof, f), os.path.join(out, "bench_under_rocprof.json"))
print(open(os.path.join(out, "kernel_stats.csv")).read())
