#!/usr/bin/env python3
"""Condense the rocprofv3 --pmc CSVs written by tools/profile_bench.sh into one JSON.
usage: tools/summarise_pmc.py gpurun_out/prof_<tag> <kernel-name-substring> <kernel_seconds> <out.json>"""
import collections
import csv
import glob
import json
import sys

prof, kname, ksec, outp = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]
out, disp = {}, {}
for d in ("pmc_fetch", "pmc_write", "pmc_sq"):
    for f in glob.glob("%s/%s/*/*_counter_collection.csv" % (prof, d)):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            out.setdefault(k, collections.defaultdict(float))[r["Counter_Name"]] += float(r["Counter_Value"])
            disp.setdefault((k, r["Counter_Name"]), set()).add(r["Dispatch_Id"])
# per-launch averages: a counter row exists per dispatch (and per XCD/SE slice), so divide by dispatches
out = {k: {c: v / max(1, len(disp[(k, c)])) for c, v in cs.items()} for k, cs in out.items()}
key = [k for k in out if kname in k][0]
r = out[key]
fetch_b, write_b = r.get("FETCH_SIZE", 0) * 1024, r.get("WRITE_SIZE", 0) * 1024
summary = {
    "command": "rocprofv3 --pmc <counters> --output-format csv -- python3 bench.py --steps 1 --warmup 0 "
               "--no-cpu-baseline (one pass per counter group; tools/profile_bench.sh)",
    "kernel": key, "kernel_seconds": ksec, "per_launch_counters": r,
    "derived": {
        "FETCH_SIZE_bytes": fetch_b,
        "FETCH_SIZE_bytes_x2_gfx950_correction": 2 * fetch_b,   # MI355X_MICROARCH.md §HBM
        "WRITE_SIZE_bytes": write_b,
        "hbm_traffic_bytes_corrected": 2 * fetch_b + write_b,
        "effective_clock_GHz": r.get("GRBM_GUI_ACTIVE", 0) / 8 / ksec / 1e9,
        "valu_wave_insts_per_clk_per_simd": r.get("SQ_INSTS_VALU", 0) / ksec / 1024 / 2.4e9,
    },
    "all_kernels": out,
}
json.dump(summary, open(outp, "w"), indent=1)
print(json.dumps(summary["derived"], indent=1))
