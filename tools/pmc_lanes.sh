#!/bin/bash
# Lane utilisation of a workload's kernels (GPU box): tools/pmc_lanes.sh <tag> <script.py> [args]
# One PMC pass on its own: SQ_THREAD_CYCLES_VALU (lane-cycles) against SQ_ACTIVE_INST_VALU (wave-cycles of VALU issue).
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/pmc_lanes_$1; shift; mkdir -p $OUT
PROG=$REPO/$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 180 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/w -- python3 $PROG "$@" > $OUT/out.txt 2> $OUT/err.txt || { tail -5 $OUT/err.txt; exit 1; }
tail -1 $OUT/out.txt
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/w/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in sorted(acc):
    if "crt::" not in k: continue
    a = acc[k]; l = len(n[k])
    print("%-62s launches %d  " % (k, l) + "  ".join("%s %.4g" % (c, a[c] / l) for c in sorted(a)))
    if a.get("SQ_ACTIVE_INST_VALU"): print("    lanes active per VALU cycle: %.1f of 64" % (a["SQ_THREAD_CYCLES_VALU"] / a["SQ_ACTIVE_INST_VALU"]))
PY
