#!/bin/bash
# WRITE_SIZE of the bench frame's kernels, one pass on its own (GPU box): tools/pmc_write_only.sh <tag>
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/pmc_write_$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -- python3 $REPO/bench.py --steps 1 --warmup 0 --clock-warm-s 0 --no-cpu-baseline --no-extras > $OUT/bench.json 2> $OUT/err.txt
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(float); n = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/w/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:70]
        acc[k] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in sorted(acc):
    if "crt::" in k: print("%-72s WRITE_SIZE %.1f MB per launch (%d launches)" % (k, acc[k] / len(n[k]) * 1024 / 1e6, len(n[k])))
PY
