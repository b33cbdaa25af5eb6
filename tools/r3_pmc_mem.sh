#!/bin/bash
# Memory-path counters of one f-workload (GPU box): tools/r3_pmc_mem.sh <tag> <workload> [args]
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
OUT=$REPO/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_ta -- python3 $REPO/tools/f_workload.py "$@" > $OUT/pmc_ta.out 2> $OUT/pmc_ta.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/pmc_tcc -- python3 $REPO/tools/f_workload.py "$@" > $OUT/pmc_tcc.out 2> $OUT/pmc_tcc.err
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for sub in ("pmc_ta", "pmc_tcc"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for f in glob.glob("%s/%s/*/*counter_collection.csv" % (out, sub)):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:70]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k in acc:
        if "render_generic" in k or "render_bvh" in k or "shade_tiles" in k:
            print(sub, k, len(n[k]), "launches:", {c: "%.4g" % (acc[k][c] / len(n[k])) for c in sorted(acc[k])})
PY
tail -2 $OUT/pmc_ta.err $OUT/pmc_tcc.err | grep -v amdgpu
