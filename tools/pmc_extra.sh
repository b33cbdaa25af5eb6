#!/bin/bash
# Extra PMC passes for the shading kernel (run through gpurun): where the wave time goes
# (parked / issue-stalled / active) and what the LDS does.   tools/pmc_extra.sh <tag>
set -o pipefail
TAG=${1:-r2}; shift
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d "$OUT/pmc_wait" -- python3 "$REPO/bench.py" --steps 1 --warmup 0 --clock-warm-s 0 --no-cpu-baseline --no-extras "$@" > "$OUT/bench_pmc_wait.json" 2> "$OUT/pmc_wait.err" || { tail -5 "$OUT/pmc_wait.err"; exit 1; }
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH --output-format csv -d "$OUT/pmc_lds" -- python3 "$REPO/bench.py" --steps 1 --warmup 0 --clock-warm-s 0 --no-cpu-baseline --no-extras "$@" > "$OUT/bench_pmc_lds.json" 2> "$OUT/pmc_lds.err" || { tail -5 "$OUT/pmc_lds.err"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for sub in ("pmc_wait", "pmc_lds"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
    for f in glob.glob("%s/%s/*/*counter_collection.csv" % (out, sub)):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
    for k in acc:
        if "shade_tiles" in k or "bin_tri" in k:
            print(sub, k, {c: acc[k][c] / max(n[k][c], 1) for c in sorted(acc[k])})
PY
