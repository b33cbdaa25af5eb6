#!/bin/bash
# WRITE_SIZE of the brute-force kernel (same state / colour layout, no scratch) against the tiled shading kernel:
# which part of the write traffic is layout, which is scratch?   (run through gpurun)
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_wv
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/brute" -- python3 "$REPO/bench.py" --variant 1 --spp 1 --steps 1 --warmup 0 --clock-warm-s 0 --no-cpu-baseline --no-extras > "$OUT/b.json" 2> "$OUT/b.err" || { tail -5 "$OUT/b.err"; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/tiled1" -- python3 "$REPO/bench.py" --spp 1 --steps 1 --warmup 0 --clock-warm-s 0 --no-cpu-baseline --no-extras > "$OUT/t.json" 2> "$OUT/t.err" || { tail -5 "$OUT/t.err"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for sub in ("brute", "tiled1"):
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob("%s/%s/*/*counter_collection.csv" % (out, sub)):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:50]
            acc[k] += float(r["Counter_Value"]); n[k] += 1
    for k in acc:
        if "render_list" in k or "shade_tiles" in k:
            print(sub, k, "WRITE_SIZE per launch %.0f KiB-units -> %.1f MB" % (acc[k] / n[k], acc[k] / n[k] * 1024 / 1e6))
PY
