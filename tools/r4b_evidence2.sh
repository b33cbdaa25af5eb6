#!/bin/bash
# Round 4, second evidence set, part 2: rocprofv3 trace + PMC passes of the bench command and of the f-row workloads.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
O=gpurun_out/r4b; mkdir -p $O
bash tools/profile_bench.sh r4b > $O/profile_bench.log 2>&1 && bash tools/pmc_extra.sh r4b >> $O/profile_bench.log 2>&1 && python3 tools/collect_profile.py r4b r4b/tiled 138391584 > $O/collect.log 2>&1; echo "profile rc=$?"
tools/profile_any.sh r4b_shade tools/f_workload.py shade 4 3 > $O/profile_f.log 2>&1 && tools/profile_any.sh r4b_bvh_tiled tools/f_workload.py bvh_tiled 64 5 >> $O/profile_f.log 2>&1
mkdir -p $O/f
for w in shade:render_generic bvh_tiled:shade_tiles_kernel; do
  python3 tools/summarise_pmc.py gpurun_out/prof_r4b_${w%%:*} ${w##*:} $O/f/pmc_summary_${w%%:*}.json > $O/f/pmc_summary_${w%%:*}.txt || echo "summarise ${w%%:*} failed"
  cp gpurun_out/prof_r4b_${w%%:*}/trace/*/*kernel_stats.csv $O/f/kernel_stats_${w%%:*}.csv
done
ls $O $O/f $O/tiled 2>/dev/null | head -40
