#!/bin/bash
# Round 4: kernel-trace + PMC evidence for the f1/f2 kernels (run through gpurun): tools/r4_profile_f.sh
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"
tools/profile_any.sh r4_shade tools/f_workload.py shade 4 3 && \
tools/profile_any.sh r4_bvh_tiled tools/f_workload.py bvh_tiled 64 5 && \
for w in shade:render_generic bvh_tiled:shade_tiles_kernel; do
  python3 tools/summarise_pmc.py gpurun_out/prof_r4_${w%%:*} ${w##*:} gpurun_out/prof_r4_${w%%:*}/pmc_summary.json > gpurun_out/prof_r4_${w%%:*}/pmc_summary.txt || exit 1
  cp gpurun_out/prof_r4_${w%%:*}/trace/*/*kernel_stats.csv gpurun_out/prof_r4_${w%%:*}/kernel_stats.csv
done
