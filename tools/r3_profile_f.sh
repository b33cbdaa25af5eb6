#!/bin/bash
# Round 3: kernel-trace + PMC evidence for the f1/f2 kernels (run through gpurun): tools/r3_profile_f.sh <suffix>
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
S=${1:-a}
cd "$REPO"
tools/profile_any.sh r3${S}_shade tools/f_workload.py shade 4 3 && \
tools/profile_any.sh r3${S}_bvh_walk tools/f_workload.py bvh_walk 4 3 && \
tools/profile_any.sh r3${S}_bvh_tiled tools/f_workload.py bvh_tiled 64 5 && \
for w in shade:render_generic bvh_walk:render_bvh_kernel bvh_tiled:shade_tiles_kernel; do
  python3 tools/summarise_pmc.py gpurun_out/prof_r3${S}_${w%%:*} ${w##*:} gpurun_out/prof_r3${S}_${w%%:*}/pmc_summary.json > gpurun_out/prof_r3${S}_${w%%:*}/pmc_summary.txt || exit 1
  cp gpurun_out/prof_r3${S}_${w%%:*}/trace/*/*kernel_stats.csv gpurun_out/prof_r3${S}_${w%%:*}/kernel_stats.csv
done
