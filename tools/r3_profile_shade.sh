#!/bin/bash
# kernel-trace + PMC evidence for render_generic_kernel on the shade workload, plus lane occupancy (run through gpurun)
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"
tools/profile_any.sh r3z_shade tools/f_workload.py shade 4 3 && \
python3 tools/summarise_pmc.py gpurun_out/prof_r3z_shade render_generic gpurun_out/prof_r3z_shade/pmc_summary.json > gpurun_out/prof_r3z_shade/pmc_summary.txt && \
cp gpurun_out/prof_r3z_shade/trace/*/*kernel_stats.csv gpurun_out/prof_r3z_shade/kernel_stats.csv && \
tools/pmc_lanes.sh shade_final tools/f_workload.py shade 4 3 > gpurun_out/prof_r3z_shade/lanes.txt && \
python3 tools/shade_time.py 4 8 2>&1 | grep -v amdgpu > gpurun_out/prof_r3z_shade/shade_time.txt
