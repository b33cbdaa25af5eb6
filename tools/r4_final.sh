#!/bin/bash
# Round 4: the evidence set of the final kernels in one gpurun call: GPU suite, bench line, rocprofv3 trace + PMC passes of the
# bench command and of the f-row workloads, configuration times (strip times: tools/strip_time.py, profiles/r4/strip_time_*.txt).  Summaries land under gpurun_out/r4_final/.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"
O=gpurun_out/r4_final; mkdir -p $O
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" | tee -a $O/gpu_tests.log; tail -2 $O/gpu_tests.log
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python bench.py --no-cpu-baseline --no-extras > $O/bench_second_run.json 2>> $O/bench.err
bash tools/profile_bench.sh r4 > $O/profile_bench.log 2>&1 && bash tools/pmc_extra.sh r4 >> $O/profile_bench.log 2>&1 && python3 tools/collect_profile.py r4 r4_final/tiled 138391584 > $O/collect.log 2>&1; echo "profile rc=$?"
bash tools/r4_profile_f.sh > $O/profile_f.log 2>&1; echo "profile_f rc=$?"
mkdir -p $O/f && for w in shade bvh_tiled; do cp gpurun_out/prof_r4_$w/pmc_summary.json $O/f/pmc_summary_$w.json; cp gpurun_out/prof_r4_$w/kernel_stats.csv $O/f/kernel_stats_$w.csv; done
python tools/config_times.py > $O/config_times.txt 2>&1
python tools/shade_time.py > $O/shade_time.txt 2>&1
tail -3 $O/config_times.txt; tail -3 $O/shade_time.txt
