#!/bin/bash
# Round 4, second evidence set (far pairs, leaf-box chain, binning), part 1: GPU suite, bench line x2, configuration and strip times.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
O=gpurun_out/r4b; mkdir -p $O
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" | tee -a $O/gpu_tests.log; tail -2 $O/gpu_tests.log
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python bench.py --no-cpu-baseline --no-extras > $O/bench_second_run.json 2>> $O/bench.err
python tools/config_times.py > $O/config_times.txt 2>&1; tail -4 $O/config_times.txt
for f in C3 C4 C5; do for wd in list bvh; do python tools/strip_time.py $f 1,2,4,8 0 $wd 2>&1 | grep -v amdgpu >> $O/strip_time_$f.txt; done; tail -2 $O/strip_time_$f.txt | cut -c1-220; done
python tools/bvh_line.py C3B > $O/bvh_line_C3B.txt 2>&1; tail -1 $O/bvh_line_C3B.txt
