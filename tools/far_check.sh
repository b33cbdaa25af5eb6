#!/bin/bash
# Far pairs (round 4): parity of the new path, then the times it is meant to change.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
O=gpurun_out/far; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pole_tiles or row_slabs or adversarial" > $O/tests.log 2>&1; rc=$?; tail -3 $O/tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "c5 or c3_full" > $O/tests_cfg.log 2>&1; rc=$?; tail -3 $O/tests_cfg.log; [ $rc -eq 0 ] || exit $rc
python tools/config_times.py > $O/config_times.txt 2>&1; tail -6 $O/config_times.txt
python tools/strip_time.py C5 1,8 0 list > $O/strip_time_C5.txt 2>&1; tail -4 $O/strip_time_C5.txt
python bench.py --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err; cat $O/bench.json | cut -c1-300
