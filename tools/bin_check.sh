#!/bin/bash
# Binning changes: the audit and parity tests that cover the lists, then the times (A/B against libcrt_hip_A.so when present).
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
O=gpurun_out/binchk; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py tests/test_bvh.py -m gpu -x -q > $O/tests.log 2>&1; rc=$?; tail -3 $O/tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "c3_full or c4 or c5" > $O/tests_cfg.log 2>&1; rc=$?; tail -3 $O/tests_cfg.log; [ $rc -eq 0 ] || exit $rc
cp cudaraytracer_amd/libcrt_hip.so cudaraytracer_amd/libcrt_hip_N.so
if [ -f cudaraytracer_amd/libcrt_hip_A.so ]; then bash tools/ab_many.sh "A N" tools/list_line.py C3 C5 | tee $O/ab.txt; fi
python tools/strip_time.py C5 8 0 list 2>&1 | tail -1 | tee $O/strip_C5.txt
