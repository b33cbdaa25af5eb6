#!/bin/bash
# BVH mode of the tiled kernels: parity first, then the frame times (A/B when two prebuilt libraries are there) and the timeline of its last waves.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
O=gpurun_out/bvhchk; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_bvh.py tests/test_shade.py tests/test_ref_hit.py -m gpu -x -q > $O/tests.log 2>&1; rc=$?; tail -3 $O/tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/fuzz_bvh.py 60 > $O/fuzz.log 2>&1; rc=$?; tail -2 $O/fuzz.log; [ $rc -eq 0 ] || exit $rc
if [ -f cudaraytracer_amd/libcrt_hip_A.so ] && [ -f cudaraytracer_amd/libcrt_hip_B.so ]; then bash tools/ab_libs.sh tools/bvh_line.py C3 C3B C5 | tee $O/ab.txt; cp cudaraytracer_amd/libcrt_hip_B.so cudaraytracer_amd/libcrt_hip.so
else python tools/bvh_line.py C3 C3B C5 2>&1 | grep -v amdgpu.ids | tee $O/bvh_line.txt; fi
python tools/strip_time.py C5 1,8 0 bvh > $O/strip_time_C5_bvh.txt 2>&1; tail -2 $O/strip_time_C5_bvh.txt
python tools/tiled_stats.py 64 bvh C5 > $O/ts_bvh_c5.txt 2>&1; grep "^frame 5\|shared tiles' slices" $O/ts_bvh_c5.txt
