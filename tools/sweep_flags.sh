#!/bin/bash
# Compiler-flag A/B of the whole library on the bench frame (builds on the GPU box): bash tools/sweep_flags.sh
run() {
  make -C cudaraytracer_amd/csrc -B EXTRA="$1" ../libcrt_hip.so > /tmp/build.log 2>&1 || { echo "EXTRA=$1: build failed: $(grep -m1 error /tmp/build.log)"; return; }
  for k in 1 2; do python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('EXTRA=$1: ms_per_step %.4f binning %.4f shade %.4f' % (d['ms_per_step'], d['kernels']['binning']['ms'], d['kernels']['shade_tiles_kernel']['ms']))"; done
}
run ""
run "-mllvm -amdgpu-enable-max-ilp-scheduling-strategy=1"
run "-mllvm -amdgpu-schedule-relaxed-occupancy=true"
run "-mllvm -amdgpu-schedule-metric-bias=0"
run "-mllvm -amdgpu-schedule-metric-bias=100"
run "-O2"
run "-fno-unroll-loops"
run ""
make -C cudaraytracer_amd/csrc -B ../libcrt_hip.so > /dev/null 2>&1
