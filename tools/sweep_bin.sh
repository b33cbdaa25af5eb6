#!/bin/bash
# Sweep of the binning slice count (tuning build only: make -C cudaraytracer_amd/csrc tuning).
# Usage on the GPU box: bash tools/sweep_bin.sh > gpurun_out/sweep_bin.txt
for w in 8192 16384 32768 65536 131072 262144; do
  for few in 8; do
    CRT_BIN_WAVES=$w CRT_BIN_FEW=$few python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('CRT_BIN_WAVES=$w few=$few ms_per_step %.4f binning %.4f shade %.4f' % (d['ms_per_step'], d['kernels']['binning']['ms'], d['kernels']['shade_tiles_kernel']['ms']))"
  done
done
