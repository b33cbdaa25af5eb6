#!/bin/bash
# GPU box, tuning build: launch policy of mid-size launches (one rank's strips of a 2- / 4-way split: 24..96 tiles per CU), which
# the round-3 sweeps (W = 8 only) did not cover: light tiles with geometry as row slabs, heavy classes as slabs, heavy threshold.
#   tools/sweep_mid_split.sh > gpurun_out/sweep_mid_split.txt
make -C cudaraytracer_amd/csrc tuning > /dev/null 2>&1 || { echo build failed; exit 1; }
for frame in C3 C3B; do
  for ls in 0 1; do
    for sp in 200000000 222222100; do
      for hm in 24 16; do
        echo "CRT_LIGHT_SPLIT=$ls CRT_SPLIT=$sp CRT_HEAVY_MIN=$hm"
        env CRT_LIGHT_SPLIT=$ls CRT_SPLIT=$sp CRT_HEAVY_MIN=$hm timeout -k 10 200 python tools/strip_time.py $frame 2,4 2>&1 | grep -v amdgpu
      done
    done
  done
done
