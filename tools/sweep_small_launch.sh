#!/bin/bash
# Small launches (one rank's strips of an 8-way split): waves per shared tile x heavy threshold (tuning build, GPU box)
FRAME=${1:-C3}
make -C cudaraytracer_amd/csrc tuning > /dev/null 2>&1 || { echo build failed; exit 1; }
for g in 8 4; do for h in 1 4 8 16; do
  echo "CRT_SHADE_GROUP=$g CRT_HEAVY_MIN=$h"
  CRT_SHADE_GROUP=$g CRT_HEAVY_MIN=$h python tools/strip_time.py $FRAME 8 2>/dev/null | sed 's/| ranks.*//'
done; done
make -C cudaraytracer_amd/csrc -B ../libcrt_hip.so > /dev/null 2>&1
