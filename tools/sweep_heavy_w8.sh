#!/bin/bash
# Heavy-tile threshold for SMALL launches (one rank's strips of an 8-way split), tuning build on the GPU box:
#   bash tools/sweep_heavy_w8.sh [frame]
FRAME=${1:-C3}
make -C cudaraytracer_amd/csrc tuning > /dev/null 2>&1 || { echo build failed; exit 1; }
for h in 8 16 24 32 48 64; do
  echo "CRT_HEAVY_MIN=$h"
  CRT_HEAVY_MIN=$h python tools/strip_time.py $FRAME 8 2>/dev/null | sed 's/| ranks.*//'
done
make -C cudaraytracer_amd/csrc -B ../libcrt_hip.so > /dev/null 2>&1
