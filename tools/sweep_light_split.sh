#!/bin/bash
# Small launches (one rank's strips of an 8-way split): light tiles with geometry as row slabs x heavy slabs (tuning build, GPU box)
FRAME=${1:-C3}
make -C cudaraytracer_amd/csrc tuning > /dev/null 2>&1 || { echo build failed; exit 1; }
for ls in 0 1 2 3; do for hs in 200000000 222222100 222222210; do
  echo "CRT_LIGHT_SPLIT=$ls CRT_SPLIT=$hs"
  CRT_LIGHT_SPLIT=$ls CRT_SPLIT=$hs python tools/strip_time.py $FRAME 8 2>/dev/null | sed 's/| ranks.*//'
done; done
make -C cudaraytracer_amd/csrc -B ../libcrt_hip.so > /dev/null 2>&1
