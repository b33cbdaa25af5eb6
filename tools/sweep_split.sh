#!/bin/bash
# Split factors of heavy tiles (shade_tiles_kernel sub-items), tuning build on the GPU box:
#   bash tools/sweep_split.sh [frame=C3] [configs...] > gpurun_out/sweep_split.txt
# Nine digits = log2 of the slabs per weight class (> 768, 512, 384, 256, 192, 128, 96, 64, heavy_min).
FRAME=${1:-C3}; shift
CFGS=${@:-000000000 111111000 222222100 222222210 222222221 222222222 333333210 333333321 000000000}
make -C cudaraytracer_amd/csrc tuning > /dev/null 2>&1 || { echo build failed; exit 1; }
for cfg in $CFGS; do
  echo "CRT_SPLIT=$cfg"
  CRT_SPLIT=$cfg python tools/strip_time.py $FRAME 1,8 2>/dev/null | sed 's/| ranks.*//'
done
make -C cudaraytracer_amd/csrc -B ../libcrt_hip.so > /dev/null 2>&1
