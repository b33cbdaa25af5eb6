#!/bin/bash
# Split factors of heavy tiles (shade_tiles_kernel sub-items) and the light tiles' group size, tuning build on the GPU box:
#   bash tools/sweep_split.sh [frame=C3] [worlds=1,8] [configs...] > gpurun_out/sweep_split.txt
# Nine digits = log2 of the slabs per weight class (> 768, 512, 384, 256, 192, 128, 96, 64, heavy_min).
FRAME=${1:-C3}; shift
WORLDS=${1:-1,8}; shift
CFGS=${@:-000000000 222222100 222222110 222222210 222222211 333333221 000000000}
make -C cudaraytracer_amd/csrc tuning > /dev/null 2>&1 || { echo build failed; exit 1; }
for cfg in $CFGS; do
  echo "CRT_SPLIT=$cfg"
  CRT_SPLIT=$cfg python tools/strip_time.py $FRAME $WORLDS 2>/dev/null | sed 's/| ranks.*//'
done
echo "CRT_LIGHT_GROUP=4 CRT_SPLIT=000000000 (round 2's dealing)"
CRT_LIGHT_GROUP=4 CRT_SPLIT=000000000 python tools/strip_time.py $FRAME $WORLDS 2>/dev/null | sed 's/| ranks.*//'
make -C cudaraytracer_amd/csrc -B ../libcrt_hip.so > /dev/null 2>&1
