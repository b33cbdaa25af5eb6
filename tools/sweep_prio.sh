#!/bin/bash
# A/B of wave issue priorities in the shading kernel, builds on the GPU box.
# bash tools/sweep_prio.sh > gpurun_out/sweep_prio.txt
for cfg in "0 0 0 0" "1 3 1 0" "1 3 2 0" "1 3 2 1" "1 2 1 0" "1 3 3 0" "1 1 0 0" "1 3 0 0" "1 3 1 0" "0 0 0 0"; do
  set -- $cfg
  make -C cudaraytracer_amd/csrc -B EXTRA="-DCRT_PRIO=$1 -DCRT_PRIO_HEAVY=$2 -DCRT_PRIO_GEOM=$3 -DCRT_PRIO_SKY=$4" ../libcrt_hip.so > /dev/null 2>&1 || { echo build failed; exit 1; }
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('PRIO=$1 heavy $2 geom $3 sky $4: ms_per_step %.4f binning %.4f shade %.4f filling %.4f bvh %.4f' % (d['ms_per_step'], d['kernels']['binning']['ms'], d['kernels']['shade_tiles_kernel']['ms'], d['screen_filling_variant']['ms_per_frame'], d['bvh_world']['ms_per_frame']))"
done
make -C cudaraytracer_amd/csrc -B ../libcrt_hip.so > /dev/null 2>&1
python tools/strip_time.py C3 8 2>/dev/null | tail -1
