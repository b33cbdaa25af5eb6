#!/bin/bash
# A/B of the shading kernel's register budget (launch-bounds occupancy 4 or 6) with and without the ray set-up
# constants in VGPRs (-DCRT_VGPR_CAM=1); tuning builds on the GPU box.  bash tools/sweep_occ.sh > gpurun_out/sweep_occ.txt
for cam in 0 1; do
  make -C cudaraytracer_amd/csrc -B EXTRA="-DCRT_TUNING -DCRT_VGPR_CAM=$cam" > /dev/null 2>&1 || { echo build failed; exit 1; }
  for occ in 4 6; do
    CRT_SHADE_OCC=$occ python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('VGPR_CAM=$cam OCC=$occ ms_per_step %.4f binning %.4f shade %.4f filling %.4f' % (d['ms_per_step'], d['kernels']['binning']['ms'], d['kernels']['shade_tiles_kernel']['ms'], d['screen_filling_variant']['ms_per_frame']))"
  done
done
