#!/bin/bash
# A/B of the BVH tiled mode's register budget and chain batch (builds on the GPU box).
for cfg in "-DCRT_BVH_OCC=6 -DCRT_CHAIN_BATCH=4" "-DCRT_BVH_OCC=4 -DCRT_CHAIN_BATCH=4" "-DCRT_BVH_OCC=4 -DCRT_CHAIN_BATCH=8" "-DCRT_BVH_OCC=6 -DCRT_CHAIN_BATCH=4"; do
  make -C cudaraytracer_amd/csrc -B EXTRA="$cfg" ../libcrt_hip.so > /dev/null 2>&1 || { echo build failed; exit 1; }
  echo "$cfg: $(python tools/bvh_time.py C3 2>/dev/null | grep 'BVH  world')"
  echo "$cfg: $(python tools/bvh_time.py C5 2>/dev/null | grep 'BVH  world')"
done
make -C cudaraytracer_amd/csrc -B ../libcrt_hip.so > /dev/null 2>&1
