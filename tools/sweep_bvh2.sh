#!/bin/bash
# Timing experiment: what the chain-of-boxes test costs in the BVH tiled mode (CRT_DBG_NOCHAIN skips it: wrong pixels).
for cfg in "" "-DCRT_DBG_NOCHAIN"; do
  make -C cudaraytracer_amd/csrc -B EXTRA="$cfg" > /dev/null 2>&1 || { echo build failed; exit 1; }
  echo "[$cfg]: $(python tools/bvh_time.py C3 2>/dev/null | grep 'BVH  world')"
done
make -C cudaraytracer_amd/csrc -B > /dev/null 2>&1
