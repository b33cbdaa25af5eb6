#!/bin/bash
# A/B of the generic kernel's occupancy target (builds on the GPU box): bash tools/sweep_generic.sh
for occ in 0 5 6 8; do
  make -C cudaraytracer_amd/csrc -B EXTRA="-DCRT_GENERIC_OCC=$occ" ../libcrt_hip.so > /dev/null 2>&1 || { echo build failed; exit 1; }
  echo "GENERIC_OCC=$occ: $(python tools/shade_time.py 4 8 2>/dev/null | head -1)"
done
make -C cudaraytracer_amd/csrc -B ../libcrt_hip.so > /dev/null 2>&1
