#!/bin/bash
# C5 (3840x2160x64, 1 M triangles): do row slabs of the two ~1 980-entry pole tiles shorten the launch?  (tuning build, GPU box)
make -C cudaraytracer_amd/csrc tuning > /dev/null 2>&1 || { echo build failed; exit 1; }
for cfg in 000000000 200000000 300000000 322000000 333333000; do
  echo "CRT_SPLIT=$cfg"
  CRT_SPLIT=$cfg python tools/strip_time.py C5 1 2>/dev/null | sed 's/| ranks.*//'
done
make -C cudaraytracer_amd/csrc -B ../libcrt_hip.so > /dev/null 2>&1
