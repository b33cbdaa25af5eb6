#!/bin/bash
# A/B of two PREBUILT libraries (cudaraytracer_amd/libcrt_hip_A.so, _B.so) on ONE GPU box: tools/ab_libs.sh <script.py> [args]
for rep in 1 2 3; do
  for v in A B; do
    cp cudaraytracer_amd/libcrt_hip_$v.so cudaraytracer_amd/libcrt_hip.so
    echo -n "$v: "; timeout -k 10 200 python "$@" 2>&1 | grep -v amdgpu | tail -1
  done
done
