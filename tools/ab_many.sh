#!/bin/bash
# A/B/... of several PREBUILT libraries cudaraytracer_amd/libcrt_hip_<tag>.so on ONE GPU box, three alternations:
#   tools/ab_many.sh "tagA tagB ..." <script.py> [args]
TAGS=$1; shift
cp cudaraytracer_amd/libcrt_hip.so /tmp/libcrt_hip_keep.so
for rep in 1 2 3; do
  for v in $TAGS; do
    cp cudaraytracer_amd/libcrt_hip_$v.so cudaraytracer_amd/libcrt_hip.so
    echo -n "$v: "; timeout -k 10 200 python "$@" 2>&1 | grep -v amdgpu | tail -1
  done
done
cp /tmp/libcrt_hip_keep.so cudaraytracer_amd/libcrt_hip.so
