#!/bin/bash
# A/B of two builds on ONE GPU box: [AB_PATTERN=regex] tools/ab_build.sh "<EXTRA flags A>" "<EXTRA flags B>" <script.py> [args]   (three alternations;
# prints the last output line that matches AB_PATTERN)
A=$1; B=$2; shift 2
for rep in 1 2 3; do
  for v in A B; do
    if [ $v = A ]; then F="$A"; else F="$B"; fi
    if [ $rep = 1 ]; then make -C cudaraytracer_amd/csrc -B EXTRA="$F" OUT=../libcrt_hip_$v.so ../libcrt_hip_$v.so > /dev/null 2>&1 || { echo build $v failed; exit 1; }; fi
    cp cudaraytracer_amd/libcrt_hip_$v.so cudaraytracer_amd/libcrt_hip.so
    echo -n "$v [$F]: "; timeout -k 10 200 python "$@" 2>&1 | grep -v amdgpu | grep -E "${AB_PATTERN:-.}" | tail -1
  done
done
rm -f cudaraytracer_amd/libcrt_hip_A.so cudaraytracer_amd/libcrt_hip_B.so
