#!/bin/bash
# A/B of two environments on ONE GPU box, tuning build: tools/ab_env.sh "<VAR=val ...>" "<VAR=val ...>" <script.py> [args]   (three alternations)
make -C cudaraytracer_amd/csrc tuning > /dev/null 2>&1 || { echo build failed; exit 1; }
A=$1; B=$2; shift 2
for rep in 1 2 3; do
  echo -n "A [$A]: "; env $A timeout -k 10 200 python "$@" 2>&1 | grep -v amdgpu | tail -1
  echo -n "B [$B]: "; env $B timeout -k 10 200 python "$@" 2>&1 | grep -v amdgpu | tail -1
done
