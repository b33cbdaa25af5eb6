#!/bin/bash
# BVH world through the tiled kernels on the C3 frame against the launch parameters tuned for the list world: heavy-tile threshold,
# light-queue group size, row slabs of heavy tiles (tuning build, ONE GPU box)
make -C cudaraytracer_amd/csrc tuning > /dev/null 2>&1 || { echo build failed; exit 1; }
run() { echo -n "$1: "; env $1 timeout -k 10 100 python tools/f_workload.py bvh_tiled 64 10 2>&1 | grep -v amdgpu | tail -1 | sed 's/.*triangles: //'; }
run "CRT_NONE=1"
for h in 8 16 24 48 64; do run "CRT_HEAVY_MIN=$h"; done
for g in 1 2 4; do run "CRT_LIGHT_GROUP=$g"; done
for s in 000000000 200000000 220000000 222000000 111000000; do run "CRT_SPLIT=$s"; done
run "CRT_NONE=1"
