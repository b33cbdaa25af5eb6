#!/bin/bash
# shade on the C3 frame against the number of resident blocks of render_generic_kernel (tuning build, ONE GPU box)
make -C cudaraytracer_amd/csrc tuning > /dev/null 2>&1
for b in 768 512 256 768; do echo -n "blocks $b: "; CRT_GENERIC_BLOCKS=$b timeout -k 10 200 python tools/f_workload.py shade 4 5 2>&1 | grep -v amdgpu | tail -1; done
