#!/bin/bash
# shade on the C3 frame against the order of render_generic_kernel's pixel queue (tuning build, GPU box)
make -C cudaraytracer_amd/csrc tuning > /dev/null 2>&1 || { echo build failed; exit 1; }
for sc in 1; do for lg in 6 4 3 2 1 0; do
  echo -n "scatter $sc chunk 2^$lg: "; CRT_QUEUE_SCATTER=$sc CRT_QUEUE_CHUNK_LOG=$lg timeout -k 10 200 python tools/f_workload.py shade 4 5 2>&1 | grep -v amdgpu | tail -1
done; done
