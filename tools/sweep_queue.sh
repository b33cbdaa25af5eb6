#!/bin/bash
# shade on the C3 frame against the order of render_generic_kernel's pixel queue (tuning build, ONE GPU box):
# chunk size (log2) x multiplier of the scatter (1 = in order)
make -C cudaraytracer_amd/csrc tuning > /dev/null 2>&1 || { echo build failed; exit 1; }
for mul in 1 1999 7919 104729; do for lg in 5 4 3 2 1 0; do
  echo -n "mul $mul chunk 2^$lg: "; CRT_QUEUE_MUL=$mul CRT_QUEUE_CHUNK_LOG=$lg timeout -k 10 200 python tools/f_workload.py shade 4 5 2>&1 | grep -v amdgpu | tail -1
done; done
