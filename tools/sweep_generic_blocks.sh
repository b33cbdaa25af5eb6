#!/bin/bash
# shade on the C3 frame against the number of persistent blocks of render_generic_kernel (tuning build, GPU box):
# does the time follow 1/waves (latency bound) or stay (throughput bound)?
make -C cudaraytracer_amd/csrc tuning > /dev/null 2>&1 || { echo build failed; exit 1; }
for b in 256 512 768 1024 2048; do
  echo -n "blocks $b: "; CRT_GENERIC_BLOCKS=$b timeout -k 10 200 python tools/f_workload.py shade 4 3 2>&1 | grep -v amdgpu | tail -1
done
