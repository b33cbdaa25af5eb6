#!/bin/bash
# GPU box, tuning build: the binning's slice count (CRT_BIN_WAVES: the launch is cut into screen slices until it has this many
# wavefronts) for a full frame and for one rank's strips of an 8-way split — bin time of tools/strip_time.py.
make -C cudaraytracer_amd/csrc tuning > /dev/null 2>&1 || { echo build failed; exit 1; }
for frame in C3 C5; do
  for bw in 8192 16384 32768 65536 131072; do
    echo -n "CRT_BIN_WAVES=$bw  "
    env CRT_BIN_WAVES=$bw timeout -k 10 200 python tools/strip_time.py $frame 1,8 2>&1 | grep -v amdgpu | sed -e 's/.*world [0-9x]* //' -e 's/| ideal.*//' | tr '\n' ' '
    echo
  done
done
