#!/bin/bash
# tuning build on the GPU box: binning time of several frames per value of CRT_BIN_WAVES (slices of the binning walk)
make -C cudaraytracer_amd/csrc -B EXTRA=-DCRT_TUNING > /dev/null 2>&1 || exit 1
for v in 16384 4096 2048 1024 256; do
  for f in C3 C5; do echo -n "CRT_BIN_WAVES=$v "; CRT_BIN_WAVES=$v python tools/list_line.py $f 2>&1 | tail -1; done
  echo -n "CRT_BIN_WAVES=$v "; CRT_BIN_WAVES=$v python tools/config_times.py 2>&1 | grep "^C1\|^C2" | tr '\n' '|'; echo
done
make -C cudaraytracer_amd/csrc -B > /dev/null 2>&1
