#!/bin/bash
# usage: tools/sweep_env.sh VAR v1 v2 ...   (GPU box) — bench.py frame time per value of a tuning variable
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v python bench.py --no-cpu-baseline --steps 20 --warmup 5 | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', round(d['ms_per_step'],4), {k:round(x['ms'],4) for k,x in d['kernels'].items()})"
done
