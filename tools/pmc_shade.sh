#!/bin/bash
# usage: tools/pmc_shade.sh <variant> <tag>   (GPU box) — SQ counters of the tiled kernels
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/pmc_$2
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/a -- python3 $REPO/bench.py --steps 1 --warmup 0 --clock-warm-s 0 --no-cpu-baseline --variant $1 > /dev/null 2> $OUT/a.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -- python3 $REPO/bench.py --steps 1 --warmup 0 --clock-warm-s 0 --no-cpu-baseline --variant $1 > /dev/null 2> $OUT/b.err
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(float)
for f in glob.glob("$OUT/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','')
        if 'tiles' in k: agg[(k,r['Counter_Name'])]+=float(r['Counter_Value'])
for k,v in sorted(agg.items()): print(k[0][:40],k[1],"%.4g"%v)
PY
