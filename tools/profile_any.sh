#!/bin/bash
# rocprofv3 evidence for ANY program of this repository on the GPU box (run through gpurun):
#   tools/profile_any.sh <tag> <script.py> [args]        e.g.  tools/profile_any.sh r3_shade tools/f_workload.py shade
# One kernel-trace pass and four PMC passes, each on its own (never --pmc together with --stats / trace domains);
# the program goes directly after `--`.  Writes gpurun_out/prof_<tag>/{trace,pmc_fetch,pmc_write,pmc_sq,pmc_wait}/ ;
# tools/summarise_pmc.py condenses them; copy what is to be judged into profiles/.
set -o pipefail
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
PROG=$REPO/$1; shift
cd /tmp && export TMPDIR=/tmp
run() {   # run <subdir> <rocprof args...> : the program's stdout goes to <subdir>.out
    local sub=$1; shift
    rocprofv3 "$@" --output-format csv -d "$OUT/$sub" -- python3 "$PROG" "${ARGS[@]}" > "$OUT/$sub.out" 2> "$OUT/$sub.err" || { tail -5 "$OUT/$sub.err"; exit 1; }
    echo "$sub: $(tail -1 "$OUT/$sub.out")"
}
ARGS=("$@")
run trace --kernel-trace --stats
run pmc_fetch --pmc FETCH_SIZE
run pmc_write --pmc WRITE_SIZE
run pmc_sq --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
run pmc_wait --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
