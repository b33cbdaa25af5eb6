#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun):
#   tools/profile_bench.sh <tag> [extra bench.py args]
# Writes gpurun_out/prof_<tag>/{trace,pmc_fetch,pmc_write,pmc_sq}/... ; copy the summaries
# you want judged into profiles/.
set -o pipefail
TAG=${1:-r1}; shift
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" --steps 20 --warmup 5 --clock-warm-s 0.3 --no-cpu-baseline --no-extras "$@" > "$OUT/bench_trace.json" 2> "$OUT/trace.err" || { tail -5 "$OUT/trace.err"; exit 1; }
# PMC passes on their own (never together with --stats / trace domains)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$REPO/bench.py" --steps 1 --warmup 0 --clock-warm-s 0 --no-cpu-baseline --no-extras "$@" > "$OUT/bench_pmc_fetch.json" 2> "$OUT/pmc_fetch.err" || { tail -5 "$OUT/pmc_fetch.err"; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$REPO/bench.py" --steps 1 --warmup 0 --clock-warm-s 0 --no-cpu-baseline --no-extras "$@" > "$OUT/bench_pmc_write.json" 2> "$OUT/pmc_write.err" || { tail -5 "$OUT/pmc_write.err"; exit 1; }
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq" -- python3 "$REPO/bench.py" --steps 1 --warmup 0 --clock-warm-s 0 --no-cpu-baseline --no-extras "$@" > "$OUT/bench_pmc_sq.json" 2> "$OUT/pmc_sq.err" || { tail -5 "$OUT/pmc_sq.err"; exit 1; }
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/pmc_lanes" -- python3 "$REPO/bench.py" --steps 1 --warmup 0 --clock-warm-s 0 --no-cpu-baseline --no-extras "$@" > "$OUT/bench_pmc_lanes.json" 2> "$OUT/pmc_lanes.err" || { tail -5 "$OUT/pmc_lanes.err"; exit 1; }
find "$OUT" -name "*.csv" | head -30
