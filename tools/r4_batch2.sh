# GPU batch 2 of round 4: full GPU suite, same-box A/B of round 2's tree against HEAD, bench line
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4/gpu_tests_2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/gpu_tests_2.log
tail -5 gpurun_out/r4/gpu_tests_2.log
for rep in 1 2 3; do
  (cd tools/_ab_r2 && python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r2  ms_per_step %.4f kernel_ms %.4f shade %.4f' % (d['ms_per_step'], d['kernel_ms'], d['kernels']['shade_tiles_kernel']['ms']))")
  python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('HEAD ms_per_step %.4f runs %s kernel_ms %.4f shade %.4f' % (d['ms_per_step'], ['%.4f' % x for x in d['ms_per_step_runs']], d['kernel_ms'], d['kernels']['shade_tiles_kernel']['ms']))"
done > gpurun_out/r4/ab_r2_vs_head.txt 2>&1
cat gpurun_out/r4/ab_r2_vs_head.txt
python bench.py > gpurun_out/r4/bench_2.json 2> gpurun_out/r4/bench_2.err; echo "bench rc=$?"
