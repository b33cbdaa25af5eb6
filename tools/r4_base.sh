set -e
mkdir -p gpurun_out/r4
python tools/strip_time.py C4 1,2,4,8 0 list > gpurun_out/r4/strip_time_C4_base.txt 2>&1
python tools/strip_time.py C4 1,2,4,8 0 bvh >> gpurun_out/r4/strip_time_C4_base.txt 2>&1
python tools/strip_time.py C5 1,2,4,8 0 list > gpurun_out/r4/strip_time_C5_base.txt 2>&1
python tools/strip_time.py C5 1,2,4,8 0 bvh >> gpurun_out/r4/strip_time_C5_base.txt 2>&1
python tools/strip_time.py C3 1,2,4,8 0 bvh > gpurun_out/r4/strip_time_C3_bvh_base.txt 2>&1
python bench.py > gpurun_out/r4/bench_base.json 2> gpurun_out/r4/bench_base.err
python bench.py --no-cpu-baseline --no-extras > gpurun_out/r4/bench_base2.json 2>> gpurun_out/r4/bench_base.err
python bench.py --no-cpu-baseline --no-extras > gpurun_out/r4/bench_base3.json 2>> gpurun_out/r4/bench_base.err
