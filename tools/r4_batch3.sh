mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4/gpu_tests_3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/gpu_tests_3.log
tail -4 gpurun_out/r4/gpu_tests_3.log
grep -q "rc=0" gpurun_out/r4/gpu_tests_3.log || exit 1
timeout -k 10 420 python tools/fuzz_bvh.py 400 > gpurun_out/r4/fuzz_bvh.txt 2>&1; tail -3 gpurun_out/r4/fuzz_bvh.txt
