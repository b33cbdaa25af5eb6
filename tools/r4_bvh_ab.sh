mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_bvh.py tests/test_shade.py -m gpu -x -q > gpurun_out/r4/bvh_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r4/bvh_tests.log
tail -3 gpurun_out/r4/bvh_tests.log
grep -q "rc=0" gpurun_out/r4/bvh_tests.log || exit 1
bash tools/ab_build.sh "-DCRT_BVH_OCC=4" "-DCRT_BVH_OCC=6" tools/bvh_line.py C3 C5 > gpurun_out/r4/bvh_occ_ab2.txt 2>&1
cat gpurun_out/r4/bvh_occ_ab2.txt
