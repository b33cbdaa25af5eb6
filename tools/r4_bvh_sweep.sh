# round 4: what bounds the BVH tiled mode — light-tile grouping and the shared-tile threshold (tuning build, one box)
mkdir -p gpurun_out/r4
make -C cudaraytracer_amd/csrc tuning > /dev/null 2>&1 || { echo tuning build failed; exit 1; }
{
for g in 4 2 1; do for h in 32 24 16 8; do
  echo -n "CRT_LIGHT_GROUP=$g CRT_HEAVY_MIN=$h: "; env CRT_LIGHT_GROUP=$g CRT_HEAVY_MIN=$h python tools/bvh_line.py C3 2>/dev/null | tail -1
done; done
echo -n "C5 default: "; python tools/bvh_line.py C5 2>/dev/null | tail -1
echo -n "C5 CRT_LIGHT_GROUP=1: "; env CRT_LIGHT_GROUP=1 python tools/bvh_line.py C5 2>/dev/null | tail -1
echo -n "C5 CRT_LIGHT_GROUP=1 CRT_HEAVY_MIN=16: "; env CRT_LIGHT_GROUP=1 CRT_HEAVY_MIN=16 python tools/bvh_line.py C5 2>/dev/null | tail -1
} > gpurun_out/r4/bvh_sweep_group_heavy.txt 2>&1
cat gpurun_out/r4/bvh_sweep_group_heavy.txt
make -C cudaraytracer_amd/csrc -B ../libcrt_hip.so > /dev/null 2>&1
