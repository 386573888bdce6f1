#!/bin/bash
# Runs on the GPU box (inside gpurun):  bash profiles/final_round_r6.sh <part>
#   a = the C2 counter sets (step of 64) + lane utilisation     b = C4, C5 on one GPU (16 spp per step), mesh scene counter sets
#   c = one bench line per configuration + the driver's own command     d = generality, group probe
# Round 6's closing measurement with ONE build.  Raw output under gpurun_out/final_r06/; the distilled summaries are copied into profiles/.
set -o pipefail
OUT=gpurun_out/final_r06
mkdir -p $OUT
case "${1:-a}" in
a)
  bash profiles/run_profile.sh r06 20 > $OUT/profile.log 2>&1 && echo "profile r06 (step of 64) ok"
  bash profiles/run_lane_util.sh c2 > $OUT/lane_c2.log 2>&1 && echo "lane c2 ok"
  ;;
b)
  BENCH_ARGS="--scene $PWD/scenes/cornell_glass.txt --res 1920 1080 --depth 16" bash profiles/run_profile.sh r06_c4 4 > $OUT/profile_c4.log 2>&1 && echo "profile c4 ok"
  BATCH=16 BENCH_ARGS="--scene $PWD/scenes/spheres64.txt --res 4096 4096 --depth 8" bash profiles/run_profile.sh r06_c5 2 > $OUT/profile_c5.log 2>&1 && echo "profile c5 ok"
  BENCH_ARGS="--scene $PWD/scenes/cornell_mesh.txt" bash profiles/run_profile.sh r06_mesh 4 > $OUT/profile_mesh.log 2>&1 && echo "profile mesh ok"
  STEPS=4 bash profiles/run_lane_util.sh mesh "--scene $PWD/scenes/cornell_mesh.txt" > $OUT/lane_mesh.log 2>&1 && echo "lane mesh ok"
  STEPS=2 bash profiles/run_lane_util.sh c5 "--scene $PWD/scenes/spheres64.txt --res 4096 4096 --depth 8 --batch 16" > $OUT/lane_c5.log 2>&1 && echo "lane c5 ok"
  ;;
c)
  bash profiles/bench_configs.sh $OUT/configs 2>&1 | tee $OUT/configs.txt
  python bench.py > $OUT/default_run.json 2> $OUT/default_run.err && echo "default run ok"
  ;;
d)
  python profiles/generality.py 2>&1 | grep -v "amdgpu.ids" | tee $OUT/generality.txt
  python profiles/group_probe.py 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tee $OUT/group_probe.txt
  ;;
esac
