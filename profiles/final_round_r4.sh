#!/bin/bash
# Runs on the GPU box (inside gpurun):  bash profiles/final_round_r4.sh <part>   (a = suite + C2 sets, b = closed box, C4, c = C5 + mesh sets, d = bench lines)
# Round 4's closing measurement with ONE build.  Raw output under gpurun_out/final_r04/; the distilled summaries are copied into profiles/.
set -o pipefail
OUT=gpurun_out/final_r04
mkdir -p $OUT
case "${1:-a}" in
a)
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
  tail -3 $OUT/pytest.log
  bash profiles/run_profile.sh r04 20 > $OUT/profile.log 2>&1 && echo "profile r04 (step of 64) ok"
  BATCH=32 bash profiles/run_profile.sh r04_b32 32 > $OUT/profile_b32.log 2>&1 && echo "profile r04_b32 (step of 32, as rounds 1-3) ok"
  ;;
b)
  BENCH_ARGS="--scene $PWD/scenes/cornell_closed.txt" bash profiles/run_profile.sh r04_closed 20 > $OUT/profile_closed.log 2>&1 && echo "profile closed ok"
  BENCH_ARGS="--scene $PWD/scenes/cornell_glass.txt --res 1920 1080 --depth 16" bash profiles/run_profile.sh r04_c4 8 > $OUT/profile_c4.log 2>&1 && echo "profile c4 ok"
  ;;
c)
  BATCH=8 BENCH_ARGS="--scene $PWD/scenes/spheres64.txt --res 4096 4096 --depth 8" bash profiles/run_profile.sh r04_c5 6 > $OUT/profile_c5.log 2>&1 && echo "profile c5 ok"
  BATCH=32 BENCH_ARGS="--scene $PWD/scenes/cornell_mesh.txt" bash profiles/run_profile.sh r04_mesh 8 > $OUT/profile_mesh.log 2>&1 && echo "profile mesh ok"
  ;;
d)
  bash profiles/bench_configs.sh $OUT/configs 2>&1 | tee $OUT/configs.txt
  ;;
esac
