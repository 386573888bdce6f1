#!/bin/bash
# Runs on the GPU box (inside gpurun):  bash profiles/final_round_r5.sh <part>
#   a = the GPU suite + the C2 counter sets (step of 64, and of 32 as rounds 1-3)      b = closed box, C4      c = C5 on one GPU (16 spp per step) + mesh scene
#   d = one bench line per configuration + the driver's own command                     e = scenes outside the Cornell shape, phase probe, scan library
# Round 5's closing measurement with ONE build.  Raw output under gpurun_out/final_r05/; the distilled summaries are copied into profiles/.
set -o pipefail
OUT=gpurun_out/final_r05
mkdir -p $OUT
case "${1:-a}" in
a)
  timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
  tail -3 $OUT/pytest.log
  bash profiles/run_profile.sh r05 20 > $OUT/profile.log 2>&1 && echo "profile r05 (step of 64) ok"
  BATCH=32 bash profiles/run_profile.sh r05_b32 32 > $OUT/profile_b32.log 2>&1 && echo "profile r05_b32 (step of 32, as rounds 1-3) ok"
  ;;
b)
  BENCH_ARGS="--scene $PWD/scenes/cornell_closed.txt" bash profiles/run_profile.sh r05_closed 20 > $OUT/profile_closed.log 2>&1 && echo "profile closed ok"
  BENCH_ARGS="--scene $PWD/scenes/cornell_glass.txt --res 1920 1080 --depth 16" bash profiles/run_profile.sh r05_c4 4 > $OUT/profile_c4.log 2>&1 && echo "profile c4 ok"
  ;;
c)
  BATCH=16 BENCH_ARGS="--scene $PWD/scenes/spheres64.txt --res 4096 4096 --depth 8" bash profiles/run_profile.sh r05_c5 2 > $OUT/profile_c5.log 2>&1 && echo "profile c5 ok"
  BENCH_ARGS="--scene $PWD/scenes/cornell_mesh.txt" bash profiles/run_profile.sh r05_mesh 4 > $OUT/profile_mesh.log 2>&1 && echo "profile mesh ok"
  ;;
d)
  bash profiles/bench_configs.sh $OUT/configs 2>&1 | tee $OUT/configs.txt
  python bench.py > $OUT/default_run.json 2> $OUT/default_run.err && echo "default run ok"
  ;;
e)
  python profiles/generality.py 2>&1 | grep -v "amdgpu.ids" | tee $OUT/generality.txt
  python profiles/probe_phases.py cornell.txt spheres64.txt 2>&1 | grep -v "amdgpu.ids" | tee $OUT/phase_probe.txt
  bash profiles/run_scan.sh r05_scan 2>&1 | tee $OUT/scan.txt
  for lg in 20 24 26 28; do python profiles/scan_probe.py $lg 20 2>&1 | grep -v amdgpu; done | tee -a $OUT/scan.txt
  ;;
esac
