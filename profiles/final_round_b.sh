#!/bin/bash
# Runs on the GPU box (inside gpurun):  bash profiles/final_round_b.sh <tag>
# After the round's last kernel changes (sphere sweep, mesh walk): the whole GPU suite, the rocprofv3 sets of C5 and of the mesh scene,
# one bench line per configuration.  Raw output under gpurun_out/final_<tag>b/.
set -o pipefail
TAG=${1:-r03}
OUT=gpurun_out/final_${TAG}b
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -3 $OUT/pytest.log
BATCH=8 BENCH_ARGS="--scene $PWD/scenes/spheres64.txt --res 4096 4096 --depth 8" bash profiles/run_profile.sh ${TAG}_c5 6 > $OUT/profile_c5.log 2>&1 && echo "profile c5 ok"
BENCH_ARGS="--scene $PWD/scenes/cornell_mesh.txt" bash profiles/run_profile.sh ${TAG}_mesh 8 > $OUT/profile_mesh.log 2>&1 && echo "profile mesh ok"
bash profiles/bench_configs.sh $OUT/configs 2>&1 | tee $OUT/configs.txt
