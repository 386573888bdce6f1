#!/bin/bash
# Runs on the GPU box (inside gpurun):  bash profiles/final_round.sh <tag>
# The round's closing measurement with ONE build: the whole GPU suite, the rocprofv3 sets of C2 and of the closed box, one bench line per
# BASELINE configuration.  Raw output under gpurun_out/final_<tag>/; the distilled summaries are copied into profiles/ afterwards.
set -o pipefail
TAG=${1:-r03}
OUT=gpurun_out/final_$TAG
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -3 $OUT/pytest.log
bash profiles/run_profile.sh $TAG 32 > $OUT/profile.log 2>&1 && echo "profile $TAG ok"
BENCH_ARGS="--scene $PWD/scenes/cornell_closed.txt" bash profiles/run_profile.sh ${TAG}_closed 32 > $OUT/profile_closed.log 2>&1 && echo "profile closed ok"
bash profiles/bench_configs.sh $OUT/configs 2>&1 | tee $OUT/configs.txt
