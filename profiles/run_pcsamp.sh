#!/bin/bash
# Runs on the GPU box (inside gpurun):  bash profiles/run_pcsamp.sh <tag> [method] [interval]
# PC sampling (rocprofv3 beta) of bench.py against a build with line tables (make -C project3-cuda-path-tracer_amd/csrc lines):
# where the waves of k_bounce spend their time, by source line.  One batch in flight, so a sample belongs to the launch it names.
# Bounded by `timeout`: the feature is beta, a run that does not finish must not hold the box.
set -o pipefail
TAG=${1:-x}
METHOD=${2:-stochastic}
INTERVAL=${3:-65536}
UNIT=cycles
if [ "$METHOD" = host_trap ]; then UNIT=time; fi
OUT=$PWD/gpurun_out/pcs_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export PT_AMD_LIB=$PWD/project3-cuda-path-tracer_amd/csrc/libpt_amd_lines.so
BENCH="python3 $PWD/bench.py --steps ${STEPS:-3} --warmup 1 --repeats 1 --cpu-spp 0 --pipeline 1 ${BENCH_ARGS:-}"
cd /tmp
timeout -k 10 ${LIMIT:-240} rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $METHOD --pc-sampling-unit $UNIT --pc-sampling-interval $INTERVAL \
    --kernel-trace --output-format csv -d $OUT/raw -- $BENCH > $OUT/run.log 2>&1
echo "rocprofv3 rc=$?" >> $OUT/run.log
tail -5 $OUT/run.log
find $OUT/raw -name "*.csv" | head
