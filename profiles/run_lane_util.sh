#!/bin/bash
# Lane utilisation of the kernels: of the vector-instruction issue slots a launch fills, how many LANES do work?
#   VALUUtilization = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)     (rocprofiler's own derived metric; per kernel instantiation)
# plus the vector-memory and LDS issue cycles beside it.  Counters only, one --pmc pass each, never combined with a trace.
#   bash profiles/run_lane_util.sh <tag> "<bench args>"      -> gpurun_out/lane_<tag>/ ; distil with profiles/lane_util.py <tag>
set -o pipefail
TAG=${1:-c2}
ARGS=${2:-}
OUT=$PWD/gpurun_out/lane_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps ${STEPS:-8} --warmup 2 --cpu-spp 0 --pipeline 1 --per-iteration-sample 0 --configs 0 --repeats 1 $ARGS"
cd /tmp
rocprofv3 -L > $OUT/counters_available.txt 2>&1 || true
p=1
for set in "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU"; do
  rocprofv3 --pmc $set --output-format csv -d $OUT/pass$p -- $BENCH > $OUT/pass$p.log 2>&1 || { echo "pass $p failed"; tail -5 $OUT/pass$p.log; }
  p=$((p + 1))
done
ls $OUT
