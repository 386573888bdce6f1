#!/bin/bash
# Runs on the GPU box (inside gpurun):  bash profiles/run_profile.sh <round-tag> [steps]
# Writes raw output under gpurun_out/prof_<tag>/ ; copy the summaries into profiles/ afterwards.
set -o pipefail
TAG=${1:-r01}
STEPS=${2:-32}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# one iteration in flight: a dispatch then owns the GPU, so its duration and counters are the kernel's own
# (round 4: the step is 64 iterations -- BATCH=32 reproduces the sets of rounds 1-3; the per-iteration reading of config C3 is left out of the
# profiled command: its thousands of small commit launches would drown the bounce kernels in the trace)
BENCH="python3 $PWD/bench.py --steps $STEPS --warmup 4 --cpu-spp 0 --pipeline ${PIPELINE:-1} --batch ${BATCH:-64} --per-iteration-sample 0 --configs 0 ${BENCH_ARGS:-}"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1 || { echo "pmc fetch failed"; tail -5 $OUT/pmc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1 || { echo "pmc write failed"; tail -5 $OUT/pmc_write.log; exit 1; }
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1 || { echo "pmc sq failed"; tail -5 $OUT/pmc_sq.log; exit 1; }
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_lds -- $BENCH > $OUT/pmc_lds.log 2>&1 || { echo "pmc lds failed"; tail -5 $OUT/pmc_lds.log; exit 1; }
find $OUT -name "*.csv" | head -50
