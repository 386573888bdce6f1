#!/bin/bash
# Runs on the GPU box (inside gpurun):  bash profiles/run_valu.sh <tag>  -- one PMC pass, VALU / SALU instruction counts only
set -o pipefail
TAG=${1:-x}
OUT=$PWD/gpurun_out/valu_$TAG${EXP:+_exp$EXP}
mkdir -p $OUT
export TMPDIR=/tmp
PROG="$PWD/bench.py"
# EXP=n: an experiment build of the library (make -C project3-cuda-path-tracer_amd/csrc exp EXP=n)
if [ -n "$EXP" ]; then PROG="$PWD/profiles/exp_bench.py $EXP"; fi
BENCH="python3 $PROG --steps 2 --warmup 1 --cpu-spp 0 --pipeline 1 --batch ${BATCH:-32} ${BENCH_ARGS:-}"
cd /tmp
rocprofv3 --pmc ${PMC:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE} --output-format csv -d $OUT/pmc -- $BENCH > $OUT/pmc.log 2>&1 || { echo "pmc failed"; tail -5 $OUT/pmc.log; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/pmc/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = "k_bounce<true>" if "k_bounceILb1" in r["Kernel_Name"] or "k_bounce<true" in r["Kernel_Name"] else ("k_bounce<false>" if "k_bounce" in r["Kernel_Name"] else None)
    if not k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in acc:
    print(k, "dispatches", len(n[k]), {c: round(v / len(n[k])) for c, v in acc[k].items()})
PY
