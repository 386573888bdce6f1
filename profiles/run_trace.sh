#!/bin/bash
# Runs on the GPU box (inside gpurun):  [BATCH=..] [BENCH_ARGS=..] bash profiles/run_trace.sh <tag>  -- kernel trace only
set -o pipefail
TAG=${1:-x}
OUT=$PWD/gpurun_out/trace_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 2 --warmup 1 --cpu-spp 0 --pipeline 1 --batch ${BATCH:-32} ${BENCH_ARGS:-}"
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- $BENCH > $OUT/t.log 2>&1 || { echo "trace failed"; tail -5 $OUT/t.log; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/t/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
for r in rows[-24:]:
    print(r["Kernel_Name"][:40], "dur_us", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "grid", r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size"), "vgpr", r.get("VGPR_Count"), "scratch", r.get("Scratch_Size", r.get("Private_Segment_Size")), "lds", r.get("LDS_Block_Size", r.get("Group_Segment_Size")))
PY
