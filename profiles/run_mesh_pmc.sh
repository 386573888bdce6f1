#!/bin/bash
# Runs on the GPU box (inside gpurun):  [SCENE=scenes/cornell.txt] bash profiles/run_mesh_pmc.sh <tag>
# Cache- and texture-path counters of a scene's launches (default: the mesh scene), one --pmc pass per group (never with a trace
# domain).  (A group with TA_ADDR_STALLED_BY_TC_CYCLES_sum / TA_DATA_STALLED_BY_TC_CYCLES_sum aborted rocprofv3 on this image and is
# not in the list.)
set -o pipefail
TAG=${1:-r03_mesh}
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 8 --warmup 2 --cpu-spp 0 --pipeline 1 --repeats 1 --scene $PWD/${SCENE:-scenes/cornell_mesh.txt}"
cd /tmp
rocprofv3 -L > $OUT/avail.txt 2>&1 || true
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TA_BUSY_avr TA_BUSY_max TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- $BENCH > $OUT/g$i.log 2>&1 || { echo "group $i failed: $grp"; tail -3 $OUT/g$i.log; continue; }
  f=$(find $OUT/g$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$grp" >> $OUT/summary.txt <<'PY'
import sys, csv, collections
f, grp = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "k_bounce" not in k and "k_mesh_walk" not in k: continue
    k = k.split("(")[0].replace("void ptk::", "")
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in acc:
    print(k, "dispatches", len(n[k]), {c: v / len(n[k]) for c, v in acc[k].items()})
PY
done
cat $OUT/summary.txt
