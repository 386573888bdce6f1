#!/bin/bash
# A/B of experiment builds by COUNTERS on one box:  bash profiles/ab_pmc.sh "<bench args>" lib1 lib2 ...   (lib = suffix of csrc/libpt_amd<suffix>.so)
# one --pmc pass per build (vector / scalar instructions, busy and wait cycles; never combined with a trace), means per kernel instantiation
args="$1"; shift
export TMPDIR=/tmp
ROOT=$PWD
L=$ROOT/project3-cuda-path-tracer_amd/csrc
OUT=$ROOT/gpurun_out/abpmc; mkdir -p $OUT
for v in "$@"; do
  rm -rf $OUT/pmc$v
  (cd /tmp && PT_AMD_LIB=$L/libpt_amd$v.so rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc$v -- python3 $ROOT/bench.py $args --cpu-spp 0 --pipeline 1 --per-iteration-sample 0 --repeats 1 > $OUT/log$v.txt 2>&1) || { echo "pmc pass failed for lib$v"; tail -5 $OUT/log$v.txt; }
  python3 - $OUT/pmc$v "lib$v" <<'PY'
import csv, glob, sys, collections, re
files = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        m = re.search(r"k_bounce<([^>]*)>", k)
        if not m: continue
        acc["k_bounce<%s>" % m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(acc.items()):
    print(sys.argv[2], k, "launches", len(c.get("SQ_INSTS_VALU", [])), " ".join("%s %.4g" % (n, sum(v) / len(v)) for n, v in sorted(c.items())))
PY
done
