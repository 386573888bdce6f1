#!/bin/bash
# Runs on the GPU box (inside gpurun):  bash profiles/run_scan.sh <tag>  -- the stream-compaction library under rocprofv3
set -o pipefail
TAG=${1:-r02_scan}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
PROG="python3 $PWD/profiles/scan_probe.py 26 10"
cd /tmp
export SCAN_PROBE_LIGHT=1
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $PROG > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
# (FETCH_SIZE and WRITE_SIZE in separate passes: together they exceed what one pass can collect)
timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $PROG > $OUT/pmc_fetch.log 2>&1 || { echo "pmc fetch failed"; tail -5 $OUT/pmc_fetch.log; exit 1; }
timeout -k 10 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $PROG > $OUT/pmc_write.log 2>&1 || { echo "pmc write failed"; tail -5 $OUT/pmc_write.log; exit 1; }
timeout -k 10 150 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/pmc_lds -- $PROG > $OUT/pmc_lds.log 2>&1 || { echo "pmc lds failed"; tail -5 $OUT/pmc_lds.log; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
st = sorted(glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)[-1]
print("kernel stats:")
for r in csv.DictReader(open(st)):
    if "k_scan" in r["Name"] or "k_compact" in r["Name"]:
        print("  %-60s calls %s avg %.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
for sub in ("pmc_fetch", "pmc_write", "pmc_lds"):
    f = sorted(glob.glob(out + "/" + sub + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)[-1]
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_scan" not in k and "k_compact" not in k: continue
        k = k[:50]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k in acc:
        print(sub, k, "dispatches", len(n[k]), {c: round(v / len(n[k])) for c, v in acc[k].items()})
PY
