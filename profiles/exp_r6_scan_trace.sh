#!/bin/bash
# (the record of an experiment: modes 2 and 4 existed in the experiment build only)
# GPU box:  bash profiles/exp_r6_scan_trace.sh  -- kernel durations of the scan's variants under rocprofv3 (PT_AMD_SCAN modes, see exp_r6_scan.sh; 4 = tiles interleaved
# over the workgroups, a timing experiment whose results are wrong)
set -o pipefail
OUT=$PWD/gpurun_out/r6_scan
mkdir -p $OUT
export TMPDIR=/tmp SCAN_PROBE_LIGHT=1
ROOT=$PWD
cd /tmp
for lg in 24 26 28; do
for mode in 0 2; do
  if [ $mode = 0 ]; then unset PT_AMD_SCAN; else export PT_AMD_SCAN=$mode; fi
  rm -rf $OUT/tr_$mode
  timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr_$mode -- python3 $ROOT/profiles/scan_probe.py $lg 10 > $OUT/tr_$mode.log 2>&1 || { echo "trace failed"; tail -5 $OUT/tr_$mode.log; exit 1; }
  echo "n = 2^$lg mode $mode"
  python3 - $OUT/tr_$mode <<'PY'
import csv, glob, sys, os
st = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)[-1]
for r in csv.DictReader(open(st)):
    if "k_scan" in r["Name"]:
        print("  %-70s calls %s avg %.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
done
