#!/bin/bash
# (the record of an experiment: modes 2 and 3 existed in the experiment build only -- the shipped library runs what was mode 0)
# Runs on the GPU box:  bash profiles/exp_r6_scan.sh  -- the scan's variants side by side on ONE box (PT_AMD_SCAN: 2 = round 5's three launches with
# the tile-ahead loads, 3 = + the reduce's striped loads, unset = + the apply adds up the totals before its chunk itself: two launches)
set -o pipefail
mkdir -p gpurun_out/r6_scan
for rep in 1 2 3; do
  for mode in 2 3 0; do
    for lg in 22 24 26 28; do
      if [ $mode = 0 ]; then unset PT_AMD_SCAN; else export PT_AMD_SCAN=$mode; fi
      echo -n "mode $mode rep $rep: "
      python3 profiles/scan_probe.py $lg 20 2>/dev/null | grep "^scan:"
    done
  done
done
