#!/bin/bash
# (the record of an experiment: PT_AMD_COMPACT existed in the experiment build only)
# GPU box:  bash profiles/exp_r6_compact.sh  -- the compaction in two launches, with (default) and without (PT_AMD_COMPACT=1) the tile-ahead loads
for rep in 1 2 3; do
  for mode in 1 0; do
    for lg in 22 24 26 28; do
      if [ $mode = 0 ]; then unset PT_AMD_COMPACT; else export PT_AMD_COMPACT=$mode; fi
      echo -n "mode $mode rep $rep: "
      python3 profiles/scan_probe.py $lg 20 2>/dev/null | grep "^compact:"
    done
  done
done
