#!/bin/bash
# round 6: what bounds a group of 8 members on ONE device (config C2 in batch mode)?  HW queues per process and the members' persistent grids.
set -e
mkdir -p gpurun_out/r6b
P="python profiles/group_probe.py --only g8"
for q in 4 8 16; do
  for b in 0 1 2 3; do
    echo "== GPU_MAX_HW_QUEUES=$q PT_AMD_BLOCKS_PER_CU=$b"
    if [ $b = 0 ]; then GPU_MAX_HW_QUEUES=$q $P; else GPU_MAX_HW_QUEUES=$q PT_AMD_BLOCKS_PER_CU=$b $P; fi
  done
done
