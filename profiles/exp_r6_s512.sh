#!/bin/bash
# round 6: kernel by kernel, the 518-primitive scene with the grouped sweep (rocprofv3 --kernel-trace --stats; one batch in flight)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r6n; mkdir -p $OUT
B="python3 $PWD/bench.py --scene $PWD/scenes/${1:-spheres512.txt} --steps 3 --warmup 1 --repeats 2 --batch 16 --cpu-spp 0 --per-iteration-sample 0 --configs 0 --pipeline 1"
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/trace.log 2>&1
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); head -6 $f | cut -c1-220
