#!/bin/bash
# workgroups per CU x batches in flight (x iterations per step) on the round-5 build:  bash profiles/exp_r5_grid.sh
for b in 64 128; do
for p in 2 3 4; do
for w in 5 6 7 8; do
  PT_AMD_BLOCKS_PER_CU=$w python bench.py --steps $((1280 / b)) --warmup 4 --repeats 7 --pipeline $p --batch $b --cpu-spp 0 --per-iteration-sample 0 --configs 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('batch $b  in flight $p  workgroups per CU $w : %.1f G  (min %.1f max %.1f)  launch %.4f ms' % (d['value'] / 1e3, d['value_min'] / 1e3, d['value_max'] / 1e3, d['roofline']['avg_launch_ms']))"
done; done; done
