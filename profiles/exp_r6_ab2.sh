#!/bin/bash
# round 6, second A/B on one box: the mesh walk with the sentinel stack, branch-free push, two ballots and no end-of-turn bookkeeping (_r6walk) against
# the build before (_r6new): the mesh scene; C2 for control
mkdir -p gpurun_out/r6j
REPS=3 bash profiles/ab_libs.sh "--scene scenes/cornell_mesh.txt --steps 4 --warmup 1 --repeats 7 --per-iteration-sample 0 --configs 0" _r6new _r6walk > gpurun_out/r6j/ab_mesh.txt 2>&1
REPS=1 bash profiles/ab_libs.sh "--steps 16 --warmup 3 --repeats 7 --per-iteration-sample 0 --configs 0" _r6new _r6walk > gpurun_out/r6j/ab_c2.txt 2>&1
tail -n 8 gpurun_out/r6j/ab_mesh.txt gpurun_out/r6j/ab_c2.txt
