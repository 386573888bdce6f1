#!/bin/bash
# round 6, fourth A/B on one box: the walk's fold without the world distance where a tile lists ONE mesh (_fnew) against the build before (_fbase)
mkdir -p gpurun_out/r6ad
REPS=3 bash profiles/ab_libs.sh "--scene scenes/cornell_mesh.txt --steps 4 --warmup 1 --repeats 7 --per-iteration-sample 0 --configs 0" _fbase _fnew > gpurun_out/r6ad/ab_mesh.txt 2>&1
cat gpurun_out/r6ad/ab_mesh.txt
