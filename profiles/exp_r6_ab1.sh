#!/bin/bash
# round 6, first A/B on one box: the round's opening commit (lib _r6base) against the build with k_commit_one, the laundered epilogue lane id and the
# dead engine state said out loud (no kernel carries scratch any more): C2, C4, the mesh scene and C5 -- frames bit-identical
mkdir -p gpurun_out/r6i
REPS=2 bash profiles/ab_libs.sh "--steps 16 --warmup 3 --repeats 7 --per-iteration-sample 0 --configs 0" _r6base _r6new > gpurun_out/r6i/ab_c2.txt 2>&1
REPS=2 bash profiles/ab_libs.sh "--scene scenes/cornell_glass.txt --res 1920 1080 --depth 16 --steps 4 --warmup 1 --repeats 5 --per-iteration-sample 0 --configs 0" _r6base _r6new > gpurun_out/r6i/ab_c4.txt 2>&1
REPS=2 bash profiles/ab_libs.sh "--scene scenes/cornell_mesh.txt --steps 4 --warmup 1 --repeats 5 --per-iteration-sample 0 --configs 0" _r6base _r6new > gpurun_out/r6i/ab_mesh.txt 2>&1
REPS=2 bash profiles/ab_libs.sh "--scene scenes/spheres64.txt --res 4096 4096 --batch 16 --steps 2 --warmup 1 --repeats 5 --per-iteration-sample 0 --configs 0" _r6base _r6new > gpurun_out/r6i/ab_c5.txt 2>&1
tail -n 6 gpurun_out/r6i/ab_c2.txt gpurun_out/r6i/ab_c4.txt gpurun_out/r6i/ab_mesh.txt gpurun_out/r6i/ab_c5.txt
