#!/bin/bash
# round 6, third A/B on one box: k_commit / k_commit_one without the re-zeroing of consumed radiance entries (_znew) against the build before (_zbase)
mkdir -p gpurun_out/r6z
REPS=3 bash profiles/ab_libs.sh "--steps 16 --warmup 3 --repeats 9 --per-iteration-sample 0 --configs 0" _zbase _znew > gpurun_out/r6z/ab_c2.txt 2>&1
REPS=2 bash profiles/ab_libs.sh "--scene scenes/cornell_glass.txt --res 1920 1080 --depth 16 --steps 4 --warmup 1 --repeats 5 --per-iteration-sample 0 --configs 0" _zbase _znew > gpurun_out/r6z/ab_c4.txt 2>&1
REPS=2 bash profiles/ab_libs.sh "--scene scenes/spheres64.txt --res 4096 4096 --batch 16 --steps 2 --warmup 1 --repeats 5 --per-iteration-sample 0 --configs 0" _zbase _znew > gpurun_out/r6z/ab_c5.txt 2>&1
tail -n 8 gpurun_out/r6z/ab_c2.txt gpurun_out/r6z/ab_c4.txt gpurun_out/r6z/ab_c5.txt
