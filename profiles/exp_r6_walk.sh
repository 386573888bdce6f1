#!/bin/bash
# round 6: the mesh walk's compile-time parameters once more on the closing build (pipelined, the driver's own shape of the mesh block):
# quarter tiles a wave wants at least (8), idle lanes before a hand-out (24), lanes at a triangle before a triangle turn (16)
mkdir -p gpurun_out/r6w
REPS=2 bash profiles/ab_libs.sh "--scene scenes/cornell_mesh.txt --steps 4 --warmup 1 --repeats 5 --per-iteration-sample 0 --configs 0" _wbase _wq4 _wq16 _wq32 _wi16 _wi32 _wl8 _wl24 > gpurun_out/r6w/ab.txt 2>&1
cat gpurun_out/r6w/ab.txt
