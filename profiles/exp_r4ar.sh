bash profiles/ab_libs.sh "--steps 3 --warmup 1 --scene scenes/spheres64.txt --res 4096 4096 --depth 8 --batch 16 --per-iteration-sample 0 --repeats 3" _prev _lazy6 _lazy7 > gpurun_out/r4ar.txt
bash profiles/ab_libs.sh "--steps 6 --warmup 2 --scene scenes/spheres64.txt --res 4096 4096 --depth 8 --batch 8 --per-iteration-sample 0 --repeats 3" _prev _lazy6 _lazy7 >> gpurun_out/r4ar.txt
