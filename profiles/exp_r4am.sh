# round 4: the sphere-heavy scenes' LDS tables staged from ONE host-built image (a straight copy) instead of field by field: C5 in batches of 16 and of 8
bash profiles/ab_libs.sh "--steps 3 --warmup 1 --scene scenes/spheres64.txt --res 4096 4096 --depth 8 --batch 16 --per-iteration-sample 0 --repeats 3" _prev "" > gpurun_out/r4am.txt
bash profiles/ab_libs.sh "--steps 6 --warmup 2 --scene scenes/spheres64.txt --res 4096 4096 --depth 8 --batch 8 --per-iteration-sample 0 --repeats 3" _prev "" >> gpurun_out/r4am.txt
bash profiles/ab_libs.sh "--steps 24 --warmup 4 --scene scenes/spheres64.txt --res 4096 4096 --depth 8 --batch 1 --per-iteration-sample 0 --repeats 3" _prev "" >> gpurun_out/r4am.txt
