bash profiles/ab_libs.sh "--steps 3 --warmup 1 --scene scenes/spheres64.txt --res 4096 4096 --depth 8 --batch 16 --per-iteration-sample 0 --repeats 3" _prev "" > gpurun_out/r4ac.txt
bash profiles/ab_libs.sh "--steps 3 --warmup 1 --scene scenes/spheres64.txt --res 4096 4096 --depth 8 --batch 16 --per-iteration-sample 0 --repeats 3 --pipeline 1" _prev "" >> gpurun_out/r4ac.txt
