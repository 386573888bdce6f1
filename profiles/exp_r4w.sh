# round 4: 16 append-counter shards per class (PT_KSUB=16: 256 segments) against 8, on the sphere-heavy scene, the mesh scene and C2
bash profiles/ab_libs.sh "--steps 6 --warmup 2 --scene scenes/spheres64.txt --res 4096 4096 --depth 8 --batch 8 --per-iteration-sample 0 --repeats 3" _prev _ksub16 > gpurun_out/r4w.txt
bash profiles/ab_libs.sh "--steps 8 --warmup 2 --scene scenes/cornell_mesh.txt --batch 32 --per-iteration-sample 0 --repeats 3" _prev _ksub16 >> gpurun_out/r4w.txt
bash profiles/ab_libs.sh "--steps 20 --warmup 5 --repeats 7 --per-iteration-sample 0" _prev _ksub16 >> gpurun_out/r4w.txt
bash profiles/ab_libs.sh "--steps 20 --warmup 5 --repeats 7 --per-iteration-sample 0 --pipeline 1" _prev _ksub16 >> gpurun_out/r4w.txt
