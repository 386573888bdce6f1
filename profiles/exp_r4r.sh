# round 4: C5 with the spheres in two clusters (class bits 3 / 4) against the build before (libpt_amd_prev.so), then C2 and the mesh scene
bash profiles/ab_libs.sh "--steps 6 --warmup 2 --scene scenes/spheres64.txt --res 4096 4096 --depth 8 --batch 8 --per-iteration-sample 0 --repeats 3" _prev "" > gpurun_out/r4r.txt
bash profiles/ab_libs.sh "--steps 6 --warmup 2 --scene scenes/spheres64.txt --res 4096 4096 --depth 8 --batch 8 --per-iteration-sample 0 --repeats 3 --pipeline 1" _prev "" >> gpurun_out/r4r.txt
bash profiles/ab_libs.sh "--steps 20 --warmup 5 --repeats 7 --per-iteration-sample 0" _prev "" >> gpurun_out/r4r.txt
