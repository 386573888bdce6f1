# round 4: the pools' chunk size by the new rule (1/256 of the paths, at most 2^18) against the old one (1/1024, at most 2^17): every configuration on one box
A="--per-iteration-sample 0 --repeats 5"
bash profiles/ab_libs.sh "--steps 20 --warmup 5 $A" _prev "" > gpurun_out/r4ak.txt
bash profiles/ab_libs.sh "--steps 20 --warmup 5 $A --pipeline 1" _prev "" >> gpurun_out/r4ak.txt
bash profiles/ab_libs.sh "--steps 20 --warmup 5 --scene scenes/sphere.txt --res 400 400 --depth 4 $A" _prev "" >> gpurun_out/r4ak.txt
bash profiles/ab_libs.sh "--steps 20 --warmup 5 --scene scenes/cornell_closed.txt $A" _prev "" >> gpurun_out/r4ak.txt
bash profiles/ab_libs.sh "--steps 8 --warmup 2 --scene scenes/cornell_glass.txt --res 1920 1080 --depth 16 $A" _prev "" >> gpurun_out/r4ak.txt
bash profiles/ab_libs.sh "--steps 3 --warmup 1 --scene scenes/spheres64.txt --res 4096 4096 --depth 8 --batch 16 $A" _prev "" >> gpurun_out/r4ak.txt
bash profiles/ab_libs.sh "--steps 8 --warmup 2 --scene scenes/cornell_mesh.txt --batch 32 $A" _prev "" >> gpurun_out/r4ak.txt
bash profiles/ab_libs.sh "--steps 40 --warmup 5 --batch 1 $A" _prev "" >> gpurun_out/r4ak.txt
bash profiles/ab_libs.sh "--steps 20 --warmup 5 --batch 8 $A" _prev "" >> gpurun_out/r4ak.txt
