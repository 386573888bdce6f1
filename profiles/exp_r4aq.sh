A="--per-iteration-sample 0 --repeats 5"
bash profiles/ab_libs.sh "--steps 20 --warmup 5 $A" _prev "" > gpurun_out/r4aq.txt
bash profiles/ab_libs.sh "--steps 20 --warmup 5 $A --pipeline 1" _prev "" >> gpurun_out/r4aq.txt
bash profiles/ab_libs.sh "--steps 20 --warmup 5 --scene scenes/cornell_closed.txt $A" _prev "" >> gpurun_out/r4aq.txt
bash profiles/ab_libs.sh "--steps 3 --warmup 1 --scene scenes/spheres64.txt --res 4096 4096 --depth 8 --batch 16 $A" _prev "" >> gpurun_out/r4aq.txt
