# round 4: what is C5's later-bounce tile sensitive to -- one more dependent atomic round trip per reservation (PT_EXP=1), one more workgroup barrier per tile (PT_EXP=2)
bash profiles/ab_libs.sh "--steps 3 --warmup 1 --scene scenes/spheres64.txt --res 4096 4096 --depth 8 --batch 16 --per-iteration-sample 0 --repeats 3" _prev _exp1 _exp2 > gpurun_out/r4ab.txt
bash profiles/ab_libs.sh "--steps 3 --warmup 1 --scene scenes/spheres64.txt --res 4096 4096 --depth 8 --batch 16 --per-iteration-sample 0 --repeats 3 --pipeline 1" _prev _exp1 _exp2 >> gpurun_out/r4ab.txt
