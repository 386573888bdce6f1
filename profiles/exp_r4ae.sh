# round 4: what is C2's tile sensitive to -- one more dependent atomic round trip per reservation (PT_EXP=1), one more workgroup barrier per tile (PT_EXP=2)
bash profiles/ab_libs.sh "--steps 20 --warmup 5 --per-iteration-sample 0 --repeats 5" _prev _exp1 _exp2 > gpurun_out/r4ae.txt
bash profiles/ab_libs.sh "--steps 20 --warmup 5 --per-iteration-sample 0 --repeats 5 --pipeline 1" _prev _exp1 _exp2 >> gpurun_out/r4ae.txt
