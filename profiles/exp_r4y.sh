# round 4: C2's kernels compiled for 7 / 6 resident workgroups per CU (72 / 80 VGPRs) instead of 8 (64)
bash profiles/ab_libs.sh "--steps 20 --warmup 5 --repeats 7 --per-iteration-sample 0" _prev _c2lb7 _c2lb6 > gpurun_out/r4y.txt
bash profiles/ab_libs.sh "--steps 20 --warmup 5 --repeats 7 --per-iteration-sample 0 --pipeline 1" _prev _c2lb7 _c2lb6 >> gpurun_out/r4y.txt
