# round-4: tickets in the camera-ray launch too (default) against the rotated fixed stride (_nofirst), pipelined and alone; C4 as a second scene
bash profiles/ab_libs.sh "--steps 20 --warmup 5 --repeats 7 --per-iteration-sample 0" _nofirst ""
bash profiles/ab_libs.sh "--steps 20 --warmup 5 --repeats 7 --per-iteration-sample 0 --pipeline 1" _nofirst ""
bash profiles/ab_libs.sh "--steps 8 --warmup 2 --repeats 5 --per-iteration-sample 0 --scene scenes/cornell_glass.txt --res 1920 1080 --depth 16" _nofirst ""
