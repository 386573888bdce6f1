# round-4: what does the sphere sweep of the later bounces cost on C5?  (_exp128 runs it twice, same masks, same frame)
bash profiles/ab_libs.sh "--steps 6 --warmup 2 --scene scenes/spheres64.txt --res 4096 4096 --depth 8 --batch 8 --per-iteration-sample 0 --repeats 3" "" _exp128
