# round 4: the mesh scenes' later-bounce kernel compiled for 7 / 5 / 4 resident workgroups per CU instead of 6
bash profiles/ab_libs.sh "--steps 8 --warmup 2 --scene scenes/cornell_mesh.txt --batch 32 --per-iteration-sample 0 --repeats 3" _prev _mlb7 _mlb5 _mlb4 > gpurun_out/r4aa.txt
