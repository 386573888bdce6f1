# round-4: the 32-class build (mesh scenes only) against its predecessor on C2, C5 and the mesh scene
bash profiles/ab_libs.sh "--steps 20 --warmup 5 --repeats 7 --per-iteration-sample 0" _prev ""
bash profiles/ab_libs.sh "--steps 20 --warmup 5 --repeats 7 --per-iteration-sample 0 --pipeline 1" _prev ""
bash profiles/ab_libs.sh "--steps 6 --warmup 2 --scene scenes/spheres64.txt --res 4096 4096 --depth 8 --batch 8 --per-iteration-sample 0 --repeats 3" _prev ""
bash profiles/ab_libs.sh "--steps 8 --warmup 2 --scene scenes/cornell_mesh.txt --per-iteration-sample 0 --repeats 5" _prev ""
bash profiles/ab_libs.sh "--steps 8 --warmup 2 --scene scenes/cornell_mesh.txt --per-iteration-sample 0 --repeats 5 --pipeline 1" _prev ""
