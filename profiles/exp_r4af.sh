# round 4: LATE (a later-bounce tile's reservation taken a tile late) against the build before
A="--per-iteration-sample 0 --repeats 5"
bash profiles/ab_libs.sh "--steps 20 --warmup 5 $A" _prev "" > gpurun_out/r4af.txt
bash profiles/ab_libs.sh "--steps 20 --warmup 5 $A --pipeline 1" _prev "" >> gpurun_out/r4af.txt
bash profiles/ab_libs.sh "--steps 20 --warmup 5 --scene scenes/cornell_closed.txt $A" _prev "" >> gpurun_out/r4af.txt
bash profiles/ab_libs.sh "--steps 8 --warmup 2 --scene scenes/cornell_glass.txt --res 1920 1080 --depth 16 $A" _prev "" >> gpurun_out/r4af.txt
