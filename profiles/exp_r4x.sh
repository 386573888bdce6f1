# round 4: resident workgroups per CU of the sphere-heavy kernels -- launch bounds of the later-bounce kernel 8 (64 VGPRs, spills) / 7 (72, 5 spilled) / 6 (80) / 5 (96),
# of the camera-ray kernel 7 / 6  (libs: _lb<later>, _lbf6 = camera 6 + later 6, _lbf6l5 = camera 6 + later 5)
bash profiles/ab_libs.sh "--steps 6 --warmup 2 --scene scenes/spheres64.txt --res 4096 4096 --depth 8 --batch 8 --per-iteration-sample 0 --repeats 3" _prev _lb6 _lb5 _lbf6 _lbf6l5 > gpurun_out/r4x2.txt
bash profiles/ab_libs.sh "--steps 6 --warmup 2 --scene scenes/spheres64.txt --res 4096 4096 --depth 8 --batch 8 --per-iteration-sample 0 --repeats 3 --pipeline 1" _prev _lb6 _lb5 _lbf6 _lbf6l5 >> gpurun_out/r4x2.txt
