python -m pytest tests/test_gpu_camera_cull.py -x -q -m gpu -s > gpurun_out/r4m_cull.log 2>&1; grep "inflation margin\|passed\|failed" gpurun_out/r4m_cull.log
bash profiles/ab_libs.sh "--steps 20 --warmup 5 --repeats 7 --per-iteration-sample 0" _66690a7 ""
bash profiles/ab_libs.sh "--steps 20 --warmup 5 --repeats 7 --per-iteration-sample 0 --pipeline 1" _66690a7 ""
