mkdir -p gpurun_out/r3f
bash profiles/run_trace.sh r3f > gpurun_out/r3f/trace.txt 2>&1; grep k_bounce gpurun_out/r3f/trace.txt | tail -9
bash profiles/ab_libs.sh "--steps 20 --warmup 5 --repeats 7" "" _k2 _k8 _k16 > gpurun_out/r3f/ab_ksub.txt 2>&1; cat gpurun_out/r3f/ab_ksub.txt
for b in 4 6 8; do for p in 1 2; do PT_AMD_BLOCKS_PER_CU=$b python bench.py --steps 20 --warmup 5 --repeats 5 --cpu-spp 0 --pipeline $p 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('blocks/CU $b pipeline $p', d['value'], d['roofline']['avg_launch_ms'])"; done; done > gpurun_out/r3f/blocks.txt 2>&1; cat gpurun_out/r3f/blocks.txt
