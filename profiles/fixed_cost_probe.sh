mkdir -p gpurun_out/ab
for cfg in "64 64 1 1" "64 64 1 2" "256 256 1 1" "1280 720 1 1" "1280 720 1 2" "1280 720 4 1" "1280 720 32 1"; do
set -- $cfg
python bench.py --steps 400 --warmup 50 --res $1 $2 --batch $3 --pipeline $4 --cpu-spp 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', 'ms/step', d['ms_per_step'], 'avg_launch_ms', d['roofline']['avg_launch_ms'], 'share', d['roofline']['bounce_kernel_share_of_step'], 'with_events', d['roofline']['ms_per_step_with_events'])"
done
