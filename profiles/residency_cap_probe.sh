for rep in 1 2; do for cap in 5 6 7 8; do
PT_AMD_BLOCKS_PER_CU=$cap python bench.py --steps 60 --warmup 10 --cpu-spp 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cap $cap', d['value'], d['roofline']['avg_launch_ms'])"
done; done
for rep in 1 2; do for p in 2 3; do
python bench.py --steps 60 --warmup 10 --cpu-spp 0 --pipeline $p 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipeline $p', d['value'], d['roofline']['avg_launch_ms'])"
done; done
