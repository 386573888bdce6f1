# round 4: C5's rate against the iterations per wavefront batch (the path pools are 0.74 GB per iteration in flight)
for b in 1 2 4 8 16; do
  python bench.py --steps $((48 / b)) --warmup 2 --scene scenes/spheres64.txt --res 4096 4096 --depth 8 --batch $b --cpu-spp 0 --per-iteration-sample 0 --repeats 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch', $b, 'value', d['value'], 'ms/launch', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'])"
done
