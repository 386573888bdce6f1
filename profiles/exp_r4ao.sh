for p in 1 2 3 4; do
  python bench.py --steps 3 --warmup 1 --scene scenes/spheres64.txt --res 4096 4096 --depth 8 --batch 16 --pipeline $p --cpu-spp 0 --per-iteration-sample 0 --repeats 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 batch 16 pipeline', $p, 'value', d['value'], d['value_min'], d['value_max'])"
done
for p in 1 2 3; do
  python bench.py --steps 6 --warmup 2 --scene scenes/spheres64.txt --res 4096 4096 --depth 8 --batch 8 --pipeline $p --cpu-spp 0 --per-iteration-sample 0 --repeats 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 batch 8 pipeline', $p, 'value', d['value'], d['value_min'], d['value_max'])"
done
