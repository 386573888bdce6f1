# round 4: C5 against the path pools' chunk size (pt_init caps the shift at 17 = 131072 paths; PT_AMD_CHUNK_SHIFT overrides)
for cs in 15 17 18 19 20; do
  PT_AMD_CHUNK_SHIFT=$cs python bench.py --steps 3 --warmup 1 --scene scenes/spheres64.txt --res 4096 4096 --depth 8 --batch 16 --cpu-spp 0 --per-iteration-sample 0 --repeats 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunk shift', $cs, 'value', d['value'], 'ms/launch', d['roofline']['avg_launch_ms'])"
done
for cs in 14 16 17 18; do
  PT_AMD_CHUNK_SHIFT=$cs python bench.py --steps 20 --warmup 5 --cpu-spp 0 --per-iteration-sample 0 --repeats 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 chunk shift', $cs, 'value', d['value'], 'ms/launch', d['roofline']['avg_launch_ms'])"
done
