# round 4: C2 against the path pools' chunk size, interleaved (two rounds), pipelined and one batch in flight
for rep in 1 2; do for cs in 16 17 18 19 20; do
  PT_AMD_CHUNK_SHIFT=$cs python bench.py --steps 20 --warmup 5 --cpu-spp 0 --per-iteration-sample 0 --repeats 7 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 round', $rep, 'chunk shift', $cs, 'value', d['value'], 'ms/launch', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'])"
done; done
for cs in 16 18 20; do
  PT_AMD_CHUNK_SHIFT=$cs python bench.py --steps 20 --warmup 5 --cpu-spp 0 --per-iteration-sample 0 --repeats 7 --pipeline 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 one batch in flight, chunk shift', $cs, 'value', d['value'], 'ms/launch', d['roofline']['avg_launch_ms'])"
done
for cs in 16 18 20; do
  PT_AMD_CHUNK_SHIFT=$cs python bench.py --steps 20 --warmup 5 --scene scenes/cornell_closed.txt --cpu-spp 0 --per-iteration-sample 0 --repeats 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('closed, chunk shift', $cs, 'value', d['value'], 'ms/launch', d['roofline']['avg_launch_ms'])"
done
