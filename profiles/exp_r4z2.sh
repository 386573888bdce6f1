for cfg in "16 3 3" "32 2 3" "32 2 2" "64 1 2"; do set -- $cfg
  timeout -k 10 200 python bench.py --steps $2 --warmup 1 --scene scenes/spheres64.txt --res 4096 4096 --depth 8 --batch $1 --pipeline $3 --cpu-spp 0 --per-iteration-sample 0 --repeats 3 2>gpurun_out/r4z2.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch', $1, 'pipeline', $3, 'value', d['value'], 'ms/launch', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'])" || tail -3 gpurun_out/r4z2.err
done
