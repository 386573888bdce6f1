#!/bin/bash
# Runs on the GPU box (inside gpurun):  bash profiles/bench_configs.sh <outdir>  -- one bench line per BASELINE configuration that fits one GPU
OUT=${1:-gpurun_out/configs}
mkdir -p $OUT
python bench.py --steps 20 --warmup 5 > $OUT/c2.json 2> $OUT/c2.err
python bench.py --steps 20 --warmup 5 --scene scenes/cornell_closed.txt --cpu-spp 4 > $OUT/closed.json 2> $OUT/closed.err
python bench.py --steps 8 --warmup 2 --scene scenes/cornell_glass.txt --res 1920 1080 --depth 16 --cpu-spp 4 > $OUT/c4.json 2> $OUT/c4.err
python bench.py --steps 6 --warmup 2 --scene scenes/spheres64.txt --res 4096 4096 --depth 8 --batch 8 --cpu-spp 0 > $OUT/c5.json 2> $OUT/c5.err
# ... and with the configuration's 16 spp as ONE step (as C2's 64 are): a launch then carries 16 iterations of the 4096 x 4096 frame
python bench.py --steps 3 --warmup 1 --scene scenes/spheres64.txt --res 4096 4096 --depth 8 --batch 16 --cpu-spp 0 > $OUT/c5_spp16.json 2> $OUT/c5_spp16.err
python bench.py --steps 8 --warmup 2 --scene scenes/cornell_mesh.txt --cpu-spp 0 > $OUT/mesh.json 2> $OUT/mesh.err
python bench.py --steps 20 --warmup 5 --scene scenes/sphere.txt --res 400 400 --depth 4 --cpu-spp 0 > $OUT/c1.json 2> $OUT/c1.err
for f in c2 closed c4 c5 c5_spp16 mesh c1; do python - $OUT/$f.json <<'PY'
import sys, json
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1].split("/")[-1], "value", d["value"], "min", d["value_min"], "max", d["value_max"], "ms/launch", d["roofline"]["avg_launch_ms"], "frac", d["roofline"]["frac"],
          "live/bounce", [int(v) for v in d["config"]["live_per_bounce_per_iteration"]][:8])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
