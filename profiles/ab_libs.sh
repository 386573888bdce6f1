# A/B of experiment builds of the library on one box:  bash profiles/ab_libs.sh "<bench args>" lib1 lib2 ...   (lib = suffix of csrc/libpt_amd<suffix>.so)
# every build renders the same frame (checked against the first one, bit for bit) and is timed twice, interleaved
args="$1"; shift
mkdir -p gpurun_out/ab
L=$PWD/project3-cuda-path-tracer_amd/csrc
out=gpurun_out/ab/libs.txt; : > $out
for rep in $(seq 1 ${REPS:-2}); do
for v in "$@"; do
  PT_AMD_LIB=$L/libpt_amd$v.so python bench.py $args --cpu-spp 0 --dump-frame gpurun_out/ab/frame$v.npy 2>gpurun_out/ab/err$v.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib$v', $rep, d['value'], d['roofline']['avg_launch_ms'])" >> $out || echo "lib$v FAILED" >> $out
done
done
python - "$@" >> $out <<'PY'
import sys, numpy as np
a = np.load("gpurun_out/ab/frame%s.npy" % sys.argv[1])
for v in sys.argv[2:]:
    b = np.load("gpurun_out/ab/frame%s.npy" % v)
    print("frame", v, "identical to", sys.argv[1], ":", bool(np.array_equal(a.view(np.uint32), b.view(np.uint32))), "max rel diff %.2e" % float(np.max(np.abs(a - b) / np.maximum(np.abs(a), 1e-30))))
PY
rm -f gpurun_out/ab/frame*.npy
cat $out
