# per-kernel times of C5, the build before (libpt_amd_prev.so) and the current one, one batch in flight
export TMPDIR=/tmp
R=$PWD
for v in _prev ""; do
  export PT_AMD_LIB=$R/project3-cuda-path-tracer_amd/csrc/libpt_amd$v.so
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4s$v -- python3 $R/bench.py --steps 4 --warmup 2 --scene $R/scenes/spheres64.txt --res 4096 4096 --depth 8 --batch 8 --per-iteration-sample 0 --repeats 1 --pipeline 1 --cpu-spp 0 > $R/gpurun_out/r4s$v.log 2>&1)
  f=$(find $R/gpurun_out/r4s$v -name "*kernel_stats.csv" | head -1)
  echo "== lib$v" ; head -8 $f | cut -c1-200
done
