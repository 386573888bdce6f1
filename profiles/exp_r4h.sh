# round-4: the PLAIN instantiations (no refraction / lobe / direct-light code) against the general ones, same library (PT_AMD_NO_PLAIN=1)
mkdir -p gpurun_out/r4h
for rep in 1 2; do
python bench.py --steps 20 --warmup 5 --repeats 7 --cpu-spp 0 --per-iteration-sample 0 --dump-frame gpurun_out/r4h/plain.npy 2>gpurun_out/r4h/e1.txt | python profiles/line_fields.py "plain   pipelined"
PT_AMD_NO_PLAIN=1 python bench.py --steps 20 --warmup 5 --repeats 7 --cpu-spp 0 --per-iteration-sample 0 --dump-frame gpurun_out/r4h/general.npy 2>gpurun_out/r4h/e2.txt | python profiles/line_fields.py "general pipelined"
done
python -c "
import numpy as np
a=np.load('gpurun_out/r4h/plain.npy'); b=np.load('gpurun_out/r4h/general.npy'); print('frames identical:', bool(np.array_equal(a.view(np.uint32), b.view(np.uint32))))"
rm -f gpurun_out/r4h/*.npy
