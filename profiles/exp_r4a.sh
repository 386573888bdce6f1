mkdir -p gpurun_out/r4a
for b in 32 64 128; do for p in 1 2; do python bench.py --steps 20 --warmup 5 --repeats 7 --cpu-spp 0 --batch $b --pipeline $p 2>gpurun_out/r4a/err_${b}_$p.txt | python profiles/line_fields.py "batch $b pipeline $p"; done; done > gpurun_out/r4a/batch.txt 2>&1
cat gpurun_out/r4a/batch.txt
