# round-4: C5 and mesh lines of the current build (same commands as bench_configs.sh; the mesh scene at the old step of 32 and the new one)
mkdir -p gpurun_out/r4c
python bench.py --steps 6 --warmup 2 --scene scenes/spheres64.txt --res 4096 4096 --depth 8 --batch 8 --cpu-spp 0 --per-iteration-sample 0 --repeats 5 2>gpurun_out/r4c/c5.err | python profiles/line_fields.py "c5 batch 8"
python bench.py --steps 8 --warmup 2 --scene scenes/cornell_mesh.txt --cpu-spp 0 --batch 32 --per-iteration-sample 0 --repeats 5 2>gpurun_out/r4c/mesh32.err | python profiles/line_fields.py "mesh batch 32"
python bench.py --steps 8 --warmup 2 --scene scenes/cornell_mesh.txt --cpu-spp 0 --per-iteration-sample 0 --repeats 5 2>gpurun_out/r4c/mesh64.err | python profiles/line_fields.py "mesh batch 64"
