# round-4: does the mesh walk scale with the resident workgroups per CU (latency-bound) or not (texture-addresser-bound)?
for b in 2 3 4 5 6; do
PT_AMD_BLOCKS_PER_CU=$b python bench.py --steps 8 --warmup 2 --scene scenes/cornell_mesh.txt --cpu-spp 0 --batch 32 --per-iteration-sample 0 --repeats 3 --pipeline 1 2>gpurun_out/r4f.err | python profiles/line_fields.py "mesh, $b workgroups per CU, one batch in flight"
done
