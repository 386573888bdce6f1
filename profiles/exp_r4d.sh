# round-4: is the mesh scene's time the SUM of its two meshes' walks?  (both, icosphere only, torus only, neither; launches alone and pipelined)
for sc in cornell_mesh _tmp_mesh_onlyA _tmp_mesh_onlyB _tmp_mesh_none; do
python bench.py --steps 8 --warmup 2 --scene scenes/$sc.txt --cpu-spp 0 --batch 32 --per-iteration-sample 0 --repeats 5 2>gpurun_out/r4d_$sc.err | python profiles/line_fields.py "$sc batch 32"
done
