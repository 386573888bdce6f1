# round-4: is the mesh scene's time the SUM of its two meshes' walks?  (both, icosphere only, torus only, neither; launches alone and pipelined)
python - <<'PY'
s = open('scenes/cornell_mesh.txt').read()
i6, i7 = s.index('// object 6'), s.index('// object 7')
open('gpurun_out/_mesh_onlyA.txt', 'w').write(s[:i7].rstrip() + "\n")
open('gpurun_out/_mesh_onlyB.txt', 'w').write((s[:i6] + s[i7:].replace('OBJECT 7', 'OBJECT 6')).replace('mesh models/', 'mesh ../scenes/models/'))
open('gpurun_out/_mesh_none.txt', 'w').write(s[:i6].rstrip() + "\n")
open('gpurun_out/_mesh_onlyA.txt', 'w').write(open('gpurun_out/_mesh_onlyA.txt').read().replace('mesh models/', 'mesh ../scenes/models/'))
PY
for sc in scenes/cornell_mesh gpurun_out/_mesh_onlyA gpurun_out/_mesh_onlyB gpurun_out/_mesh_none; do
python bench.py --steps 8 --warmup 2 --scene $sc.txt --cpu-spp 0 --batch 32 --per-iteration-sample 0 --repeats 5 2>gpurun_out/r4d.err | python profiles/line_fields.py "$sc batch 32"
done
