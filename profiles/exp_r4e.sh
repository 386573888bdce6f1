python -m pytest tests/test_gpu_mesh.py -x -q -m gpu > gpurun_out/r4e_mesh_tests.log 2>&1; tail -3 gpurun_out/r4e_mesh_tests.log
python bench.py --steps 8 --warmup 2 --scene scenes/cornell_mesh.txt --cpu-spp 0 --batch 32 --per-iteration-sample 0 --repeats 5 2>gpurun_out/r4e.err | python profiles/line_fields.py "mesh batch 32"
