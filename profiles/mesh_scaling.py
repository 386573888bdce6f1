"""The mesh scene with its icosphere at subdivision levels 2 .. 7 (320 .. 327 680 triangles) next to the 2304-triangle torus: throughput at
1280x720, depth 8, batches of 32, three batches in flight -- how the walk scales with the triangle count.      python profiles/mesh_scaling.py
(on the GPU box; PT_AMD_LIB selects another build of the library)"""
import importlib.util, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pt = ge.load_package()
if os.environ.get("PT_AMD_LIB"):
    pt.LIB_PATH = os.environ["PT_AMD_LIB"]
spec = importlib.util.spec_from_file_location("make_scenes", os.path.join(ROOT, "scenes", "make_scenes.py"))
ms = importlib.util.module_from_spec(spec); spec.loader.exec_module(ms)


def icosphere(level):
    v, f = ms.icosphere(level, 0.5)
    v = np.array(v, np.float32)
    f = np.array(f, np.int64)
    return np.concatenate([v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]], axis=1).astype(np.float32)


W, H, D, B = 1280, 720, 8, 32
for level in (2, 3, 4, 5, 6, 7):
    sc = pt.Scene(os.path.join(ROOT, "scenes", "cornell_mesh.txt"))
    sc.set_resolution(W, H)
    tris = icosphere(level)
    sc.meshes = {6: tris, 7: sc.meshes[7]}
    pt.pathtraceFree()
    t0 = time.perf_counter()
    pt.pathtraceInit(sc, traceDepth=D, max_batch=B, pipeline_depth=3)
    t_init = time.perf_counter() - t0
    it = 1
    for _ in range(3):
        pt.pathtrace_batch(None, 0, it, B); it += B
    pt.sync()
    n = 12
    t0 = time.perf_counter()
    for _ in range(n):
        pt.pathtrace_batch(None, 0, it, B); it += B
    pt.sync()
    dt = time.perf_counter() - t0
    print("icosphere level %d: %7d + 2304 triangles  %6.1f G nominal paths/s  (pt_init incl. the hierarchies: %.2f s)"
          % (level, len(tris), W * H * D * B * n / dt / 1e9, t_init), flush=True)
pt.pathtraceFree()
