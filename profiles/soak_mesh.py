"""Soak of the mesh walk kernel: 65 536 iterations of scenes/cornell_mesh.txt (1280x720, depth 8) under two schedules -- batches of 64, three in flight;
batches of 24, two in flight -- must give the same frame bit for bit, with no device fault.      python profiles/soak_mesh.py   (GPU box)"""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pt = ge.load_package()
sc = pt.Scene(os.path.join(ROOT, "scenes", "cornell_mesh.txt")); sc.set_resolution(1280, 720)
total = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
digests = []
for batch, pipe in ((64, 3), (24, 2)):
    pt.pathtraceFree()
    pt.pathtraceInit(sc, traceDepth=8, max_batch=batch, pipeline_depth=pipe)
    t0 = time.perf_counter()
    it = 1
    while it <= total:
        n = min(batch, total - it + 1)
        pt.pathtrace_batch(None, 0, it, n)
        it += n
    img = pt.readback(1280 * 720)
    dt = time.perf_counter() - t0
    digests.append(hashlib.sha256(img.tobytes()).hexdigest()[:16])
    print("batch %d x %d in flight: %d iterations in %.1f s = %.1f G nominal paths/s, sha256 %s, mean %.6f" %
          (batch, pipe, total, dt, 1280 * 720 * 8 * total / dt / 1e9, digests[-1], float(img.mean())), flush=True)
pt.pathtraceFree()
assert len(set(digests)) == 1, digests
print("identical")
