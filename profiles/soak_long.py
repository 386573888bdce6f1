"""Long soak: 2 x 400 000 iterations of Cornell 1280x720 (32 per wavefront batch, two in flight, iteration indices up to the
seed format's limit region) -- the two runs must give the same SHA-256, the path counts must be conserved, no device fault.
python profiles/soak_long.py [iterations]   (GPU box; ~20 s per run)"""
import os, sys, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
pt = ge.load_package()
total = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
sc = pt.Scene(os.path.join(ROOT, "scenes", "cornell.txt"))
sc.set_resolution(1280, 720)
digests = []
for run in range(2):
    pt.pathtraceFree()
    pt.pathtraceInit(sc, traceDepth=8, max_batch=32, pipeline_depth=2)
    t0 = time.time()
    first = 4194303 - total          # ends at the largest iteration index the seed format allows (pathtrace.cu:43)
    it = first
    while it < first + total:
        n = min(32, first + total - it)
        pt.pathtrace_batch(None, 0, it, n)
        it += n
        if (it - first) % 100000 < 32:
            print("run %d: %d iterations, %.1f s" % (run, it - first, time.time() - t0), flush=True)
    pt.sync()
    dt = time.time() - t0
    c = pt.counters()
    live = [int(c.live[d]) for d in range(1, 10)]
    assert live[0] == total * 1280 * 720 and all(a >= b for a, b in zip(live, live[1:])), live
    img = pt.readback(1280 * 720)
    digests.append(hashlib.sha256(img.tobytes()).hexdigest()[:16])
    print("run %d: %d iterations in %.1f s = %.1f G nominal paths/s, mean %.3f, sha %s" % (
        run, total, dt, total * 1280 * 720 * 8 / dt / 1e9, float(img.mean()) / total, digests[-1]), flush=True)
pt.pathtraceFree()
assert digests[0] == digests[1], digests
print("identical")
