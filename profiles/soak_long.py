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
# ... and the same iterations through the reference's protocol, one pt_iterate call per iteration, served from batches traced
# ahead (PT_FLAG_TRACE_AHEAD): the same frame, bit for bit
pt.pathtraceFree()
pt.pathtraceInit(sc, traceDepth=8, max_batch=32, pipeline_depth=2, trace_ahead=True)
L = pt.lib()
t0 = time.time()
first = 4194303 - total
for it in range(first, first + total):
    rc = L.pt_iterate(0, it, None)
    assert rc == 0, pt.lib().pt_last_error()
    if (it - first + 1) % 100000 == 0:
        print("run 2 (trace-ahead, one call per iteration): %d iterations, %.1f s" % (it - first + 1, time.time() - t0), flush=True)
pt.sync()
dt = time.time() - t0
c = pt.counters()
assert c.iterations == total
img = pt.readback(1280 * 720)
digests.append(hashlib.sha256(img.tobytes()).hexdigest()[:16])
print("run 2: %d pt_iterate calls in %.1f s = %.3f ms per call = %.1f G nominal paths/s, sha %s" % (
    total, dt, dt / total * 1e3, total * 1280 * 720 * 8 / dt / 1e9, digests[-1]), flush=True)
pt.pathtraceFree()
assert digests[0] == digests[1] == digests[2], digests
print("identical")
