"""Long soak of the C ABI's device groups on one device: 400 000 iterations of Cornell 1280x720 through pt_group_iterate (one call and one frame
assembly per ITERATION, iterations out of batches of 32 traced ahead, iteration indices up to the seed format's limit) for three group shapes --
the frame's SHA-256 must be the single renderer's (profiles/soak_long.py: ea83831194199eb9 since round 3), the path counts conserved, no fault.
python profiles/soak_group.py [iterations]   (GPU box; ~20-40 s per shape)"""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
pt = ge.load_package()
total = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
sc = pt.Scene(os.path.join(ROOT, "scenes", "cornell.txt"))
sc.set_resolution(1280, 720)
first = 4194303 - total
digests = []
for members, collective, threads in ((1, "rccl", "1"), (8, None, "1"), (3, "rccl", "1"), (2, "rccl", "0")):
    os.environ["PT_AMD_GROUP_THREADS"] = threads
    if collective:
        os.environ["PT_AMD_COLLECTIVE"] = collective
    else:
        os.environ.pop("PT_AMD_COLLECTIVE", None)
    g = pt.Group(members)
    try:
        g.init(sc, traceDepth=8, flags=pt.PT_FLAG_TRACE_AHEAD, max_batch=32, pipeline_depth=2)
        t0 = time.time()
        for it in range(first, first + total):
            g.iterate(it)
            if (it - first) % 100000 == 99999:
                img = g.readback()                      # (a read-back in the middle: the collective stream alone is waited for)
                print("  group of %d [%s]: %d iterations, %.1f s, mean %.4f" % (members, g.collective, it - first + 1, time.time() - t0, float(img.mean())), flush=True)
        g.sync()
        dt = time.time() - t0
        img = g.readback()
        c = g.counters()
        live = [int(c.live[d]) for d in range(1, 10)]
        assert int(c.iterations) == total and all(a >= b for a, b in zip(live, live[1:])), (int(c.iterations), live)
        sha = hashlib.sha256(img.tobytes()).hexdigest()[:16]
        digests.append(sha)
        print("group of %d, threads %s [%s]: %d pt_group_iterate calls in %.1f s = %.4f ms per call = %.1f G nominal paths/s, sha %s"
              % (members, threads, g.collective, total, dt, dt / total * 1e3, 1280 * 720 * 8 * total / dt / 1e9, sha), flush=True)
    finally:
        g.destroy()
assert len(set(digests)) == 1, digests
print("identical" + (" to the single renderer's (rounds 3-6)" if digests[0] == "ea83831194199eb9" and total == 400000 else ""))
