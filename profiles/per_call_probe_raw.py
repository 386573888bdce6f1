import os, sys, time
ROOT = "/root/repo"
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pt = ge.load_package()
sc = pt.Scene(os.path.join(ROOT, "scenes", "cornell.txt")); sc.set_resolution(1280, 720)
L = pt.lib()
for label, res in (("1280x720", (1280, 720)), ("64x64", (64, 64))):
    sc.set_resolution(*res)
    pt.pathtraceFree()
    pt.pathtraceInit(sc, traceDepth=8, pipeline_depth=2, max_batch=64, trace_ahead=True)
    for it in range(1, 257): L.pt_iterate(0, it, None)
    pt.sync()
    n = 6400
    t0 = time.perf_counter()
    for it in range(257, 257 + n): L.pt_iterate(0, it, None)
    t1 = time.perf_counter(); pt.sync(); t2 = time.perf_counter()
    print(label, "raw ctypes pt_iterate: host %.4f ms per call, with sync %.4f" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3), flush=True)
pt.pathtraceFree()
