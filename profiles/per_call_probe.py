"""What does ONE pt_iterate call of the reference's protocol cost with trace-ahead on?  (python profiles/per_call_probe.py [calls]; under
rocprofv3 --kernel-trace --stats it also gives the single-iteration k_commit launches' durations)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pt = ge.load_package()
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 6400
sc = pt.Scene(os.path.join(ROOT, "scenes", "cornell.txt"))
sc.set_resolution(1280, 720)
for batch, pipe in ((64, 2), (64, 3), (32, 3)):
    pt.pathtraceFree()
    pt.pathtraceInit(sc, traceDepth=8, pipeline_depth=pipe, max_batch=batch, trace_ahead=True)
    for it in range(1, 257):
        pt.pathtrace(None, 0, it, readback=False)
    pt.sync()
    t0 = time.perf_counter()
    for it in range(257, 257 + calls):
        pt.pathtrace(None, 0, it, readback=False)
    t1 = time.perf_counter()
    pt.sync()
    t2 = time.perf_counter()
    print("trace ahead %d x %d slots: %d calls: host loop %.4f ms per call, with the final sync %.4f ms per call = %.1f G nominal paths/s" % (
        batch, pipe, calls, (t1 - t0) / calls * 1e3, (t2 - t0) / calls * 1e3, 1280 * 720 * 8 * calls / (t2 - t0) / 1e9), flush=True)
pt.pathtraceFree()
