"""The per-iteration device-to-host copy of the reference protocol (src/pathtrace.cu:170-171: 11.06 MB at 1280x720): time of
pt_readback into a pageable and into a page-locked (pt_pin_host) host buffer, and of pt_iterate + pt_readback per iteration.
python profiles/readback_probe.py"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
pt = ge.load_package()
L = pt.lib()
L.pt_pin_host.argtypes = [C.c_void_p, C.c_size_t]
sc = pt.Scene(os.path.join(ROOT, "scenes", "cornell.txt"))
sc.set_resolution(1280, 720)
pt.pathtraceFree()
pt.pathtraceInit(sc, traceDepth=8)
buf = np.zeros(1280 * 720 * 3, np.float32)
for pinned in (False, True):
    if pinned:
        assert L.pt_pin_host(buf.ctypes.data, buf.nbytes) == 0
    pt.pathtrace(None, 0, 1, readback=False); pt.sync()
    L.pt_readback(buf.ctypes.data_as(C.c_void_p))
    t0 = time.perf_counter()
    for _ in range(100):
        L.pt_readback(buf.ctypes.data_as(C.c_void_p))
    dt = (time.perf_counter() - t0) / 100
    print("pt_readback into %s memory: %.3f ms (%.1f GB/s)" % ("page-locked" if pinned else "pageable", dt * 1e3, buf.nbytes / dt / 1e9))
    t0 = time.perf_counter()
    for it in range(2, 102):
        L.pt_iterate(0, it, None)
        L.pt_readback(buf.ctypes.data_as(C.c_void_p))
    dt = (time.perf_counter() - t0) / 100
    print("  pt_iterate + pt_readback: %.3f ms per iteration = %.2f G nominal paths/s" % (dt * 1e3, 1280 * 720 * 8 / dt / 1e9))
L.pt_unpin_host()
pt.pathtraceFree()
# the same protocol served from batches traced ahead (PT_FLAG_TRACE_AHEAD, what the shim's pathtraceInit switches on), page-locked
# buffer; and without the copy: what a caller pays per pt_iterate call
for ahead, depth in ((32, 2), (32, 3), (64, 3)):
    pt.pathtraceInit(sc, traceDepth=8, max_batch=ahead, pipeline_depth=depth, trace_ahead=True)
    assert L.pt_pin_host(buf.ctypes.data, buf.nbytes) == 0
    for it in range(1, 65):
        L.pt_iterate(0, it, None)
    pt.sync()
    n = 1024
    t0 = time.perf_counter()
    for it in range(65, 65 + n):
        L.pt_iterate(0, it, None)
        L.pt_readback(buf.ctypes.data_as(C.c_void_p))
    dt = (time.perf_counter() - t0) / n
    print("trace ahead %d x %d slots: pt_iterate + pt_readback %.3f ms per iteration = %.2f G nominal paths/s" % (ahead, depth, dt * 1e3, 1280 * 720 * 8 / dt / 1e9))
    t0 = time.perf_counter()
    for it in range(65 + n, 65 + 2 * n):
        L.pt_iterate(0, it, None)
    pt.sync()
    dt = (time.perf_counter() - t0) / n
    print("                          pt_iterate alone          %.3f ms per iteration = %.2f G nominal paths/s" % (dt * 1e3, 1280 * 720 * 8 / dt / 1e9))
    L.pt_unpin_host()
    pt.pathtraceFree()
