"""Wave-level steps of the mesh walk (instrumented build, NOT the product library):
    make -C project3-cuda-path-tracer_amd/csrc probe && python profiles/probe_mesh.py [scene]
Prints how many times a wave ran each part of ptd::meshIntersectionTest and with how many of its 64 lanes."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pt = ge.load_package()
pt.LIB_PATH = os.path.join(ROOT, "project3-cuda-path-tracer_amd", "csrc", "libpt_amd_probe.so")   # before the first call
L = pt.lib()
L.pt_probe_read.argtypes = [C.POINTER(C.c_uint64)]
scene = sys.argv[1] if len(sys.argv) > 1 else "cornell_mesh.txt"
sc = pt.Scene(os.path.join(ROOT, "scenes", scene))
sc.set_resolution(1280, 720)
pt.pathtraceFree()
pt.pathtraceInit(sc, traceDepth=8, max_batch=8, pipeline_depth=3)
out = (C.c_uint64 * 64)()
L.pt_probe_read(out)
for it in range(1, 33, 8):
    pt.pathtrace_batch(None, 0, it, 8)
pt.sync()
L.pt_probe_read(out)
v = [int(x) for x in out]
paths = 32 * 1280 * 720
print("%s 1280x720 depth 8, 32 iterations = %.1f M camera rays" % (scene, paths / 1e6))
for k, what in ((21, "results folded in (lanes = finished walks)"), (26, "hand-outs (lanes = jobs taken)"), (22, "inner-node steps"), (24, "triangle steps"), (29, "loop turns with lanes at an inner node"), (30, "loop turns with lanes at a triangle"),
                (31, "loop turns with idle lanes"), (23, "loop turns with nothing left to hand out"), (27, "queueing: quarter-tile visits"),
                (28, "queueing: (quarter, mesh) with jobs (lanes = jobs)"), (7, "tiles of the later bounces (waves)"), (8, "tiles of the camera bounce (waves)")):
    w, l = v[2 * k], v[2 * k + 1]
    print("  %-58s %11d wave executions, %5.1f lanes each (%.2f per camera ray)" % (what, w, l / max(w, 1), l / paths))
pt.pathtraceFree()
