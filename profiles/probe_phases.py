"""Wave-level executions of the intersection phases (instrumented build, NOT the product library):
    make -C project3-cuda-path-tracer_amd/csrc probe && python profiles/probe_phases.py
Prints, per non-first bounce wave (tile of 64 paths): how many times each phase ran and with how many lanes."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pt = ge.load_package()
pt.LIB_PATH = os.path.join(ROOT, "project3-cuda-path-tracer_amd", "csrc", "libpt_amd_probe.so")   # before the first call
L = pt.lib()
L.pt_probe_read.argtypes = [C.POINTER(C.c_uint64)]
names = ["box: transform + early miss", "box: normalize + slabs", "box: hit phase", "sphere: cull test",
         "sphere: transform + radicand", "sphere: roots", "sphere: hit phase", "tile (later bounces)",
         "tile (camera bounce, in scene)", "shading (a hit)", "scatter", "hemisphere sample", "bounding-ball certificate", "wall certificate"]
SCENES = (("cornell.txt", (1280, 720), 1), ("cornell.txt", (1280, 720), 8), ("cornell_glass.txt", (1920, 1080), 16), ("spheres64.txt", (1024, 1024), 8), ("cornell_mesh.txt", (1280, 720), 8),
          ("cubes64.txt", (1024, 1024), 8), ("spheres512.txt", (1024, 1024), 8))
if len(sys.argv) > 1: SCENES = tuple(x for x in SCENES if x[0] in sys.argv[1:])
for scene_name, res, depth in SCENES:
    sc = pt.Scene(os.path.join(ROOT, "scenes", scene_name))
    sc.set_resolution(*res)
    pt.pathtraceFree()
    pt.pathtraceInit(sc, traceDepth=depth, max_batch=8, pipeline_depth=3)
    out = (C.c_uint64 * 64)()
    L.pt_probe_read(out)
    for it in range(1, 33, 8):
        pt.pathtrace_batch(None, 0, it, 8)
    pt.sync()
    L.pt_probe_read(out)
    v = [int(x) for x in out]
    tiles = max(v[14], 1)
    print("%s %dx%d depth %d: %d wave-tiles after the first bounce (%.1f valid paths per wave); the FIRST bounce's phases are included in the counts below" % (scene_name, res[0], res[1], depth, v[14], v[15] / tiles))
    for k in list(range(7)) + list(range(8, 14)):
        print("  %-30s %10d wave executions, %5.1f active lanes each" % (names[k], v[2 * k], v[2 * k + 1] / max(v[2 * k], 1)))
    for k, what in ((16, "box fast path: runs"), (17, "  guards fail (range, NaN)"), (18, "  hit / miss within the margin"), (19, "  a hit's axis within the margin"), (20, "  falls back on the exact loop")):
        print("  %-30s %10d wave executions, %5.1f lanes each" % (what, v[2 * k], v[2 * k + 1] / max(v[2 * k], 1)))
pt.pathtraceFree()
