"""Where a tile's TIME goes (instrumented build, NOT the product library):
    make -C project3-cuda-path-tracer_amd/csrc timeline && python profiles/timeline_phases.py
Wave 0 of every workgroup stamps s_memtime at the tile-level marks of k_bounce; printed per phase: shader cycles per visit (wave 0's
wall time between the mark and the next one, whatever the wave did or waited for in between) and the share of the tile."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pt = ge.load_package()
pt.LIB_PATH = os.path.join(ROOT, "project3-cuda-path-tracer_amd", "csrc", "libpt_amd_timeline.so")   # before the first call
L = pt.lib()
L.pt_probe_timeline.argtypes = [C.POINTER(C.c_uint64)]
names = {14: "tile start: set-up / camera ray", 15: "nearest-hit loop", 24: "sphere-heavy: sweep of the packed spheres", 25: "sphere-heavy: the candidates' passes", 16: "after the loop -> a hit's record", 9: "shading: hit record, normal, material",
         10: "scatter: engine, branch by material", 11: "hemisphere sample", 12: "bounding-ball certificates", 13: "wall certificates, class", 17: "next tile: segment look-up (LDS)",
         26: "next tile: chunk look-up (scalar cache)", 27: "next tile: 3 loads issued",
         18: "compaction: ballots, ranks", 21: "wait at the first barrier", 22: "reservation (atomic round trip) / waves 1-3 idle", 23: "wait at the second barrier",
         19: "stores", 20: "tile end -> next tile start", 30: "prologue (staging, scan of the segment counts)"}
order = [30, 14, 15, 24, 25, 16, 9, 10, 11, 12, 13, 17, 26, 27, 18, 21, 22, 23, 19, 20]
scene = sys.argv[1] if len(sys.argv) > 1 else "cornell.txt"
res = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1280, 720)
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 32
for depth, pipeline in ((8, 2),):
    sc = pt.Scene(os.path.join(ROOT, "scenes", scene))
    sc.set_resolution(*res)
    pt.pathtraceFree()
    pt.pathtraceInit(sc, traceDepth=depth, max_batch=batch, pipeline_depth=pipeline)
    out = (C.c_uint64 * 128)()
    pt.pathtrace_batch(None, 0, 1, batch)
    pt.sync()
    L.pt_probe_timeline(out)                                  # (warm-up discarded)
    for it in range(1 + batch, 1 + batch + 8 * batch, batch):
        pt.pathtrace_batch(None, 0, it, batch)
    pt.sync()
    L.pt_probe_timeline(out)
    v = [int(x) for x in out]
    for kind, off in (("later bounces", 0), ("camera-ray bounce", 32)):
        T, N = v[off:off + 32], v[64 + off:64 + off + 32]
        tiles = max(N[14], 1)
        total = sum(T[k] for k in order if k != 30)
        print("%s %dx%d depth %d, %d batch(es) in flight, %s: %d tiles stamped (wave 0 of each workgroup), %.0f cycles per tile" % (
            scene, res[0], res[1], depth, pipeline, kind, tiles, total / tiles))
        for k in order:
            if N[k]:
                print("  %-52s %9.0f cycles per visit x %8d visits  %5.1f %% of the tiles' time" % (names[k], T[k] / N[k], N[k], 100.0 * T[k] / max(total, 1) if k != 30 else 0.0))
pt.pathtraceFree()
