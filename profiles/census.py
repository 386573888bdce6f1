"""Residency census of k_bounce (instrumented build: make -C project3-cuda-path-tracer_amd/csrc probe): how many of its
workgroups were ever resident together on a CU -- what the register counts promise (78 SGPRs / <= 64 VGPRs: 8) against what the
hardware admitted.    python profiles/census.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pt = ge.load_package()
pt.LIB_PATH = os.path.join(ROOT, "project3-cuda-path-tracer_amd", "csrc", "libpt_amd_probe.so")
L = pt.lib()
L.pt_probe_census.argtypes = [C.POINTER(C.c_uint32)]
out = (C.c_uint32 * 16)()
for scene, res, depth, pipe in (("cornell.txt", (1280, 720), 8, 1), ("cornell.txt", (1280, 720), 8, 2), ("spheres64.txt", (2048, 2048), 8, 1),
                                ("cornell_mesh.txt", (1280, 720), 8, 1)):
    sc = pt.Scene(os.path.join(ROOT, "scenes", scene))
    sc.set_resolution(*res)
    pt.pathtraceFree()
    pt.pathtraceInit(sc, traceDepth=depth, max_batch=8, pipeline_depth=pipe)
    L.pt_probe_census(out)
    for it in range(1, 65, 8):
        pt.pathtrace_batch(None, 0, it, 8)
    pt.sync()
    L.pt_probe_census(out)
    hist = {k: int(out[k]) for k in range(16) if out[k]}
    print("%s %dx%d, %d batch(es) in flight: CUs by most workgroups of k_bounce resident together: %s" % (scene, res[0], res[1], pipe, hist))
pt.pathtraceFree()
