"""One-off soak: the same 192 iterations of Cornell 1280x720 under very different schedules must give the same bits.
python profiles/soak_determinism.py   (GPU box)"""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as ge
pt = ge.load_package()
ref = None
for scene_name, res, depth in (("cornell.txt", (1280, 720), 8), ("cornell_glass.txt", (960, 540), 16), ("spheres64.txt", (512, 512), 8),
                               ("cornell_mesh.txt", (640, 360), 8)):
    sc = pt.Scene(os.path.join(ROOT, "scenes", scene_name))
    sc.set_resolution(*res)
    digests = []
    for batch, pipe in ((1, 1), (1, 3), (7, 2), (32, 2), (64, 3), (13, 4)):
        pt.pathtraceFree()
        pt.pathtraceInit(sc, traceDepth=depth, max_batch=batch, pipeline_depth=pipe)
        it, total = 1, 192
        while it <= total:
            n = min(batch, total - it + 1)
            pt.pathtrace_batch(None, 0, it, n)
            it += n
        img = pt.readback(res[0] * res[1])
        digests.append(hashlib.sha256(img.tobytes()).hexdigest()[:16])
        print(scene_name, "batch", batch, "pipeline", pipe, digests[-1], "mean", float(img.mean()), flush=True)
    assert len(set(digests)) == 1, digests
pt.pathtraceFree()
print("identical under all schedules")
