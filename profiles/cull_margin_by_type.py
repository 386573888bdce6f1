import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/oracle")
import numpy as np
import __graft_entry__ as ge
pt = ge.load_package()
import oracle
from test_gpu_camera_cull import _case
class G: pass
gpu = pt
gpu.LIB_PATH = None
for typ, name in ((1, "cubes"), (0, "spheres")):
    rng = np.random.default_rng(20261004)
    worst, needed, at = 0.0, 0, -1
    for k in range(10500):
        cam, geoms = _case(oracle, rng)
        g = geoms[geoms["type"] == typ]
        if len(g) == 0:
            continue
        w, n = gpu.test_camera_cull_margin(cam.view(gpu.CAMERA_DTYPE), g.view(gpu.GEOM_DTYPE), samples=1)
        needed += n
        if w > worst:
            worst, at = w, k
    print(name, "worst fraction %.4f (case %d), %d hits needed some of the inflation" % (worst, at, needed), flush=True)
