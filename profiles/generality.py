"""Scenes outside the Cornell shape: bit-identity with the oracle at a small frame, then throughput at 1280x720.
    python profiles/generality.py            (on the GPU box)"""
import json, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as ge
pt = ge.load_package()
import oracle as orc
SCENES = ["cornell.txt", "room_tilted.txt", "cubes64.txt", "spheres64.txt", "spheres512.txt"]
for name in SCENES:
    sc = pt.Scene(os.path.join(ROOT, "scenes", name)); sc.set_resolution(96, 64)
    ref = orc.Renderer(sc.camera.view(orc.CAMERA_DTYPE), sc.geoms.view(orc.GEOM_DTYPE), sc.materials.view(orc.MATERIAL_DTYPE), sc.traceDepth)
    want = np.zeros(96 * 64 * 3, np.float32)
    for it in (1, 2): ref.iterate(it, want)
    pt.pathtraceFree(); pt.pathtraceInit(sc, max_batch=2); pt.pathtrace_batch(None, 0, 1, 2); got = pt.readback(96 * 64); pt.pathtraceFree()
    same = bool(np.array_equal(got.view(np.uint32), want.view(np.uint32)))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--scene", os.path.join(ROOT, "scenes", name), "--steps", "3", "--warmup", "1", "--repeats", "3",
                        "--batch", "16", "--cpu-spp", "0", "--per-iteration-sample", "0", "--configs", "0"], capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print("%-16s oracle-identical %s  bench FAILED: %s" % (name, same, r.stderr.strip().splitlines()[-1:] )); continue
    d = json.loads(line[-1])
    print("%-16s %4d primitives  oracle-identical %s  %9.1f Mpaths/s  %.3f ms per launch  live segments per iteration %.0f  hbm frac %.3f"
          % (name, len(sc.geoms), same, d["value"], d["roofline"]["avg_launch_ms"], d["config"]["live_segments_per_iteration"], d["roofline"]["frac"]), flush=True)
