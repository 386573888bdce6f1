"""Bounce-kernel time and wall time per wavefront batch for a rank of an N-way row shard (one GPU, no collective) next to the\nunsharded frame with the same number of paths per launch.   python profiles/shard_kernel_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pt = ge.load_package()
scene = pt.Scene(os.path.join(ROOT, "scenes", "cornell.txt")); scene.set_resolution(1280, 720)
for world, batch in ((1, 32), (8, 256), (8, 32), (4, 128)):
    pt.pathtraceFree()
    pt.pathtraceInit(scene, shard_rank=0, shard_count=world, pipeline_depth=1, max_batch=batch, flags=pt.PT_FLAG_KERNEL_TIMING)
    it = 1
    for _ in range(3):
        pt.pathtrace_batch(None, 0, it, batch); it += batch
    pt.sync(); pt.counters_reset()
    t0 = time.perf_counter()
    nb = 8
    for _ in range(nb):
        pt.pathtrace_batch(None, 0, it, batch); it += batch
    pt.sync()
    dt = time.perf_counter() - t0
    c = pt.counters()
    print("world %d batch %d: wall %.3f ms per batch, bounce kernels %.3f ms per batch (%d launches), live %s" % (
        world, batch, dt / nb * 1e3, c.bounce_kernel_ms / nb, c.bounce_launches / nb, [int(c.live[d]) // nb for d in (1, 2, 8)]), flush=True)
pt.pathtraceFree()
