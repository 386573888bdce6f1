"""Per-rank cost of an N-way row shard, measured on ONE GPU without any collective: an upper bound for
the strong-scaling curve (the driver runs the real 8-GPU bench).  python profiles/shard_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pt = ge.load_package()
scene = pt.Scene(os.path.join(ROOT, "scenes", "cornell.txt"))
scene.set_resolution(1280, 720)
P = 1280 * 720
for world in (1, 2, 4, 8):
    # what bench.py gives a rank: batches of 32 x min(N, 8) iterations, two in flight (and the old cap of 128 for comparison)
    for depth_pipe, batch in ((2, 32), (2, min(32 * world, 128)), (2, 32 * world)):
        acc = torch.zeros(P * 3, device="cuda")
        pt.pathtraceFree()
        pt.pathtraceInit(scene, shard_rank=0, shard_count=world, stream=torch.cuda.current_stream().cuda_stream,
                         accum_dev=acc.data_ptr(), pipeline_depth=depth_pipe, max_batch=batch)
        for it in range(1, 1 + 4 * batch, batch):
            pt.pathtrace_batch(None, 0, it, batch)
        torch.cuda.synchronize()
        N = 1024
        t0 = time.perf_counter()
        for it in range(100, 100 + N, batch):
            pt.pathtrace_batch(None, 0, it, batch)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        us = (t2 - t0) / N * 1e6
        print("shard 1/%d pipeline %d batch %d: enqueue %.1f us/iter, total %.1f us/iter -> whole-job bound %.1f Gpaths/s (x%d ranks)"
              % (world, depth_pipe, batch, (t1 - t0) / N * 1e6, us, P * 8 / us / 1e3 * world, world), flush=True)
pt.pathtraceFree()
