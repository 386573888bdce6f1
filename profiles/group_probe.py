"""pt_group on ONE device: what VERDICT round 5 item 1 asks to be measured.

  (a) config C2 in batch mode (steps of 64 iterations of 1280x720, depth 8) by ONE context and by groups of 2 / 4 / 8 members that share
      the device, with a host thread per member and without (PT_AMD_GROUP_THREADS), pipeline_depth 1 / 2 / 3 per member: G paths/s and
      the HOST's enqueue time per step (the calls alone, nothing waited for inside the loop);
  (b) config C3 as written (one call + one frame assembly per ITERATION, iterations out of batches of 64 traced ahead) through the
      LIBRARY's own collective -- a group of one member, PT_AMD_COLLECTIVE=rccl: snapshot -> ncclReduce in a one-rank communicator on the
      collective stream -- against pt_iterate alone: ms per iteration, host enqueue per iteration.

    python profiles/group_probe.py [--steps 12] [--iters 2048]        (GPU box; one JSON-ish line per measurement)
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=16)
ap.add_argument("--iters", type=int, default=2048)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--only", default="")
args = ap.parse_args()

pt = ge.load_package()
W, H, D, B = 1280, 720, 8, args.batch
sc = pt.Scene(os.path.join(ROOT, "scenes", "cornell.txt"))
sc.set_resolution(W, H)
P = W * H


def rate(seconds, iters):
    return P * D * iters / seconds / 1e9


def one_context(pipeline):
    pt.pathtraceFree()
    pt.pathtraceInit(sc, traceDepth=D, pipeline_depth=pipeline, max_batch=B)
    it = 1
    for _ in range(3):
        pt.pathtrace_batch(None, 0, it, B)
        it += B
    pt.sync()
    best = None
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(args.steps):
            pt.pathtrace_batch(None, 0, it, B)
            it += B
        t1 = time.perf_counter()
        pt.sync()
        t2 = time.perf_counter()
        if best is None or t2 - t0 < best[1]:
            best = (t1 - t0, t2 - t0)
    pt.pathtraceFree()
    print("one context, pipeline %d: %.1f G paths/s, host enqueue %.1f us per step of %d" % (pipeline, rate(best[1], args.steps * B), best[0] / args.steps * 1e6, B), flush=True)
    return rate(best[1], args.steps * B)


def group(members, pipeline, threads, collective=None, reduce_every_step=False, fuse=1):
    """`fuse`: the members trace `fuse` steps as ONE wavefront batch (a member of 8 carries 1/8 of a step's paths per launch: bench.py's ranks
    fuse steps the same way), the frame reduced per fused batch"""
    global B
    B0, B = B, B * fuse
    try:
        return _group(members, pipeline, threads, collective, reduce_every_step, fuse)
    finally:
        B = B0


def _group(members, pipeline, threads, collective, reduce_every_step, fuse):
    os.environ["PT_AMD_GROUP_THREADS"] = "1" if threads else "0"
    if collective:
        os.environ["PT_AMD_COLLECTIVE"] = collective
    else:
        os.environ.pop("PT_AMD_COLLECTIVE", None)
    g = pt.Group(members)
    try:
        g.init(sc, traceDepth=D, pipeline_depth=pipeline, max_batch=B)
        it = 1
        for _ in range(3):
            g.iterate_batch(it, B)
            it += B
        g.sync()
        best = None
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(max(1, args.steps // fuse)):
                g.iterate_batch(it, B)
                if reduce_every_step:
                    g.reduce()
                it += B
            t1 = time.perf_counter()
            g.sync()
            t2 = time.perf_counter()
            if best is None or t2 - t0 < best[1]:
                best = (t1 - t0, t2 - t0)
        how = g.collective
    finally:
        g.destroy()
    nb = max(1, args.steps // fuse)
    print("group of %d on one device, pipeline %d, threads %d%s [%s]: %.1f G paths/s, host enqueue %.1f us per batch of %d"
          % (members, pipeline, threads, ", reduce per batch" if reduce_every_step else "", how, rate(best[1], nb * B), best[0] / nb * 1e6, B), flush=True)
    return rate(best[1], nb * B)


def c3(members, collective, threads=1, pipeline=2):
    os.environ["PT_AMD_GROUP_THREADS"] = "1" if threads else "0"
    if collective:
        os.environ["PT_AMD_COLLECTIVE"] = collective
    else:
        os.environ.pop("PT_AMD_COLLECTIVE", None)
    g = pt.Group(members)
    try:
        g.init(sc, traceDepth=D, flags=pt.PT_FLAG_TRACE_AHEAD, pipeline_depth=pipeline, max_batch=B)
        it = 1
        for _ in range(256):
            g.iterate(it)
            it += 1
        g.sync()
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(args.iters):
                g.iterate(it)
                it += 1
            t1 = time.perf_counter()
            g.sync()
            t2 = time.perf_counter()
            if best is None or t2 - t0 < best[1]:
                best = (t1 - t0, t2 - t0)
        how = g.collective
    finally:
        g.destroy()
    print("C3 as written, group of %d, threads %d, pipeline %d [%s]: %.4f ms per iteration (%.1f G paths/s), host enqueue %.1f us per iteration"
          % (members, threads, pipeline, how, best[1] / args.iters * 1e3, rate(best[1], args.iters), best[0] / args.iters * 1e6), flush=True)


def c3_plain(pipeline=2):
    pt.pathtraceFree()
    pt.pathtraceInit(sc, traceDepth=D, flags=pt.PT_FLAG_TRACE_AHEAD, pipeline_depth=pipeline, max_batch=B)
    it = 1
    for _ in range(256):
        pt.pathtrace(None, 0, it, readback=False)
        it += 1
    pt.sync()
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(args.iters):
            pt.pathtrace(None, 0, it, readback=False)
            it += 1
        t1 = time.perf_counter()
        pt.sync()
        t2 = time.perf_counter()
        if best is None or t2 - t0 < best[1]:
            best = (t1 - t0, t2 - t0)
    pt.pathtraceFree()
    print("pt_iterate alone (trace-ahead, no frame assembly), pipeline %d: %.4f ms per iteration (%.1f G paths/s), host enqueue %.1f us per iteration"
          % (pipeline, best[1] / args.iters * 1e3, rate(best[1], args.iters), best[0] / args.iters * 1e6), flush=True)


if args.only == "g8":
    base = one_context(3)
    for pl, fuse in ((1, 1), (2, 1), (2, 2), (2, 4), (3, 4), (1, 4)):
        v = group(8, pl, 1, fuse=fuse)
        print("    = %.3f of the one-context rate" % (v / base), flush=True)
    v = group(8, 2, 0, fuse=4)
    print("    = %.3f of the one-context rate" % (v / base), flush=True)
    v = group(8, 2, 1, collective="rccl", reduce_every_step=True, fuse=4)
    print("    = %.3f of the one-context rate" % (v / base), flush=True)
    sys.exit(0)
if args.only == "c3":
    c3_plain()
    c3(1, "rccl")
    c3(1, None)
    sys.exit(0)
if "a" in (args.only or "ab"):
    base = one_context(3)
    one_context(2)
    for m, pl, th in ((8, 1, 1), (8, 2, 1), (8, 3, 1), (8, 2, 0), (8, 1, 0), (4, 2, 1), (2, 3, 1), (1, 3, 1)):
        v = group(m, pl, th)
        print("    = %.3f of the one-context rate" % (v / base), flush=True)
    group(8, 2, 1, collective="rccl", reduce_every_step=True)
    group(1, 3, 1, collective="rccl", reduce_every_step=True)
if "b" in (args.only or "ab"):
    c3_plain()
    c3(1, "rccl")
    c3(1, None)
    c3(1, "rccl", pipeline=3)
    c3(8, None, threads=1)
    c3(8, None, threads=0)
    c3(8, "rccl", threads=1)
