"""C3 as written at N = 1: how much of an iteration is the HOST's time?  The same loop as bench.py's value_c3_as_written (one pt_iterate, one snapshot,
one one-rank RCCL reduce per iteration; batches of 64 traced ahead), timed twice: enqueue time (the host's calls alone, no synchronisation inside the
loop) and total.      python profiles/c3_host_probe.py      (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
import __graft_entry__ as ge
pt = ge.load_package()
ptdist = __import__("importlib").import_module(pt.__name__ + ".distributed")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29535")
dist.init_process_group("nccl", rank=0, world_size=1)
sc = pt.Scene(os.path.join(ROOT, "scenes", "cornell.txt")); sc.set_resolution(1280, 720)
accum = torch.zeros(1280 * 720 * 3, device="cuda")
pt.pathtraceInit(sc, traceDepth=8, max_batch=64, pipeline_depth=2, trace_ahead=True, accum_dev=accum.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
N = 2048
for mode in ("pt_iterate alone", "+ snapshot", "+ snapshot + reduce (the protocol)"):
    red = ptdist.PerIterationReducer(accum, dst=0, always_collective=mode.endswith("(the protocol)"))
    it0 = 1 if mode.startswith("pt_") else (1 + N * (2 if mode.startswith("+ snapshot +") else 1)) + 256 * 0
    # (iterations go on from where the last mode stopped: the sequence must not break)
    start = getattr(sys.modules[__name__], "_next", 1)
    for it in range(start, start + 128):
        pt.pathtrace(None, 0, it, readback=False)
        if not mode.startswith("pt_"): red.collect()
    red.finish(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(start + 128, start + 128 + N):
        pt.pathtrace(None, 0, it, readback=False)
        if not mode.startswith("pt_"): red.collect()
    t1 = time.perf_counter()
    red.finish(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    sys.modules[__name__]._next = start + 128 + N
    print("%-36s host enqueue %.1f us per iteration, total %.1f us per iteration" % (mode, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6), flush=True)
pt.pathtraceFree()
dist.destroy_process_group()
