"""Which device copies does ONE iteration of the C3-as-written protocol issue?  (under rocprofv3 --kernel-trace, on the GPU box)
    mode 0: pt_iterate alone      1: + the snapshot (torch copy_)      2: + the one-rank RCCL reduce of the snapshot"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
import __graft_entry__ as ge
pt = ge.load_package()
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 2
if mode >= 2:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
    dist.init_process_group("nccl", rank=0, world_size=1)
sc = pt.Scene(os.path.join(ROOT, "scenes", "cornell.txt")); sc.set_resolution(1280, 720)
accum = torch.zeros(1280 * 720 * 3, device="cuda")
pt.pathtraceInit(sc, traceDepth=8, max_batch=64, pipeline_depth=2, trace_ahead=True, accum_dev=accum.data_ptr())
snap = [torch.empty_like(accum) for _ in range(2)]
work = [None, None]
N = 256
for it in range(1, N + 1):
    pt.pathtrace(None, 0, it, readback=False)
    if mode >= 1:
        k = it & 1
        if work[k] is not None: work[k].wait()
        snap[k].copy_(accum, non_blocking=True)
        if mode >= 2: work[k] = dist.reduce(snap[k], dst=0, async_op=True)
torch.cuda.synchronize()
pt.pathtraceFree()
print("done mode", mode, "iterations", N)
