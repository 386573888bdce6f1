"""What does a ONE-rank RCCL reduce of an 11 MB frame launch on the device?   (under rocprofv3 --kernel-trace --stats, on the GPU box)
The C3-as-written reading at N = 1 issues one per iteration (distributed.PerIterationReducer, always_collective)."""
import os, sys
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
buf = torch.zeros(1280 * 720 * 3, device="cuda")
mode = sys.argv[1] if len(sys.argv) > 1 else "reduce"
for _ in range(50):
    if mode == "reduce": dist.reduce(buf, dst=0)
    elif mode == "allreduce": dist.all_reduce(buf)
torch.cuda.synchronize()
dist.destroy_process_group()
print("done", mode)
