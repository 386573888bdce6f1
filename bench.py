#!/usr/bin/env python3
"""bench.py -- headline benchmark of the path-tracing hot path on MI355X.

Metric (BASELINE.json): Mpaths/s, paths = pixels x bounces x spp (NOMINAL segments), Cornell box at
1280x720, 8 bounces.  One "step" = one iteration (1 spp) of the whole frame: camera rays, 8 fused
intersect+shade+compact bounces, ordered accumulation.  Steps are issued as wavefront batches of
--batch iterations (pt_iterate_batch: their paths share the 8 launches; results are identical to
one call per iteration) with 2 batches in flight on internal streams.  Scene, accumulator
and path state are resident in HBM before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (one rank per GPU)

N > 1: image rows are sharded round-robin over the ranks (row y -> rank y % N), every rank accumulates
only its own rows (packed), and ONE RCCL collective per committed batch -- a gather of the row blocks to
rank 0 over xGMI -- assembles the frame (disjoint rows: bit-identical to 1 GPU; it moves 1/N of the bytes the
reduce of zero-padded full frames would).  Fixed total work -> "scaling": "strong".

Prints ONE JSON line on rank 0, with `roofline` (dominant kernel = the fused bounce kernel, HIP-event
timed, against the 8 TB/s HBM peak) and `cpu_baseline` (the single-thread CPU oracle on a bounded
sample of the same workload; N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PATH_BYTES = 44                # SoA PathSegment: origin 12 + dir 12 + throughput 12 + pixelIndex 4 + remainingBounces 4
ACCUM_BYTES = 24               # accumulator read + write of one emitter hit (vec3 fp32)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="iterations timed; default 1024 on one GPU (16 x the 64 spp of BASELINE config C2: the 5.5 ms a "
                         "single 64-spp render takes is too short a timed region), 5000 on N GPUs (C3)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed iterations first (default 128, N GPUs: 256)")
    ap.add_argument("--scene", default=os.path.join(ROOT, "scenes", "cornell.txt"))
    ap.add_argument("--res", type=int, nargs=2, default=[1280, 720])
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--cpu-spp", type=int, default=40, help="spp of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--pipeline", type=int, default=2, help="batches in flight (PtOptions.pipeline_depth; 0 = library default 3)")
    ap.add_argument("--batch", type=int, default=32,
                    help="iterations traced as one wavefront per pt_iterate_batch call on ONE GPU; N GPUs trace N x as many "
                         "(at most 64), so that a launch keeps covering as many paths when the rows are sharded")
    ap.add_argument("--pmc-traffic-json", default=os.path.join(ROOT, "profiles", "pmc_traffic.json"),
                    help="HBM bytes per bounce-kernel launch from a rocprofv3 --pmc run (profiles/README.md)")
    return ap.parse_args()


def cpu_baseline(args, scene):
    """Single-thread CPU oracle (oracle/, kind 'port') on a bounded sample of the same workload."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    W, H = args.res
    ref = orc.Renderer(scene.camera.view(orc.CAMERA_DTYPE), scene.geoms.view(orc.GEOM_DTYPE),
                       scene.materials.view(orc.MATERIAL_DTYPE), args.depth)
    img = np.zeros(W * H * 3, np.float32)
    ref.iterate(1, img)                                    # warm the caches / page in
    t0 = time.perf_counter()
    for it in range(2, 2 + args.cpu_spp):
        ref.iterate(it, img)
    dt = time.perf_counter() - t0
    return {"value": round(W * H * args.depth * args.cpu_spp / dt / 1e6, 3), "unit": "Mpaths/s", "cores": 1,
            "kind": "port",
            "sample": "%d spp of %s %dx%d depth %d (%.1f s, single thread, g++ -O2 -ffp-contract=off, host has %d cores)"
                      % (args.cpu_spp, os.path.basename(args.scene), W, H, args.depth, dt, os.cpu_count())}


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    if args.steps is None:
        args.steps = 1024 if world == 1 else 5000
    if args.warmup is None:
        args.warmup = 128 if world == 1 else 256
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the hot path has no CPU fallback)")
    # BENCH_BACKEND=gloo lets the N > 1 path be rehearsed with several ranks on ONE GPU (RCCL needs one
    # GPU per rank); the real runs use nccl = RCCL over xGMI.
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    device_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    import __graft_entry__ as ge
    pt = ge.load_package()
    ptdist = ge.load_submodule("distributed")
    if world > 1:
        ptdist.init_process_group(backend)

    W, H = args.res
    scene = pt.Scene(args.scene)
    scene.set_resolution(W, H)
    P = W * H
    # accumulator as a torch tensor so RCCL can move it.  N = 1: the full frame.  N > 1: this rank's rows
    # only (packed, padded to the largest shard); rank 0 assembles `frame` from the gathered blocks.
    if world > 1:
        accum = torch.zeros(ptdist.padded_block_floats(W, H, world), dtype=torch.float32, device="cuda")
        frame = torch.zeros(P * 3, dtype=torch.float32, device="cuda") if rank == 0 else None
        bufs = ptdist.make_gather_buffers(accum, world, rank)
        shard_flag = pt.PT_FLAG_ACCUM_SHARD_ROWS
    else:
        accum = torch.zeros(P * 3, dtype=torch.float32, device="cuda")
        frame, bufs, shard_flag = None, None, 0
    stream = torch.cuda.current_stream()

    # path buffers grow with the batch (44 B x 16 class-worst-case x 2 ping-pong x slots per path): keep them
    # under ~128 GB of the 288: batch 32 with 2 slots at 1280x720 (83 GB), 21 at 1080p, 2 for a 4096x4096 frame on one GPU
    n_local = ptdist.local_pixel_count(W, H, rank, world)
    B = max(1, min(args.batch * world, pt.PT_MAX_BATCH,
                   int(128e9 // (max(n_local, 1) * 44 * 16 * 2 * (args.pipeline if args.pipeline > 0 else 3)))))

    def init(flags, pipeline):
        pt.pathtraceFree()
        pt.pathtraceInit(scene, shard_rank=rank, shard_count=world, stream=stream.cuda_stream,
                         accum_dev=accum.data_ptr(), device=device_index, flags=flags | shard_flag,
                         traceDepth=args.depth, pipeline_depth=pipeline, max_batch=B)

    def run_steps(first_iter, steps):
        """`steps` iterations (1 spp each), issued as wavefront batches of up to B iterations; the frame is
        assembled at rank 0 after every batch."""
        it, end = first_iter, first_iter + steps
        while it < end:
            n = min(B, end - it)
            pt.pathtrace_batch(None, 0, it, n)
            if world > 1:
                # the single collective of the data path: gather of the row blocks over xGMI
                ptdist.gather_frame(accum, bufs, frame, W, H, dst=0)
            it += n

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(first_iter, steps):
        barrier()
        t0 = time.perf_counter()
        run_steps(first_iter, steps)
        barrier()
        return time.perf_counter() - t0

    # ---- pass A: the headline number ----------------------------------------------------------
    init(0, args.pipeline)
    run_steps(1, args.warmup)
    barrier()
    pt.counters_reset()
    dt = timed(1 + args.warmup, args.steps)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    cntA = pt.counters()

    # ---- pass B: same steps with HIP events around every launch (roofline of the bounce kernel); one
    #      iteration in flight, so that a launch's duration is the kernel's own and not its share of a GPU it
    #      co-occupies with the neighbouring iterations' launches
    accum.zero_()
    init(pt.PT_FLAG_KERNEL_TIMING, 1)
    run_steps(1, min(args.warmup, B))
    barrier()
    pt.counters_reset()
    dtB = timed(1 + args.warmup, args.steps)
    cnt = pt.counters()
    pt.pathtraceFree()

    D = args.depth
    live = [int(cnt.live[d]) for d in range(D + 2)]
    hits = int(cnt.light_hits)
    # algorithmic HBM bytes of the bounce launches: read every live path (bounce 1 builds its camera
    # rays in registers and reads nothing), write every survivor (nothing is written after the last
    # bounce), read+write the accumulator for every emitter hit
    bounce_bytes = sum(PATH_BYTES * live[d] for d in range(2, D + 1)) \
        + sum(PATH_BYTES * live[d + 1] for d in range(1, D)) + ACCUM_BYTES * hits
    launches = max(int(cnt.bounce_launches), 1)
    avg_ms = cnt.bounce_kernel_ms / launches
    achieved = bounce_bytes / launches / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    # PMC counters of the bounce kernel from the committed rocprofv3 passes -- only when they were collected on this
    # very configuration (frame, depth and iterations per launch), else null
    traffic, pmc = None, {}
    try:
        pmc = json.load(open(args.pmc_traffic_json))
        here = ["%s %dx%d" % (os.path.relpath(args.scene, ROOT), W, H), " %d bounces" % D]
        if world == 1 and pmc.get("workload") == here and pmc.get("iterations_per_wavefront_batch") == B:
            traffic = pmc.get("hbm_bytes_per_bounce_launch")
        else:
            pmc = {}
    except Exception:
        pmc = {}

    if rank == 0:
        nominal = P * D * args.steps
        out = {
            "metric": "Mpaths/sec (paths = pixels x bounces x spp) at 1280x720, 8 bounces",
            "value": round(nominal / dt / 1e6, 2), "unit": "Mpaths/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s %dx%d, %d spp, %d bounces%s" % (
                           os.path.relpath(args.scene, ROOT), W, H, args.steps, D,
                           "" if world == 1 else ", rows sharded y%%%d + RCCL gather of the row blocks per batch" % world),
                       "paths_per_step_nominal": P * D,
                       "iterations_per_wavefront_batch": B,
                       "batches_in_flight": args.pipeline if args.pipeline > 0 else 3,
                       "live_segments_per_step": round(sum(live[1:D + 1]) / max(args.steps, 1), 1),
                       "live_Msegments_per_s": round(sum(int(cntA.live[d]) for d in range(1, D + 1)) / dt / 1e6, 2)},
            "roofline": {"bound": "hbm", "kernel": "k_bounce (fused [camera rays+]intersect+shade+compact, one launch per bounce)",
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": traffic,
                         "algorithmic_bytes_per_launch": round(bounce_bytes / launches, 1),
                         "avg_launch_ms": round(avg_ms, 5), "launches": launches,
                         "bounce_kernel_share_of_step": round(cnt.bounce_kernel_ms / (dtB * 1e3), 4),
                         "ms_per_step_with_events": round(dtB / args.steps * 1e3, 4),
                         # the kernel is VALU-issue-bound, not HBM-bound (profiles/): VALU
                         # wave-instructions per launch from the PMC pass x 4 cycles / (1024 SIMDs x 2.4 GHz)
                         "valu_issue_bound_ms_per_launch": pmc.get("valu_issue_bound_ms_per_launch"),
                         "lds_bank_conflict_cycles_per_launch": pmc.get("lds_bank_conflict_cycles_per_launch")},
        }
        if world == 1 and args.cpu_spp > 0:
            out["cpu_baseline"] = cpu_baseline(args, scene)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
