#!/usr/bin/env python3
"""bench.py -- headline benchmark of the path-tracing hot path on MI355X.

Metric (BASELINE.json): Mpaths/s, paths = pixels x bounces x spp (NOMINAL segments), Cornell box at
1280x720, 8 bounces.

One STEP = one pass of the hot path over one batch of input = `--batch` (32) consecutive iterations (spp) of the
whole frame: camera rays, 8 fused intersect+shade+compact bounces, ordered accumulation, issued as ONE wavefront
batch (pt_iterate_batch: the 32 iterations' paths share the 8 launches; results are identical to one call per
iteration), with 2 batches in flight on internal streams.  So `--steps 20` times 640 iterations, and
ms_per_step x steps is the timed wall.  Scene, accumulator and path state are resident in HBM before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W]
        N > 1 without a torchrun environment: bench.py starts its N ranks itself (a child `python -m
        torch.distributed.run`, before this process has touched the GPU) and relays rank 0's JSON line
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (what the driver does)

N > 1: image rows are sharded round-robin over the ranks (row y -> rank y % N), every rank accumulates only its
own rows (packed), and ONE collective -- a gather of the row blocks to rank 0, RCCL over xGMI -- assembles the frame
(disjoint rows: bit-identical to 1 GPU; it moves 1/N of the bytes the reduce of zero-padded full frames would;
`--collective reduce` runs that reduce instead).  `--collective-every batch` (default) issues it after every
committed wavefront batch, `--collective-every 1` after every single iteration (BASELINE config C3 as written:
one pt_iterate + one collective per iteration).  The default run also times a bounded sample of the per-iteration
mode and reports it in config.per_iteration_collective.
`--scaling weak` (default): a step on N GPUs is `--batch` x N iterations of the whole frame -- every rank traces its 1/N of
the rows for N times the iterations, as many paths per step as the single GPU (a renderer's weak scaling: N times the
samples per pixel in the same time); value = all ranks' paths / the slowest rank's time.  `--scaling strong`: a step is
`--batch` iterations whatever N (config C3 as written: a fixed number of samples, divided; ranks then fuse steps into
fatter wavefront batches).  At N = 1 the two are the same run.

Prints ONE JSON line on rank 0, with `roofline` (dominant kernel = the fused bounce kernel, HIP-event timed on the
streams it runs on, against the 8 TB/s HBM peak) and `cpu_baseline` (the single-thread CPU oracle on a bounded
sample of the same workload; N = 1 only).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

# dmabuf IPC: read when the HSA runtime starts, so it must be in the environment before the first GPU call
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PATH_BYTES = 44                # SoA PathSegment: origin 12 + dir 12 + throughput 12 + pixelIndex 4 + remainingBounces 4
ACCUM_BYTES = 12               # radiance of one emitter hit parked in the batch's buffer (vec3 fp32 write)
SIMDS, CLOCK_HZ = 1024, 2.4e9  # 256 CUs x 4 SIMDs; max shader clock


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="steps timed (one step = --batch iterations of the whole frame, times N with --scaling weak); default 32 "
                         "on one GPU (1024 spp = 16 x the 64 spp of BASELINE config C2); on N GPUs config C3's 5000 spp rounded up to "
                         "whole steps (157 with --scaling strong)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps first (default 4)")
    ap.add_argument("--scene", default=os.path.join(ROOT, "scenes", "cornell.txt"))
    ap.add_argument("--res", type=int, nargs=2, default=[1280, 720])
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--cpu-spp", type=int, default=40, help="spp of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--pipeline", type=int, default=2, help="batches in flight (PtOptions.pipeline_depth; 0 = library default 3)")
    ap.add_argument("--batch", type=int, default=32, help="iterations per step = iterations traced as one wavefront batch")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1.  weak (default): a step is --batch x N iterations of the whole frame, so a rank's share of a step -- "
                         "its 1/N of the rows of N times the iterations -- is as many paths as the single GPU's step (more GPUs = "
                         "more samples per pixel in the same time); strong: a step is --batch iterations whatever N (BASELINE "
                         "config C3 as written: a fixed number of samples, divided)")
    ap.add_argument("--collective-every", default="batch", choices=["batch", "1"],
                    help="N > 1: assemble the frame at rank 0 after every wavefront batch, or after every iteration")
    ap.add_argument("--collective", default="gather", choices=["gather", "reduce"],
                    help="N > 1: gather of the packed row blocks (default) or reduce(sum) of zero-padded full frames")
    ap.add_argument("--per-iteration-sample", type=int, default=2,
                    help="N > 1 with --collective-every batch: also time this many steps in per-iteration mode (0 = skip)")
    ap.add_argument("--pmc-traffic-json", default=os.path.join(ROOT, "profiles", "pmc_traffic.json"),
                    help="PMC counters per bounce-kernel launch from the rocprofv3 --pmc passes (profiles/README.md)")
    ap.add_argument("--valu-rate-json", default=os.path.join(ROOT, "profiles", "valu_issue_rate.json"),
                    help="measured cycles per wave64 vector instruction per SIMD (profiles/valu_issue_rate.hip)")
    ap.add_argument("--dump-frame", default=None, help="rank 0 writes the final frame (float32 W*H*3 running sum) to this .npy file")
    return ap.parse_args(argv)


def self_launch(args):
    """`python bench.py --gpus N` outside torchrun: start the N ranks as children of a `torch.distributed.run` child
    process -- before this process has initialised the GPU -- relay their output and exit with their code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd)
    sys.exit(r.returncode)


def cpu_baseline(args, scene):
    """Single-thread CPU oracle (oracle/, kind 'port') on a bounded sample of the same workload."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    W, H = args.res
    ref = orc.Renderer(scene.camera.view(orc.CAMERA_DTYPE), scene.geoms.view(orc.GEOM_DTYPE),
                       scene.materials.view(orc.MATERIAL_DTYPE), args.depth, meshes=getattr(scene, "meshes", None))
    img = np.zeros(W * H * 3, np.float32)
    K = 64                                                 # a probe first: every 64th row of one sample
    t0 = time.perf_counter()
    ref.iterate(1, img, 0, K)                              # (also warms the caches / pages the library in)
    probe = time.perf_counter() - t0
    if probe * K * args.cpu_spp <= 45.0:
        t0 = time.perf_counter()
        for it in range(2, 2 + args.cpu_spp):
            ref.iterate(it, img)
        dt = time.perf_counter() - t0
        pixels, what = W * H * args.cpu_spp, "%d spp" % args.cpu_spp
    else:
        # a slow scene (brute-force meshes): bound the sample to ~15 s of row slices (rows y with y % 64 == r) of sample 2
        nslices = max(1, min(K - 1, int(15.0 / max(probe, 1e-3))))
        t0 = time.perf_counter()
        for r in range(1, 1 + nslices):
            ref.iterate(2, img, r, K)
        dt = time.perf_counter() - t0
        pixels = sum((H - r + K - 1) // K for r in range(1, 1 + nslices)) * W
        what = "%d of every %d rows of 1 spp" % (nslices, K)
    return {"value": round(pixels * args.depth / dt / 1e6, 3), "unit": "Mpaths/s", "cores": 1,
            "kind": "port",
            "sample": "%s of %s %dx%d depth %d (%.1f s, single thread, g++ -O2 -ffp-contract=off, host has %d cores)"
                      % (what, os.path.basename(args.scene), W, H, args.depth, dt, os.cpu_count())}


def valu_issue_rate(path, waves_per_simd):
    """cycles per wave64 vector instruction per SIMD from the committed microbenchmark: the kernel-like instruction mix at
    the residency nearest to the kernel's, and the plain v_fma_f32 rate (wall-clock based columns)."""
    try:
        rows = json.load(open(path))["rows"]
    except Exception:
        return None
    def pick(prefix):
        c = [r for r in rows if r["op"].startswith(prefix)]
        if not c:
            return None
        r = min(c, key=lambda r: abs(r["waves_per_simd"] - waves_per_simd))
        return r["cycles_per_instruction_per_simd_from_wall"], r["waves_per_simd"]
    mix, fma = pick("mix"), pick("v_fma_f32")
    if not mix or not fma:
        return None
    return {"mix": mix[0], "fma": fma[0], "waves_per_simd": mix[1]}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1:
        self_launch(args)                                   # never returns
    if world != args.gpus:
        args.gpus = world

    import numpy as np
    import torch
    import torch.distributed as dist

    B = args.batch
    weak = args.scaling == "weak"
    I = B * (world if weak else 1)                          # iterations of the whole frame per step
    if args.steps is None:
        args.steps = 32 if world == 1 else (5000 + I - 1) // I
    if args.warmup is None:
        args.warmup = 4
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
    D = args.depth
    if B < 1 or B > pt.PT_MAX_BATCH:
        sys.exit("bench.py: --batch must be 1..%d" % pt.PT_MAX_BATCH)
    # accumulator as a torch tensor so RCCL can move it.  N = 1: the full frame.  N > 1: this rank's rows
    # only (packed, padded to the largest shard); rank 0 assembles `frame` from the gathered blocks.
    if world > 1:
        accum = torch.zeros(ptdist.padded_block_floats(W, H, world), dtype=torch.float32, device="cuda")
        frame = torch.zeros(P * 3, dtype=torch.float32, device="cuda") if rank == 0 else None
        bufs = ptdist.make_gather_buffers(accum, world, rank) if args.collective == "gather" else None
        shard_flag = pt.PT_FLAG_ACCUM_SHARD_ROWS
    else:
        accum = torch.zeros(P * 3, dtype=torch.float32, device="cuda")
        frame, bufs, shard_flag = None, None, 0
    stream = torch.cuda.current_stream()

    def init(flags, pipeline, max_batch):
        pt.pathtraceFree()
        pt.pathtraceInit(scene, shard_rank=rank, shard_count=world, stream=stream.cuda_stream,
                         accum_dev=accum.data_ptr(), device=device_index, flags=flags | shard_flag,
                         traceDepth=D, pipeline_depth=pipeline, max_batch=max_batch)

    def collect():
        # the single collective of the data path: the row blocks travel to rank 0 over xGMI
        ptdist.gather_frame(accum, bufs, frame, W, H, dst=0, collective=args.collective)

    def run_steps(first_iter, steps, every):
        """`steps` steps of I iterations from `first_iter`.  every == "batch": wavefront batches of `maxb` iterations (one or
        several steps each, or a part of one), the frame assembled at rank 0 after every batch; every == "1": I single-iteration
        calls per step, each followed by the collective."""
        it = first_iter
        end = first_iter + steps * I
        while it < end:
            if every == "batch":
                n = min(maxb, end - it)
                pt.pathtrace_batch(None, 0, it, n)
                if world > 1:
                    collect()
                it += n
            else:
                pt.pathtrace(None, 0, it, readback=False)
                if world > 1:
                    collect()
                it += 1
        return it

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(first_iter, steps, every):
        barrier()
        t0 = time.perf_counter()
        run_steps(first_iter, steps, every)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    every = args.collective_every
    # N ranks share the frame's rows, so a rank's launches cover 1/N of the paths: with the collective once per batch, a rank
    # traces `fuse` consecutive steps as ONE wavefront batch (at most PT_MAX_BATCH iterations), which keeps its launches fat
    # -- but not so few batches that the two in flight never overlap: at least 8 per timed run (measured on a rank of 8 and of
    # 4 at the driver's 20 steps: batches of 64 beat 128 and 256 by 2-5 %)
    # Weak scaling: a step is I = B x N iterations, traced as one wavefront batch (in pieces of PT_MAX_BATCH should it be larger).
    if weak:
        fuse = 1
        maxb = min(I, pt.PT_MAX_BATCH) if every == "batch" else 1
    else:
        fuse = max(1, min(world, pt.PT_MAX_BATCH // B, max(1, args.steps // 8))) if every == "batch" else 1
        maxb = B * fuse if every == "batch" else 1
    # ---- pass A: the headline number ----------------------------------------------------------
    init(0, args.pipeline, maxb)
    nxt = run_steps(1, args.warmup, every)
    barrier()
    pt.counters_reset()
    dt = timed(nxt, args.steps, every)
    cntA = pt.counters()
    if args.dump_frame and rank == 0:
        np.save(args.dump_frame, (frame if world > 1 else accum).cpu().numpy())

    # ---- pass A': N > 1, the same workload with the collective after EVERY iteration (config C3 as written), bounded
    per_iter = None
    if world > 1 and every == "batch" and args.per_iteration_sample > 0:
        accum.zero_()
        init(0, args.pipeline, 1)
        nxt1 = run_steps(1, 1, "1")
        dt1 = timed(nxt1, args.per_iteration_sample, "1")
        per_iter = {"value": round(P * D * I * args.per_iteration_sample / dt1 / 1e6, 2), "unit": "Mpaths/s",
                    "steps": args.per_iteration_sample,
                    "mode": "pt_iterate + one %s per iteration (BASELINE config C3 as written)" % args.collective}

    # ---- pass B: same steps with HIP events around every launch (roofline of the bounce kernel); one
    #      batch in flight, so that a launch's duration is the kernel's own and not its share of a GPU it
    #      co-occupies with the neighbouring batches' launches
    accum.zero_()
    init(pt.PT_FLAG_KERNEL_TIMING, 1, maxb)
    nxt = run_steps(1, min(args.warmup, 2), every)
    barrier()
    pt.counters_reset()
    dtB = timed(nxt, args.steps, every)
    cnt = pt.counters()
    pt.pathtraceFree()

    iters_timed = args.steps * I
    live = [int(cnt.live[d]) for d in range(D + 2)]
    # paths that actually travel through the path pools: a survivor that certainly misses everything ends at its scatter and
    # is counted in live[d + 1] (it does enter that bounce and miss) without ever being written or read
    moved = [int(cnt.live[d]) - int(cnt.ended_early[d]) for d in range(D + 2)]
    hits = int(cnt.light_hits)
    # algorithmic HBM bytes of the bounce launches: read every queued path (bounce 1 builds its camera
    # rays in registers and reads nothing), write every queued survivor (nothing is written after the last
    # bounce), park the radiance of every emitter hit
    bounce_bytes = sum(PATH_BYTES * moved[d] for d in range(2, D + 1)) \
        + sum(PATH_BYTES * moved[d + 1] for d in range(1, D)) + ACCUM_BYTES * hits
    launches = max(int(cnt.bounce_launches), 1)
    iters_per_launch = maxb      # (N > 1: the last launch of a run may carry fewer steps; the PMC figures are N = 1 only)
    avg_ms = cnt.bounce_kernel_ms / launches
    achieved = bounce_bytes / launches / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    # PMC counters of the bounce kernel from the committed rocprofv3 passes.  They are stored PER ITERATION of a launch's
    # batch together with the configuration they were collected on; used only for that very configuration (frame,
    # depth, iterations per launch), scaled by the iterations a launch of THIS run carries -- else null.
    traffic, valu_insts, lds_conf, pmc_src = None, None, None, None
    try:
        pmc = json.load(open(args.pmc_traffic_json))
        here = ["%s %dx%d" % (os.path.relpath(args.scene, ROOT), W, H), "%d bounces" % D]
        if world == 1 and pmc.get("workload") == here and pmc.get("iterations_per_launch") == iters_per_launch:
            traffic = pmc["hbm_bytes_per_launch_iteration"] * iters_per_launch
            valu_insts = pmc["valu_wave_insts_per_launch_iteration"] * iters_per_launch
            lds_conf = pmc.get("lds_bank_conflict_cycles_per_launch")
            pmc_src = pmc.get("source")
    except Exception:
        pass
    rate = valu_issue_rate(args.valu_rate_json, 6)

    if rank == 0:
        nominal = P * D * iters_timed
        rf = {"bound": "hbm", "kernel": "k_bounce (fused [camera rays+]intersect+shade+compact, one launch per bounce and batch)",
              "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
              "frac": round(achieved / HBM_PEAK_GBS, 5),
              "traffic": traffic,
              "algorithmic_bytes_per_launch": round(bounce_bytes / launches, 1),
              "traffic_over_algorithmic": round(traffic / (bounce_bytes / launches), 3) if traffic else None,
              "avg_launch_ms": round(avg_ms, 5), "launches": launches, "iterations_per_launch": iters_per_launch,
              "bounce_kernel_share_of_step": round(cnt.bounce_kernel_ms / (dtB * 1e3), 4),
              "ms_per_step_with_events": round(dtB / args.steps * 1e3, 4),
              "pmc_source": pmc_src}
        if valu_insts and rate:
            # the kernel is VALU-issue-bound, not HBM-bound: wave64 vector instructions per launch (PMC)
            # x MEASURED cycles per instruction per SIMD (profiles/valu_issue_rate.json) / (1024 SIMDs x 2.4 GHz)
            t_mix = valu_insts * rate["mix"] / (SIMDS * CLOCK_HZ) * 1e3
            t_fma = valu_insts * rate["fma"] / (SIMDS * CLOCK_HZ) * 1e3
            rf["valu"] = {"wave_instructions_per_launch": round(valu_insts),
                          "cycles_per_instruction_per_simd_measured": {"kernel_like_mix": rate["mix"], "v_fma_f32": rate["fma"],
                                                                       "waves_per_simd": rate["waves_per_simd"]},
                          "issue_bound_ms_per_launch": {"kernel_like_mix": round(t_mix, 5), "v_fma_f32": round(t_fma, 5)},
                          "frac_of_issue_bound": {"kernel_like_mix": round(t_mix / avg_ms, 4), "v_fma_f32": round(t_fma / avg_ms, 4)},
                          "lds_bank_conflict_cycles_per_launch": lds_conf}
        out = {
            "metric": "Mpaths/sec (paths = pixels x bounces x spp) at %dx%d, %d bounces" % (W, H, D),
            "value": round(nominal / dt / 1e6, 2), "unit": "Mpaths/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s %dx%d, %d bounces, %d spp per step x %d steps%s" % (
                           os.path.relpath(args.scene, ROOT), W, H, D, I, args.steps,
                           "" if world == 1 else ", rows sharded y%%%d + RCCL %s of the row blocks per %s" % (
                               world, args.collective, "batch" if every == "batch" else "iteration")),
                       "iterations_per_step": I,
                       "paths_per_step_nominal": P * D * I,
                       "ms_per_iteration": round(dt / iters_timed * 1e3, 5),
                       "iterations_per_wavefront_batch": maxb,
                       "batches_in_flight": args.pipeline if args.pipeline > 0 else 3,
                       "collective_every": None if world == 1 else every,
                       "live_segments_per_iteration": round(sum(live[1:D + 1]) / max(iters_timed, 1), 1),
                       "queued_segments_per_iteration": round(sum(moved[1:D + 1]) / max(iters_timed, 1), 1),
                       "live_Msegments_per_s": round(sum(int(cntA.live[d]) for d in range(1, D + 1)) / dt / 1e6, 2)},
            "roofline": rf,
        }
        if per_iter:
            out["config"]["per_iteration_collective"] = per_iter
        if world == 1 and args.cpu_spp > 0:
            out["cpu_baseline"] = cpu_baseline(args, scene)
        print(json.dumps(out), flush=True)
    if world > 1:
        # orderly exit: nobody leaves (and tears its HIP context down) while another rank still works
        torch.cuda.synchronize()
        dist.barrier()
        del accum, frame, bufs
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
