"""The N > 1 data path with the HIP kernels: every rank renders its row shard on the GPU into a packed torch
accumulator (PT_FLAG_ACCUM_SHARD_ROWS), the SAME `gather_frame` bench.py uses assembles the frame at rank 0, and
rank 0 compares it bit for bit with the unsharded CPU oracle.

The GPU box has one card and RCCL wants one card per rank, so the collective runs over gloo here (what
BENCH_BACKEND=gloo does for bench.py); the rest -- sharding, packed accumulation, batching, the caller's stream,
the interleave at rank 0 -- is the code the 8-GPU run executes."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, SCENES

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, res, batches, out_path):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import __graft_entry__ as ge
    pt = ge.load_package()
    ptdist = ge.load_submodule("distributed")
    ptdist.init_process_group("gloo")
    torch.cuda.set_device(0)
    W, H = res
    depth = 6
    sc = pt.Scene(os.path.join(SCENES, "cornell_glass.txt"))
    sc.set_resolution(W, H)
    accum = torch.zeros(ptdist.padded_block_floats(W, H, world), dtype=torch.float32, device="cuda")
    frame = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda") if rank == 0 else None
    bufs = ptdist.make_gather_buffers(accum, world, rank)
    pt.pathtraceInit(sc, shard_rank=rank, shard_count=world, stream=torch.cuda.current_stream().cuda_stream,
                     accum_dev=accum.data_ptr(), device=0, flags=pt.PT_FLAG_ACCUM_SHARD_ROWS, traceDepth=depth,
                     pipeline_depth=3, max_batch=max(batches))
    ok, it = True, 1
    if rank == 0:
        import oracle as orc
        ref = orc.Renderer(sc.camera.view(orc.CAMERA_DTYPE), sc.geoms.view(orc.GEOM_DTYPE),
                           sc.materials.view(orc.MATERIAL_DTYPE), depth)
        want = np.zeros(W * H * 3, np.float32)
    for n in batches:
        pt.pathtrace_batch(None, 0, it, n)
        ptdist.gather_frame(accum, bufs, frame, W, H, dst=0)       # after every committed batch, like bench.py
        if rank == 0:
            for k in range(n):
                ref.iterate(it + k, want)
            got = frame.cpu().numpy()
            ok = ok and np.array_equal(got.view(np.uint32), want.view(np.uint32)) and want.max() > 0
        it += n
    pt.pathtraceFree()
    if rank == 0:
        np.save(out_path, np.array([1 if ok else 0]))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world,res", [(2, (96, 54)), (3, (80, 50))])      # even shards (one strided copy) and ragged ones
def test_ranks_render_their_rows_on_the_gpu_and_gather_the_frame(pt, tmp_path, world, res):
    if pt.device_count() < 1:
        pytest.fail("no HIP device: GPU tests must run on the MI355X box")
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "ok.npy")
    mp.spawn(_worker, args=(world, port, res, [4, 1, 7], out), nprocs=world, join=True)
    assert np.load(out)[0] == 1


def _single_rank_frame(pt, iterations, batch):
    """The unsharded 1280x720 Cornell render of iterations 1..iterations on this process's GPU (running sum)."""
    sc = pt.Scene(os.path.join(SCENES, "cornell.txt"))
    sc.set_resolution(1280, 720)
    pt.pathtraceFree()
    pt.pathtraceInit(sc, traceDepth=8, pipeline_depth=2, max_batch=batch)
    it = 1
    while it <= iterations:
        n = min(batch, iterations - it + 1)
        pt.pathtrace_batch(None, 0, it, n)
        it += n
    got = pt.readback(1280 * 720)
    pt.pathtraceFree()
    return got


@pytest.mark.parametrize("every,extra,ranks,scaling", [("batch", ["--group-blocks", "1"], 2, "weak"),       # the default collective: the reduce
                                                       ("1", ["--collective", "gather", "--batch", "4"], 2, "weak"),
                                                       ("batch", ["--batch", "8"], 2, "strong"),
                                                       ("1", ["--batch", "4"], 2, "strong"),   # config C3 as written
                                                       ("batch", ["--collective", "gather", "--batch", "16"], 4, "strong"),
                                                       ("batch", ["--collective", "gather", "--batch", "96"], 3, "weak")])
def test_bench_two_ranks_contract(pt, tmp_path, every, extra, ranks, scaling):
    # bench.py --gpus 2 (and 4) WITHOUT a torchrun environment: it starts its ranks itself (torch.distributed.run as a
    # child, one process per rank -- exactly what the driver launches); gloo stands in for RCCL because both ranks
    # share the box's single GPU.  Rank 0's assembled 1280x720 frame must equal the single-rank render BIT FOR BIT,
    # with the collective after every batch, after every iteration, and as the reduce of zero-padded frames.
    if pt.device_count() < 1:
        pytest.fail("no HIP device: GPU tests must run on the MI355X box")
    env = dict(os.environ, BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    dump, dump_c3 = str(tmp_path / "frame.npy"), str(tmp_path / "frame_c3.npy")
    steps, warmup, repeats = 3, 1, 2
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", str(steps), "--warmup", str(warmup),
                        "--repeats", str(repeats), "--collective-every", every, "--per-iteration-sample", "1", "--dump-frame", dump,
                        "--dump-c3-frame", dump_c3] + extra + ([] if "--group-blocks" in extra else ["--group-blocks", "0"])
                       + ([] if scaling == "weak" else ["--scaling", "strong"]),                # (weak is the default)
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                          # rank 0 only
    d = json.loads(lines[0])
    B = int(extra[extra.index("--batch") + 1]) if "--batch" in extra else 64
    # weak scaling: a step is B x N iterations of the whole frame (a rank's share = the single GPU's step); strong: B iterations
    I = B * ranks if scaling == "weak" else B
    assert d["n_gpus"] == ranks and d["steps"] == steps and d["warmup"] == warmup and d["scaling"] == scaling
    assert "cpu_baseline" not in d and d["value"] > 0
    assert "rows sharded y%%%d" % ranks in d["config"]["workload"] and d["config"]["iterations_per_step"] == I
    assert d["config"]["paths_per_step_nominal"] == 1280 * 720 * 8 * I
    assert abs(d["value"] - 1280 * 720 * 8 * I * steps / (d["ms_per_step"] * steps * 1e-3) / 1e6) <= 1e-3 * d["value"]
    assert d["config"]["collective_every"] == every
    # strong, collective per batch: a rank fuses min(N, PT_MAX_BATCH // B, steps // 8) steps (at least one) into one wavefront batch;
    # weak: a step is one wavefront batch, in pieces of PT_MAX_BATCH iterations should it be larger (3 x 96 = 288 -> 256 + 32)
    if every != "batch":
        wb = B                                                      # (single-iteration calls draw on batches of B traced ahead)
    elif scaling == "weak":
        wb = min(I, pt.PT_MAX_BATCH)
    else:
        wb = B * max(1, min(ranks, pt.PT_MAX_BATCH // B, steps // 8))
    assert d["config"]["iterations_per_wavefront_batch"] == wb
    # the timed block is repeated inside the run: median / min / max, every block listed
    assert d["repeats"] == repeats and len(d["ms_per_step_blocks"]) == repeats
    assert d["ms_per_step_min"] <= d["ms_per_step"] <= d["ms_per_step_max"] and d["value_min"] <= d["value"] <= d["value_max"]
    # ONE N > 1 run carries all three readings of "N GPUs", each over whole steps, and the collective's size and share
    for k in ("value_weak", "value_strong", "value_c3_as_written"):
        assert d[k]["value"] > 0 and d[k]["steps"] >= 1 and d[k]["unit"] == "Mpaths/s", k
    assert d["value_" + scaling]["value"] == d["value"]             # `value` = the reading --scaling names
    assert d["value_weak"]["iterations_per_step"] == B * ranks and d["value_strong"]["iterations_per_step"] == B
    c3 = d["value_c3_as_written"]
    assert c3["iterations_per_step"] == B and c3["iterations_per_wavefront_batch"] == B and c3["steps"] == 1
    assert "reduce per iteration" in c3["mode"] and c3["collective_bytes_per_call"] == 1280 * 720 * 12 and c3["ms_per_iteration"] > 0
    # ... and the frame its LAST reduce delivered at rank 0 (1 warm-up step + 1 step of B iterations) equals the single-rank render
    c3_got = np.load(dump_c3)
    c3_want = _single_rank_frame(pt, 2 * B, min(B, 64))
    assert c3_want.max() > 0 and np.array_equal(c3_got.view(np.uint32), c3_want.view(np.uint32))
    mg = d["multi_gpu"]
    block = -(-720 // ranks) * 1280 * 12                            # a rank's packed rows, padded to the largest shard
    assert mg["collective"] == ("gather" if "gather" in extra else "reduce")                      # (the contract's reduce is the default since round 6)
    assert mg["collective_bytes_per_call"]["sent_by_each_rank"] == (block if "gather" in extra else 1280 * 720 * 12)
    # ... and the line carries the OTHER collective's reading of the same pass beside it
    oc = d["multi_gpu"]["value_collective_" + ("reduce" if "gather" in extra else "gather")]
    assert oc["value"] > 0 and ("one reduce per" in oc["mode"]) == ("gather" in extra)
    if "--group-blocks" in extra:
        # round 6: beside the ranks, rank 0 measures the C ABI's own multi-device path in child processes -- a pt_group of as many members as there
        # are ranks (here on the one device: a rehearsal, and the line says so), in batch mode and as config C3 as written
        gb, gc = d["group"], d["group_c3_as_written"]
        assert "error" not in gb and "error" not in gc, (gb, gc)
        assert gb["members"] == ranks and gb["devices"] == 1 and gb["collective"] == "shared accumulator" and gb["value"] > 0
        assert gc["members"] == ranks and gc["ms_per_iteration"] > 0 and "unmeasured" in d["multi_device_note"]
    else:
        assert "group" not in d
    # (the collective is timed on its own, outside the steps: with gloo ranks sharing one GPU and the host's cores its share of a step
    # is whatever the box's load makes it -- 0.3 .. 1.6 seen; the field must be there and sane, its size is not the test's business)
    assert mg["collective_ms_per_call"] > 0 and 0 < mg["collective_share_of_step"] < 100
    assert mg["backend"] == "gloo" and mg["ranks_per_device"] == ranks          # (a rehearsal says so in its line)
    assert d["roofline"]["scope"].startswith("rank 0")
    got = np.load(dump)
    want = _single_rank_frame(pt, (steps * repeats + warmup) * I, min(I, 64))
    assert want.max() > 0
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_rccl_needs_one_gpu_per_rank_and_says_so(pt):
    # two ranks under the real backend on the box's single GPU: RCCL cannot serve ranks stacked on one device, so bench.py must
    # refuse at once, with a clear message and a non-zero exit code that the self-launched parent relays
    if pt.device_count() < 1:
        pytest.fail("no HIP device: GPU tests must run on the MI355X box")
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("more than one GPU visible")
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "BENCH_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0
    assert "2 ranks need 2 GPUs under RCCL" in (r.stderr + r.stdout)
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def _rccl_single_rank(_index, port, out_path):
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    pt = ge.load_package()
    ptdist = ge.load_submodule("distributed")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", world_size=1, rank=0, device_id=torch.device("cuda", 0))
    W, H = 160, 90
    sc = pt.Scene(os.path.join(SCENES, "cornell.txt"))
    sc.set_resolution(W, H)
    accum = torch.zeros(ptdist.padded_block_floats(W, H, 1), dtype=torch.float32, device="cuda")
    ok = True
    for collective in ("gather", "reduce"):
        accum.zero_()
        frame = torch.full((W * H * 3,), -1.0, dtype=torch.float32, device="cuda")
        bufs = ptdist.make_gather_buffers(accum, 1, 0)
        pt.pathtraceInit(sc, shard_rank=0, shard_count=1, stream=torch.cuda.current_stream().cuda_stream,
                         accum_dev=accum.data_ptr(), device=0, flags=pt.PT_FLAG_ACCUM_SHARD_ROWS, traceDepth=8, max_batch=4)
        pt.pathtrace_batch(None, 0, 1, 4)
        # the collective bench.py issues after every batch, through RCCL (backend "nccl" on ROCm) -- here in a one-rank group
        ptdist.gather_frame(accum, bufs, frame, W, H, dst=0, collective=collective, always_collective=True)
        t = torch.ones(1, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)                   # the timing reduction of bench.py
        dist.barrier()
        torch.cuda.synchronize()
        want = pt.readback(W * H)
        pt.pathtraceFree()
        ok = ok and want.max() > 0 and np.array_equal(frame.cpu().numpy().view(np.uint32), want.view(np.uint32)) and float(t) == 1.0
    np.save(out_path, np.array([1 if ok else 0]))
    dist.destroy_process_group()


def test_rccl_call_path_on_one_gpu(pt, tmp_path):
    # The box has one GPU, so RCCL cannot move data between ranks here; what CAN run is everything else of the N > 1
    # path on the real backend: communicator creation, gather / reduce / all_reduce / barrier kernels on the caller's
    # stream, ordered behind the renderer's commits, in a one-rank group.
    if pt.device_count() < 1:
        pytest.fail("no HIP device: GPU tests must run on the MI355X box")
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "ok.npy")
    mp.spawn(_rccl_single_rank, args=(port, out), nprocs=1, join=True)
    assert np.load(out)[0] == 1
