"""World-size-2 test of the N>1 path on CPU (gloo): row sharding + the single collective (gather of the
packed row blocks).

No GPU here, so each rank renders ITS rows with the CPU oracle (checker standing in for the kernels),
packs them the way PT_FLAG_ACCUM_SHARD_ROWS does, and then runs the SAME `gather_frame` bench.py uses;
rank 0 must end up with a frame that is bit-identical to the unsharded render after every iteration."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, SCENES


def _worker(rank, world, port, res, iters, out_path):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import __graft_entry__ as ge
    import oracle as orc
    ptdist = ge.load_submodule("distributed")
    r, w = ptdist.init_process_group("gloo")
    assert (r, w) == (rank, world)
    W, H = res
    sc = orc.Scene(os.path.join(SCENES, "cornell.txt"))
    sc.set_resolution(W, H)
    ren = orc.Renderer(sc.camera, sc.geoms, sc.materials, 8)
    sparse = np.zeros(W * H * 3, np.float32)                     # oracle output: full frame, own rows only
    block = torch.zeros(ptdist.padded_block_floats(W, H, world), dtype=torch.float32)
    bufs = ptdist.make_gather_buffers(block, world, rank)
    frame = torch.zeros(W * H * 3, dtype=torch.float32)
    full = np.zeros(W * H * 3, np.float32)
    mine = list(ptdist.shard_rows(H, rank, world))
    ok = True
    npix = 0
    for it in iters:
        c = ren.iterate(it, sparse, rank, world)                 # this rank's rows only
        npix = c.live[1]
        packed = sparse.reshape(H, W * 3)[mine].reshape(-1)      # what PT_FLAG_ACCUM_SHARD_ROWS holds
        block[:packed.size] = torch.from_numpy(packed)
        ptdist.gather_frame(block, bufs, frame, W, H, dst=0, collective="gather")
        if rank == 0:
            ren.iterate(it, full)
            ok = ok and np.array_equal(frame.numpy().view(np.uint32), full.view(np.uint32))
    assert npix == ptdist.local_pixel_count(W, H, rank, world)
    rows = sparse.reshape(H, W, 3)
    other = np.ones(H, bool)
    other[mine] = False
    assert not np.any(rows[other])                               # other ranks' rows stay exactly zero
    if rank == 0:
        np.save(out_path, np.array([1 if ok else 0]))
    torch.distributed.destroy_process_group()


# ragged shards (19 / 18 rows: per-rank copies) and even ones (one strided copy) on two ranks; and the world size of the real
# run, eight ranks: 40 rows = five each (the strided copy), 27 rows = ragged with ranks that hold three or four rows
@pytest.mark.parametrize("world,res", [(2, (48, 37)), (2, (48, 36)), (8, (32, 40)), (8, (24, 27))])
def test_ranks_gloo_assemble_the_frame_bit_exactly(tmp_path, world, res):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "ok.npy")
    mp.spawn(_worker, args=(world, port, res, [1, 2, 3], out), nprocs=world, join=True)
    assert np.load(out)[0] == 1


def _worker_reduce(rank, world, port, out_path):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch
    import __graft_entry__ as ge
    ptdist = ge.load_submodule("distributed")
    ptdist.init_process_group("gloo")
    W, H = 5, 7
    block = torch.zeros(ptdist.padded_block_floats(W, H, world))
    rows = list(ptdist.shard_rows(H, rank, world))
    vals = torch.arange(H * W * 3, dtype=torch.float32).view(H, W * 3)[rows].reshape(-1)
    block[:vals.numel()] = vals
    frame = torch.full((H * W * 3,), -1.0) if rank == 0 else None
    ptdist.gather_frame(block, None, frame, W, H, dst=0, collective="reduce")   # the collective BASELINE.json sketches
    if rank == 0:
        np.save(out_path, np.array([int(torch.equal(frame, torch.arange(H * W * 3, dtype=torch.float32)))]))
    torch.distributed.destroy_process_group()


def test_reduce_collective_assembles_the_same_frame(tmp_path):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "ok.npy")
    mp.spawn(_worker_reduce, args=(3, port, out), nprocs=3, join=True)
    assert np.load(out)[0] == 1


def _worker_per_iteration(rank, world, port, res, iters, out_path):
    """config C3 as written: every rank accumulates ITS rows into a full frame (zeros elsewhere) and the frame is reduced after
    EVERY iteration through PerIterationReducer -- the oracle stands in for the kernels, the collective is the product's."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import __graft_entry__ as ge
    import oracle as orc
    ptdist = ge.load_submodule("distributed")
    ptdist.init_process_group("gloo", timeout_s=120)
    W, H = res
    sc = orc.Scene(os.path.join(SCENES, "cornell.txt"))
    sc.set_resolution(W, H)
    ren = orc.Renderer(sc.camera, sc.geoms, sc.materials, 8)
    sparse = np.zeros(W * H * 3, np.float32)                     # full frame, own rows only: the accumulator itself
    accum = torch.from_numpy(sparse)                             # (shares the memory: what the renderer's commit writes)
    red = ptdist.PerIterationReducer(accum, dst=0)
    assert red.bytes_per_call() == W * H * 12
    full = np.zeros(W * H * 3, np.float32)
    ok = True
    for it in iters:
        ren.iterate(it, sparse, rank, world)
        red.collect()
        if rank == 0:
            ren.iterate(it, full)
            ok = ok and np.array_equal(red.frame().numpy().view(np.uint32), full.view(np.uint32))
    ok = ok and red.calls == len(iters)
    # the accumulator itself is never touched by the collective: still this rank's rows only
    rows = sparse.reshape(H, W, 3)
    other = np.ones(H, bool)
    other[list(ptdist.shard_rows(H, rank, world))] = False
    assert not np.any(rows[other])
    if rank == 0:
        np.save(out_path, np.array([1 if ok else 0]))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world,res", [(2, (48, 37)), (3, (40, 30))])
def test_per_iteration_reduce_of_full_frames_is_bit_exact(tmp_path, world, res):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "ok.npy")
    mp.spawn(_worker_per_iteration, args=(world, port, res, [1, 2, 3, 4], out), nprocs=world, join=True)
    assert np.load(out)[0] == 1


def test_shard_rows_partition():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    d = ge.load_submodule("distributed")
    for H in (1, 7, 8, 720, 1080, 4096):
        for world in (1, 2, 3, 4, 8):
            rows = sorted(y for r in range(world) for y in d.shard_rows(H, r, world))
            assert rows == list(range(H))
            counts = [len(d.shard_rows(H, r, world)) for r in range(world)]
            assert max(counts) - min(counts) <= 1
