"""Multi-GPU plumbing of the hot path: one process per GPU, rows sharded round-robin, ONE collective.

Image rows are dealt to the ranks like cards (row y -> rank y % world), which balances the black
border rows and the bright light rows.  Every rank accumulates ONLY its own rows, packed
(PT_FLAG_ACCUM_SHARD_ROWS), and the frame is assembled at rank 0 by a single collective (RCCL over xGMI on
the GPUs, gloo in the CPU tests) issued by the caller after every iteration (BASELINE config C3 as written;
bench.py --collective-every 1) or after every wavefront batch of iterations (--collective-every batch:
accumulation is additive, so the frame at rank 0 is the same whenever it is assembled).

Two collectives, chosen by the caller (bench.py --collective; since round 6 its default is the contract's):
  "reduce"  reduce(sum, float32, 3 W H, root 0) of zero-padded full frames -- what north_star and SURVEY 8e name;
  "gather"  gather of the packed row blocks: the rows are disjoint, so the reduce adds x + 0 + ... + 0 -- the same
            bits (both are a single-GPU render's) for world x the bytes.  xGMI is point-to-point: in the
            gather every rank sends its 1/world of the frame straight to rank 0 over its own link (1.4 MB per rank
            for 1280x720 at world = 8) instead of pushing 11 MB around a ring.
One N > 1 bench line carries both readings.  (The C ABI's own multi-device path -- csrc/pt_group.h, ONE process --
uses the reduce.)
"""
import os


def shard_rows(height, rank, world):
    """Rows rendered by `rank`: y % world == rank."""
    return range(rank, height, world)


def local_pixel_count(width, height, rank, world):
    return width * len(shard_rows(height, rank, world))


def padded_block_floats(width, height, world):
    """Size of one rank's packed row block, padded to the largest shard so every rank sends the same."""
    return ((height + world - 1) // world) * width * 3


def init_process_group(backend=None, timeout_s=600, single_rank=False):
    """Rendezvous from the torchrun environment (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).  Every collective of the
    group -- the exit barrier included -- gives up after `timeout_s` seconds instead of waiting for a rank that died (the same bound
    holds for the rendezvous itself: ranks of a fresh box can be a minute or two apart in their first `import torch`).
    `single_rank`: form a ONE-rank group on 127.0.0.1 when there is no torchrun environment (bench.py at N = 1 sends its
    per-iteration reduce through RCCL's call path: a hardware number for the protocol's fixed cost without a second GPU)."""
    import datetime
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and not single_rank:
        return 0, 1
    file_store = None
    if world == 1:
        # a one-rank group needs no rendezvous over the network: a FileStore in a fresh temporary directory (probing a free TCP port,
        # closing it and letting c10d bind it again left a window in which a parallel bench or test on the box could take it)
        import tempfile
        file_store = os.path.join(tempfile.mkdtemp(prefix="pt_amd_store_"), "store")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    # HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC, the only kind this driver supports) is read when the HSA runtime
    # starts, i.e. at the first GPU call of the process: the caller exports it before touching the GPU (bench.py does
    # so at its very top); setting it here would be too late.
    if backend == "nccl" and os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") != "0":
        raise RuntimeError("export HSA_ENABLE_IPC_MODE_LEGACY=0 before the first GPU call (RCCL needs dmabuf IPC here)")
    kw = {}
    if backend == "nccl":
        kw["device_id"] = torch.device("cuda", torch.cuda.current_device())
    if file_store is not None:
        dist.init_process_group(backend=backend, init_method="file://" + file_store, rank=0, world_size=1,
                                timeout=datetime.timedelta(seconds=timeout_s), **kw)
    else:
        dist.init_process_group(backend=backend, timeout=datetime.timedelta(seconds=timeout_s), **kw)
    return dist.get_rank(), dist.get_world_size()


def make_gather_buffers(block, world, rank, dst=0):
    """Receive buffers at the destination rank: one block per rank, views of ONE contiguous (world, block) tensor so
    that the rows can be interleaved into the frame with a single strided copy."""
    import torch
    if rank != dst:
        return None
    recv = torch.empty((world, block.numel()), dtype=block.dtype, device=block.device)
    return list(recv.unbind(0))


def _interleave_rows(bufs, frame, width, height, world):
    """frame row y = row y // world of rank y % world's block."""
    per = height // world
    base = bufs[0]
    contiguous = all(b.data_ptr() == base.data_ptr() + r * base.numel() * base.element_size() for r, b in enumerate(bufs))
    if height % world == 0 and contiguous and base.numel() == per * width * 3:
        import torch
        recv = torch.as_strided(base, (per, world, width * 3), (width * 3, base.numel(), 1))
        frame.view(per, world, width * 3).copy_(recv)      # one kernel instead of `world`
        return
    rows_view = frame.view(height, width * 3)
    for r in range(world):
        n = len(shard_rows(height, r, world))
        if n:
            rows_view[r::world].copy_(bufs[r][:n * width * 3].view(n, width * 3))


def gather_frame(block, bufs, frame, width, height, dst=0, collective="reduce", always_collective=False):
    """The data path's single collective: rank `dst` receives every rank's packed rows and interleaves
    them into `frame` (H*W*3 floats).  `block` is this rank's packed accumulator, padded to
    padded_block_floats(); it keeps accumulating, the collective only reads it.

    `collective` is chosen by the caller, once, identically on every rank (a command-line option in bench.py);
    nothing here switches it at run time, and an error of the backend propagates -- a failed RCCL collective
    leaves the communicator unusable, so there is nothing to fall back to:
      "reduce"  the reduce(sum) of zero-padded full frames that BASELINE.json names (default): x + 0 is exact;
      "gather"  every rank sends its 1/world of the frame straight to `dst`: same bits, 1/world of the bytes.
    `always_collective`: issue the collective even in a one-rank group (tests: the RCCL call path on a single GPU)."""
    import torch
    import torch.distributed as dist
    if collective not in ("gather", "reduce"):
        raise ValueError("gather_frame: collective must be 'gather' or 'reduce', not %r" % (collective,))
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if world == 1 and not (always_collective and dist.is_initialized()):
        frame.copy_(block[:frame.numel()])
        return frame
    if collective == "gather":
        dist.gather(block, bufs if rank == dst else None, dst=dst)
        if rank == dst:
            _interleave_rows(bufs, frame, width, height, world)
        return frame
    # reduce(sum) of full frames: this rank's rows in place, zeros elsewhere
    full = torch.zeros(height * width * 3, dtype=block.dtype, device=block.device)
    n = len(shard_rows(height, rank, world))
    if n:
        full.view(height, width * 3)[rank::world].copy_(block[:n * width * 3].view(n, width * 3))
    dist.reduce(full, dst=dst, op=dist.ReduceOp.SUM)
    if rank == dst:
        frame.copy_(full)
    return frame


class PerIterationReducer:
    """BASELINE config C3 AS WRITTEN: after EVERY iteration one reduce(sum) of zero-padded full frames to rank `dst` -- the
    reference's own per-iteration full-frame transfer (src/pathtrace.cu:170-171, src/main.cpp:97-106) with RCCL in the place of
    its cudaMemcpy -- as a pipeline instead of a stop-and-go loop:

      * the renderer accumulates into a FULL frame, its own rows in place and zeros everywhere else (a row shard WITHOUT
        PT_FLAG_ACCUM_SHARD_ROWS: k_commit indexes by the global pixel), so the accumulator itself is the zero-padded frame:
        no per-call torch.zeros, no strided scatter;
      * `collect()` takes a SNAPSHOT of it on the caller's stream (one contiguous device-to-device copy, ordered behind the
        iteration's commit) into one of two buffers and issues the reduce of that buffer asynchronously: the backend runs it on a
        stream of its own, ordered behind the snapshot (torch's ProcessGroupNCCL synchronises its stream with the caller's at the
        call), so iteration i's reduce runs while iteration i + 1 is committed and while the batches traced ahead keep tracing;
      * a buffer is reused two calls later, behind the completion of the reduce that read it (`Work.wait()`: a stream-level
        wait under RCCL, nothing the host blocks on).
    Three calls into torch per iteration.

    At rank `dst` the buffer of the latest call holds the summed frame once its reduce has finished (`frame()`).  Rows are
    disjoint, every other rank adds +0.0 to them: bit-identical to a single-GPU render.
    CPU tensors (the gloo tests) take the same calls.  `always_collective`: issue the reduce in a one-rank group too (N = 1:
    the RCCL call path on one GPU)."""

    def __init__(self, accum_full, dst=0, always_collective=False):
        import torch
        import torch.distributed as dist
        self.accum = accum_full
        self.dst = dst
        self.collective = dist.is_initialized() and (dist.get_world_size() > 1 or always_collective)
        self.snap = [torch.empty_like(accum_full) for _ in range(2)]
        self.work = [None, None]
        self.k = 0
        self.last = None
        self.calls = 0
        self._reduce = dist.reduce
        self._sum = dist.ReduceOp.SUM

    def collect(self):
        k = self.k
        self.k = k ^ 1
        w = self.work[k]
        if w is not None:
            w.wait()                                        # the reduce that last read this buffer
        buf = self.snap[k]
        buf.copy_(self.accum, non_blocking=True)
        if self.collective:
            self.work[k] = self._reduce(buf, dst=self.dst, op=self._sum, async_op=True)
        self.last = k
        self.calls += 1

    def finish(self):
        """every reduce issued so far has completed, as far as the caller's stream is concerned"""
        for k in (0, 1):
            if self.work[k] is not None:
                self.work[k].wait()
                self.work[k] = None

    def frame(self):
        """rank `dst`: the frame of the latest collect(), summed over the ranks.  Every other rank: None -- a reduce leaves the
        contents of its buffer off the root unspecified (gloo and RCCL may park partial sums there)."""
        import torch.distributed as dist
        self.finish()
        if self.last is None or (self.collective and dist.get_rank() != self.dst):
            return None
        return self.snap[self.last]

    def bytes_per_call(self):
        return self.accum.numel() * self.accum.element_size()
