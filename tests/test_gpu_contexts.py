"""Renderer contexts and device groups (include/pt_amd.h: pt_ctx_*, pt_group_*; round 5): several renderers in one host process.
The reference keeps ONE renderer in file-static globals bound to device 0 (src/pathtrace.cu:70-71, src/preview.cpp:107); the library
now drives one per context -- one per device of a node, or, on this one-GPU box, several on the same device.  Everything is
compared bit for bit with the one-renderer frame and with the CPU oracle."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import SCENES

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(pt):
    if pt.device_count() < 1:
        pytest.fail("no HIP device: GPU tests must run on the MI355X box")
    return pt


def _scene(gpu, name="cornell.txt", res=(320, 200)):
    sc = gpu.Scene(os.path.join(SCENES, name))
    sc.set_resolution(*res)
    return sc


def _single(gpu, sc, batches, **init):
    W, H = (int(v) for v in sc.camera["resolution"][0])
    gpu.pathtraceFree()
    gpu.pathtraceInit(sc, **init)
    for first, count in batches:
        gpu.pathtrace_batch(None, 0, first, count)
    img = gpu.readback(W * H)
    cnt = gpu.counters()
    gpu.pathtraceFree()
    return img, cnt


def _oracle_frame(oracle, sc, iters):
    W, H = (int(v) for v in sc.camera["resolution"][0])
    ref = oracle.Renderer(sc.camera.view(oracle.CAMERA_DTYPE), sc.geoms.view(oracle.GEOM_DTYPE), sc.materials.view(oracle.MATERIAL_DTYPE), sc.traceDepth)
    want = np.zeros(W * H * 3, np.float32)
    for it in iters:
        ref.iterate(it, want)
    return want


def test_abi_version_and_sized_mesh_registration(gpu):
    assert gpu.lib().pt_abi_version() == gpu.PT_AMD_ABI_VERSION
    # a host built against the 16-byte PtMesh of round 3 is refused instead of being strided through as 32-byte structs
    arr = (gpu.PtMesh * 1)()
    assert gpu.lib().pt_set_meshes_sized(arr, 0, 16) == -1
    assert b"PtMesh is 16 bytes" in gpu.lib().pt_last_error()
    assert gpu.lib().pt_set_meshes_sized(arr, 0, C.sizeof(gpu.PtMesh)) == 0


def test_two_contexts_on_one_device_render_complementary_shards(gpu, oracle):
    # two contexts on the same device render the row shards y % 2 == 0 / 1 of one frame CONCURRENTLY (their
    # batches interleaved call by call, each with its own streams, pools and counters) and sum, bit for bit, to the unsharded frame
    sc = _scene(gpu)
    W, H = 320, 200
    batches = [(1, 4), (5, 4), (9, 2)]
    whole, cnt = _single(gpu, sc, batches, max_batch=4, pipeline_depth=2)
    assert np.array_equal(whole.view(np.uint32), _oracle_frame(oracle, sc, range(1, 11)).view(np.uint32))
    a, b = gpu.Context(), gpu.Context()
    try:
        with a:
            gpu.pathtraceInit(sc, shard_rank=0, shard_count=2, max_batch=4, pipeline_depth=2)
        with b:
            gpu.pathtraceInit(sc, shard_rank=1, shard_count=2, max_batch=4, pipeline_depth=2)
        for first, count in batches:                       # nothing waits in here: both renderers' launches are in flight together
            with a:
                gpu.pathtrace_batch(None, 0, first, count)
            with b:
                gpu.pathtrace_batch(None, 0, first, count)
        with a:
            ia, ca = gpu.readback(W * H), gpu.counters()
        with b:
            ib, cb = gpu.readback(W * H), gpu.counters()
        rows = np.arange(H)
        fa, fb = ia.reshape(H, W * 3), ib.reshape(H, W * 3)
        assert not fa[rows % 2 == 1].any() and not fb[rows % 2 == 0].any()          # each holds its own rows only
        assert np.array_equal((ia + ib).view(np.uint32), whole.view(np.uint32))     # x + 0 is exact
        assert [int(ca.live[d]) + int(cb.live[d]) for d in range(1, 9)] == [int(cnt.live[d]) for d in range(1, 9)]
        assert int(ca.light_hits) + int(cb.light_hits) == int(cnt.light_hits)
        # the default context was never touched by any of this
        assert gpu.lib().pt_ctx_current() is None
        with pytest.raises(gpu.PtError, match="before pt_init"):
            gpu.pathtrace(None, 0, 1)
    finally:
        a.destroy()
        b.destroy()


@pytest.mark.parametrize("members", [2, 3, 8])
def test_group_assembles_the_one_device_frame(gpu, oracle, members, monkeypatch):
    # pt_group_*: n renderers behind one call each, a host thread per member; on this box they share the device, so they commit their rows
    # into ONE full-frame accumulator and there is nothing to collect
    monkeypatch.delenv("PT_AMD_COLLECTIVE", raising=False)
    sc = _scene(gpu, res=(257, 131))                       # (ragged: 131 rows over 2 / 3 / 8 members, a width that is no multiple of the tile)
    whole, cnt = _single(gpu, sc, [(1, 3), (4, 3)], max_batch=3)
    g = gpu.Group(members)
    try:
        assert g.collective == "shared accumulator"
        g.init(sc, max_batch=3)
        g.iterate_batch(1, 3)
        g.iterate_batch(4, 3)
        got = g.readback()
        c = g.counters()
        assert np.array_equal(got.view(np.uint32), whole.view(np.uint32))
        assert [int(c.live[d]) for d in range(1, 9)] == [int(cnt.live[d]) for d in range(1, 9)] and int(c.iterations) == 6
        # ... and a second scene through the same group (the reference's Free -> Init restart), with meshes
        sm = _scene(gpu, "mesh_small.txt", (96, 64))
        wm, _ = _single(gpu, sm, [(1, 2)], max_batch=2)
        g.init(sm, max_batch=2)
        g.iterate_batch(1, 2)
        assert np.array_equal(g.readback().view(np.uint32), wm.view(np.uint32))
    finally:
        g.destroy()


def test_group_reduces_over_rccl_in_a_one_rank_communicator(gpu, oracle, monkeypatch):
    # The RCCL leg -- ncclCommInitAll in ONE process, every member's zero-padded full frame reduced (sum) to member 0's device -- needs
    # distinct devices; what a one-GPU box can run of it is the one-member group: the library is found and loaded, the communicator
    # made, the reduce issued on its own stream and its result read back.  ACROSS devices it is unmeasured (no node was available).
    monkeypatch.setenv("PT_AMD_COLLECTIVE", "rccl")
    sc = _scene(gpu, res=(200, 120))
    whole, _ = _single(gpu, sc, [(1, 4)], max_batch=4)
    g = gpu.Group(1)
    try:
        if not g.collective.startswith("rccl reduce"):
            pytest.skip("librccl.so could not be loaded in this process: %s" % g.collective)
        assert "one-rank communicator" in g.collective
        g.init(sc, max_batch=4)
        g.iterate_batch(1, 4)
        got = g.readback()
        assert np.array_equal(got.view(np.uint32), whole.view(np.uint32))
        g.iterate_batch(5, 4)                               # the reduce leaves the accumulators alone: the render goes on
        again = g.readback()
        whole8, _ = _single(gpu, sc, [(1, 4), (5, 4)], max_batch=4)
        assert np.array_equal(again.view(np.uint32), whole8.view(np.uint32))
    finally:
        g.destroy()


def _frames_call_by_call(gpu, sc, n_iter):
    """the one-renderer frame after EVERY iteration 1 .. n_iter (plain protocol: one pt_iterate per iteration)"""
    W, H = (int(v) for v in sc.camera["resolution"][0])
    gpu.pathtraceFree()
    gpu.pathtraceInit(sc)
    out = []
    for it in range(1, n_iter + 1):
        gpu.pathtrace(None, 0, it, readback=False)
        out.append(gpu.readback(W * H).copy())
    gpu.pathtraceFree()
    return out


@pytest.mark.parametrize("members,collective,threads", [(1, "rccl", "1"), (3, "rccl", "1"), (3, "rccl", "0"), (3, None, "1"), (8, None, "1"), (2, "host", "0")])
def test_group_per_iteration_reduce_is_the_one_device_frame_after_every_call(gpu, members, collective, threads, monkeypatch):
    # BASELINE config C3 as written through the LIBRARY's own collective: pt_group_iterate = one iteration on
    # every member -- a commit out of batches traced ahead -- and the frame's asynchronous assembly; pt_group_readback waits for the
    # collective stream alone.  After EVERY call the frame is the one-renderer frame bit for bit: the snapshot never runs under a
    # co-member's next commit, a result buffer is never overwritten under its read-back, the trace-ahead batches never leak into it.
    # (PT_AMD_COLLECTIVE=rccl on this one-GPU box: the members' shared accumulator goes through the snapshot -> ncclReduce pipeline in a
    # one-rank communicator; PT_AMD_GROUP_THREADS=0: the calling thread issues every member's work.)
    if collective:
        monkeypatch.setenv("PT_AMD_COLLECTIVE", collective)
    else:
        monkeypatch.delenv("PT_AMD_COLLECTIVE", raising=False)
    monkeypatch.setenv("PT_AMD_GROUP_THREADS", threads)
    sc = _scene(gpu, res=(193, 101))
    n_iter = 11
    want = _frames_call_by_call(gpu, sc, n_iter)
    g = gpu.Group(members)
    try:
        if collective == "rccl":
            if not g.collective.startswith("rccl reduce"):
                pytest.skip("librccl.so could not be loaded in this process: %s" % g.collective)
        else:
            assert g.collective == "shared accumulator"          # (one device: "host" has nothing to gather either)
        g.init(sc, flags=gpu.PT_FLAG_TRACE_AHEAD, max_batch=4, pipeline_depth=2)
        for it in range(1, n_iter + 1):
            g.iterate(it)
            got = g.readback()
            assert np.array_equal(got.view(np.uint32), want[it - 1].view(np.uint32)), "iteration %d" % it
        assert int(g.counters().iterations) == n_iter
        # the same group in batch mode afterwards, reduced explicitly, then per iteration again (the parked batches are discarded)
        g.init(sc, flags=gpu.PT_FLAG_TRACE_AHEAD, max_batch=4, pipeline_depth=2)
        g.iterate_batch(1, 4)
        g.reduce()
        assert np.array_equal(g.readback().view(np.uint32), want[3].view(np.uint32))
        g.iterate(5)
        g.iterate(6)                                              # two assemblies without a read-back between them
        assert np.array_equal(g.readback().view(np.uint32), want[5].view(np.uint32))
        g.sync()
    finally:
        g.destroy()


def test_group_survives_a_failed_reduce(gpu, monkeypatch):
    # an error between ncclGroupStart and ncclGroupEnd used to leave RCCL's call group open.  The test
    # library's hook issues the next reduce with a NULL communicator (ncclInvalidArgument): the call fails with the library's message,
    # ncclGroupEnd has run -- the NEXT assembly succeeds -- and the render goes on, bit-identical.
    monkeypatch.setenv("PT_AMD_COLLECTIVE", "rccl")
    sc = _scene(gpu, res=(160, 90))
    whole4, _ = _single(gpu, sc, [(1, 4)], max_batch=4)
    whole8, _ = _single(gpu, sc, [(1, 4), (5, 4)], max_batch=4)
    with gpu.renderer_from_test_library():
        g = gpu.Group(2)
        try:
            if not g.collective.startswith("rccl reduce"):
                pytest.skip("librccl.so could not be loaded in this process: %s" % g.collective)
            g.init(sc, max_batch=4)
            g.iterate_batch(1, 4)
            assert gpu.test_lib().pt_test_group_fail_next_reduce(g.handle, 1) == 0
            with pytest.raises(gpu.PtError, match="ncclReduce failed"):
                g.readback()
            assert np.array_equal(g.readback().view(np.uint32), whole4.view(np.uint32))       # the same frame, assembled on the second try
            g.iterate_batch(5, 4)
            assert gpu.test_lib().pt_test_group_fail_next_reduce(g.handle, 2) == 0
            for _ in range(2):
                with pytest.raises(gpu.PtError, match="ncclReduce failed"):
                    g.reduce()
            g.reduce()
            assert np.array_equal(g.readback().view(np.uint32), whole8.view(np.uint32))
        finally:
            g.destroy()


def test_context_cannot_be_destroyed_while_current_on_another_thread(gpu):
    # pt_ctx_destroy of a context that is another thread's current one left that thread's t_ctx dangling
    import threading
    L = gpu.lib()
    ctx = gpu.Context()
    entered, leave, seen = threading.Event(), threading.Event(), {}

    def other():
        seen["make"] = L.pt_ctx_make_current(ctx.handle)
        entered.set()
        leave.wait(30)
        seen["back"] = L.pt_ctx_make_current(None)

    t = threading.Thread(target=other)
    t.start()
    try:
        assert entered.wait(30) and seen["make"] == 0
        assert L.pt_ctx_destroy(ctx.handle) == -1                  # PT_ERR_INVALID
        assert b"current on another thread" in L.pt_last_error()
    finally:
        leave.set()
        t.join()
    assert seen["back"] == 0
    with ctx:                                                      # current on THIS thread alone: allowed, the thread falls back to the default
        ctx.destroy()
    assert L.pt_ctx_current() is None
    # a thread that ENDS with a context current releases it too
    ctx2 = gpu.Context()
    t = threading.Thread(target=lambda: L.pt_ctx_make_current(ctx2.handle))
    t.start()
    t.join()
    ctx2.destroy()


def test_scan_workspaces_outlive_another_contexts_free(gpu):
    # the scan library's workspaces are keyed by (device, stream) and pt_free waits for every device that owns
    # one before it releases them: a thread that scans on a stream of its own while another thread initialises and frees renderers
    # must get correct results every time (a released workspace is allocated again on the next call)
    import threading
    import torch
    n = (1 << 20) + 77
    x = torch.randint(0, 3, (n,), dtype=torch.int32, device="cuda")
    want = (torch.cumsum(x, 0, dtype=torch.int64) - x).to(torch.int32)
    sc = _scene(gpu, res=(64, 48))
    stop, errors = threading.Event(), []

    def churn():
        c = gpu.Context()
        try:
            with c:
                while not stop.is_set():
                    gpu.pathtraceInit(sc, max_batch=2)
                    gpu.pathtrace_batch(None, 0, 1, 2)
                    gpu.pathtraceFree()                              # pt_free: releases the scan workspaces too
        except Exception as e:                                       # noqa: BLE001
            errors.append(repr(e))
        finally:
            with c:
                pass
            c.destroy()

    t = threading.Thread(target=churn)
    t.start()
    try:
        st = torch.cuda.Stream()
        y = torch.empty_like(x)
        for _ in range(60):
            with torch.cuda.stream(st):
                gpu.scan_exclusive_dev(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
            st.synchronize()
            assert torch.equal(y, want)
    finally:
        stop.set()
        t.join()
    assert errors == []
