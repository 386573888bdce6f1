"""PT_FLAG_TRACE_AHEAD: the reference's protocol -- pathtrace(pbo, frame, iter) once per iteration (src/main.cpp:97-103) -- served
from wavefront batches traced ahead of the calls.  The accumulator after EVERY call must be what the call-by-call renderer (and
the oracle) has, bit for bit, whatever the caller does next: carry on, skip, repeat, go back, switch to explicit batches, move
the camera (Free / Init), read back in between."""
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


def _scene(gpu, name, res):
    sc = gpu.Scene(os.path.join(SCENES, name))
    sc.set_resolution(*res)
    return sc


def _oracle_for(oracle, sc, depth):
    return oracle.Renderer(sc.camera.view(oracle.CAMERA_DTYPE), sc.geoms.view(oracle.GEOM_DTYPE), sc.materials.view(oracle.MATERIAL_DTYPE), depth)


@pytest.mark.parametrize("max_batch,pipeline", [(2, 1), (5, 2), (32, 3), (8, 4)])
def test_image_after_every_call_equals_the_oracle(gpu, oracle, max_batch, pipeline):
    res, depth, n = (96, 54), 6, 23
    sc = _scene(gpu, "cornell_glass.txt", res)
    ref = _oracle_for(oracle, sc, depth)
    want = np.zeros(res[0] * res[1] * 3, np.float32)
    gpu.pathtraceFree()
    gpu.pathtraceInit(sc, traceDepth=depth, max_batch=max_batch, pipeline_depth=pipeline, trace_ahead=True)
    for it in range(1, n + 1):
        gpu.pathtrace(None, 0, it, readback=False)
        ref.iterate(it, want)
        got = gpu.readback(res[0] * res[1])                 # the reference copies the image back after every iteration
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), it
    cnt = gpu.counters()
    gpu.pathtraceFree()
    assert cnt.iterations == n                               # committed iterations; live[] also covers what was traced ahead
    assert cnt.live[1] >= n * res[0] * res[1] and cnt.live[1] % (res[0] * res[1]) == 0


def test_calls_that_break_the_sequence(gpu, oracle):
    # skip ahead inside the parked batch, beyond it, go back, repeat an iteration, an explicit batch in between, the last
    # admissible iterations: what is parked is dropped and never reaches the image
    res, depth = (80, 60), 5
    sc = _scene(gpu, "cornell.txt", res)
    ref = _oracle_for(oracle, sc, depth)
    want = np.zeros(res[0] * res[1] * 3, np.float32)
    gpu.pathtraceFree()
    gpu.pathtraceInit(sc, traceDepth=depth, max_batch=6, pipeline_depth=2, trace_ahead=True)
    last = (1 << 22) - 1
    calls = [1, 2, 3, 5, 6, 40, 41, 4, 4, 4, 5, ("batch", 7, 6), 13, 14, 15, 16, 17, 18, 19, 20, ("batch", 100, 1), 101, last - 1, last, 9]
    for c in calls:
        if isinstance(c, tuple):
            _, first, count = c
            gpu.pathtrace_batch(None, 0, first, count)       # count 1 = pt_iterate: continues or restarts the sequence
            for k in range(count):
                ref.iterate(first + k, want)
        else:
            gpu.pathtrace(None, 0, c, readback=False)
            ref.iterate(c, want)
        got = gpu.readback(res[0] * res[1])
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), c
    with pytest.raises(gpu.PtError, match="iter must be"):
        gpu.pathtrace(None, 0, last + 1, readback=False)
    got = gpu.readback(res[0] * res[1])
    gpu.pathtraceFree()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_camera_move_restart_and_row_shards(gpu, oracle):
    # src/main.cpp:91-95: a camera move is Free + Init + iteration 1 again -- nothing traced ahead for the old camera survives;
    # and a row shard with its packed accumulator (what a rank of a multi-GPU run holds) is served the same way
    res, depth = (64, 48), 4
    sc = _scene(gpu, "cornell.txt", res)
    gpu.pathtraceFree()
    gpu.pathtraceInit(sc, traceDepth=depth, max_batch=16, pipeline_depth=3, trace_ahead=True)
    for it in range(1, 8):
        gpu.pathtrace(None, 0, it, readback=False)
    gpu.pathtraceFree()
    sc.camera["position"][0] += np.float32([0.5, 0.25, -1.0])
    ref = _oracle_for(oracle, sc, depth)
    world, rank = 3, 1
    want = np.zeros(res[0] * res[1] * 3, np.float32)
    gpu.pathtraceInit(sc, traceDepth=depth, max_batch=16, pipeline_depth=3, trace_ahead=True, shard_rank=rank, shard_count=world,
                      flags=gpu.PT_FLAG_ACCUM_SHARD_ROWS)
    for it in range(1, 21):
        gpu.pathtrace(None, 0, it, readback=False)
        ref.iterate(it, want, rank, world)
    got = gpu.readback(res[0] * res[1])
    gpu.pathtraceFree()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_flag_without_a_batch_is_the_plain_protocol(gpu, oracle):
    res, depth = (48, 32), 3
    sc = _scene(gpu, "sphere.txt", res)
    ref = _oracle_for(oracle, sc, depth)
    want = np.zeros(res[0] * res[1] * 3, np.float32)
    gpu.pathtraceFree()
    gpu.pathtraceInit(sc, traceDepth=depth, trace_ahead=True)          # max_batch 1: nothing to trace ahead
    live = 0
    for it in range(1, 6):
        gpu.pathtrace(None, 0, it, readback=False)
        live += ref.iterate(it, want).live[1]
    got = gpu.readback(res[0] * res[1])
    cnt = gpu.counters()
    gpu.pathtraceFree()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert cnt.iterations == 5 and cnt.live[1] == live


def test_pbo_conversion_uses_the_committed_image(gpu, oracle):
    # pathtrace(pbo, frame, iter) with a PBO: sendImageToPBO of the image after iteration iter (src/pathtrace.cu:48-68), not of
    # anything traced ahead
    torch = pytest.importorskip("torch")
    res, depth = (64, 40), 4
    sc = _scene(gpu, "cornell.txt", res)
    ref = _oracle_for(oracle, sc, depth)
    want = np.zeros(res[0] * res[1] * 3, np.float32)
    pbo = torch.zeros(res[0] * res[1] * 4, dtype=torch.uint8, device="cuda")
    gpu.pathtraceFree()
    gpu.pathtraceInit(sc, traceDepth=depth, max_batch=8, pipeline_depth=2, trace_ahead=True)
    for it in range(1, 12):
        gpu.pathtrace(pbo.data_ptr(), 0, it, readback=False)
        ref.iterate(it, want)
        gpu.sync()
        rgba = pbo.cpu().numpy().reshape(-1, 4)
        assert np.array_equal(rgba, oracle.to_rgba8(want, it)), it
    gpu.pathtraceFree()


@pytest.mark.parametrize("seed", range(int(os.environ.get("PT_SEQ_SEEDS", "12"))))      # PT_SEQ_SEEDS=<n>: a longer one-off run
def test_random_call_sequences_equal_the_plain_protocol(gpu, seed):
    # a random walk over the API -- next iteration, any iteration, explicit batches, read-backs, counter resets, re-initialisation --
    # with trace-ahead on against the very same calls with it off (GPU against GPU: the image after every call must agree bit
    # for bit, and so must the number of committed iterations)
    rng = np.random.default_rng(9100 + seed)
    res = (int(rng.choice([40, 256, 300])), int(rng.choice([24, 33])))
    depth = int(rng.integers(2, 7))
    sc = _scene(gpu, str(rng.choice(["cornell.txt", "cornell_glass.txt", "spheres64.txt", "mesh_small.txt"])), res)
    max_batch, slots = int(rng.choice([2, 3, 8, 32])), int(rng.choice([1, 2, 3, 4]))
    ops, it = [], 1
    for _ in range(int(rng.integers(20, 60))):
        r = rng.random()
        if r < 0.55:
            ops.append(("iter", it)); it += 1
        elif r < 0.65:
            it = int(rng.integers(1, 200)); ops.append(("iter", it)); it += 1
        elif r < 0.75:
            n = int(rng.integers(1, max_batch + 1)); ops.append(("batch", it, n)); it += n
        elif r < 0.85:
            ops.append(("read",))
        elif r < 0.92:
            ops.append(("reset",))
        else:
            ops.append(("init",)); it = 1
    def play(ahead):
        images, iters = [], []
        gpu.pathtraceFree()
        gpu.pathtraceInit(sc, traceDepth=depth, max_batch=max_batch, pipeline_depth=slots, trace_ahead=ahead)
        for op in ops:
            if op[0] == "iter":
                gpu.pathtrace(None, 0, op[1], readback=False)
            elif op[0] == "batch":
                gpu.pathtrace_batch(None, 0, op[1], op[2])
            elif op[0] == "read":
                images.append(gpu.readback(res[0] * res[1]).copy())
            elif op[0] == "reset":
                iters.append(int(gpu.counters().iterations))
                gpu.counters_reset()
            else:
                images.append(gpu.readback(res[0] * res[1]).copy())
                gpu.pathtraceFree()
                gpu.pathtraceInit(sc, traceDepth=depth, max_batch=max_batch, pipeline_depth=slots, trace_ahead=ahead)
        images.append(gpu.readback(res[0] * res[1]).copy())
        iters.append(int(gpu.counters().iterations))
        gpu.pathtraceFree()
        return images, iters
    plain, n_plain = play(False)
    ahead, n_ahead = play(True)
    assert n_plain == n_ahead, (seed, ops)
    assert len(plain) == len(ahead)
    for k, (a, b) in enumerate(zip(plain, ahead)):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (seed, k, ops)
    assert max(float(a.max()) for a in plain) > 0 or all(o[0] in ("read", "reset", "init") for o in ops)   # (something was rendered)
