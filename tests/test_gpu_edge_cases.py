"""GPU edge cases against the oracle (bit-exact): degenerate frames and scenes, limits of the API,
rotated geometry with an oblique camera, and the full BASELINE size."""
import os
import types

import numpy as np
import pytest

from conftest import SCENES

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(pt):
    if pt.device_count() < 1:
        pytest.fail("no HIP device: GPU tests must run on the MI355X box")
    return pt


def _scene(gpu, oracle, geoms, mats, res, depth, eye=(0, 5, 10.5), view=(0, 0, -1), up=(0, 1, 0), fovy=45.0):
    cam = np.zeros(1, oracle.CAMERA_DTYPE)
    cam["resolution"] = res
    cam["position"], cam["view"], cam["up"] = eye, view, up
    cam["fov"] = (0.0, fovy)
    oracle.lib().orc_camera_set_resolution(cam.ctypes.data, res[0], res[1])
    return types.SimpleNamespace(geoms=np.ascontiguousarray(geoms).view(gpu.GEOM_DTYPE) if len(geoms) else np.zeros(0, gpu.GEOM_DTYPE),
                                 materials=np.ascontiguousarray(mats).view(gpu.MATERIAL_DTYPE),
                                 camera=cam.view(gpu.CAMERA_DTYPE), traceDepth=depth,
                                 image=np.zeros((res[1], res[0], 3), np.float32))


def _compare(gpu, oracle, sc, iters, **init):
    W, H = (int(v) for v in sc.camera["resolution"][0])
    ref = oracle.Renderer(sc.camera.view(oracle.CAMERA_DTYPE), sc.geoms.view(oracle.GEOM_DTYPE),
                          sc.materials.view(oracle.MATERIAL_DTYPE), sc.traceDepth)
    want = np.zeros(W * H * 3, np.float32)
    gpu.pathtraceFree()
    gpu.pathtraceInit(sc, **init)
    for it in iters:
        gpu.pathtrace(None, 0, it, readback=False)
        ref.iterate(it, want)
    got = gpu.readback(W * H)
    gpu.pathtraceFree()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    return got


def _mat(oracle, color=(1, 1, 1), emit=0.0, refl=0.0, refr=0.0, ior=0.0, spec=(0, 0, 0)):
    m = np.zeros(1, oracle.MATERIAL_DTYPE)
    m["color"], m["specColor"], m["emittance"] = color, spec, emit
    m["hasReflective"], m["hasRefractive"], m["indexOfRefraction"] = refl, refr, ior
    return m


def test_one_pixel_frame_and_depth_one(gpu, oracle):
    light = oracle.make_geom(0, 0, (0, 5, 0), (0, 0, 0), (30, 30, 30))          # camera inside an emissive sphere
    sc = _scene(gpu, oracle, light, _mat(oracle, emit=2.0), (1, 1), 1)
    img = _compare(gpu, oracle, sc, [1, 2, 3])
    assert img.tolist() == [6.0, 6.0, 6.0]


def test_empty_scene_is_black(gpu, oracle):
    sc = _scene(gpu, oracle, [], _mat(oracle), (33, 17), 4)
    img = _compare(gpu, oracle, sc, [1])
    assert not img.any()


def test_more_shards_than_rows(gpu, oracle):
    sc = gpu.Scene(os.path.join(SCENES, "cornell.txt"))
    sc.set_resolution(40, 3)
    ref = oracle.Renderer(sc.camera.view(oracle.CAMERA_DTYPE), sc.geoms.view(oracle.GEOM_DTYPE),
                          sc.materials.view(oracle.MATERIAL_DTYPE), 8)
    want = np.zeros(40 * 3 * 3, np.float32)
    ref.iterate(1, want)
    acc = np.zeros_like(want)
    for r in range(5):                                    # ranks 3 and 4 own no row at all
        gpu.pathtraceFree()
        gpu.pathtraceInit(sc, shard_rank=r, shard_count=5)
        gpu.pathtrace(None, 0, 1, readback=False)
        acc += gpu.readback(40 * 3)
        assert gpu.counters().live[1] == (40 if r < 3 else 0)
    gpu.pathtraceFree()
    assert np.array_equal(acc.view(np.uint32), want.view(np.uint32))


def test_maximum_depth_and_iteration_index(gpu, oracle):
    sc = gpu.Scene(os.path.join(SCENES, "cornell_glass.txt"))
    sc.set_resolution(48, 40)
    sc.traceDepth = gpu.PT_MAX_DEPTH
    _compare(gpu, oracle, sc, [4194302, 4194303], traceDepth=gpu.PT_MAX_DEPTH)
    with pytest.raises(gpu.PtError, match="traceDepth"):
        gpu.pathtraceInit(sc, traceDepth=gpu.PT_MAX_DEPTH + 1)
    gpu.pathtraceInit(sc)
    with pytest.raises(gpu.PtError, match="iter"):
        gpu.pathtrace(None, 0, 4194304)
    gpu.pathtraceFree()


def test_rotated_geometry_oblique_camera(gpu, oracle):
    sc = gpu.Scene(os.path.join(SCENES, "rotated.txt"))
    img = _compare(gpu, oracle, sc, [1, 2, 3, 4])
    assert img.max() > 0
    # per-bounce state too (arrival order on the device, pixel order here)
    ref = oracle.Renderer(sc.camera.view(oracle.CAMERA_DTYPE), sc.geoms.view(oracle.GEOM_DTYPE),
                          sc.materials.view(oracle.MATERIAL_DTYPE), sc.traceDepth)
    gpu.pathtraceInit(sc)
    for b in (0, 1, 3, 6):
        o, d, c, pix = gpu.debug_trace_paths(7, b, 320 * 200)
        wo, wd, wc, wpix = ref.dump_paths(7, b)
        assert np.array_equal(pix, wpix)
        assert np.array_equal(o.view(np.uint32), wo.view(np.uint32)) and np.array_equal(d.view(np.uint32), wd.view(np.uint32))
        assert np.array_equal(c.view(np.uint32), wc.view(np.uint32))
    gpu.pathtraceFree()


def test_two_hundred_spheres_in_lds_and_scalar_path(gpu, oracle):
    rng = np.random.default_rng(5)
    geoms = [oracle.make_geom(1, 1, (0, -1, 0), (0, 0, 0), (40, 1, 40)), oracle.make_geom(0, 0, (0, 14, 0), (0, 0, 0), (8, 1, 8))]
    for _ in range(200):
        geoms.append(oracle.make_geom(0, int(rng.integers(1, 4)), rng.uniform(-7, 7, 3) + (0, 6, 0), rng.uniform(-90, 90, 3),
                                      rng.uniform(0.3, 1.5, 3)))
    mats = np.concatenate([_mat(oracle, emit=4.0), _mat(oracle, (.8, .8, .8)), _mat(oracle, (.9, .3, .3), refl=1.0, spec=(.9, .9, .9)),
                           _mat(oracle, (.95, .95, .95), refr=1.0, ior=1.5, spec=(.95, .95, .95))])
    sc = _scene(gpu, oracle, np.concatenate(geoms), mats, (96, 64), 8, eye=(0, 6, 16))
    _compare(gpu, oracle, sc, [1, 2])


def test_coincident_and_nested_spheres_keep_file_order(gpu, oracle):
    # Sphere-heavy scenes record a lane's candidate spheres (8 per lane, in-place test beyond) and test them after the
    # loop over the primitives: 12 COINCIDENT spheres with different materials (equal distances: the first in file order
    # must win, whether it was recorded or tested in place), 10 nested ones around them, a cube in between the indices.
    geoms = [oracle.make_geom(1, 1, (0, -1, 0), (0, 0, 0), (30, 1, 30)), oracle.make_geom(0, 0, (0, 12, 0), (0, 0, 0), (6, 1, 6))]
    for k in range(12):
        geoms.append(oracle.make_geom(0, 1 + k % 3, (0, 4, 0), (0, 0, 0), (3, 3, 3)))
    geoms.append(oracle.make_geom(1, 2, (0, 4, 0), (0, 0, 0), (3, 3, 3)))              # a cube circumscribing them
    for k in range(10):
        geoms.append(oracle.make_geom(0, 3, (0, 4, 0), (10 * k, 5 * k, 0), (3.2 + 0.4 * k, 3.2 + 0.3 * k, 3.2 + 0.5 * k)))
    mats = np.concatenate([_mat(oracle, emit=4.0), _mat(oracle, (.8, .8, .8)), _mat(oracle, (.9, .3, .3), refl=1.0, spec=(.9, .9, .9)),
                           _mat(oracle, (.95, .95, .95), refr=1.0, ior=1.5, spec=(.95, .95, .95))])
    sc = _scene(gpu, oracle, np.concatenate(geoms), mats, (96, 64), 10, eye=(0, 5, 14))
    _compare(gpu, oracle, sc, [1, 2, 3], max_batch=1)
    _compare(gpu, oracle, sc, [1, 2, 3], max_batch=4, pipeline_depth=2)


def test_full_baseline_frame_is_bit_identical(gpu, oracle):
    # BASELINE config C2 at its real size (1280x720, depth 8), 2 spp: every one of the 2.7 M floats
    sc = gpu.Scene(os.path.join(SCENES, "cornell.txt"))
    sc.set_resolution(1280, 720)
    _compare(gpu, oracle, sc, [1, 2], max_batch=8)


def test_large_frame_properties(gpu):
    # 4096x4096 (config C5's frame): too slow for the CPU oracle, so size-independent properties:
    # conservation of paths per bounce, and the row-sharded halves sum to the unsharded frame bit for bit
    sc = gpu.Scene(os.path.join(SCENES, "spheres64.txt"))
    sc.set_resolution(4096, 4096)
    P = 4096 * 4096
    gpu.pathtraceFree()
    gpu.pathtraceInit(sc, pipeline_depth=1)
    gpu.pathtrace(None, 0, 1, readback=False)
    full = gpu.readback(P)
    c = gpu.counters()
    live = [int(c.live[d]) for d in range(10)]
    assert live[1] == P and all(live[d + 1] <= live[d] for d in range(1, 8))
    # every path ends exactly once: on an emitter, in the void, or at the depth limit (survivors of bounce 8)
    assert c.light_hits + c.misses <= P and c.light_hits > 0 and c.misses > 0
    acc = np.zeros_like(full)
    for r in range(2):
        gpu.pathtraceFree()
        gpu.pathtraceInit(sc, shard_rank=r, shard_count=2, pipeline_depth=1)
        gpu.pathtrace(None, 0, 1, readback=False)
        acc += gpu.readback(P)
    gpu.pathtraceFree()
    assert np.array_equal(acc.view(np.uint32), full.view(np.uint32))
    assert np.isfinite(full).all() and full.min() >= 0


def test_config_c3_as_written_eight_row_shards(gpu):
    # BASELINE config C3 as written -- cornell.txt at 1280x720, 5000 spp, depth 8, rows dealt to 8 ranks -- on one GPU: every
    # rank's share (256 iterations per wavefront batch, as bench.py fuses them for 8 ranks) equals the rows of the unsharded
    # 5000-spp render bit for bit, and so does a second unsharded render under another schedule
    sc = gpu.Scene(os.path.join(SCENES, "cornell.txt"))
    sc.set_resolution(1280, 720)
    P, spp = 1280 * 720, 5000

    def render(batch, pipe, **kw):
        gpu.pathtraceFree()
        gpu.pathtraceInit(sc, traceDepth=8, pipeline_depth=pipe, max_batch=batch, **kw)
        it = 1
        while it <= spp:
            n = min(batch, spp - it + 1)
            gpu.pathtrace_batch(None, 0, it, n)
            it += n
        return gpu.readback(P).reshape(720, 1280 * 3)

    full = render(32, 2)
    assert full.max() > 0 and np.array_equal(render(64, 3).view(np.uint32), full.view(np.uint32))
    for r in range(8):
        part = render(256, 2, shard_rank=r, shard_count=8, flags=gpu.PT_FLAG_ACCUM_SHARD_ROWS)
        mine = np.arange(720) % 8 == r
        assert np.array_equal(part[mine].view(np.uint32), full[mine].view(np.uint32)), r
        assert not part[~mine].any()
    c = gpu.counters()
    assert int(c.live[1]) == spp * 1280 * 90
    gpu.pathtraceFree()


def test_config_c5_as_written_eight_row_shards(gpu):
    # BASELINE config C5 as written -- spheres64 at 4096x4096, 16 spp, depth 8, rows dealt to 8 ranks -- on one GPU: every rank's
    # share (packed accumulator, shard-local radiance buffers, batches of 8 fused like bench.py fuses them) scattered back
    # into the frame equals the unsharded render bit for bit
    sc = gpu.Scene(os.path.join(SCENES, "spheres64.txt"))
    sc.set_resolution(4096, 4096)
    P, spp = 4096 * 4096, 16
    gpu.pathtraceFree()
    gpu.pathtraceInit(sc, pipeline_depth=2, max_batch=8)
    for it in range(1, spp + 1, 8):
        gpu.pathtrace_batch(None, 0, it, 8)
    full = gpu.readback(P).reshape(4096, 4096 * 3)
    assert full.max() > 0
    for r in range(8):
        gpu.pathtraceFree()
        gpu.pathtraceInit(sc, shard_rank=r, shard_count=8, pipeline_depth=2, max_batch=16, flags=gpu.PT_FLAG_ACCUM_SHARD_ROWS)
        gpu.pathtrace_batch(None, 0, 1, spp)
        part = gpu.readback(P).reshape(4096, 4096 * 3)
        mine = np.arange(4096) % 8 == r
        assert np.array_equal(part[mine].view(np.uint32), full[mine].view(np.uint32)), r
        assert not part[~mine].any()
    gpu.pathtraceFree()


def test_headless_driver_renders_the_reference_protocol(gpu, oracle, tmp_path):
    # pt_render = main()/runCuda()/saveImage() of the reference (src/main.cpp:21-113) over the C++ shim:
    # Free -> Init at iteration 0, 1-based iterations with a D2H copy each, save at the end (X mirror, /samples)
    import subprocess
    from test_host import _decode_png
    from conftest import ROOT
    exe = os.path.join(ROOT, "project3-cuda-path-tracer_amd", "host", "pt_render")
    base = str(tmp_path / "render")
    r = subprocess.run([exe, os.path.join(SCENES, "cornell.txt"), "--res", "96", "64", "--iterations", "5", "--depth", "8",
                        "--out", base, "--hdr"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = _decode_png(base + ".png")
    sc = oracle.Scene(os.path.join(SCENES, "cornell.txt"))
    sc.set_resolution(96, 64)
    ref = oracle.Renderer(sc.camera, sc.geoms, sc.materials, 8)
    img = np.zeros(96 * 64 * 3, np.float32)
    for it in range(1, 6):
        ref.iterate(it, img)
    want = (np.clip(img.reshape(64, 96, 3) / np.float32(5), 0, 1) * np.float32(255)).astype(np.uint8)[:, ::-1]
    assert np.array_equal(got, want)
    assert os.path.getsize(base + ".hdr") > 4 * 96 * 64
    # --batch: iterations traced 4 at a time through the C ABI, one D2H copy before the save -- the same file, byte for byte
    r = subprocess.run([exe, os.path.join(SCENES, "cornell.txt"), "--res", "96", "64", "--iterations", "5", "--depth", "8",
                        "--out", base + "_b", "--batch", "4"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "one D2H copy" in r.stdout
    assert open(base + "_b.png", "rb").read() == open(base + ".png", "rb").read()


def test_headless_driver_renders_a_frame_beyond_2_24_pixels(gpu, oracle, tmp_path):
    # A host written against the reference's pathtraceInit knows nothing of batches: the shim's trace-ahead (32 iterations per
    # wavefront batch by default) must never be the reason an Init fails.  8192 x 2064 = 16.9 M pixels: 32 x that many paths exceed
    # the library's 2^29 per batch, so the shim clamps the batch (to 16) -- round 2's shim exited in pathtraceInit here.
    import subprocess
    from test_host import _decode_png
    from conftest import ROOT
    W, H, its, depth = 8192, 2064, 3, 2
    assert W * H > 1 << 24
    exe = os.path.join(ROOT, "project3-cuda-path-tracer_amd", "host", "pt_render")
    base = str(tmp_path / "big")
    r = subprocess.run([exe, os.path.join(SCENES, "cornell.txt"), "--res", str(W), str(H), "--iterations", str(its), "--depth", str(depth),
                        "--out", base], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    got = _decode_png(base + ".png")
    assert got.shape == (H, W, 3)
    # every 64th row against the oracle (rows y with y % 64 == 7), X mirrored like the file
    sc = oracle.Scene(os.path.join(SCENES, "cornell.txt"))
    sc.set_resolution(W, H)
    ref = oracle.Renderer(sc.camera, sc.geoms, sc.materials, depth)
    img = np.zeros(W * H * 3, np.float32)
    for it in range(1, its + 1):
        ref.iterate(it, img, 7, 64)
    want = (np.clip(img.reshape(H, W, 3) / np.float32(its), 0, 1) * np.float32(255)).astype(np.uint8)[:, ::-1]
    assert want[7::64].max() > 0 and np.array_equal(got[7::64], want[7::64])


@pytest.mark.parametrize("eye,view,up,fovy", [
    ((0, 5, 4.9), (0, 0, -1), (0, 1, 0), 45.0),          # inside the room: every wall's bounding cube surrounds the eye
    ((0, 5, 30), (0.02, -0.01, -1), (0, 1, 0), 12.0),    # far away, narrow: the room covers a small pixel rectangle
    ((12, 9, 9), (-1.2, -0.5, -1), (0.05, 1, 0), 30.0),  # oblique, non-unit view, skewed up: partly off-screen
    ((0, 5, 10.5), (0, 0, 1), (0, 1, 0), 45.0),          # looking away: everything behind the camera
    ((-4.99, 5, 0), (1, 0.2, 0.1), (0, 1, 0), 50.0),     # eye almost touching the left wall
    ((0, 9.6, 0), (0, -1, 0.001), (0, 0, -1), 40.0),     # just below the light, looking down
])
def test_camera_ray_pixel_rectangles_never_cull_a_hit(gpu, oracle, eye, view, up, fovy):
    # first-bounce culling by projected bounding cubes (GeomDev::rect / KParams::sceneRect) must be invisible
    z = oracle.Scene(os.path.join(SCENES, "cornell_glass.txt"))
    sc = _scene(gpu, oracle, z.geoms, z.materials, (150, 90), 6, eye=eye, view=view, up=up, fovy=fovy)
    _compare(gpu, oracle, sc, [1, 2], max_batch=2)


# ----------------------------------------------------------------------------- chunked path pools
def _one_class_scene(gpu, oracle, res, depth):
    """Nearly every path in ONE class of the queue for several bounces: a narrow camera looks along (1, 1, -1) through a
    stack of large index-1 "glass" slabs (Schlick r0 = 0, refraction leaves the direction unchanged: every survivor keeps
    the camera ray's octant), an emitter behind them.  There is no small primitive, so the candidate bit is constant too."""
    geoms = []
    for k in range(3):                                   # 3 slabs = 6 refractions, the emitter ends the path at bounce 7
        geoms.append(oracle.make_geom(1, 1, (4 + 3 * k, 4 + 3 * k, -4 - 3 * k), (0, 0, 0), (40, 40, 0.5)))
    geoms.append(oracle.make_geom(1, 0, (30, 30, -30), (0, 0, 0), (80, 80, 1)))
    mats = np.concatenate([_mat(oracle, emit=3.0), _mat(oracle, (.9, .95, .97), refr=1.0, ior=1.0, spec=(.9, .9, .9))])
    return _scene(gpu, oracle, np.concatenate(geoms), mats, res, depth, eye=(0, 0, 0), view=(1, 1, -1), up=(0, 1, 0), fovy=8.0)


def test_every_path_in_one_class_fits_the_pools(gpu, oracle):
    # 512 x 512 x 4 iterations = 1 M paths = 512 chunks of 2048, and four segments (one class x four counter shards)
    # receive practically all of them: the chunk lists grow on demand, far beyond an even split over the 64 segments
    sc = _one_class_scene(gpu, oracle, (512, 512), 8)
    W = H = 512
    ref = oracle.Renderer(sc.camera.view(oracle.CAMERA_DTYPE), sc.geoms.view(oracle.GEOM_DTYPE),
                          sc.materials.view(oracle.MATERIAL_DTYPE), sc.traceDepth)
    want = np.zeros(W * H * 3, np.float32)
    live = np.zeros(16, np.int64)
    for it in range(1, 5):
        c = ref.iterate(it, want)
        live += np.array(c.live[:16])
    gpu.pathtraceFree()
    gpu.pathtraceInit(sc, max_batch=4, pipeline_depth=2)
    gpu.pathtrace_batch(None, 0, 1, 4)
    got = gpu.readback(W * H)
    cnt = gpu.counters()
    assert [int(cnt.live[d]) for d in range(1, 9)] == live[1:9].tolist()
    assert live[5] > 0.9 * live[1]                       # the stack really keeps (nearly) every path alive ...
    o, d, c, pix = gpu.debug_trace_paths(1, 3, W * H)
    octant = (d[:, 0] < 0).astype(int) | ((d[:, 1] < 0).astype(int) << 1) | ((d[:, 2] < 0).astype(int) << 2)
    assert np.bincount(octant, minlength=8).max() > 0.97 * len(d)      # ... and in one octant
    gpu.pathtraceFree()
    assert want.max() > 0 and np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_undersized_pool_fails_loudly(gpu, oracle, monkeypatch):
    # PT_AMD_POOL_CHUNKS (tests only) shrinks the pools below what the render needs: the kernels must stay in bounds,
    # and the fault must surface as PT_ERR_DEVICE -- at pt_sync, at pt_counters, and stay sticky until the next pt_init
    sc = _one_class_scene(gpu, oracle, (256, 256), 6)
    monkeypatch.setenv("PT_AMD_POOL_CHUNKS", "134")       # 128 static chunks (16 classes x 8 shards) + 5: 256 x 256 paths in ONE class need ~24 more
    gpu.pathtraceFree()
    gpu.pathtraceInit(sc, pipeline_depth=1)
    gpu.pathtrace(None, 0, 1, readback=False)
    with pytest.raises(gpu.PtError, match="path pool exhausted"):
        gpu.sync()
    with pytest.raises(gpu.PtError, match="device fault"):
        gpu.counters()
    gpu.pathtrace(None, 0, 2, readback=False)             # still memory-safe, still flagged
    with pytest.raises(gpu.PtError, match="device fault"):
        gpu.sync()
    # ... and by the calls a host of the reference protocol makes: the image copies report the faulted render too
    with pytest.raises(gpu.PtError, match="device fault"):
        gpu.readback(256 * 256)
    with pytest.raises(gpu.PtError, match="device fault"):
        gpu.readback_rgba8(2, 256 * 256)
    with pytest.raises(gpu.PtError, match="device fault"):
        gpu.pathtrace(None, 0, 3)                          # pathtrace() = pt_iterate + the copy into scene.image
    monkeypatch.delenv("PT_AMD_POOL_CHUNKS")
    _compare(gpu, oracle, sc, [1, 2])                     # a fresh pt_init with real pools renders correctly again


@pytest.mark.parametrize("grid", [8, 24, 56, 72])
def test_grids_smaller_than_the_ticket_shards_visit_every_tile(gpu, oracle, monkeypatch, grid):
    # tile tickets are drawn from min(kTicketShards, grid) shards -- with a grid below the 64 shards (a small or
    # partitioned device, one workgroup per CU; PT_AMD_MAX_GRID: tests only) a shard nobody draws from would leave its tiles, their
    # paths and their radiance out.  A frame of 1200 tiles per iteration on 8 .. 72 workgroups against the oracle, bit for bit,
    # and every path accounted for.
    sc = gpu.Scene(os.path.join(SCENES, "cornell.txt"))
    sc.set_resolution(640, 480)
    monkeypatch.setenv("PT_AMD_MAX_GRID", str(grid))
    _compare(gpu, oracle, sc, [1, 2, 3], max_batch=2, pipeline_depth=2)
    monkeypatch.delenv("PT_AMD_MAX_GRID")


def test_forced_fault_words_are_reported(gpu):
    import torch
    # renderer: a fault word set by hand is reported by pt_sync and pt_counters and survives a counter reset
    # (the hook that sets the word lives in the test library only -- the product exports no such thing -- so this one test
    # drives the renderer instance of libpt_amd_test.so)
    sc = gpu.Scene(os.path.join(SCENES, "cornell.txt"))
    sc.set_resolution(32, 32)
    gpu.pathtraceFree()
    assert not hasattr(gpu.lib(), "pt_test_force_fault")
    with gpu.renderer_from_test_library():
        gpu.pathtraceInit(sc)
        gpu.sync()
        gpu.force_fault(2)
        with pytest.raises(gpu.PtError, match="device fault"):
            gpu.sync()
        gpu.counters_reset()
        with pytest.raises(gpu.PtError, match="device fault"):
            gpu.sync()
        with pytest.raises(gpu.PtError, match="device fault"):
            gpu.readback(32 * 32)
        gpu.force_fault(0)                                    # (diagnostics: cleared, the renderer answers again)
        gpu.sync()
        gpu.readback(32 * 32)
        gpu.pathtraceFree()
    # the scan library has no fault word: no workgroup of it ever waits for another one.  Without a renderer pt_sync waits for it.
    x = torch.ones(5000, dtype=torch.int32, device="cuda")
    y = torch.empty_like(x)
    gpu.scan_exclusive_dev(x.data_ptr(), y.data_ptr(), x.numel())
    gpu.sync()
    assert int(y[-1]) == 4999
    with pytest.raises(gpu.PtError):
        gpu.force_fault(1)


def test_rejected_call_leaves_the_image_untouched(gpu, oracle):
    # a PBO conversion cannot be served from a row-sharded accumulator: the call must be refused BEFORE anything is
    # enqueued, so that a retry does not add the samples twice
    import torch
    sc = gpu.Scene(os.path.join(SCENES, "cornell.txt"))
    sc.set_resolution(64, 48)
    pbo = torch.zeros(64 * 48, dtype=torch.int32, device="cuda")
    gpu.pathtraceFree()
    gpu.pathtraceInit(sc, shard_rank=0, shard_count=2, flags=gpu.PT_FLAG_ACCUM_SHARD_ROWS)
    with pytest.raises(gpu.PtError, match="row-sharded"):
        gpu.pathtrace(pbo.data_ptr(), 0, 1, readback=False)
    assert gpu.counters().iterations == 0 and not gpu.readback(64 * 48).any()
    gpu.pathtrace(None, 0, 1, readback=False)
    got = gpu.readback(64 * 48)
    gpu.pathtraceFree()
    ref = oracle.Renderer(sc.camera.view(oracle.CAMERA_DTYPE), sc.geoms.view(oracle.GEOM_DTYPE),
                          sc.materials.view(oracle.MATERIAL_DTYPE), sc.traceDepth)
    want = np.zeros(64 * 48 * 3, np.float32)
    ref.iterate(1, want, 0, 2)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_large_frame_batches_of_eight(gpu):
    # chunked pools: a 4096 x 4096 frame traces batches of 8 iterations on one GPU (134 M paths per launch: 12 GB of
    # path state per slot, where the worst-case-per-class provisioning of round 1 allowed batches of 1-2)
    sc = gpu.Scene(os.path.join(SCENES, "spheres64.txt"))
    sc.set_resolution(4096, 4096)
    P = 4096 * 4096
    gpu.pathtraceFree()
    gpu.pathtraceInit(sc, pipeline_depth=1, max_batch=8)
    gpu.pathtrace_batch(None, 0, 1, 8)
    a = gpu.readback(P)
    c = gpu.counters()
    assert int(c.live[1]) == 8 * P and c.light_hits > 0
    gpu.pathtraceFree()
    gpu.pathtraceInit(sc, pipeline_depth=1, max_batch=2)
    for it in (1, 3, 5, 7):
        gpu.pathtrace_batch(None, 0, it, 2)
    b = gpu.readback(P)
    gpu.pathtraceFree()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and a.max() > 0


def test_config_c4_frame_properties(gpu):
    # BASELINE config C4 at its real size (cornell_glass, 1920x1080, depth 16; the oracle covers it at 240x135): path
    # conservation per bounce, and three row shards traced as wavefront batches sum to the unsharded frame bit for bit
    sc = gpu.Scene(os.path.join(SCENES, "cornell_glass.txt"))
    sc.set_resolution(1920, 1080)
    P = 1920 * 1080
    gpu.pathtraceFree()
    gpu.pathtraceInit(sc, traceDepth=16, pipeline_depth=2, max_batch=4)
    gpu.pathtrace_batch(None, 0, 1, 4)
    full = gpu.readback(P)
    c = gpu.counters()
    live = [int(c.live[d]) for d in range(18)]
    assert live[1] == 4 * P and all(live[d + 1] <= live[d] for d in range(1, 16)) and live[16] > 0 and live[17] == 0
    assert c.light_hits > 0 and c.misses > 0 and c.light_hits + c.misses <= 4 * P
    early = sum(int(c.ended_early[d]) for d in range(18))
    assert 0 < early < c.misses                               # the open front: paths that end at their scatter
    acc = np.zeros_like(full)
    for r in range(3):
        gpu.pathtraceFree()
        gpu.pathtraceInit(sc, traceDepth=16, shard_rank=r, shard_count=3, pipeline_depth=2, max_batch=2)
        gpu.pathtrace_batch(None, 0, 1, 2)
        gpu.pathtrace_batch(None, 0, 3, 2)
        acc += gpu.readback(P)
    gpu.pathtraceFree()
    assert np.array_equal(acc.view(np.uint32), full.view(np.uint32))
    assert np.isfinite(full).all() and full.min() >= 0 and full.max() > 0


def test_closed_cornell_box_matches_the_oracle_and_lets_nothing_escape(gpu, oracle):
    # the reference's analysis scene (README.md:284-293): Cornell closed by a front wall, camera inside.  No light can escape, so a
    # path ends only on the light or at the depth limit: bit-identical to the oracle, zero misses, and every bounce's queue is the
    # previous one minus the paths that reached the light
    sc = gpu.Scene(os.path.join(SCENES, "cornell_closed.txt"))
    W, H = 200, 120
    sc.set_resolution(W, H)
    ref = oracle.Renderer(sc.camera.view(oracle.CAMERA_DTYPE), sc.geoms.view(oracle.GEOM_DTYPE), sc.materials.view(oracle.MATERIAL_DTYPE), 8)
    want = np.zeros(W * H * 3, np.float32)
    live = np.zeros(16, np.int64)
    hits = 0
    for it in range(1, 5):
        c = ref.iterate(it, want)
        live += np.array(c.live[:16])
        hits += int(c.lightHits)
    gpu.pathtraceFree()
    gpu.pathtraceInit(sc, traceDepth=8, max_batch=4, pipeline_depth=2)
    gpu.pathtrace_batch(None, 0, 1, 4)
    got = gpu.readback(W * H)
    cnt = gpu.counters()
    gpu.pathtraceFree()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)) and want.max() > 0
    assert [int(cnt.live[d]) for d in range(1, 10)] == live[1:10].tolist()
    assert int(cnt.misses) == 0 and int(cnt.light_hits) == hits
    assert all(int(cnt.ended_early[d]) == 0 for d in range(10))          # nothing to end early in a closed box
    assert live[8] > 0.5 * live[1]                                       # the closed box keeps most paths alive to the last bounce


@pytest.mark.gpu
def test_matrix_rows_from_global_memory_in_scenes_of_hundreds_of_primitives(gpu, oracle, monkeypatch):
    # round 5: a sphere-heavy workgroup stages every primitive's matrix row (112 B) in LDS -- at 518 primitives 58 KB of a 103 KB
    # workgroup, one per CU.  Beyond 40 KB of LDS the rows stay in global memory (KParams::ldsRowFloats = 0) and a per-lane test reads its
    # row from there.  The 518-sphere scene takes that path by itself; PT_AMD_ROWS_GLOBAL=1 forces it on the 64-sphere and the
    # 64-cube scene (the per-lane sphere AND box tests, camera rays and later bounces): the oracle's frames, bit for bit.
    sc = gpu.Scene(os.path.join(SCENES, "spheres512.txt"))
    sc.set_resolution(96, 64)
    _compare(gpu, oracle, sc, [1, 2])
    monkeypatch.setenv("PT_AMD_ROWS_GLOBAL", "1")
    for name in ("spheres64.txt", "cubes64.txt"):
        sc = gpu.Scene(os.path.join(SCENES, name))
        sc.set_resolution(96, 64)
        _compare(gpu, oracle, sc, [1, 2, 3])
        _compare(gpu, oracle, sc, [4, 5, 6, 7], max_batch=4, pipeline_depth=2)
    monkeypatch.delenv("PT_AMD_ROWS_GLOBAL")


@pytest.mark.gpu
def test_grouped_sweep_of_scenes_of_hundreds_of_primitives(gpu, oracle, monkeypatch):
    # round 6: beyond 128 swept primitives the later bounces take k_bounce<..., GROUPS> -- the table in spatial groups
    # of 16 with a bounding ball each, a two-level sweep (the groups' balls wave-uniform, then per LANE the members of the groups its ray may
    # reach), hit records / frames / matrix rows read from global memory.  The 518-sphere scene and the 200 random ellipsoids take it by
    # themselves; PT_AMD_GROUPS=1 forces it on the 64-sphere scene (two clusters, a light behind the first candidate bit, last-bounce bits)
    # and on the 64-cube scene (swept cubes: the per-lane box test, the frames' NaN row in global memory); PT_AMD_GROUPS=0 renders the
    # 518-sphere scene with the flat sweep.  All against the oracle, bit for bit -- frames and live counts.
    sc = gpu.Scene(os.path.join(SCENES, "spheres512.txt"))
    sc.set_resolution(96, 64)
    _compare(gpu, oracle, sc, [1, 2])
    _compare(gpu, oracle, sc, [3, 4, 5, 6], max_batch=4, pipeline_depth=2)
    monkeypatch.setenv("PT_AMD_GROUPS", "0")
    _compare(gpu, oracle, sc, [1, 2])
    monkeypatch.setenv("PT_AMD_GROUPS", "1")
    for name in ("spheres64.txt", "cubes64.txt"):
        sc = gpu.Scene(os.path.join(SCENES, name))
        sc.set_resolution(96, 64)
        _compare(gpu, oracle, sc, [1, 2, 3])
        _compare(gpu, oracle, sc, [4, 5, 6, 7], max_batch=4, pipeline_depth=2)
    # coincident and nested spheres (every one a candidate of the same groups: file order decides the ties), as a grouped scene
    geoms = [oracle.make_geom(1, 1, (0, -1, 0), (0, 0, 0), (30, 1, 30)), oracle.make_geom(0, 0, (0, 12, 0), (0, 0, 0), (6, 1, 6))]
    for k in range(12):
        geoms.append(oracle.make_geom(0, 1 + k % 3, (0, 4, 0), (0, 0, 0), (3, 3, 3)))
    for k in range(10):
        geoms.append(oracle.make_geom(0, 3, (0, 4, 0), (10 * k, 5 * k, 0), (3.2 + 0.4 * k, 3.2 + 0.3 * k, 3.2 + 0.5 * k)))
    mats = np.concatenate([_mat(oracle, emit=4.0), _mat(oracle, (.8, .8, .8)), _mat(oracle, (.9, .3, .3), refl=1.0, spec=(.9, .9, .9)),
                           _mat(oracle, (.95, .95, .95), refr=1.0, ior=1.5, spec=(.95, .95, .95))])
    sc = _scene(gpu, oracle, np.concatenate(geoms), mats, (96, 64), 10, eye=(0, 5, 14))
    _compare(gpu, oracle, sc, [1, 2, 3], max_batch=4, pipeline_depth=2)
    monkeypatch.delenv("PT_AMD_GROUPS")
