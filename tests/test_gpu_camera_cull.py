"""The screen-space culling of camera rays -- per-primitive pixel rectangles (GeomDev::rect), their union (KParams::sceneRect:
whole tiles skipped outside it) and the per-row primitive lists with their hull spans (pt_init: build_camera_cull) -- proven the
way every other shortcut of the kernels is:

  * a device sweep over >= 10^4 random (camera, primitive set) pairs (pt_test_camera_cull_sweep, test library): every pixel of the
    frame sends its camera rays through the FULL reference test (src/intersections.h:47-143) of EVERY primitive; a hit from a pixel
    the culling would have skipped is a violation.  Half-angles 1 - 85 degrees, frame widths 17 ... 8192, eyes inside bounding
    cubes, corners on and behind the eye plane, tilted up vectors, needles and plates;
  * the camera-ray bounce rendered with the culling ON and OFF (PT_AMD_NO_CAMERA_CULL, a tests-only switch of pt_init): live
    counts, tallies, the surviving paths of bounce 1 (which pin every pixel's hit: origin = hit point, colour = material) and the
    image are identical, bit for bit -- random scenes plus the hand-picked extremes at full width.
"""
import os
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(pt):
    if pt.device_count() < 1:
        pytest.fail("no HIP device: GPU tests must run on the MI355X box")
    return pt


# frame shapes: the widths the verdict names, heights chosen so that a case stays around 10^5 pixels
SHAPES = [(17, 33), (64, 64), (150, 90), (255, 40), (257, 30), (1920, 48), (4096, 24), (8192, 12), (1280, 72), (33, 1000)]


def _camera(oracle, rng, shape, eye, target, half_angle, tilt=True):
    cam = np.zeros(1, oracle.CAMERA_DTYPE)
    view = np.asarray(target, np.float64) - np.asarray(eye, np.float64)
    view /= max(np.linalg.norm(view), 1e-9)
    view *= rng.choice([1.0, 1.0, 0.37, 2.5])                      # the reference never normalises `view` (src/scene.cpp:118-128)
    helper = np.array([0.0, 1.0, 0.0]) if abs(view[1]) < 0.9 * np.linalg.norm(view) else np.array([1.0, 0.0, 0.0])
    up = helper - view * (helper @ view) / (view @ view)
    up /= np.linalg.norm(up)
    if tilt:                                                      # ... nor makes `up` perpendicular to it: rolled and tilted up vectors
        up = up + view / np.linalg.norm(view) * rng.uniform(-0.4, 0.4)
        right = np.cross(view, up)
        up = up * np.cos(r := rng.uniform(-0.5, 0.5)) + right / np.linalg.norm(right) * np.sin(r)
    cam["resolution"] = shape
    cam["position"], cam["view"], cam["up"] = eye, view, up
    cam["fov"] = (0.0, half_angle)
    oracle.lib().orc_camera_set_resolution(cam.ctypes.data, shape[0], shape[1])
    return cam


def _case(oracle, rng):
    """one random (camera, primitive set): the primitives are placed RELATIVE to the camera so that the frame sees the interesting
    configurations -- in front, at the frame's edges, straddling the eye plane, around the eye, behind it"""
    shape = SHAPES[int(rng.integers(len(SHAPES)))]
    half = float(np.exp(rng.uniform(np.log(1.0), np.log(85.0)))) if rng.random() < 0.6 else float(rng.uniform(1.0, 85.0))
    eye = rng.normal(size=3) * rng.choice([0.0, 3.0, 30.0])
    fwd = rng.normal(size=3)
    fwd /= np.linalg.norm(fwd)
    if rng.random() < 0.3:
        fwd = np.eye(3)[int(rng.integers(3))] * rng.choice([-1.0, 1.0])        # axis-parallel views (Cornell looks down -z)
    cam = _camera(oracle, rng, shape, eye, eye + fwd, half, tilt=rng.random() < 0.5)
    side = np.cross(fwd, [0.3, 0.9, 0.2])
    side /= np.linalg.norm(side)
    upv = np.cross(side, fwd)
    # half extents of the frustum at unit distance (vertical half angle `half`, horizontal by the aspect ratio)
    ty = np.tan(np.radians(half))
    tx = ty * shape[0] / shape[1]
    geoms = []
    for _ in range(int(rng.integers(1, 7))):
        where = rng.choice(["inside", "edge", "plane", "around", "behind", "far"], p=[0.3, 0.25, 0.15, 0.1, 0.1, 0.1])
        dist = float(np.exp(rng.uniform(np.log(0.3), np.log(60.0))))
        size = dist * float(np.exp(rng.uniform(np.log(0.02), np.log(1.5)))) * max(min(ty, 3.0), 0.05)
        if where == "inside":
            c = eye + fwd * dist + side * rng.uniform(-1, 1) * min(tx, 20) * dist + upv * rng.uniform(-1, 1) * min(ty, 20) * dist
        elif where == "edge":                                     # centred on a frame border: partly visible
            ex, ey = (rng.choice([-1.0, 1.0]), rng.uniform(-1, 1)) if rng.random() < 0.5 else (rng.uniform(-1, 1), rng.choice([-1.0, 1.0]))
            c = eye + fwd * dist + side * ex * min(tx, 50) * dist + upv * ey * min(ty, 50) * dist
        elif where == "plane":                                    # straddling the plane through the eye: corners on and behind it
            c = eye + fwd * rng.uniform(-0.5, 0.5) * size + (side * rng.uniform(-1, 1) + upv * rng.uniform(-1, 1)) * dist
            size *= rng.uniform(1.0, 4.0)
        elif where == "around":                                   # the eye inside the primitive's bounding cube
            size = max(size, 0.5)
            c = eye + rng.uniform(-0.4, 0.4, 3) * size
        elif where == "behind":
            c = eye - fwd * dist + side * rng.uniform(-1, 1) * dist
        else:
            c = eye + fwd * dist * 50 + side * rng.uniform(-1, 1) * tx * dist * 50
        rot = tuple(rng.choice([0, 90, 180, -90], 3)) if rng.random() < 0.35 else tuple(rng.uniform(-180, 180, 3))
        scl = size * np.where(rng.random(3) < 0.3, rng.choice([0.01, 0.1, 10.0]), 1.0) * rng.uniform(0.5, 1.5, 3)
        geoms.append(oracle.make_geom(int(rng.integers(0, 2)), 0, tuple(c), rot, tuple(scl)))
    return cam, np.concatenate(geoms)


def test_camera_culling_never_skips_a_hit(gpu, oracle):
    # >= 10^4 random (camera, primitive set) pairs, every pixel, the full tests of every primitive: 0 violations
    rng = np.random.default_rng(20261004)
    cases = int(os.environ.get("PT_CULL_SWEEP_CASES", "10500"))
    hits = culled = bad = pairs_seen = 0
    widest = 0.0
    per_width = {}
    for k in range(cases):
        cam, geoms = _case(oracle, rng)
        h, c, v = gpu.test_camera_cull_sweep(cam.view(gpu.CAMERA_DTYPE), geoms.view(gpu.GEOM_DTYPE), samples=2 if k % 4 == 0 else 1)
        hits, culled, bad = hits + h, culled + c, bad + v
        assert v == 0, "case %d: %d hit(s) from pixels the culling skips (camera %s)" % (k, v, cam)
        widest = max(widest, float(cam["fov"][0][1]))
        w = int(cam["resolution"][0][0])
        per_width[w] = per_width.get(w, 0) + (1 if h and c else 0)
    assert bad == 0
    # the sweep is not vacuous: it saw hundreds of millions of hits and of culled pairs, at every width, up to 85 degrees
    assert hits > 5e7 and culled > 5e7 and widest > 84.0
    assert all(per_width.get(w, 0) > 100 for w in (17, 1920, 4096, 8192)), per_width


def test_inflation_margin_of_the_culling_tables(gpu, oracle):
    # How much of inflated_object_box's inflation do the reference's hits need?  Per hit of the full tests, the smallest fraction of the
    # inflation at which the EXACT object-space half-line (double precision) meets the grown primitive: 0 for a geometric hit, (0, 1] for
    # a hit the fp32 test's rounding created, > 1 would be a hit outside the inflated box.  The reciprocal of the largest fraction is the
    # margin of the error model -- stated next to the function like certainMiss's 40-50 x.  The cases of the soundness sweep above, per
    # primitive type (the two types carry different factors: a cube's first-order terms x 8, a sphere's x 128).
    cases = int(os.environ.get("PT_CULL_MARGIN_CASES", "10500"))
    worst = {}
    for typ, name in ((1, "cubes"), (0, "spheres")):
        rng = np.random.default_rng(20261004)
        w_max, needed, at = 0.0, 0, -1
        for k in range(cases):
            cam, geoms = _case(oracle, rng)
            g = geoms[geoms["type"] == typ]
            if len(g) == 0:
                continue
            w, n = gpu.test_camera_cull_margin(cam.view(gpu.CAMERA_DTYPE), g.view(gpu.GEOM_DTYPE), samples=1)
            needed += n
            if w > w_max:
                w_max, at = w, k
        print("inflation margin, %s: worst fraction %.4f (case %d), %d hits needed some of the inflation" % (name, w_max, at, needed))
        worst[name] = (w_max, needed, at)
    # measured: spheres 0.105 (a sphere seen from 10^4 object units, where the inflation is dominated by the radicand's term
    # sqrt(512 eps) R: in terms of that term's factor the margin is the square, ~90 x); cubes 0.003 (a handful of hits in ~10^9)
    assert worst["spheres"][1] > 1000                             # (the sweep does meet hits that only the rounding explains)
    assert worst["spheres"][0] <= 0.125, "a sphere hit used %.3f of the inflation (case %d): margin below 8 x" % (worst["spheres"][0], worst["spheres"][2])
    assert worst["cubes"][0] <= 0.02, "a cube hit used %.4f of the inflation (case %d): margin below 50 x" % (worst["cubes"][0], worst["cubes"][2])


def _scene(gpu, oracle, cam, geoms, rng):
    mats = np.zeros(3, oracle.MATERIAL_DTYPE)
    mats["color"] = rng.uniform(0.3, 1.0, (3, 3))
    mats["emittance"][0] = 5.0
    mats["hasReflective"][2], mats["specColor"][2] = 1.0, (0.9, 0.9, 0.9)
    g = geoms.copy()
    g["materialid"] = rng.integers(0, 3, len(g))
    W, H = (int(v) for v in cam["resolution"][0])
    return types.SimpleNamespace(geoms=g.view(gpu.GEOM_DTYPE), materials=mats.view(gpu.MATERIAL_DTYPE), camera=cam.view(gpu.CAMERA_DTYPE),
                                 traceDepth=3, meshes={}, image=np.zeros((H, W, 3), np.float32)), W, H


def _render(gpu, sc, W, H, cull_off, monkeypatch):
    if cull_off:
        monkeypatch.setenv("PT_AMD_NO_CAMERA_CULL", "1")
    else:
        monkeypatch.delenv("PT_AMD_NO_CAMERA_CULL", raising=False)
    gpu.pathtraceFree()
    gpu.pathtraceInit(sc, traceDepth=3, max_batch=2, pipeline_depth=1)
    gpu.pathtrace_batch(None, 0, 1, 2)
    img = gpu.readback(W * H)
    cnt = gpu.counters()
    tallies = ([int(cnt.live[d]) for d in range(1, 5)], int(cnt.light_hits), int(cnt.misses))
    paths = gpu.debug_trace_paths(2, 1, W * H)               # survivors of the camera-ray bounce of iteration 2: every pixel's hit
    gpu.pathtraceFree()
    monkeypatch.delenv("PT_AMD_NO_CAMERA_CULL", raising=False)
    return img, tallies, paths


def _same(a, b):
    (ia, ta, pa), (ib, tb, pb) = a, b
    assert ta == tb
    assert np.array_equal(pa[3], pb[3])                        # the same pixels survive ...
    for x, y in zip(pa[:3], pb[:3]):                           # ... from the same hit points, in the same directions, with the same colours
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    assert np.array_equal(ia.view(np.uint32), ib.view(np.uint32))


def test_camera_bounce_is_identical_with_the_culling_switched_off(gpu, oracle, monkeypatch):
    rng = np.random.default_rng(77)
    seen_hits = 0
    for k in range(int(os.environ.get("PT_CULL_RENDER_CASES", "160"))):
        cam, geoms = _case(oracle, rng)
        sc, W, H = _scene(gpu, oracle, cam, geoms, rng)
        on = _render(gpu, sc, W, H, False, monkeypatch)
        off = _render(gpu, sc, W, H, True, monkeypatch)
        _same(on, off)
        seen_hits += on[1][1] + len(on[2][3])
    assert seen_hits > 100000


@pytest.mark.parametrize("shape,half", [((17, 29), 85.0), ((1920, 1080), 84.0), ((4096, 512), 80.0), ((8192, 256), 85.0), ((8192, 2048), 1.0),
                                        ((1280, 720), 45.0)])
def test_cornell_at_the_extremes_with_and_without_culling(gpu, oracle, monkeypatch, shape, half):
    # the reference's own scene at full widths: the shipped camera (45 degrees), the widest and the narrowest half-angle, and an eye
    # INSIDE the box close to a wall (corners of the walls behind the eye plane)
    from conftest import SCENES
    base = gpu.Scene(os.path.join(SCENES, "cornell.txt"))
    rng = np.random.default_rng(shape[0])
    for eye, target in (((0.0, 5.0, 10.5), (0.0, 5.0, 0.0)), ((3.9, 1.0, 3.5), (-2.0, 6.0, -4.0)), ((0.0, 9.6, 0.0), (0.3, 0.0, -0.2))):
        cam = _camera(oracle, rng, shape, eye, target, half, tilt=False)
        sc = types.SimpleNamespace(geoms=base.geoms, materials=base.materials, camera=cam.view(gpu.CAMERA_DTYPE), traceDepth=3, meshes={},
                                   image=np.zeros((shape[1], shape[0], 3), np.float32))
        on = _render(gpu, sc, shape[0], shape[1], False, monkeypatch)
        off = _render(gpu, sc, shape[0], shape[1], True, monkeypatch)
        _same(on, off)
        assert on[1][0][0] == shape[0] * shape[1] * 2 and on[1][0][1] > 0
