"""Randomised scenes on the HIP path against the oracle, bit for bit: every combination the kernels branch on -- walls or no
walls, few or many spheres (the per-lane lists), cubes small enough to be binned, meshes, nested and overlapping primitives,
exact and arbitrary rotations, flat and tiny scales, every material kind, the extras -- drawn from seeded generators, so a
failure names its seed."""
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


def _material(oracle, rng, kind):
    m = np.zeros(1, oracle.MATERIAL_DTYPE)
    m["color"] = rng.uniform(0.2, 1.0, 3)
    if kind == "light":
        m["emittance"] = rng.uniform(1, 8)
    elif kind == "mirror":
        m["hasReflective"], m["specColor"] = 1.0, rng.uniform(0.5, 1.0, 3)
        m["specExponent"] = rng.choice([0.0, 0.0, 5.0, 300.0])
    elif kind == "glass":
        m["hasRefractive"], m["indexOfRefraction"], m["specColor"] = 1.0, rng.choice([1.33, 1.5, 2.4]), rng.uniform(0.7, 1.0, 3)
    return m


def _random_scene(gpu, oracle, seed, tris_by_name):
    rng = np.random.default_rng(seed)
    kinds = ["light", "diffuse", "diffuse", "mirror", "glass", "diffuse"]
    mats = np.concatenate([_material(oracle, rng, k) for k in kinds])
    geoms, meshes = [], {}
    room = rng.random() < 0.6
    if room:                                                  # walls: flat cubes around the scene, one of them maybe missing or oblique
        S = rng.uniform(6, 14)
        walls = [((0, 0, 0), (S, 0.01 * rng.choice([1, 10]), S)), ((0, S, 0), (S, 0.02, S)), ((0, S / 2, -S / 2), (S, S, 0.01)),
                 ((-S / 2, S / 2, 0), (0.01, S, S)), ((S / 2, S / 2, 0), (0.01, S, S))]
        for k, (t, s) in enumerate(walls):
            if rng.random() < 0.15:
                continue
            r = (0, 0, 0) if rng.random() < 0.7 else tuple(rng.choice([0, 90, 180, 3.0, -17.0], 3))
            geoms.append(oracle.make_geom(1, int(rng.integers(1, 6)), t, r, s))
        geoms.append(oracle.make_geom(1, 0, (0, S - 0.2, 0), (0, 0, 0), (S / 3, 0.3, S / 3)))          # a ceiling light
        centre, spread = np.array([0, S / 2, 0]), S / 3
    else:
        geoms.append(oracle.make_geom(rng.choice([0, 1]), 0, (0, 9, 0), tuple(rng.uniform(-30, 30, 3)), (8, 1, 8)))   # a light in the open
        centre, spread = np.array([0, 4.0, 0]), 4.0
    n = int(rng.choice([2, 4, 7, 12, 20]))
    for _ in range(n):
        t = centre + rng.normal(size=3) * spread * 0.5
        r = tuple(rng.choice([0, 90, -90, 45, 0, 0], 3)) if rng.random() < 0.4 else tuple(rng.uniform(-180, 180, 3))
        sc_ = rng.uniform(0.3, 3.0) * np.where(rng.random(3) < 0.2, rng.choice([0.05, 4.0]), 1.0) * rng.uniform(0.5, 1.5, 3)
        mat = int(rng.integers(0, 6)) if rng.random() < 0.15 else int(rng.integers(1, 6))
        kind = rng.choice(["sphere", "sphere", "cube", "mesh"], p=[0.35, 0.15, 0.3, 0.2])
        if kind == "mesh":
            meshes[len(geoms)] = tris_by_name[rng.choice(sorted(tris_by_name))]
            geoms.append(oracle.make_geom(2, mat, tuple(t), r, tuple(sc_ * 1.5)))
        else:
            geoms.append(oracle.make_geom(0 if kind == "sphere" else 1, mat, tuple(t), r, tuple(sc_)))
    if rng.random() < 0.3 and len(geoms) > 3:                 # coincident duplicates: file order decides
        geoms.append(geoms[-1].copy())
        if len(geoms) - 2 in meshes:
            meshes[len(geoms) - 1] = meshes[len(geoms) - 2]
    res = (int(rng.choice([48, 64, 256])), int(rng.choice([36, 50])))
    cam = np.zeros(1, oracle.CAMERA_DTYPE)
    cam["resolution"] = res
    eye = centre + np.array([rng.uniform(-2, 2), rng.uniform(-1, 2), spread * 2.2 + 2])
    cam["position"], cam["view"], cam["up"] = eye, (rng.uniform(-0.15, 0.15), rng.uniform(-0.15, 0.05), -1), (0, 1, 0)
    cam["fov"] = (0.0, rng.uniform(25, 50))
    oracle.lib().orc_camera_set_resolution(cam.ctypes.data, res[0], res[1])
    # mesh attributes (round 4), from a generator of their own so that the scenes themselves stay the ones of rounds 2-3: vertex normals --
    # the face normal bent by up to ~40 degrees per vertex, some of them through the surface (the blend is turned to the face's side) -- for
    # two meshes in five, a material per face (any of the scene's six, or -1 = the object's) for two in five
    arng = np.random.default_rng(seed + 977)
    mesh_normals, mesh_materials = {}, {}
    for g in sorted(meshes):
        t = meshes[g].reshape(-1, 3, 3).astype(np.float64)
        if arng.random() < 0.4:
            fn = np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0])
            fn /= np.maximum(np.linalg.norm(fn, axis=1, keepdims=True), 1e-30)
            vn = fn[:, None, :] + arng.uniform(-0.8, 0.8, t.shape)
            mesh_normals[g] = (vn / np.maximum(np.linalg.norm(vn, axis=2, keepdims=True), 1e-30)).reshape(-1, 9).astype(np.float32)
        if arng.random() < 0.4:
            mesh_materials[g] = arng.integers(-1, 6, len(t)).astype(np.int32)
    sc = types.SimpleNamespace(geoms=np.concatenate(geoms).view(gpu.GEOM_DTYPE), materials=mats.view(gpu.MATERIAL_DTYPE),
                               camera=cam.view(gpu.CAMERA_DTYPE), traceDepth=int(rng.integers(2, 9)), meshes=meshes,
                               mesh_normals=mesh_normals, mesh_materials=mesh_materials,
                               image=np.zeros((res[1], res[0], 3), np.float32))
    extras = {}
    if rng.random() < 0.25:
        extras.update(lens_radius=float(rng.uniform(0.05, 0.4)), focal_distance=float(spread * 2.2 + 2))
    if rng.random() < 0.25:
        extras.update(direct_lighting=True)
    return sc, res, extras, rng


def _sphere_field(gpu, oracle, seed):
    """Sphere-heavy scenes without meshes (k_bounce's CLUSTER variants: the survivors binned by which of two clusters of spheres their ray
    can hit): 5 .. 40 spheres of similar size in a box-shaped region, in a room or in the open; the emitter a ceiling light (binned: the
    last bounce then visits its candidates alone), an emissive sphere among the others (not binned), or both."""
    rng = np.random.default_rng(seed)
    kinds = ["light", "diffuse", "diffuse", "mirror", "glass", "diffuse"]
    mats = np.concatenate([_material(oracle, rng, k) for k in kinds])
    geoms = []
    S = rng.uniform(8, 14)
    room = rng.random() < 0.7
    if room:
        walls = [((0, 0, 0), (S, 0.01, S)), ((0, S, 0), (S, 0.01, S)), ((0, S / 2, -S / 2), (S, S, 0.01)),
                 ((-S / 2, S / 2, 0), (0.01, S, S)), ((S / 2, S / 2, 0), (0.01, S, S))]
        for t, sc_ in walls:
            if rng.random() < 0.1:
                continue
            geoms.append(oracle.make_geom(1, int(rng.integers(1, 6)), t, (0, 0, 0), sc_))
    emitter = rng.choice(["light", "sphere", "both"])
    if emitter != "sphere":
        geoms.append(oracle.make_geom(1, 0, (0, S - 0.2, 0), (0, 0, 0), (S / 3, 0.3, S / 3)))
    n = int(rng.choice([5, 6, 9, 16, 27, 40]))
    ext = rng.uniform(0.3, 0.45, 3) * S * np.where(rng.random(3) < 0.3, 0.4, 1.0)          # (sometimes a slab or a column of spheres)
    base = rng.uniform(0.15, 0.5) * S / n ** (1 / 3)
    for k in range(n):
        t = np.array([0, S / 2, 0]) + rng.uniform(-1, 1, 3) * ext
        d = base * rng.uniform(0.6, 1.4)
        scale = (d, d, d) if rng.random() < 0.8 else tuple(d * rng.uniform(0.7, 1.3, 3))
        mat = 0 if (emitter != "light" and k == 0) else int(rng.integers(1, 6))
        geoms.append(oracle.make_geom(0, mat, tuple(t), tuple(rng.uniform(-180, 180, 3)), scale))
    if rng.random() < 0.3:                                    # coincident duplicates: file order decides
        geoms.append(geoms[-1].copy())
    order = rng.permutation(len(geoms)) if rng.random() < 0.5 else np.arange(len(geoms))        # spheres between the walls in file order too
    geoms = [geoms[i] for i in order]
    res = (int(rng.choice([64, 256])), int(rng.choice([36, 50])))
    cam = np.zeros(1, oracle.CAMERA_DTYPE)
    cam["resolution"] = res
    cam["position"], cam["view"], cam["up"] = (rng.uniform(-1, 1), S / 2 + rng.uniform(-1, 1), S * 1.05), (rng.uniform(-0.1, 0.1), rng.uniform(-0.1, 0.05), -1), (0, 1, 0)
    cam["fov"] = (0.0, rng.uniform(30, 45))
    oracle.lib().orc_camera_set_resolution(cam.ctypes.data, res[0], res[1])
    sc = types.SimpleNamespace(geoms=np.concatenate(geoms).view(gpu.GEOM_DTYPE), materials=mats.view(gpu.MATERIAL_DTYPE),
                               camera=cam.view(gpu.CAMERA_DTYPE), traceDepth=int(rng.integers(2, 9)), meshes={}, mesh_normals={}, mesh_materials={},
                               image=np.zeros((res[1], res[0], 3), np.float32))
    extras = {}
    if rng.random() < 0.2:
        extras.update(direct_lighting=True)
    return sc, res, extras, rng


def _check(gpu, oracle, sc, W, H, extras, rng, seed):
    depth = sc.traceDepth
    ref = oracle.Renderer(sc.camera.view(oracle.CAMERA_DTYPE), sc.geoms.view(oracle.GEOM_DTYPE), sc.materials.view(oracle.MATERIAL_DTYPE),
                          depth, meshes=sc.meshes, mesh_normals=sc.mesh_normals, mesh_materials=sc.mesh_materials)
    ref.set_extras(**extras)
    world = int(rng.choice([1, 1, 2, 3]))
    rank = int(rng.integers(0, world))
    batch = int(rng.choice([1, 3, 8]))
    iters = list(range(1, 1 + int(rng.choice([2, 3, 8]))))
    want = np.zeros(W * H * 3, np.float32)
    live = np.zeros(64, np.int64)
    pipeline = int(rng.choice([1, 2, 3]))
    # a third of the scenes through the reference's protocol -- one call per iteration -- served from batches traced ahead
    ahead = bool(rng.random() < 0.33)
    gpu.pathtraceFree()
    gpu.pathtraceInit(sc, traceDepth=depth, max_batch=batch, pipeline_depth=pipeline, shard_rank=rank, shard_count=world, trace_ahead=ahead, **extras)
    it = iters[0]
    while it <= iters[-1]:
        n = 1 if ahead else min(batch, iters[-1] - it + 1)
        gpu.pathtrace_batch(None, 0, it, n)
        it += n
    for k in iters:
        live += np.array(ref.iterate(k, want, rank, world).live[:64])
    got = gpu.readback(W * H)
    cnt = gpu.counters()
    nd = depth + (1 if extras.get("direct_lighting") else 0)
    if not (ahead and batch > 1):                             # (the tallies also cover iterations traced ahead)
        assert [int(cnt.live[d]) for d in range(1, nd + 2)] == live[1:nd + 2].tolist(), seed
    assert cnt.iterations == len(iters)
    b = int(rng.integers(1, depth + 1))
    o, d, c, pix = gpu.debug_trace_paths(iters[0], b, W * H)
    wo, wd, wc, wpix = ref.dump_paths(iters[0], b, rank, world)
    gpu.pathtraceFree()
    assert np.array_equal(pix, wpix), seed
    assert np.array_equal(o.view(np.uint32), wo.view(np.uint32)) and np.array_equal(d.view(np.uint32), wd.view(np.uint32)), seed
    assert np.array_equal(c.view(np.uint32), wc.view(np.uint32)), seed
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), seed


# PT_FUZZ_SEEDS=<n>: a longer one-off run (profiles/r02_fuzz.txt holds the last one)
@pytest.mark.parametrize("seed", range(int(os.environ.get("PT_FUZZ_SEEDS", "40"))))
def test_random_scene(gpu, oracle, seed):
    small = gpu.Scene(os.path.join(SCENES, "mesh_small.txt"))
    tris = {"icosphere": small.meshes[3], "torus": small.meshes[4]}
    sc, (W, H), extras, rng = _random_scene(gpu, oracle, 5650 + seed, tris)
    _check(gpu, oracle, sc, W, H, extras, rng, seed)


@pytest.mark.parametrize("seed", range(max(16, int(os.environ.get("PT_FUZZ_SEEDS", "40")) // 4)))
def test_random_sphere_field(gpu, oracle, seed):
    sc, (W, H), extras, rng = _sphere_field(gpu, oracle, 8800 + seed)
    _check(gpu, oracle, sc, W, H, extras, rng, seed)
