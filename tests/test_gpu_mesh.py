"""Triangle meshes (scene-format object type "mesh", reference README.md:112-116, 236) on the HIP path against the oracle,
bit for bit.  The reference holds no mesh code: the semantics are the oracle's brute-force loop over every triangle
(oracle/pt_oracle.cpp, mesh_intersection_test; its triangle test is pinned against the reference's vendored
glm::intersectRayTriangle in tests/test_golden.py).  The kernels walk a bounding-volume hierarchy instead; these tests hold
them to the brute-force result per ray, per path and per pixel."""
import os

import numpy as np
import pytest

from conftest import SCENES

pytestmark = pytest.mark.gpu
_SW = int(os.environ.get("PT_SWEEP_SCALE", "1"))      # (see tests/test_gpu_parity.py)


@pytest.fixture(scope="module")
def gpu(pt):
    if pt.device_count() < 1:
        pytest.fail("no HIP device: GPU tests must run on the MI355X box")
    return pt


def mesh_rays(rng, geom, tris, n):
    """Ray mix for one mesh geom: aimed at points of its triangles (hits, edges, vertices), origins inside its box, far
    origins, grazes of its bounding ball, random misses; mostly unit directions, some not."""
    xf = geom["transform"][0].reshape(4, 4).T.astype(np.float64)          # column-major mat4
    to_world = lambda p: (xf[:3, :3] @ p.T).T + xf[:3, 3]
    tr = tris.reshape(-1, 3, 3).astype(np.float64)
    lo, hi = tr.reshape(-1, 3).min(0), tr.reshape(-1, 3).max(0)
    centre_w = to_world(((lo + hi) / 2)[None])[0]
    radius_w = np.linalg.norm(to_world(np.array([hi])) - centre_w)
    rays = np.zeros((n, 6), np.float32)
    for i in range(n):
        kind = i % 8
        t = tr[rng.integers(len(tr))]
        if kind in (0, 1, 2):          # aimed at a point of a triangle (kind 1: on an edge, kind 2: at a vertex)
            a, b = rng.uniform(0, 1, 2)
            if a + b > 1:
                a, b = 1 - a, 1 - b
            if kind == 1:
                b = 0.0
            if kind == 2:
                a = b = 0.0
            tgt = to_world((t[0] + a * (t[1] - t[0]) + b * (t[2] - t[0]))[None])[0]
            o = centre_w + rng.normal(size=3) * radius_w * rng.choice([1.5, 4, 40])
            d = tgt - o
        elif kind == 3:                # origin inside the mesh's box
            o = to_world((lo + rng.uniform(0.2, 0.8, 3) * (hi - lo))[None])[0]
            d = rng.normal(size=3)
        elif kind == 4:                # grazing the bounding ball
            o = centre_w + rng.normal(size=3) * radius_w * 6
            u = rng.normal(size=3)
            u -= u @ (centre_w - o) * (centre_w - o) / ((centre_w - o) @ (centre_w - o))
            tgt = centre_w + u / np.linalg.norm(u) * radius_w * rng.uniform(0.9, 1.1)
            d = tgt - o
        elif kind == 5:                # axis-parallel direction with exact zeros
            o = centre_w + rng.uniform(-1, 1, 3) * radius_w * 0.5
            ax = rng.integers(3)
            o[ax] += radius_w * 3 * rng.choice([-1, 1])
            d = np.zeros(3)
            d[ax] = -np.sign(o[ax] - centre_w[ax])
        elif kind == 6:                # from a point just off the surface (a scattered ray's origin)
            c = to_world(t.mean(0)[None])[0]
            o = c + rng.normal(size=3) * 1e-3
            d = rng.normal(size=3)
        else:
            o = rng.uniform(-12, 12, 3)
            d = rng.normal(size=3)
        nd = np.linalg.norm(d)
        if nd > 0 and rng.random() < 0.85:
            d = d / nd
        rays[i, :3], rays[i, 3:] = o, d
    return rays


@pytest.mark.parametrize("flat", [False, True])
def test_mesh_test_against_the_oracle_per_ray(gpu, oracle, flat):
    sc = gpu.Scene(os.path.join(SCENES, "mesh_small.txt"))
    rng = np.random.default_rng(1116)
    hits = culled_total = 0
    cases = [(sc.geoms[g:g + 1], sc.meshes[g]) for g in sorted(sc.meshes)]
    # the same triangles under an anisotropic, rotated transform and far off their own origin
    g2 = oracle.make_geom(2, 0, (3, -2, 1), (25, 70, -40), (0.7, 2.5, 1.2))
    cases.append((g2, sc.meshes[3]))
    cases.append((oracle.make_geom(2, 0, (0, 0, 0), (0, 0, 0), (1, 1, 1)), sc.meshes[4] + np.float32(3.0)))
    for geom, tris in cases:
        rays = mesh_rays(rng, geom, tris, 1536)
        t, p, n, o, culled = gpu.test_mesh_intersect(geom, tris, rays, flat=flat)
        assert not np.isnan(t[culled != 0]).any(), "the bounding-ball test rejected a ray the full test hits"
        for i in range(len(rays)):
            wt, wp, wn, wo, _ = oracle.mesh_intersect(geom.view(oracle.GEOM_DTYPE), tris, rays[i])
            assert np.float32(t[i]).view(np.uint32) == np.float32(wt).view(np.uint32), (i, rays[i], t[i], wt)
            assert np.array_equal(p[i].view(np.uint32), wp.view(np.uint32)) and o[i] == wo, (i, p[i], wp)
            if wt != -1.0:
                assert np.array_equal(n[i].view(np.uint32), wn.view(np.uint32)), (i, n[i], wn)
                assert culled[i] == 0
        hits += int((t > 0).sum())
        culled_total += int(culled.sum())
    assert hits > 1500 and culled_total > 300


def test_mesh_half_precision_planes_at_the_extremes(gpu, oracle):
    # The inner nodes keep their children's boxes as half-precision planes rounded outwards (pt_mesh.h: halfBitsDirected).  Object
    # coordinates beyond the half range (planes at the largest finite half or at infinity), in the subnormal halves, straddling
    # zero, and the smallest hierarchies (one triangle: a root with one child; two; three) -- per ray against the oracle's loop
    # over every triangle, bit for bit, with the hierarchy and with the plain list.
    sc = gpu.Scene(os.path.join(SCENES, "mesh_small.txt"))
    rng = np.random.default_rng(20261004)
    base = sc.meshes[4]                                                   # the 128-triangle torus
    cases = []
    for k, (obj_scale, label) in enumerate(((1.0e5, "beyond the half range"), (3.0e-7, "subnormal halves"), (70000.0, "around 65504"))):
        tris = (base * np.float32(obj_scale)).astype(np.float32)
        inv = 1.0 / obj_scale
        cases.append((oracle.make_geom(2, 0, (1.0, 2.0, -0.5), (20 * k, 35, -10), (2 * inv, 3 * inv, 2.5 * inv)), tris))
    cases.append((oracle.make_geom(2, 0, (0, 0, 0), (0, 0, 0), (1, 1, 1)), base - np.float32(0.25)))      # boxes straddling zero
    for nt in (1, 2, 3, 5):
        cases.append((oracle.make_geom(2, 0, (0.5, 0, 0), (10, 20, 30), (2, 2, 2)), base[:nt].copy()))
    hits = 0
    for geom, tris in cases:
        rays = mesh_rays(rng, geom, tris, 768)
        for flat in (False, True):
            t, p, n, o, culled = gpu.test_mesh_intersect(geom, tris, rays, flat=flat)
            assert not np.isnan(t[culled != 0]).any()
            for i in range(len(rays)):
                wt, wp, wn, wo, _ = oracle.mesh_intersect(geom.view(oracle.GEOM_DTYPE), tris, rays[i])
                assert np.float32(t[i]).view(np.uint32) == np.float32(wt).view(np.uint32), (len(tris), flat, i, rays[i], t[i], wt)
                assert np.array_equal(p[i].view(np.uint32), wp.view(np.uint32)) and o[i] == wo
                if wt != -1.0:
                    assert np.array_equal(n[i].view(np.uint32), wn.view(np.uint32))
            hits += int((t > 0).sum())
    assert hits > 1500


def test_bounding_ball_never_rejects_a_mesh_hit(gpu, oracle):
    """2^24 rays per case, dense in grazes of the bounding ball, origins 1/64 .. 64 radii away: the world-space test that
    lets tiles skip a mesh (and classes the queue) never rejects a ray the full walk hits."""
    sc = gpu.Scene(os.path.join(SCENES, "cornell_mesh.txt"))
    cases = [(sc.geoms[g:g + 1], sc.meshes[g]) for g in sorted(sc.meshes)]
    cases.append((oracle.make_geom(2, 0, (3, -2, 1), (25, 70, -40), (0.7, 2.5, 1.2)), sc.meshes[7]))     # anisotropic, rotated
    cases.append((oracle.make_geom(2, 0, (0, 0, 0), (0, 0, 0), (1, 1, 1)), sc.meshes[6] + np.float32(20.0)))   # far off its origin
    cases.append((oracle.make_geom(2, 0, (-4, 1, 2), (0, 45, 0), (40, 1.0, 40)), sc.meshes[6]))          # a 40:1 pancake
    for k, (geom, tris) in enumerate(cases):
        culled, violations, hits = gpu.test_mesh_cull_sweep(geom, tris, 565 + k + 100 * (_SW - 1), (1 << 24) * _SW)
        assert violations == 0, (k, violations)
        assert hits > (1 << 24) // 50, (k, hits)
        assert culled > (1 << 24) // (20 if k != 4 else 2000), (k, culled)      # (the pancake's margin term leaves little to cull)
    # a transform too anisotropic for the test's margins switches the culling off altogether (never an unsound answer)
    with pytest.raises(gpu.PtError):
        gpu.test_mesh_cull_sweep(oracle.make_geom(2, 0, (0, 0, 0), (0, 0, 0), (40, 0.02, 40)), sc.meshes[6], 1, 1 << 10)


def _render_both(gpu, oracle, sc, depth, iters, res, dump_bounces=(), rank=0, count=1, **extras):
    W, H = res
    ref = oracle.Renderer(sc.camera.view(oracle.CAMERA_DTYPE), sc.geoms.view(oracle.GEOM_DTYPE),
                          sc.materials.view(oracle.MATERIAL_DTYPE), depth, meshes=sc.meshes, mesh_normals=getattr(sc, "mesh_normals", None),
                          mesh_materials=getattr(sc, "mesh_materials", None))
    ref.set_extras(**extras)
    want = np.zeros(W * H * 3, np.float32)
    live = np.zeros(64, np.int64)
    gpu.pathtraceFree()
    gpu.pathtraceInit(sc, traceDepth=depth, max_batch=4, pipeline_depth=2, shard_rank=rank, shard_count=count, **extras)
    gpu.pathtrace_batch(None, 0, iters[0], len(iters))
    for k in iters:
        live += np.array(ref.iterate(k, want, rank, count).live[:64])
    got = gpu.readback(W * H)
    cnt = gpu.counters()
    assert [int(cnt.live[d]) for d in range(1, depth + 3)] == live[1:depth + 3].tolist()
    for b in dump_bounces:
        o, d, c, pix = gpu.debug_trace_paths(iters[0], b, W * H)
        wo, wd, wc, wpix = ref.dump_paths(iters[0], b, rank, count)
        assert np.array_equal(pix, wpix)
        assert np.array_equal(o.view(np.uint32), wo.view(np.uint32)) and np.array_equal(d.view(np.uint32), wd.view(np.uint32))
        assert np.array_equal(c.view(np.uint32), wc.view(np.uint32))
    gpu.pathtraceFree()
    assert want.max() > 0 and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    return got


def test_mesh_scene_render_against_the_oracle(gpu, oracle):
    sc = gpu.Scene(os.path.join(SCENES, "mesh_small.txt"))
    assert sorted(sc.meshes) == [3, 4] and list(sc.geoms["type"]) == [1, 1, 1, 2, 2]
    _render_both(gpu, oracle, sc, 6, [1, 2, 3], (96, 96), dump_bounces=(1, 2, 4))


def test_mesh_attributes_render_against_the_oracle(gpu, oracle):
    # vertex normals (`vn`: the barycentric blend at the hit, PtMesh::normals) and a material per face (`usemtl <k>`, PtMesh::materials --
    # one face of the cube mesh is emissive, one a mirror mix, one glass): per-bounce path state and frame bit-identical to the oracle's
    sc = gpu.Scene(os.path.join(SCENES, "mesh_attributes.txt"))
    assert sorted(sc.meshes) == [3, 4, 5] and sorted(sc.mesh_normals) == [3] and sorted(sc.mesh_materials) == [5]
    got = _render_both(gpu, oracle, sc, 6, [1, 2, 3, 4], (96, 96), dump_bounces=(1, 2, 3, 5))
    # ... and the attributes do reach the picture: without them the same scene renders differently
    import types
    plain = types.SimpleNamespace(geoms=sc.geoms, materials=sc.materials, camera=sc.camera, traceDepth=6, meshes=sc.meshes, image=sc.image)
    other = _render_both(gpu, oracle, plain, 6, [1, 2, 3, 4], (96, 96))
    assert not np.array_equal(got, other)
    # row shards and the thin lens + direct lighting on top
    _render_both(gpu, oracle, sc, 5, [7, 8], (96, 96), rank=1, count=3)
    _render_both(gpu, oracle, sc, 5, [3, 4], (96, 96), dump_bounces=(2,), lens_radius=0.3, focal_distance=9.0, direct_lighting=True)


def test_direct_lighting_samples_emissive_meshes(gpu, oracle):
    # round 5 (README.md:107-108 x :112-116; VERDICT round 4 "missing" #5): the direct-lighting bounce also aims at emissive MESHES --
    # a uniformly chosen point of the object-space bounds of the mesh's vertices, as it does at the unit cube of a sphere or cube.  A
    # scene whose ONLY emitter is a mesh (the icosphere takes the light's material, the ceiling light a diffuse one): path state after
    # the aiming bounce and the frame are the oracle's bit for bit (and not the frame of the same scene without the flag).
    import types
    sc = gpu.Scene(os.path.join(SCENES, "mesh_small.txt"))
    geoms = sc.geoms.copy()
    geoms["materialid"][3] = 0                      # the icosphere emits
    geoms["materialid"][0] = 1                      # the ceiling light does not
    s2 = types.SimpleNamespace(geoms=geoms, materials=sc.materials, camera=sc.camera, traceDepth=3, meshes=sc.meshes, image=sc.image)
    lit = _render_both(gpu, oracle, s2, 3, [1, 2, 3, 4], (96, 96), dump_bounces=(3,), direct_lighting=True)
    plain = _render_both(gpu, oracle, s2, 3, [1, 2, 3, 4], (96, 96))
    assert lit.max() > 0 and not np.array_equal(lit, plain)          # (the aiming bounce changes the picture: it reaches the mesh)
    # ... next to an emissive face of a mesh with per-face materials and an emissive cube (file order decides who is picked)
    sa = gpu.Scene(os.path.join(SCENES, "mesh_attributes.txt"))
    _render_both(gpu, oracle, sa, 4, [1, 2], (96, 96), dump_bounces=(4,), direct_lighting=True)


def test_face_material_beyond_the_scenes_materials_is_rejected(gpu):
    sc = gpu.Scene(os.path.join(SCENES, "mesh_attributes.txt"))
    bad = dict(sc.mesh_materials)
    bad[5] = bad[5].copy()
    bad[5][3] = len(sc.materials)
    import types
    s2 = types.SimpleNamespace(geoms=sc.geoms, materials=sc.materials, camera=sc.camera, traceDepth=4, meshes=sc.meshes,
                               mesh_normals=sc.mesh_normals, mesh_materials=bad, image=sc.image)
    gpu.pathtraceFree()
    with pytest.raises(gpu.PtError):
        gpu.pathtraceInit(s2, traceDepth=4)
    gpu.pathtraceFree()


def test_mesh_scene_render_with_the_median_rebuild(gpu, oracle, monkeypatch):
    # pt_init rebuilds a hierarchy that would need more stack levels than its threshold by median splits alone (pt_mesh.h); the
    # tests' switch lowers the threshold so that the small scene's meshes take that path: the same frame, bit for bit
    sc = gpu.Scene(os.path.join(SCENES, "mesh_small.txt"))
    monkeypatch.setenv("PT_AMD_MESH_STACK_MAX", "3")
    _render_both(gpu, oracle, sc, 6, [1, 2, 3], (96, 96), dump_bounces=(1, 3))
    monkeypatch.delenv("PT_AMD_MESH_STACK_MAX")


def test_mesh_scene_row_shards(gpu, oracle):
    sc = gpu.Scene(os.path.join(SCENES, "mesh_small.txt"))
    total = np.zeros(96 * 96 * 3, np.float32)
    for rank in range(3):
        total += _render_both(gpu, oracle, sc, 5, [7, 8], (96, 96), rank=rank, count=3)
    whole = _render_both(gpu, oracle, sc, 5, [7, 8], (96, 96))
    assert np.array_equal(total.view(np.uint32), whole.view(np.uint32))


def test_mesh_scene_with_lens_and_direct_lighting(gpu, oracle):
    sc = gpu.Scene(os.path.join(SCENES, "mesh_small.txt"))
    sc.set_resolution(80, 64)
    _render_both(gpu, oracle, sc, 4, [1, 2], (80, 64), dump_bounces=(1,), lens_radius=0.3, focal_distance=9.0)
    _render_both(gpu, oracle, sc, 4, [1, 2], (80, 64), direct_lighting=True)


def test_cornell_mesh_hierarchy_equals_triangle_list_at_full_size(gpu):
    """cornell_mesh.txt (1280 + 2304 triangles) at 1280x720: the hierarchy's frame equals the plain triangle list's (the
    brute-force rule on the device) bit for bit, and the path counts are conserved."""
    sc = gpu.Scene(os.path.join(SCENES, "cornell_mesh.txt"))
    sc.set_resolution(1280, 720)
    frames, counts = [], []
    for flat in ("0", "1"):
        os.environ["PT_AMD_MESH_FLAT"] = flat
        try:
            gpu.pathtraceFree()
            gpu.pathtraceInit(sc, traceDepth=8, max_batch=2, pipeline_depth=2)
            gpu.pathtrace_batch(None, 0, 1, 2)
            frames.append(gpu.readback(1280 * 720))
            c = gpu.counters()
            counts.append([int(c.live[d]) for d in range(1, 10)] + [int(c.light_hits), int(c.misses)])
        finally:
            del os.environ["PT_AMD_MESH_FLAT"]
            gpu.pathtraceFree()
    assert frames[0].max() > 0 and np.array_equal(frames[0].view(np.uint32), frames[1].view(np.uint32))
    assert counts[0] == counts[1] and counts[0][0] == 2 * 1280 * 720
    live = counts[0][:9]
    assert all(a >= b for a, b in zip(live, live[1:]))


def test_mesh_registration_errors(gpu):
    sc = gpu.Scene(os.path.join(SCENES, "mesh_small.txt"))
    gpu.pathtraceFree()
    keep = dict(sc.meshes)
    try:
        sc.meshes = {3: keep[3]}                       # geom 4 is a mesh without triangles
        with pytest.raises(gpu.PtError):
            gpu.pathtraceInit(sc)
        sc.meshes = dict(keep)
        sc.meshes[1] = keep[3]                         # geom 1 is a cube
        with pytest.raises(gpu.PtError):
            gpu.pathtraceInit(sc)
        bad = keep[3].copy()
        bad[0, 0] = np.nan
        with pytest.raises(gpu.PtError):
            gpu.set_meshes({3: bad})
    finally:
        sc.meshes = keep
        gpu.set_meshes({})
        gpu.pathtraceFree()
    plain = gpu.Scene(os.path.join(SCENES, "cornell.txt"))   # a scene without meshes after one with: nothing is left over
    plain.set_resolution(64, 64)
    gpu.pathtraceInit(plain, traceDepth=2)
    gpu.pathtrace(None, 0, 1, readback=False)
    gpu.sync()
    gpu.pathtraceFree()


def test_headless_driver_renders_a_mesh_scene(gpu, oracle, tmp_path):
    # pt_render = the reference's main()/runCuda()/saveImage() over the shim (which registers the scene's meshes before
    # pt_init), and its --batch path over the C ABI: both write the oracle's picture
    import subprocess
    from test_host import _decode_png
    from conftest import ROOT
    exe = os.path.join(ROOT, "project3-cuda-path-tracer_amd", "host", "pt_render")
    for scene_name in ("mesh_small.txt", "mesh_attributes.txt"):            # (the second: vertex normals and face materials from the OBJ files)
        sc = oracle.Scene(os.path.join(SCENES, scene_name))
        ref = oracle.Renderer(sc.camera, sc.geoms, sc.materials, 4, meshes=sc.meshes, mesh_normals=sc.mesh_normals, mesh_materials=sc.mesh_materials)
        img = np.zeros(96 * 96 * 3, np.float32)
        for it in range(1, 4):
            ref.iterate(it, img)
        want = (np.clip(img.reshape(96, 96, 3) / np.float32(3), 0, 1) * np.float32(255)).astype(np.uint8)[:, ::-1]
        for extra in ([], ["--batch", "3"]):
            base = str(tmp_path / (scene_name[:-4] + str(len(extra))))
            r = subprocess.run([exe, os.path.join(SCENES, scene_name), "--iterations", "3", "--depth", "4", "--out", base] + extra,
                               capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
            assert np.array_equal(_decode_png(base + ".png"), want)


def _icosphere(subdiv, radius=0.5):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_scenes", os.path.join(SCENES, "make_scenes.py"))
    ms = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ms)
    v, f = ms.icosphere(subdiv, radius)
    v = np.array(v, np.float32)
    return np.array([np.concatenate([v[a], v[b], v[c]]) for a, b, c in f], np.float32)


def test_scene_of_meshes_only_with_an_emissive_mesh(gpu, oracle):
    """No cube, no sphere: the light is a mesh too (octant queue classes, no wall certificates, emitter inside the walk)."""
    sc = gpu.Scene(os.path.join(SCENES, "mesh_small.txt"))
    tris = _icosphere(1)
    geoms = np.concatenate([oracle.make_geom(2, 0, (0, 6, 0), (0, 0, 0), (4, 1, 4)),            # emissive, flattened icosphere
                            oracle.make_geom(2, 1, (0, 1.5, 0), (20, 40, 0), (3, 3, 3)),         # diffuse
                            oracle.make_geom(2, 5, (2.5, 2.5, 1.5), (0, 0, 30), (2, 3, 2)),      # glass
                            oracle.make_geom(2, 4, (-2.5, 2.0, 1.0), (0, 0, 0), (2, 2, 2))])     # mirror mix
    sc.geoms = geoms.view(sc.geoms.dtype)
    sc.meshes = {0: tris, 1: tris, 2: sc.meshes[4], 3: tris}
    sc.set_resolution = None
    got = _render_both(gpu, oracle, sc, 5, [1, 2, 3], (96, 96), dump_bounces=(1, 3))
    assert (got > 0).mean() > 0.005      # (open space: most paths escape; the emitter itself and a few lucky bounces are lit)


def test_twenty_thousand_triangles(gpu, oracle):
    """A 20480-triangle icosphere: per-ray parity with the oracle's loop over every triangle, and a frame through the hierarchy
    equal to the frame through the plain triangle list."""
    tris = _icosphere(5)
    assert tris.shape == (20480, 9)
    geom = oracle.make_geom(2, 4, (-1, 4, -1), (10, 20, 30), (3, 2.5, 3))
    rng = np.random.default_rng(2048)
    rays = mesh_rays(rng, geom, tris, 384)
    t, p, n, o, culled = gpu.test_mesh_intersect(geom, tris, rays)
    hits = 0
    for i in range(len(rays)):
        wt, wp, wn, wo, _ = oracle.mesh_intersect(geom, tris, rays[i])
        assert np.float32(t[i]).view(np.uint32) == np.float32(wt).view(np.uint32), (i, t[i], wt)
        assert np.array_equal(p[i].view(np.uint32), wp.view(np.uint32)) and o[i] == wo
        hits += wt > 0
    assert hits > 150
    sc = gpu.Scene(os.path.join(SCENES, "cornell_mesh.txt"))
    sc.set_resolution(320, 240)
    sc.meshes = {6: tris, 7: sc.meshes[7]}
    frames = []
    for flat in ("0", "1"):
        os.environ["PT_AMD_MESH_FLAT"] = flat
        try:
            gpu.pathtraceFree()
            gpu.pathtraceInit(sc, traceDepth=6, max_batch=2, pipeline_depth=2)
            gpu.pathtrace_batch(None, 0, 1, 2)
            frames.append(gpu.readback(320 * 240))
        finally:
            del os.environ["PT_AMD_MESH_FLAT"]
            gpu.pathtraceFree()
    assert frames[0].max() > 0 and np.array_equal(frames[0].view(np.uint32), frames[1].view(np.uint32))


def test_sphere_heavy_scene_with_a_mesh(gpu, oracle):
    """64 spheres AND a mesh: the sphere-heavy variants of the kernel (packed sphere sweep, per-lane lists) with the mesh walk."""
    sc = gpu.Scene(os.path.join(SCENES, "spheres64.txt"))
    sc.set_resolution(96, 72)
    tris = _icosphere(1)
    extra = oracle.make_geom(2, 4, (0.5, 5.0, 3.0), (15, 30, 45), (2.5, 2.0, 2.5))
    small = gpu.Scene(os.path.join(SCENES, "mesh_small.txt")).meshes[4]
    extra2 = oracle.make_geom(2, 5, (-2.0, 2.5, 3.5), (60, 0, 20), (3, 3, 3))
    sc.geoms = np.concatenate([sc.geoms.view(oracle.GEOM_DTYPE), extra, extra2]).view(sc.geoms.dtype)
    n = len(sc.geoms)
    sc.meshes = {n - 2: tris, n - 1: small}
    sc.set_resolution = None
    _render_both(gpu, oracle, sc, 6, [1, 2, 3], (96, 72), dump_bounces=(1, 3))
    _render_both(gpu, oracle, sc, 4, [5, 6], (96, 72), rank=1, count=2, lens_radius=0.2, focal_distance=8.0)


@pytest.mark.gpu
def test_walks_ahead_of_the_bounce_on_small_grids_and_ragged_rows(gpu, oracle, monkeypatch):
    # round 5: the mesh walks run in a kernel of their own ahead of every bounce launch (pt_mesh_walk.h): persistent waves that draw
    # quarter tiles from sharded tickets.  Rows of 300 pixels (two tiles per row, the second one ragged: a quarter of 44 pixels and two
    # empty ones), a camera-ray-only render (trace depth 1: the bounce takes its tiles in rotated order, the walk in ticket order), and
    # grids smaller than the ticket shards (one shard per wave) -- the oracle's frame and per-bounce path state, bit for bit
    sc = gpu.Scene(os.path.join(SCENES, "mesh_small.txt"))
    sc.set_resolution(300, 70)
    _render_both(gpu, oracle, sc, 1, [1, 2, 3], (300, 70))
    _render_both(gpu, oracle, sc, 4, [1, 2, 3], (300, 70), dump_bounces=(1, 3))
    for grid in ("8", "24"):
        monkeypatch.setenv("PT_AMD_MAX_GRID", grid)
        _render_both(gpu, oracle, sc, 4, [5, 6, 7], (300, 70), dump_bounces=(2,))
        _render_both(gpu, oracle, sc, 3, [1, 2], (300, 70), lens_radius=0.3, focal_distance=9.0)
    monkeypatch.delenv("PT_AMD_MAX_GRID")


def test_scene_of_forty_meshes_reads_the_walks_rows_from_global_memory(gpu, oracle, monkeypatch):
    """the walk kernel staged a 128-byte row per mesh in every workgroup's LDS -- a hundred meshes cost it its residency, a
    thousand failed pt_init.  Beyond 32 meshes the rows now stay in global memory (k_mesh_walk<., ., false>): forty small icospheres over a
    floor, one of them the light, against the oracle's loop over every triangle; and the small scene through BOTH forms, bit for bit."""
    sc = gpu.Scene(os.path.join(SCENES, "mesh_small.txt"))
    tris = _icosphere(0)
    rng = np.random.default_rng(5)
    geoms = [oracle.make_geom(1, 1, (0, -0.1, 0), (0, 0, 0), (14, 0.2, 14)),                    # a floor (cube, diffuse)
             oracle.make_geom(2, 0, (0, 7, 0), (0, 0, 0), (5, 0.6, 5))]                          # the light: a flattened icosphere
    for k in range(39):
        x, z = (k % 7 - 3) * 1.6 + rng.uniform(-0.3, 0.3), (k // 7 - 2.5) * 1.6 + rng.uniform(-0.3, 0.3)
        geoms.append(oracle.make_geom(2, [1, 4, 5][k % 3], (x, rng.uniform(0.6, 2.5), z), tuple(rng.uniform(0, 90, 3)), tuple(rng.uniform(0.7, 1.4, 3))))
    sc.geoms = np.concatenate(geoms).view(sc.geoms.dtype)
    sc.meshes = {i: tris for i in range(1, 41)}
    sc.set_resolution = None
    got = _render_both(gpu, oracle, sc, 4, [1, 2], (96, 96), dump_bounces=(1, 2))
    assert (got > 0).mean() > 0.01
    small = gpu.Scene(os.path.join(SCENES, "mesh_small.txt"))
    lds = _render_both(gpu, oracle, small, 5, [1, 2, 3], (96, 96))
    monkeypatch.setenv("PT_AMD_WALK_ROWS_GLOBAL", "1")
    glob = _render_both(gpu, oracle, small, 5, [1, 2, 3], (96, 96))
    assert np.array_equal(lds.view(np.uint32), glob.view(np.uint32))
