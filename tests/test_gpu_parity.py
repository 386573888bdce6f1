"""GPU parity tests (pytest -m gpu): the HIP path, called through the C ABI, against the CPU oracle
and the committed golden fixtures.  Integer work and reference-defined fp32 primitives must be
BIT-EXACT; rendered pixels must agree within the north-star tolerance of 1e-3 relative (they are in
fact bit-identical, which is asserted as well and reported separately)."""
import json
import os

import numpy as np
import pytest

# PT_SWEEP_SCALE=<n>: the certificates' sweeps with n times the rays and other seeds (a longer one-off run: profiles/r03_sweeps.txt)
_SW = int(os.environ.get("PT_SWEEP_SCALE", "1"))

from conftest import GOLD, SCENES

pytestmark = pytest.mark.gpu

REL_TOL = 1e-3   # BASELINE.json north_star: per-pixel RGB within 1e-3 relative of the CPU path


def bits(a):
    a = np.ascontiguousarray(a, np.float32)
    b = a.view(np.uint32).copy()
    b[np.isnan(a)] = 0x7FC00000
    return b


def rel_err(got, want):
    return np.abs(got - want) / np.maximum(np.abs(want), 1e-6)


@pytest.fixture(scope="module")
def gpu(pt):
    if pt.device_count() < 1:
        pytest.fail("no HIP device: GPU tests must run on the MI355X box")
    return pt


def _render_both(gpu, oracle, scene_name, res, depth, iters, shard=(0, 1)):
    sc = gpu.Scene(os.path.join(SCENES, scene_name))
    sc.set_resolution(*res)
    W, H = res
    ref = oracle.Renderer(sc.camera.view(oracle.CAMERA_DTYPE), sc.geoms.view(oracle.GEOM_DTYPE),
                          sc.materials.view(oracle.MATERIAL_DTYPE), depth)
    want = np.zeros(W * H * 3, np.float32)
    gpu.pathtraceFree()
    gpu.pathtraceInit(sc, shard_rank=shard[0], shard_count=shard[1], traceDepth=depth)
    live = np.zeros(64, np.int64)
    hits = 0
    for it in iters:
        gpu.pathtrace(None, 0, it, readback=False)
        c = ref.iterate(it, want, shard[0], shard[1])
        live += np.array(c.live[:64])
        hits += c.lightHits
    got = gpu.readback(W * H)
    cnt = gpu.counters()
    gpu.pathtraceFree()
    return got, want, cnt, live, hits


# ----------------------------------------------------------------------------- primitives (rows a6-a12, a19)
def test_utilhash_bit_exact(gpu):
    z = np.load(os.path.join(GOLD, "utilhash.npz"))
    assert np.array_equal(gpu.test_utilhash(z["x"]), z["h"])


def test_rng_bit_exact_vs_thrust(gpu):
    z = np.load(os.path.join(GOLD, "rng_thrust.npz"))
    u = gpu.test_rng(z["seeds"], z["u01_bits"].shape[1])
    assert np.array_equal(u.view(np.uint32), z["u01_bits"])


def test_intersections_bit_exact_vs_reference_vectors(gpu, oracle):
    z = np.load(os.path.join(GOLD, "intersections.npz"))
    G = np.frombuffer(z["geoms"].tobytes(), gpu.GEOM_DTYPE)
    ng, n = z["rays"].shape[0], z["rays"].shape[1]
    gi = np.repeat(np.arange(ng, dtype=np.int32), n)
    t, p, nn, o = gpu.test_intersect(G, gi, z["rays"].reshape(-1, 6))
    assert np.array_equal(bits(t), bits(z["t"].reshape(-1)))
    assert np.array_equal(bits(p), bits(z["p"].reshape(-1, 3)))     # includes 'untouched on a miss'
    assert np.array_equal(bits(nn), bits(z["n"].reshape(-1, 3)))
    assert np.array_equal(o, z["outside"].reshape(-1))


def test_slab_quotients_equal_ieee_division(gpu):
    # The box test's two quotients per axis share one reciprocal and run as packed Newton steps inside a
    # guarded exponent range; they must equal the correctly rounded `/` bit for bit, everywhere.
    rng = np.random.default_rng(11)
    n = 1 << 20
    o = np.concatenate([rng.uniform(-2000, 2000, n // 2), rng.uniform(-1, 1, n // 4),
                        rng.integers(0, 2**32, n // 4, dtype=np.uint64).astype(np.uint32).view(np.float32)]).astype(np.float32)
    d = np.concatenate([rng.uniform(-1, 1, n // 2), rng.normal(size=n // 4) * 1e-6,
                        rng.integers(0, 2**32, n // 4, dtype=np.uint64).astype(np.uint32).view(np.float32)]).astype(np.float32)
    special = np.array([0.0, -0.0, 0.5, -0.5, 1.0, np.inf, -np.inf, np.nan, 1e-45, 1e-38, 2.0**-40, 2.0**-41, 2.0**40,
                        2.0**41, 2.0**39, 3.4e38, 1e-30], np.float32)
    so, sd = np.meshgrid(special, special)
    o = np.concatenate([o, so.ravel()])
    d = np.concatenate([d, sd.ravel()])
    t1, t2, r1, r2 = gpu.test_slab_quotients(o, d)
    assert np.array_equal(bits(t1), bits(r1)) and np.array_equal(bits(t2), bits(r2))
    # and against numpy's IEEE division on the host
    with np.errstate(all="ignore"):
        assert np.array_equal(bits(r1), bits((np.float32(-0.5) - o) / d))
        assert np.array_equal(bits(r2), bits((np.float32(0.5) - o) / d))
    assert gpu.test_slab_quotients_sweep(12345, 1 << 30) == 0        # 2^30 more pairs on the device


def test_box_fast_path_equals_the_exact_one(gpu, oracle):
    # round 5: the box test decides its slab comparisons on approximate quotients where their outcome is beyond doubt and divides
    # exactly ONCE (pt_device.h: boxSlabsFast); any doubt or operand outside its guards runs the reference's loop.  Both forms of the
    # test, bit for bit, on rays dense in edges, corners, grazes, surface origins and degenerate directions -- thin walls, rotated and
    # sheared boxes, tiny and huge ones, boxes far from the origin.
    S = oracle.make_geom
    sets = {
        "cornell walls": [S(1, 0, (0, 0, 0), (0, 0, 0), (10, .01, 10)), S(1, 0, (0, 10, 0), (0, 0, 90), (.01, 10, 10)), S(1, 0, (0, 5, -5), (0, 90, 0), (.01, 10, 10)),
                          S(1, 0, (-5, 5, 0), (0, 0, 0), (.01, 10, 10)), S(1, 0, (5, 5, 0), (0, 0, 0), (.01, 10, 10)), S(1, 0, (0, 10, 0), (0, 0, 0), (3, .3, 3))],
        "rotated": [S(1, 0, (1, 2, 3), (30, 45, 60), (1, 2, 3)), S(1, 0, (-2, 1, 0), (10, 200, -75), (.5, .5, 4)), S(1, 0, (0, 0, 0), (45, 45, 45), (1, 1, 1))],
        "extremes": [S(1, 0, (0, 0, 0), (0, 0, 0), (1e-3, 1e-3, 1e-3)), S(1, 0, (300, -200, 100), (5, 5, 5), (400, 1, 400)), S(1, 0, (1e4, 1e4, 1e4), (0, 30, 0), (2, 2, 2)),
                     S(1, 0, (0, 0, 0), (0, 0, 0), (1e4, 1e-4, 1))],
    }
    total_fast = 0
    for name, geoms in sets.items():
        g = np.concatenate(geoms).view(gpu.GEOM_DTYPE)
        rays, fast, hits, bad, bad_rcp, bad_div = gpu.test_box_fast_sweep(g, 31337 + _SW - 1, (1 << 26) * _SW)
        print("%-14s rays %d, decided by the fast path %d (%.1f %%), hits among those %d, mismatches %d" % (name, rays, fast, 100.0 * fast / rays, hits, bad))
        assert bad == 0 and bad_rcp == 0 and bad_div == 0, name
        assert fast > rays // 2 and hits > rays // 16, name          # the fast path actually decides, hits and misses
        total_fast += fast
    assert total_fast > 1 << 26


def test_unscaled_sqrt_exhaustive(gpu):
    # The hemisphere sampler issues the compiler's correctly rounded sqrt WITHOUT the instructions that only act near
    # the exponent limits (pt_device.h: sqrtUnscaled); its operands (u01 and 1 - up^2) are 0 or >= 2^-31 by
    # construction.  Compared with __builtin_sqrtf on every fp32 bit pattern of the range +-0, [2^-96, inf).
    bad, checked, bad_inv, short_path = gpu.test_unscaled_sqrt_sweep()
    assert checked == 2 + ((0x7f800000 - 0x0f800000))      # +-0 and every positive float from 2^-96 up to FLT_MAX
    assert bad == 0
    # inverseSqrtNearOne (getPointOnRay's re-normalisation: integer arithmetic within 256 ulps of 1, sqrt + division
    # elsewhere) against 1.0f / sqrtf(x) on all 2^32 bit patterns
    assert bad_inv == 0 and short_path == 513


def test_sphere_culling_never_rejects_a_hit(gpu, oracle):
    # certainMiss is a sufficient condition for the reference's `radicand < 0` exit; 2^28 rays (dense in
    # grazing cases, origins 1/64 .. 64 units away) against uniform spheres, ellipsoids, tiny and huge ones
    geoms = np.concatenate([
        oracle.make_geom(0, 0, (-1, 4, -1), (0, 0, 0), (3, 3, 3)),            # Cornell sphere
        oracle.make_geom(0, 0, (1, 2, 3), (30, 45, 60), (1, 2, 3)),           # ellipsoid
        oracle.make_geom(0, 0, (0, 0, 0), (0, 0, 0), (0.6, 0.6, 0.6)),        # C5-sized
        oracle.make_geom(0, 0, (2.5, 6, -2), (10, 20, 30), (0.05, 0.05, 0.05)),
        oracle.make_geom(0, 0, (-3, 1, 2), (75, -20, 130), (8, 0.5, 3)),      # 16:1 anisotropy
        oracle.make_geom(0, 0, (100, -50, 25), (0, 0, 0), (40, 40, 40)),
    ]).view(gpu.GEOM_DTYPE)
    culled, bad = gpu.test_sphere_cull_sweep(geoms, 2024 + _SW - 1, (1 << 28) * _SW)
    assert bad == 0
    assert culled > (1 << 28) // 10           # the shortcut actually fires (most sweep rays are aimed at the sphere)


def test_sphere_halfline_certificate_never_rejects_a_hit(gpu, oracle):
    # sphereHalfLineExcess -- what the sphere-heavy sweep of the later bounces certifies a miss with: the centre's distance from the
    # HALF-line, the direction normalised approximately -- is a sufficient condition for the reference's miss, `radicand < 0` or
    # `t1 < 0 && t2 < 0` (src/intersections.h:114, 121-123).  2^28 rays: aimed near the ball from 1/64 .. 64 units, leaving the
    # sphere's own surface the way a scatter does (+-1e-3 along the normal), starting within 2 % of the bounding ball's surface;
    # directions unit, nearly unit and far from unit.
    geoms = np.concatenate([
        oracle.make_geom(0, 0, (-1, 4, -1), (0, 0, 0), (3, 3, 3)),            # Cornell sphere
        oracle.make_geom(0, 0, (1, 2, 3), (30, 45, 60), (1, 2, 3)),           # ellipsoid
        oracle.make_geom(0, 0, (0, 0, 0), (0, 0, 0), (0.6, 0.6, 0.6)),        # C5-sized
        oracle.make_geom(0, 0, (2.5, 6, -2), (10, 20, 30), (0.05, 0.05, 0.05)),
        oracle.make_geom(0, 0, (-3, 1, 2), (75, -20, 130), (8, 0.5, 3)),      # 16:1 anisotropy
        oracle.make_geom(0, 0, (100, -50, 25), (0, 0, 0), (40, 40, 40)),
        oracle.make_geom(0, 0, (4.1, 8.7, -3.3), (0, 0, 0), (1.1, 1.1, 1.1)),
    ]).view(gpu.GEOM_DTYPE)
    culled, behind, bad = gpu.test_sphere_halfline_sweep(geoms, 2026 + _SW - 1, (1 << 28) * _SW)
    assert bad == 0
    assert culled > (1 << 28) // 5 and behind > (1 << 28) // 16      # both branches of the certificate fire, by the tens of millions


def test_halfline_certificate_of_swept_cubes_never_rejects_a_hit(gpu, oracle):
    # round 5: scenes with many small primitives sweep their small CUBES like their spheres -- the same certificate against the cube's
    # bounding ball (rho = sqrt(3) / 2: its corners), i.e. a sufficient condition for the reference's `tmax >= tmin && tmax > 0` to fail
    # (src/intersections.h:70).  2^28 rays of the same families against cubes: small, rotated, flat, far away.
    geoms = np.concatenate([
        oracle.make_geom(1, 0, (0, 10, 0), (0, 0, 0), (3, .3, 3)),            # Cornell's light
        oracle.make_geom(1, 0, (1, 2, 3), (30, 45, 60), (1, 2, 3)),
        oracle.make_geom(1, 0, (0, 0, 0), (0, 0, 0), (0.6, 0.6, 0.6)),
        oracle.make_geom(1, 0, (2.5, 6, -2), (10, 20, 30), (0.05, 0.05, 0.05)),
        oracle.make_geom(1, 0, (-3, 1, 2), (75, -20, 130), (8, 0.5, 3)),
        oracle.make_geom(1, 0, (100, -50, 25), (0, 0, 0), (40, 40, 40)),
        oracle.make_geom(1, 0, (4.1, 8.7, -3.3), (123, 231, 312), (1.1, 0.7, 1.9)),
    ]).view(gpu.GEOM_DTYPE)
    culled, behind, bad = gpu.test_sphere_halfline_sweep(geoms, 2027 + _SW - 1, (1 << 28) * _SW)
    assert bad == 0
    assert culled > (1 << 28) // 8 and behind > (1 << 28) // 32


def test_sphere_cluster_boxes_never_reject_a_hit(gpu, oracle):
    # Sphere-heavy scenes without meshes bin the survivors by which of TWO spatial clusters of spheres their ray can hit (class bits 3 / 4,
    # a slab certificate against each cluster's inflated box: pt_api.hip build_sphere_clusters, k_bounce CLUSTER); a tile then sweeps only
    # the clusters its class names.  A box certified as missed must imply the reference's miss (src/intersections.h:101-143) for EVERY
    # sphere behind it.  2^28 rays per scene: scatters off the spheres, rays from the scene's extent at the boxes and their shells, from
    # close to the boxes, axis- and plane-parallel ones.  Scenes: config C5's own; ellipsoids, tiny and large spheres, rotated, far off the
    # origin and between walls that are not axis-parallel.
    sc = oracle.Scene(os.path.join(SCENES, "spheres64.txt"))
    rng = np.random.default_rng(41)
    odd = [oracle.make_geom(1, 1, (0, 0, 0), (0, 0, 0), (30, 0.02, 30)), oracle.make_geom(1, 1, (0, 12, 0), (0, 0, 30), (30, 0.02, 30)),
           oracle.make_geom(1, 1, (-9, 6, 0), (0, 0, 90), (14, 0.02, 30))]
    for i in range(23):
        c = rng.uniform(-6, 6, 3) + np.array([3.0, 6.0, -2.0])
        scale = np.exp(rng.uniform(np.log(0.05), np.log(3.0), 3)) if i % 3 else np.full(3, np.exp(rng.uniform(np.log(0.05), np.log(4.0))))
        odd.append(oracle.make_geom(0, 1, tuple(c), tuple(rng.uniform(-180, 180, 3)), tuple(scale)))
    far = [oracle.make_geom(1, 1, (100, -50, 25), (0, 0, 0), (60, 0.1, 60))]
    for i in range(9):
        far.append(oracle.make_geom(0, 1, tuple(np.array([100.0, -40.0, 25.0]) + rng.uniform(-20, 20, 3)), (0, 0, 0), (6 + i,) * 3))
    total = 0
    for name, geoms in (("spheres64", sc.geoms), ("odd", np.concatenate(odd)), ("far", np.concatenate(far))):
        cert, bad, info = gpu.test_sphere_cluster_sweep(geoms.view(gpu.GEOM_DTYPE), 4242 + _SW - 1, (1 << 28) * _SW)
        assert bad == 0, (name, bad, info)
        # both clusters' certificates fire: by the tens of millions in C5's scene, by the millions where anisotropic spheres (K |oc|^2 is
        # large for them) make the boxes wide
        assert min(cert) > ((1 << 28) // 16 if name == "spheres64" else (1 << 20)), (name, cert, info)
        assert info["omax"] > 0 and info["n0"] >= 2 and info["n0"] % 2 == 0
        total += sum(cert)
    assert total > 0


def test_sphere_group_balls_never_reject_a_hit(gpu, oracle):
    # round 6: scenes of hundreds of swept primitives keep their packed table in spatial groups of 16 with a bounding ball each; a later
    # tile sweeps a group's members only for the lanes whose half-line may reach the ball (k_bounce<..., GROUPS>; pt_host_scene.h:
    # build_sphere_groups).  A ball certified as missed must imply the reference's miss (src/intersections.h:101-143) for EVERY member.
    # 2^28 rays per scene: the 512-sphere lattice; 300 random ellipsoids (anisotropic: K |oc|^2 is large for them), tiny and large,
    # rotated; a cloud far off the origin.
    sc = oracle.Scene(os.path.join(SCENES, "spheres512.txt"))
    rng = np.random.default_rng(43)
    odd = [oracle.make_geom(1, 1, (0, 0, 0), (0, 0, 0), (30, 0.02, 30)), oracle.make_geom(1, 1, (0, 12, 0), (0, 0, 30), (30, 0.02, 30))]
    for i in range(300):
        c = rng.uniform(-6, 6, 3) + np.array([3.0, 6.0, -2.0])
        scale = np.exp(rng.uniform(np.log(0.05), np.log(2.0), 3)) if i % 3 else np.full(3, np.exp(rng.uniform(np.log(0.05), np.log(3.0))))
        odd.append(oracle.make_geom(0, 1, tuple(c), tuple(rng.uniform(-180, 180, 3)), tuple(scale)))
    far = [oracle.make_geom(1, 1, (100, -50, 25), (0, 0, 0), (60, 0.1, 60))]
    for i in range(150):
        far.append(oracle.make_geom(0, 1, tuple(np.array([100.0, -40.0, 25.0]) + rng.uniform(-20, 20, 3)), (0, 0, 0), (0.5 + 0.02 * i,) * 3))
    for name, geoms in (("spheres512", sc.geoms), ("odd", np.concatenate(odd)), ("far", np.concatenate(far))):
        cert, bad, ng = gpu.test_sphere_group_sweep(geoms.view(gpu.GEOM_DTYPE), 4343 + _SW - 1, (1 << 28) * _SW)
        assert bad == 0, (name, bad, cert, ng)
        assert ng >= 9 and cert > (1 << 28), (name, cert, ng)        # (every ray is tried against every group: most groups are certified)


def test_cube_culling_never_rejects_a_hit(gpu, oracle):
    # the same bounding-ball test decides which queue tiles skip the scene's small cubes (k_bounce bins survivors by
    # it): it must imply the reference's own miss for cubes of every shape -- the Cornell light and walls, thin plates
    # seen edge-on, rotated and sheared-looking boxes, tiny and huge ones
    geoms = np.concatenate([
        oracle.make_geom(1, 0, (0, 10, 0), (0, 0, 0), (3, 0.3, 3)),           # Cornell light
        oracle.make_geom(1, 0, (0, 5, -5), (0, 90, 0), (0.01, 10, 10)),       # Cornell back wall (1000:1)
        oracle.make_geom(1, 0, (1, 2, 3), (30, 45, 60), (1, 2, 3)),
        oracle.make_geom(1, 0, (0, 0, 0), (0, 0, 0), (1, 1, 1)),              # the unit cube itself
        oracle.make_geom(1, 0, (2.5, 6, -2), (10, 20, 30), (0.05, 0.05, 0.05)),
        oracle.make_geom(1, 0, (-3, 1, 2), (75, -20, 130), (8, 0.5, 3)),
        oracle.make_geom(1, 0, (100, -50, 25), (45, 45, 45), (40, 40, 40)),
        oracle.make_geom(1, 0, (0.5, 0.25, -0.75), (0, 0, 45), (2, 2, 0.2)),
        oracle.make_geom(1, 0, (-2, 3, 1), (20, 70, -35), (5, 0.1, 5)),       # 50:1 plate, near the anisotropy limit of the test
        oracle.make_geom(1, 0, (4, -1, 2), (-60, 15, 80), (0.12, 6, 0.12)),   # 50:1 rod
    ]).view(gpu.GEOM_DTYPE)
    culled, bad = gpu.test_sphere_cull_sweep(geoms, 77 + _SW - 1, (1 << 28) * _SW)
    assert bad == 0
    assert culled > (1 << 28) // 16            # (the elongated shapes carry wide margins and are rarely culled)


def test_wall_boxes_never_reject_a_hit(gpu, oracle):
    # the world-space slab test against a large cube's inflated bounding box (ptd::wallCertainMiss) classes the queue by the
    # one wall a path can still hit and ends paths that can hit nothing: it must imply the reference's own miss for every
    # cube it may be handed -- the Cornell walls (1000:1 plates, one of them rotated by 90 degrees with its 4e-8 matrix
    # entries), the light, rotated and oblique boxes, tiny and huge ones, far from the origin
    geoms = np.concatenate([
        oracle.make_geom(1, 0, (0, 0, 0), (0, 0, 90), (0.01, 10, 10)),        # Cornell floor
        oracle.make_geom(1, 0, (0, 10, 0), (0, 0, 90), (0.01, 10, 10)),       # ceiling
        oracle.make_geom(1, 0, (0, 5, -5), (0, 90, 0), (0.01, 10, 10)),       # back wall
        oracle.make_geom(1, 0, (-5, 5, 0), (0, 0, 0), (0.01, 10, 10)),        # left wall
        oracle.make_geom(1, 0, (0, 10, 0), (0, 0, 0), (3, 0.3, 3)),           # light
        oracle.make_geom(1, 0, (1, 2, 3), (30, 45, 60), (1, 2, 3)),
        oracle.make_geom(1, 0, (0, 0, 0), (0, 0, 0), (1, 1, 1)),              # the unit cube itself
        oracle.make_geom(1, 0, (2.5, 6, -2), (10, 20, 30), (0.05, 0.05, 0.05)),
        oracle.make_geom(1, 0, (-3, 1, 2), (75, -20, 130), (8, 0.5, 3)),
        oracle.make_geom(1, 0, (100, -50, 25), (45, 45, 45), (40, 40, 40)),
        oracle.make_geom(1, 0, (-2, 3, 1), (20, 70, -35), (5, 0.1, 5)),
        oracle.make_geom(1, 0, (0, -1, 0), (0, 0, 0), (30, 1, 30)),           # a ground slab
    ]).view(gpu.GEOM_DTYPE)
    culled, bad = gpu.test_wall_box_sweep(geoms, 4242 + _SW - 1, (1 << 28) * _SW)
    assert bad == 0
    assert culled > (1 << 28) // 16            # the certificate actually fires (most sweep rays are aimed at the cube)


def test_wall_planes_never_reject_a_hit(gpu, oracle):
    # round 3: most walls are certified by ONE plane of their inflated box and the point where the ray leaves the box around all the
    # walls (ptd::wallPlanesPossible) instead of the slab test -- same statement (the exact ray misses the inflated box), a fifth of
    # the instructions.  Swept over rooms as pt_init would number their walls: Cornell open and closed, a rotated and a tilted room,
    # walls that overlap in the corners, a slab across the middle (which gets no plane), rooms far off the origin, a huge one
    S = oracle.make_geom
    cornell = [S(1, 0, (0, 0, 0), (0, 0, 0), (10, .01, 10)), S(1, 0, (0, 10, 0), (0, 0, 90), (.01, 10, 10)), S(1, 0, (0, 5, -5), (0, 90, 0), (.01, 10, 10)),
               S(1, 0, (-5, 5, 0), (0, 0, 0), (.01, 10, 10)), S(1, 0, (5, 5, 0), (0, 0, 0), (.01, 10, 10))]
    rooms = {
        "cornell": cornell,
        "cornell closed": cornell + [S(1, 0, (0, 5, 5), (0, 90, 0), (.01, 10, 10))],
        "thick walls": [S(1, 0, (0, -1, 0), (0, 0, 0), (12, 2, 12)), S(1, 0, (0, 9, 0), (0, 0, 0), (12, 2, 12)), S(1, 0, (-6, 4, 0), (0, 0, 0), (2, 12, 12)),
                        S(1, 0, (6, 4, 0), (0, 0, 0), (2, 12, 12)), S(1, 0, (0, 4, -6), (0, 0, 0), (12, 12, 2))],
        "tilted": [S(1, 0, (0, 0, 0), (3, 0, -2), (10, .05, 10)), S(1, 0, (0, 8, 0), (-4, 10, 0), (10, .05, 10)), S(1, 0, (-5, 4, 0), (0, 5, 88), (8, .05, 10)),
                   S(1, 0, (5, 4, 0), (7, 0, 93), (8, .05, 10)), S(1, 0, (0, 4, -5), (85, 3, 0), (10, .05, 8))],
        # round 5: rotated walls are certified by the plane of their own inner face, whatever its direction (ptd::wallPlanesOriented), and
        # the ray's segment ends where it leaves the half-spaces that hold every wall: the walls of scenes/room_tilted.txt with its tilted light
        # (small: it keeps the slab test and gets no plane), a room turned as a whole by 30 / 20 degrees, one with walls leaning 25 degrees
        "room_tilted.txt": [S(1, 0, (0, 0, 0), (3, 17, -2), (12, .05, 12)), S(1, 0, (0, 10, 0), (-4, 10, 2), (12, .05, 12)), S(1, 0, (0, 5, -5.5), (85, 3, 12), (12, .05, 11)),
                            S(1, 0, (-5.5, 5, 0), (0, 15, 88), (11, .05, 12)), S(1, 0, (5.5, 5, 0), (7, -12, 93), (11, .05, 12)), S(1, 0, (0.3, 9.4, -0.5), (6, 20, -4), (3.5, .3, 3))],
        "turned room": [S(1, 0, (0, -0.9, -1.6), (20, 30, 0), (10, .05, 10)), S(1, 0, (0, 8.5, 1.8), (20, 30, 0), (10, .05, 10)),
                        S(1, 0, (-4.3, 3.0, -2.8), (20, 30, 90), (10, .05, 10)), S(1, 0, (4.3, 4.6, 2.2), (20, 30, 90), (10, .05, 10)),
                        S(1, 0, (-2.3, 3.0, -4.2), (110, 30, 0), (10, .05, 10))],
        "leaning walls": [S(1, 0, (0, 0, 0), (0, 0, 0), (14, .1, 14)), S(1, 0, (-5, 4, 0), (0, 0, 65), (10, .1, 12)), S(1, 0, (5, 4, 0), (0, 0, 115), (10, .1, 12)),
                          S(1, 0, (0, 4, -5), (65, 0, 0), (12, .1, 10)), S(1, 0, (0, 8.5, 0), (5, 40, 3), (8, .1, 8))],
        "slab across": cornell + [S(1, 0, (0, 5, 0), (0, 0, 0), (10, .5, 3))],
        "far away": [S(1, 0, (100, 50, -30), (0, 0, 0), (10, .01, 10)), S(1, 0, (100, 60, -30), (0, 0, 0), (10, .01, 10)),
                     S(1, 0, (95, 55, -30), (0, 0, 90), (10, .01, 10)), S(1, 0, (105, 55, -30), (0, 0, 90), (10, .01, 10))],
        "huge": [S(1, 0, (0, 0, 0), (0, 0, 0), (400, 1, 400)), S(1, 0, (0, 300, 0), (0, 0, 0), (400, 1, 400)), S(1, 0, (-200, 150, 0), (0, 0, 0), (1, 300, 400)),
                 S(1, 0, (200, 150, 0), (0, 0, 0), (1, 300, 400))],
    }
    total = 0
    for name, geoms in rooms.items():
        g = np.concatenate(geoms).view(gpu.GEOM_DTYPE)
        nplane, certified, bad, single = gpu.test_wall_plane_sweep(g, 977 + _SW - 1, (1 << 25) * _SW)
        print("%-15s walls with a plane %d of %d, certificates %d, rays with one possible wall %d, violations %d" % (name, nplane, len(g), certified, single, bad))
        assert bad == 0, name
        assert nplane == (len(g) - 1 if name in ("slab across", "room_tilted.txt") else len(g)), name
        assert certified > (1 << 25) and single > (1 << 25) // 8, name          # it certifies, and mostly down to one wall
        total += certified
    assert total > 1 << 28


def test_reflect_refract_bit_exact(gpu):
    z = np.load(os.path.join(GOLD, "glm_ops.npz"))
    r1, r2 = gpu.test_reflect_refract(z["An"], z["Bn"], z["eta"])
    assert np.array_equal(bits(r1), bits(z["reflect"]))
    assert np.array_equal(bits(r2), bits(z["refract"]))


def test_sincos_and_hemisphere_bit_exact_vs_oracle(gpu, oracle):
    xs = np.concatenate([np.linspace(0, 2 * np.pi, 50001), [0.0, np.pi / 4, np.pi / 2, np.pi, 2 * np.pi]]).astype(np.float32)
    s, c = gpu.test_sincos(xs)
    ws, wc = np.empty_like(xs), np.empty_like(xs)
    for i, x in enumerate(xs):
        ws[i], wc[i] = oracle.sincos(float(x))
    assert np.array_equal(bits(s), bits(ws)) and np.array_equal(bits(c), bits(wc))

    rng = np.random.default_rng(7)
    n = 4096
    nrm = rng.normal(size=(n, 3)).astype(np.float32)
    nrm = np.stack([oracle.normalize(v) for v in nrm])
    nrm[:6] = [[0, 1, 0], [1, 0, 0], [0, 0, 1], [0, -1, 0], [0.6, 0, 0.8], oracle.normalize([1, 1, 1])]
    iid = np.stack([rng.integers(1, 5000, n), rng.integers(0, 921600, n), rng.integers(1, 9, n)], axis=1).astype(np.int32)
    got = gpu.test_hemisphere(nrm, iid)
    want = np.stack([oracle.hemisphere_seeded(nrm[i], *[int(v) for v in iid[i]]) for i in range(n)])
    assert np.array_equal(bits(got), bits(want))


def test_hemisphere_survey_kats(gpu, oracle):
    k = json.load(open(os.path.join(GOLD, "survey_kats.json")))["hemisphere"]
    nrm = np.array([c["n"] if "n" in c else oracle.normalize(c["n_unnormalized"]) for c in k], np.float32)
    iid = np.array([[c["iter"], c["index"], c["depth"]] for c in k], np.int32)
    got = gpu.test_hemisphere(nrm, iid)
    assert np.allclose(got, [c["out"] for c in k], rtol=0, atol=4e-7)


# ----------------------------------------------------------------------------- per-bounce state parity
@pytest.mark.parametrize("scene,res", [("cornell.txt", (160, 120)), ("cornell_glass.txt", (128, 96))])
def test_path_state_after_each_bounce_equals_oracle(gpu, oracle, scene, res):
    sc = gpu.Scene(os.path.join(SCENES, scene))
    sc.set_resolution(*res)
    depth = 8
    ref = oracle.Renderer(sc.camera.view(oracle.CAMERA_DTYPE), sc.geoms.view(oracle.GEOM_DTYPE),
                          sc.materials.view(oracle.MATERIAL_DTYPE), depth)
    gpu.pathtraceFree()
    gpu.pathtraceInit(sc, traceDepth=depth)
    npix = res[0] * res[1]
    for it in (1, 37):
        for b in (0, 1, 2, 5, 8):
            o, d, c, pix = gpu.debug_trace_paths(it, b, npix)
            wo, wd, wc, wpix = ref.dump_paths(it, b)
            assert len(pix) == len(wpix), (it, b)
            assert np.array_equal(pix, wpix), (it, b)                   # stable (pixel-order) compaction
            for g, w, what in ((o, wo, "origin"), (d, wd, "direction"), (c, wc, "throughput")):
                assert np.array_equal(bits(g), bits(w)), (it, b, what)
    gpu.pathtraceFree()


# ----------------------------------------------------------------------------- rendered image parity
@pytest.mark.parametrize("scene,res,depth,iters", [
    ("sphere.txt", (400, 400), 4, [1]),                       # BASELINE config C1
    ("cornell.txt", (320, 180), 8, [1, 2, 3, 4]),             # config C2's scene at reduced size
    ("cornell_glass.txt", (240, 135), 16, [1, 2, 5000]),      # config C4's scene (refraction + Schlick)
    ("spheres64.txt", (128, 128), 8, [1, 2]),                 # config C5's scene (70 primitives in LDS)
    ("cornell.txt", (333, 77), 1, [9]),                       # ragged size, depth 1
])
def test_render_matches_oracle(gpu, oracle, scene, res, depth, iters):
    got, want, cnt, live, hits = _render_both(gpu, oracle, scene, res, depth, iters)
    err = rel_err(got, want)
    assert float(err.max()) <= REL_TOL, f"max rel err {err.max()} over {int(np.sum(err > REL_TOL))} values"
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "not bit-identical to the oracle"
    assert list(cnt.live[1:depth + 1]) == list(live[1:depth + 1])      # per-bounce live counts (README.md:286-289)
    assert cnt.light_hits == hits and cnt.iterations == len(iters)


def test_sharded_render_sums_to_the_unsharded_frame(gpu, oracle):
    # rows y % 3 == r rendered separately into zeroed full frames; x + 0 is exact, so the sum of the
    # three accumulators is bit-identical to the single-GPU frame
    res, depth, iters = (200, 111), 8, [1, 2]
    full, want, _, _, _ = _render_both(gpu, oracle, "cornell.txt", res, depth, iters)
    acc = np.zeros_like(full)
    for r in range(3):
        part, pwant, _, _, _ = _render_both(gpu, oracle, "cornell.txt", res, depth, iters, shard=(r, 3))
        assert np.array_equal(part.view(np.uint32), pwant.view(np.uint32))
        rows = part.reshape(res[1], res[0], 3)
        assert not np.any(rows[np.arange(res[1]) % 3 != r])            # untouched rows stay zero
        acc += part
    assert np.array_equal(acc.view(np.uint32), full.view(np.uint32))
    assert np.array_equal(full.view(np.uint32), want.view(np.uint32))


def test_row_sharded_packed_accumulator(gpu, oracle):
    # PT_FLAG_ACCUM_SHARD_ROWS: the accumulator holds only this shard's rows; pt_readback scatters them back
    res, depth, iters = (96, 50), 8, [1, 2, 3]
    W, H = res
    sc = gpu.Scene(os.path.join(SCENES, "cornell.txt"))
    sc.set_resolution(*res)
    ref = oracle.Renderer(sc.camera.view(oracle.CAMERA_DTYPE), sc.geoms.view(oracle.GEOM_DTYPE),
                          sc.materials.view(oracle.MATERIAL_DTYPE), depth)
    full = np.zeros(W * H * 3, np.float32)
    for it in iters:
        ref.iterate(it, full)
    acc = np.zeros_like(full)
    for r in range(3):
        gpu.pathtraceFree()
        gpu.pathtraceInit(sc, shard_rank=r, shard_count=3, traceDepth=depth, flags=gpu.PT_FLAG_ACCUM_SHARD_ROWS)
        for it in iters:
            gpu.pathtrace(None, 0, it, readback=False)
        part = gpu.readback(W * H)
        rows = part.reshape(H, W, 3)
        assert not np.any(rows[np.arange(H) % 3 != r])
        acc += part
    gpu.pathtraceFree()
    assert np.array_equal(acc.view(np.uint32), full.view(np.uint32))


@pytest.mark.parametrize("pipeline", [1, 2, 4])
def test_pipeline_depth_does_not_change_the_image(gpu, oracle, pipeline):
    # iterations overlap on internal streams; radiance is committed in iteration order, so the running sum is
    # bit-identical to the sequential oracle for every depth of the pipeline
    res, depth = (200, 120), 8
    sc = gpu.Scene(os.path.join(SCENES, "cornell.txt"))
    sc.set_resolution(*res)
    ref = oracle.Renderer(sc.camera.view(oracle.CAMERA_DTYPE), sc.geoms.view(oracle.GEOM_DTYPE),
                          sc.materials.view(oracle.MATERIAL_DTYPE), depth)
    want = np.zeros(res[0] * res[1] * 3, np.float32)
    gpu.pathtraceFree()
    gpu.pathtraceInit(sc, traceDepth=depth, pipeline_depth=pipeline)
    for it in range(1, 12):
        gpu.pathtrace(None, 0, it, readback=False)
        ref.iterate(it, want)
    got = gpu.readback(res[0] * res[1])
    gpu.pathtraceFree()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("batches", [[3, 1, 4], [8], [2, 2, 2, 2]])
def test_batched_iterations_equal_sequential_iterations(gpu, oracle, batches):
    # pt_iterate_batch traces `count` iterations as one wavefront; every pixel must still receive its samples in
    # iteration order, so the running sum stays bit-identical to the sequential oracle
    res, depth = (160, 90), 8
    sc = gpu.Scene(os.path.join(SCENES, "cornell_glass.txt"))
    sc.set_resolution(*res)
    ref = oracle.Renderer(sc.camera.view(oracle.CAMERA_DTYPE), sc.geoms.view(oracle.GEOM_DTYPE),
                          sc.materials.view(oracle.MATERIAL_DTYPE), depth)
    want = np.zeros(res[0] * res[1] * 3, np.float32)
    gpu.pathtraceFree()
    gpu.pathtraceInit(sc, traceDepth=depth, max_batch=8, pipeline_depth=2)
    it, live = 1, 0
    for count in batches:
        gpu.pathtrace_batch(None, 0, it, count)
        for k in range(count):
            c = ref.iterate(it + k, want)
            live += c.live[3]
        it += count
    got = gpu.readback(res[0] * res[1])
    cnt = gpu.counters()
    with pytest.raises(gpu.PtError, match="count must be"):
        gpu.pathtrace_batch(None, 0, it, 9)
    gpu.pathtraceFree()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert cnt.iterations == sum(batches) and cnt.live[3] == live


def test_largest_batch_and_row_shard_equal_sequential_iterations(gpu, oracle):
    # what an 8-GPU run does on every rank: PT_MAX_BATCH iterations of its row shard as one wavefront, packed accumulator
    res, depth, world, rank = (96, 64), 6, 8, 3
    sc = gpu.Scene(os.path.join(SCENES, "cornell.txt"))
    sc.set_resolution(*res)
    ref = oracle.Renderer(sc.camera.view(oracle.CAMERA_DTYPE), sc.geoms.view(oracle.GEOM_DTYPE),
                          sc.materials.view(oracle.MATERIAL_DTYPE), depth)
    want = np.zeros(res[0] * res[1] * 3, np.float32)
    gpu.pathtraceFree()
    gpu.pathtraceInit(sc, traceDepth=depth, max_batch=gpu.PT_MAX_BATCH, pipeline_depth=3, shard_rank=rank,
                      shard_count=world, flags=gpu.PT_FLAG_ACCUM_SHARD_ROWS)
    it = 1
    for count in (gpu.PT_MAX_BATCH, 5, gpu.PT_MAX_BATCH):
        gpu.pathtrace_batch(None, 0, it, count)
        for k in range(count):
            ref.iterate(it + k, want)
        it += count
    got = gpu.readback(res[0] * res[1]).reshape(res[1], res[0] * 3)      # pt_readback scatters the shard's rows into a zeroed frame
    gpu.pathtraceFree()
    mine = np.arange(res[1]) % world == rank
    want = want.reshape(res[1], res[0] * 3)
    assert np.array_equal(got[mine].view(np.uint32), want[mine].view(np.uint32)) and want[mine].max() > 0
    assert not got[~mine].any()


def test_rgba8_conversion_matches_reference_formula(gpu, oracle):
    sc = gpu.Scene(os.path.join(SCENES, "cornell.txt"))
    sc.set_resolution(64, 48)
    gpu.pathtraceFree()
    gpu.pathtraceInit(sc)
    for it in (1, 2, 3):
        gpu.pathtrace(None, 0, it, readback=False)
    img = gpu.readback(64 * 48)
    got = gpu.readback_rgba8(3, 64 * 48)
    gpu.pathtraceFree()
    assert np.array_equal(got, oracle.to_rgba8(img, 3))
    assert got[:, :3].max() > 0 and np.all(got[:, 3] == 0)


def test_api_protocol_and_errors(gpu):
    gpu.pathtraceFree()
    gpu.pathtraceFree()                                   # Free before Init, twice (src/main.cpp:91-94)
    with pytest.raises(gpu.PtError, match="before pt_init"):
        gpu.pathtrace(None, 0, 1)
    sc = gpu.Scene(os.path.join(SCENES, "sphere.txt"))
    sc.set_resolution(32, 32)
    with pytest.raises(gpu.PtError, match="traceDepth"):
        gpu.pathtraceInit(sc, traceDepth=0)
    gpu.pathtraceInit(sc)
    gpu.pathtraceInit(sc)                                 # re-init without Free (camera move restarts)
    with pytest.raises(gpu.PtError, match="iter"):
        gpu.pathtrace(None, 0, 0)
    gpu.pathtrace(None, 0, 1)
    assert sc.image.sum() > 0
    gpu.pathtraceFree()


def _full_spec_blocks(gpu, name):
    """The reference's shipped configuration (scene file as is: 800x800, 5000 iterations, depth 8 -- scenes/cornell.txt:53-56,
    scenes/sphere.txt:13-16) on the HIP path -> 8-bit PNG values (saveImage + image.cpp: /samples, X mirror, clamp, x255,
    truncate) -> 50x50 means of 16x16-pixel blocks, the form tests/golden/reference_png_stats.npz holds the staff renders in."""
    sc = gpu.Scene(os.path.join(SCENES, name + ".txt"))
    W, H = (int(v) for v in sc.camera["resolution"][0])
    assert (W, H, sc.iterations, sc.traceDepth) == (800, 800, 5000, 8)
    gpu.pathtraceFree()
    gpu.pathtraceInit(sc, max_batch=64, pipeline_depth=2)
    it = 1
    while it <= sc.iterations:
        k = min(64, sc.iterations - it + 1)
        gpu.pathtrace_batch(None, 0, it, k)
        it += k
    img = gpu.readback(W * H).reshape(H, W, 3) / np.float32(sc.iterations)
    gpu.pathtraceFree()
    png = (np.clip(img, 0, 1) * np.float32(255)).astype(np.uint8)[:, ::-1].astype(np.float64)
    return png, png.reshape(50, 16, 50, 16, 3).mean(axis=(1, 3))


def test_full_spec_cornell_against_the_reference_png(gpu):
    # The reference's only end-to-end golden: img/REFERENCE_cornell.5000samp.png (staff solution; the reference has no
    # ray generation / scatterRay / accumulation to compare with, src/pathtrace.cu:160, src/interactions.h:77).  At
    # 5000 spp the Monte-Carlo noise of a 256-pixel block mean is ~0.1 % -- below the PNG's own quantisation -- so what
    # remains is the systematic difference between two implementations of the README's prose.  Measured (printed below,
    # profiles/png_stats_probe.py): global means -0.1 / -0.2 / -0.3 %, the 921 smooth lit blocks 0.8 % on average,
    # 1.9 % at the 95th percentile, 3.6 % at most; the bounds are twice that.
    z = np.load(os.path.join(GOLD, "reference_png_stats.npz"))
    png, blocks = _full_spec_blocks(gpu, "cornell")
    ref = z["cornell"].astype(np.float64)
    mean = png.reshape(-1, 3).mean(axis=0)
    print("global 8-bit mean", mean, "reference", z["cornell_mean"], "ratio", mean / z["cornell_mean"])
    assert np.all(np.abs(mean / z["cornell_mean"] - 1) < 0.01), (mean, z["cornell_mean"])
    # ALL 2500 block means: smooth lit blocks (the reference varies by < 20 % over the 3x3 neighbourhood) must agree
    # closely; the others are silhouettes, where a sub-pixel difference of the edge position moves a block mean a lot
    lum = ref.sum(-1)
    pad = np.pad(ref, ((1, 1), (1, 1), (0, 0)), mode="edge")
    nb = np.stack([pad[i:i + 50, j:j + 50] for i in range(3) for j in range(3)])
    smooth = ((nb.max(0) - nb.min(0)).sum(-1) < 0.2 * np.maximum(lum, 1)) & (lum > 24)
    rel = np.abs(blocks - ref).sum(-1) / np.maximum(lum, 1)
    print("smooth lit blocks %d: mean %.4f p95 %.4f max %.4f; all lit blocks p95 %.4f" % (
        smooth.sum(), rel[smooth].mean(), np.percentile(rel[smooth], 95), rel[smooth].max(), np.percentile(rel[lum > 24], 95)))
    assert smooth.sum() > 800
    assert rel[smooth].mean() < 0.016 and np.percentile(rel[smooth], 95) < 0.04 and rel[smooth].max() < 0.075
    assert np.percentile(rel[lum > 24], 95) < 0.09                      # silhouette blocks included
    dark = lum == 0
    assert dark.sum() > 150 and not blocks[dark].any()                   # outside the box both are exactly black
    regions = {"back wall": (slice(20, 30), slice(20, 30)), "left wall": (slice(20, 30), slice(3, 8)),
               "right wall": (slice(20, 30), slice(42, 47)), "floor": (slice(42, 47), slice(20, 30)),
               "ceiling": (slice(3, 6), slice(8, 15)), "sphere": (slice(26, 32), slice(17, 23))}
    for name, (ys, xs) in regions.items():
        ratio = blocks[ys, xs].mean(axis=(0, 1)) / ref[ys, xs].mean(axis=(0, 1))
        print("region %-10s ratio to the reference (r, g, b) %s" % (name, np.round(ratio, 3)))
        # the sphere region is where the build-defined 50/50 mirror/diffuse mixture shows: within 3 % of
        # the staff render in every channel (a pure mirror is 10-20 % darker there)
        assert np.all(np.abs(ratio - 1) < 0.04), (name, ratio)


def test_full_spec_sphere_against_the_reference_png(gpu):
    # img/REFERENCE_sphere.5000samp.png: an emissive sphere (emittance 5 -> clamped to 255) on black.  Every path ends
    # at its first bounce, so this pins camera rays + the sphere test + emission end to end.  The staff render's disc is
    # ~3.5 pixels wider per side (an extra anti-aliasing blur): the interior and the centre must agree
    # exactly, the footprint must lie inside the reference's, the total differs by that rim only.
    z = np.load(os.path.join(GOLD, "reference_png_stats.npz"))
    png, blocks = _full_spec_blocks(gpu, "sphere")
    ref = z["sphere"].astype(np.float64)
    assert np.array_equal(blocks[..., 0], blocks[..., 1]) and np.array_equal(blocks[..., 0], blocks[..., 2])
    b, r = blocks[..., 0], ref[..., 0]
    inside = r == 255
    assert inside.sum() >= 30 and np.array_equal(b[inside], r[inside])                    # the same fully lit blocks
    assert not b[r == 0].any()                                                            # nothing where the reference is black
    cy, cx = (np.indices(b.shape) * b).sum(axis=(1, 2)) / b.sum()
    ry, rx = (np.indices(r.shape) * r).sum(axis=(1, 2)) / r.sum()
    print("centre of mass (block units) ours (%.3f, %.3f) reference (%.3f, %.3f); mass ratio %.4f" % (cy, cx, ry, rx, b.sum() / r.sum()))
    assert abs(cy - ry) < 0.05 and abs(cx - rx) < 0.05
    assert 0.88 < b.sum() / r.sum() < 0.93                                                # the rim of the wider staff disc
