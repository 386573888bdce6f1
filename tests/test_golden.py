"""CPU oracle vs golden vectors (no GPU).

tests/golden/*.npz were produced by oracle/gen_golden.py from the REFERENCE's own sources
compiled in place (oracle/_ref); tests/golden/survey_kats.json holds the known-answer values
recorded in SURVEY.md section 8(a).  Integer work and every fp32 primitive that the reference
defines without libm (everything except sin/cos in calculateRandomDirectionInHemisphere) must
match BIT FOR BIT.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLD, SCENES


def bits(a):
    a = np.ascontiguousarray(a, np.float32)
    b = a.view(np.uint32).copy()
    b[np.isnan(a)] = 0x7FC00000  # any NaN == any NaN
    return b


def assert_bits(got, want, what):
    g, w = bits(got), bits(want)
    bad = np.argwhere(g != w)
    assert bad.size == 0, f"{what}: {len(bad)} mismatching lanes, first {bad[:3].tolist()} " \
                          f"got {np.asarray(got).reshape(-1)[:0]}"


@pytest.fixture(scope="module")
def kats():
    return json.load(open(os.path.join(GOLD, "survey_kats.json")))


def test_utilhash_reference_vectors(oracle):
    z = np.load(os.path.join(GOLD, "utilhash.npz"))
    got = np.array([oracle.utilhash(int(x)) for x in z["x"]], np.uint32)
    assert np.array_equal(got, z["h"])


def test_utilhash_survey_kats(oracle, kats):
    for x, h in kats["utilhash"]:
        assert oracle.utilhash(x) == h


def test_glm_ops_bit_exact(oracle):
    z = np.load(os.path.join(GOLD, "glm_ops.npz"))
    n = len(z["A"])
    got = {k: np.empty((n, 3), np.float32) for k in ("normalize", "reflect", "refract", "mulmv", "point_on_ray")}
    for i in range(n):
        got["normalize"][i] = oracle.normalize(z["A"][i])
        got["reflect"][i] = oracle.reflect(z["An"][i], z["Bn"][i])
        got["refract"][i] = oracle.refract(z["An"][i], z["Bn"][i], float(z["eta"][i]))
        got["mulmv"][i] = oracle.mulmv(z["M"][i], z["V4"][i])
        got["point_on_ray"][i] = oracle.point_on_ray(np.concatenate([z["A"][i], z["B"][i]]), float(z["t"][i]))
    for k, v in got.items():
        assert_bits(v, z[k], k)


def test_transform_builder_bit_exact(oracle):
    z = np.load(os.path.join(GOLD, "transforms.npz"))
    for i in range(len(z["T"])):
        xf, inv, it = oracle.build_transform(z["T"][i], z["R"][i], z["S"][i])
        assert_bits(xf, z["transform"][i], f"transform[{i}]")
        assert_bits(inv, z["inverse"][i], f"inverse[{i}]")
        assert_bits(it, z["invTranspose"][i], f"invTranspose[{i}]")


@pytest.mark.parametrize("name", ["cornell", "sphere", "cornell_glass", "spheres64", "rotated"])
def test_scene_loader_matches_reference_loader(oracle, name):
    z = np.load(os.path.join(GOLD, f"scene_{name}.npz"))
    sc = oracle.Scene(os.path.join(SCENES, f"{name}.txt"))
    meta = json.loads(str(z["meta"]))
    assert sc.geoms.tobytes() == z["geoms"].tobytes()
    assert sc.materials.tobytes() == z["materials"].tobytes()
    assert sc.camera.tobytes() == z["camera"].tobytes()
    assert (sc.iterations, sc.depth, sc.image_name) == (meta["iterations"], meta["depth"], meta["image_name"])
    assert meta["image_len"] == int(sc.camera["resolution"][0][0]) * int(sc.camera["resolution"][0][1])


def test_fov_override_1280x720(oracle, kats):
    sc = oracle.Scene(os.path.join(SCENES, "cornell.txt"))
    sc.set_resolution(1280, 720)
    fov = sc.camera["fov"][0]
    assert abs(float(fov[0]) - kats["fov_1280x720"][0]) < 5e-6 and float(fov[1]) == 45.0
    sc.set_resolution(1920, 1080)
    assert abs(float(sc.camera["fov"][0][0]) - kats["fov_1280x720"][0]) < 5e-6


def test_intersections_bit_exact(oracle):
    z = np.load(os.path.join(GOLD, "intersections.npz"))
    G = np.frombuffer(z["geoms"].tobytes(), oracle.GEOM_DTYPE)
    nhit = nmiss = 0
    for gi in range(len(G)):
        g = G[gi:gi + 1]
        rays = z["rays"][gi]
        n = len(rays)
        t = np.empty(n, np.float32)
        P = np.empty((n, 3), np.float32)
        N = np.empty((n, 3), np.float32)
        O = np.empty(n, np.int32)
        for i in range(n):
            t[i], P[i], N[i], O[i] = oracle.intersect(g, rays[i])
        assert_bits(t, z["t"][gi], f"t geom {gi}")
        assert_bits(P, z["p"][gi], f"p geom {gi}")   # includes 'untouched on miss' (-7 sentinel)
        assert_bits(N, z["n"][gi], f"n geom {gi}")
        assert np.array_equal(O, z["outside"][gi]), f"outside geom {gi}"
        nhit += int(np.sum(z["t"][gi] > 0))
        nmiss += int(np.sum(z["t"][gi] < 0))
    assert nhit > 5000 and nmiss > 3000  # the vector mix exercises both outcomes


def _geom(oracle, d):
    return oracle.make_geom(d["type"], 0, d["trans"], d["rot"], d["scale"])


@pytest.mark.parametrize("which", ["box", "sphere", "ellipsoid"])
def test_intersection_survey_kats(oracle, kats, which):
    g = _geom(oracle, kats[which]["geom"])
    for c in kats[which]["cases"]:
        d = oracle.normalize(c["dir_unnormalized"])
        t, p, n, o = oracle.intersect(g, list(c["origin"]) + list(d))
        assert t == pytest.approx(c["t"], rel=2e-7, abs=1e-9)
        if "p" in c:
            assert np.allclose(p, c["p"], rtol=3e-7, atol=1e-12)
        if "n" in c:
            assert np.allclose(n, c["n"], rtol=3e-7, atol=1e-12)
        if "outside" in c:
            assert o == c["outside"]


def test_rng_matches_thrust(oracle):
    z = np.load(os.path.join(GOLD, "rng_thrust.npz"))
    for sd, want in zip(z["seeds"], z["u01_bits"]):
        u, _ = oracle.rng_stream_from_seed(int(sd), len(want))
        assert np.array_equal(u.view(np.uint32), want), f"seed {sd}"


def test_rng_survey_kats(oracle, kats):
    for k in kats["rng"]:
        if "seed" in k:
            assert oracle.seed(k["iter"], k["index"], k["depth"]) == k["seed"]
        u, _ = oracle.rng_stream(k["iter"], k["index"], k["depth"], len(k["u01"]))
        assert np.allclose(u, k["u01"], rtol=2e-7, atol=0)


def test_u01_can_reach_one_but_not_exceed(oracle):
    # state m-1 -> float(2^31-3) rounds to 2^31 -> u01 == 1.0f exactly (SURVEY S1 note)
    u, s = oracle.rng_stream_from_seed(1, 4)
    assert s[0] == 48271 and 0.0 <= u.min() and u.max() <= 1.0


def test_hemisphere_survey_kats(oracle, kats):
    # sin/cos are libm-dependent in the reference; the oracle's fixed polynomial must agree
    # with the recorded values to a few ulp of the unit-length result.
    for k in kats["hemisphere"]:
        n = k["n"] if "n" in k else oracle.normalize(k["n_unnormalized"])
        got = oracle.hemisphere_seeded(n, k["iter"], k["index"], k["depth"])
        assert np.allclose(got, k["out"], rtol=0, atol=4e-7), (got, k["out"])


def test_reflect_refract_survey_kats(oracle, kats):
    k = kats["reflect_refract"]
    I = oracle.normalize(k["I_unnormalized"])
    assert np.allclose(oracle.reflect(I, k["N"]), k["reflect"], rtol=2e-7)
    assert np.allclose(oracle.refract(I, k["N"], np.float32(k["eta"])), k["refract"], rtol=2e-7)
    # glm::refract returns NaN*0 when k < 0: callers must test k first
    out = oracle.refract(oracle.normalize([1, -0.05, 0]), [0, 1, 0], 1.5)
    assert np.all(np.isnan(out))


def test_sincos_polynomial_accuracy(oracle):
    xs = np.linspace(0, 2 * np.pi, 20001).astype(np.float32)
    worst = 0.0
    for x in xs:
        s, c = oracle.sincos(float(x))
        es = abs(float(s) - np.sin(np.float64(x)))
        ec = abs(float(c) - np.cos(np.float64(x)))
        worst = max(worst, es, ec)
    # absolute error below 1.2e-7 (= 1 ulp at 1.0): same class as CUDA sinf/cosf (2 ulp)
    assert worst < 1.2e-7, worst


def test_to_rgba8_conversion(oracle):
    # pathtrace.cu:58-66 : clamp((int)(pix / iter * 255.0), 0, 255), w = 0
    img = np.array([[0.0, 0.5, 1.0], [2.0, 7.0, -1.0], [3.0, 2.999, 1e-9]], np.float32)
    out = oracle.to_rgba8(img, 3)
    want = [[0, int(np.float32(0.5) / np.float32(3) * 255.0), 85], [170, 255, 0],
            [255, int(float(np.float32(2.999) / np.float32(3)) * 255.0), 0]]
    assert out[:, :3].tolist() == want and np.all(out[:, 3] == 0)


def test_triangle_test_against_glm_intersect_ray_triangle(oracle):
    """The oracle's two-sided triangle test (meshes, README.md:112-116) next to the reference's vendored
    glm::intersectRayTriangle (glm/gtx/intersect.inl:36-72; vectors by oracle/gen_golden_mesh.py): where glm reports a hit the
    restatement reports the same front-side hit with the same (u, v, t) bit for bit; where glm does not, it reports a miss or a
    hit on the back side -- the only difference between the two (glm culls back faces, `a < epsilon`)."""
    z = np.load(os.path.join(GOLD, "triangles.npz"))
    n = len(z["hit"])
    front_hits = back_hits = 0
    for i in range(n):
        hit, tuv, front = oracle.mesh_triangle(z["origin"][i], z["direction"][i], z["v"][i, 0], z["v"][i, 1], z["v"][i, 2])
        if z["hit"][i]:
            assert hit and front, i
            got = np.array([tuv[1], tuv[2], tuv[0]], np.float32)       # glm's baryPosition is (u, v, t)
            assert np.array_equal(bits(got), bits(z["bary"][i])), (i, got, z["bary"][i])
            front_hits += 1
        else:
            assert (not hit) or (not front), i
            back_hits += int(hit)
    assert front_hits == int(z["hit"].sum()) and front_hits > 500 and back_hits > 500
