"""CPU checks of the README extras in the oracle (imperfect specular README.md:171-185, depth of field
:100-101, direct lighting :107-108 -- named by the reference, implemented nowhere in it, so build-defined and pinned
only by their own properties)."""
import os

import numpy as np

from conftest import SCENES


def test_pow_poly_accuracy(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(0, 1, 4000), 2.0 ** rng.uniform(-31, 0, 4000)]).astype(np.float32)
    e = np.concatenate([1 / (rng.integers(0, 2000, 4000) + 1), rng.uniform(1e-4, 1, 4000)]).astype(np.float32)
    got = np.array([L.orc_pow(float(a), float(b)) for a, b in zip(x, e)], np.float32)
    want = np.power(x.astype(np.float64), e.astype(np.float64))
    assert np.max(np.abs(got - want) / want) < 3e-6
    assert L.orc_pow(0.0, 0.5) == 0.0 and L.orc_pow(1.0, 0.5) == 1.0 and L.orc_pow(0.25, 1.0) == 0.25


def _render(oracle, sc, depth, iters, mats=None, **extras):
    ren = oracle.Renderer(sc.camera, sc.geoms, sc.materials if mats is None else mats, depth)
    if extras:
        ren.set_extras(**extras)
    W, H = (int(v) for v in sc.camera["resolution"][0])
    img = np.zeros(W * H * 3, np.float32)
    for it in iters:
        ren.iterate(it, img)
    return img.reshape(H, W, 3) / np.float32(len(iters)), ren


def test_extras_off_is_the_pinned_renderer(oracle):
    sc = oracle.Scene(os.path.join(SCENES, "cornell.txt"))
    sc.set_resolution(40, 30)
    a, _ = _render(oracle, sc, 5, [1, 2])
    b, _ = _render(oracle, sc, 5, [1, 2], lens_radius=0.0, focal_distance=3.0, direct_lighting=False)
    assert np.array_equal(a, b)


def test_depth_of_field_keeps_the_focal_plane_sharp(oracle):
    sc = oracle.Scene(os.path.join(SCENES, "cornell.txt"))
    sc.set_resolution(64, 64)
    ren = oracle.Renderer(sc.camera, sc.geoms, sc.materials, 4)
    pin = np.array([ren.camera_ray(1, i) for i in (0, 2000, 4095)])
    ren.set_extras(lens_radius=0.5, focal_distance=15.5)             # focus on the back wall (z = -5 from z = 10.5)
    dof = np.array([ren.camera_ray(1, i) for i in (0, 2000, 4095)])
    eye = sc.camera["position"][0]
    # origins lie on the lens disc in the plane through the eye spanned by right and up ...
    assert np.all(np.abs(dof[:, 2] - eye[2]) < 1e-6) and np.all(np.linalg.norm(dof[:, :3] - eye, axis=1) <= 0.5 + 1e-6)
    assert np.any(np.linalg.norm(dof[:, :3] - eye, axis=1) > 0.05)
    # ... and every lens ray passes through the point its pinhole ray reaches on the focal plane
    for p, d in zip(pin, dof):
        t = 15.5 / -p[5]
        focus = p[:3] + t * p[3:]
        s = (focus[2] - d[2]) / d[5]
        assert np.allclose(d[:3] + s * d[3:], focus, atol=2e-4)


def test_direct_lighting_brightens_shallow_renders(oracle):
    sc = oracle.Scene(os.path.join(SCENES, "cornell.txt"))
    sc.set_resolution(48, 48)
    its = list(range(1, 9))
    plain, _ = _render(oracle, sc, 2, its)
    direct, ren = _render(oracle, sc, 2, its, direct_lighting=True)
    deep, _ = _render(oracle, sc, 8, its)
    # two bounces reach the light only by luck; a final ray aimed at it collects most of what eight bounces find
    assert direct.mean() > 1.4 * plain.mean()
    assert 0.7 * deep.mean() < direct.mean() < 1.3 * deep.mean()
    c = ren.iterate(9, np.zeros(48 * 48 * 3, np.float32))
    assert c.live[3] > 0 and c.live[4] == 0                             # depth 2 + the one collecting bounce


def test_imperfect_specular_blurs_the_mirror(oracle):
    sc = oracle.Scene(os.path.join(SCENES, "cornell.txt"))
    sc.set_resolution(64, 64)
    mats = sc.materials.copy()
    assert mats["hasReflective"][4] > 0 and mats["specExponent"][4] == 0
    mirror, _ = _render(oracle, sc, 6, [1, 2, 3, 4])
    mats["specExponent"][4] = 5.0
    glossy, _ = _render(oracle, sc, 6, [1, 2, 3, 4], mats=mats)
    assert not np.array_equal(mirror, glossy)
    assert abs(glossy.mean() / mirror.mean() - 1) < 0.1                 # same energy, different directions
    # the lobe narrows towards the mirror as the exponent grows: directions after the first bounce off the sphere
    ren0 = oracle.Renderer(sc.camera, sc.geoms, sc.materials, 6)
    o0, d0, c0, p0 = ren0.dump_paths(1, 1)
    spread = []
    for n in (5.0, 500.0):
        mats["specExponent"][4] = n
        ren = oracle.Renderer(sc.camera, sc.geoms, mats, 6)
        o, d, c, p = ren.dump_paths(1, 1)
        assert np.array_equal(p, p0)
        changed = np.any(d != d0, axis=1)                               # the specular half of the sphere's hits
        assert changed.sum() > 10
        spread.append(np.mean(1 - np.sum(d[changed] * d0[changed], axis=1)))
    assert spread[1] < 0.1 * spread[0]
