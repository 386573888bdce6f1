"""CPU tests of the oracle's per-iteration pipeline (spec S0-S9) -- the thing the GPU
path is compared with: sharding invariance, live-path statistics the survey recorded, camera
convention, and agreement with the reference's staff-solution PNG statistics."""
import os

import numpy as np
import pytest

from conftest import GOLD, SCENES


def _renderer(oracle, name, res, depth):
    sc = oracle.Scene(os.path.join(SCENES, name))
    sc.set_resolution(*res)
    return oracle.Renderer(sc.camera, sc.geoms, sc.materials, depth), sc


def test_config_c1_sphere_every_path_ends_at_bounce_one(oracle):
    # BASELINE config C1: scenes/sphere.txt, 400x400, 1 spp, 4 bounces, single-thread CPU path
    R, _ = _renderer(oracle, "sphere.txt", (400, 400), 4)
    img = np.zeros(400 * 400 * 3, np.float32)
    c = R.iterate(1, img)
    assert c.live[1] == 160000 and c.live[2] == 0                   # every path ends at bounce 1
    assert c.lightHits + c.misses == 160000 and c.lightHits > 0
    lit = img.reshape(400, 400, 3)
    assert np.all((lit == 0) | (lit == 5.0))                        # emittance 5, white: 1*1*5
    ys, xs = np.nonzero(lit[:, :, 0])
    assert abs(xs.mean() - 199.5) < 1.0                              # centred horizontally
    # vertical position equals the staff-solution render's (img/REFERENCE_sphere.5000samp.png)
    z = np.load(os.path.join(GOLD, "reference_png_stats.npz"))["sphere"][:, :, 0]
    ry, rx = np.nonzero(z > 1)
    wgt = z[ry, rx]
    assert abs((ys.mean() + 0.5) / 400 - ((ry * wgt).sum() / wgt.sum() + 0.5) / 50) < 0.005
    assert abs((xs.mean() + 0.5) / 400 - (1 - ((rx * wgt).sum() / wgt.sum() + 0.5) / 50)) < 0.005   # PNG is X-mirrored


def test_cornell_live_fractions_match_survey(oracle):
    # live-path fraction entering bounce 1..8 = 1.000 0.817 0.564 0.435 0.347 0.280 0.228 0.187
    R, _ = _renderer(oracle, "cornell.txt", (200, 200), 8)
    img = np.zeros(200 * 200 * 3, np.float32)
    live = np.zeros(9)
    n = 8
    for it in range(1, n + 1):
        c = R.iterate(it, img)
        live += np.array(c.live[:9])
    frac = live[1:] / (200 * 200 * n)
    assert np.allclose(frac, [1.000, 0.817, 0.564, 0.435, 0.347, 0.280, 0.228, 0.187], atol=0.004)


def test_row_sharding_is_bit_exact(oracle):
    R, _ = _renderer(oracle, "cornell.txt", (64, 37), 8)
    full = np.zeros(64 * 37 * 3, np.float32)
    parts = [np.zeros_like(full) for _ in range(4)]
    tot = 0
    for it in (1, 2):
        cf = R.iterate(it, full)
        for r in range(4):
            cp = R.iterate(it, parts[r], r, 4)
            tot += cp.live[1]
    assert tot == 2 * 64 * 37 and cf.live[1] == 64 * 37
    acc = parts[0] + parts[1] + parts[2] + parts[3]
    assert np.array_equal(acc.view(np.uint32), full.view(np.uint32))


def test_camera_convention_x0_is_camera_right_y0_is_top(oracle):
    R, _ = _renderer(oracle, "cornell.txt", (100, 100), 8)
    r = R.camera_ray(1, 0)                 # pixel (0, 0)
    assert r[3] > 0 and r[4] > 0 and r[5] < 0      # +x (camera right), +y (up), looking down -z
    r = R.camera_ray(1, 99 + 99 * 100)     # pixel (99, 99)
    assert r[3] < 0 and r[4] < 0
    assert np.allclose(r[:3], [0, 5, 10.5])
    assert abs(np.linalg.norm(r[3:]) - 1) < 1e-6


def test_depth_limit_and_dump_consistency(oracle):
    R, _ = _renderer(oracle, "cornell.txt", (48, 32), 3)
    img = np.zeros(48 * 32 * 3, np.float32)
    c = R.iterate(5, img)
    o, d, col, pix = R.dump_paths(5, 3)
    assert len(pix) == c.depthKilled                              # survivors of the last bounce contribute black
    assert c.lightHits + c.misses + c.depthKilled == 48 * 32
    assert np.all(np.diff(pix) > 0)                                # pixel order
    assert np.all(col <= 1.0) and np.all(col >= 0)


def test_statistics_against_reference_png(oracle):
    # img/REFERENCE_cornell.5000samp.png (staff solution) is the reference's only end-to-end golden.
    z = np.load(os.path.join(GOLD, "reference_png_stats.npz"))
    R, _ = _renderer(oracle, "cornell.txt", (100, 100), 8)
    img = np.zeros(100 * 100 * 3, np.float32)
    n = 48
    for it in range(1, n + 1):
        R.iterate(it, img)
    png = (np.clip(img.reshape(100, 100, 3) / np.float32(n), 0, 1) * np.float32(255)).astype(np.uint8)[:, ::-1]
    mean = png.reshape(-1, 3).astype(np.float64).mean(axis=0)
    assert np.all(np.abs(mean / z["cornell_mean"] - 1) < 0.04), (mean, z["cornell_mean"])
    # the red wall is on the LEFT of the saved image (X mirror, src/main.cpp:58)
    blocks = png.astype(np.float64).reshape(50, 2, 50, 2, 3).mean(axis=(1, 3))
    assert blocks[20:30, 3:8, 0].mean() > 1.5 * blocks[20:30, 3:8, 1].mean()
    assert blocks[20:30, 42:47, 1].mean() > 1.5 * blocks[20:30, 42:47, 0].mean()


def test_glass_scene_refracts(oracle):
    R, _ = _renderer(oracle, "cornell_glass.txt", (64, 64), 16)
    o, d, col, pix = R.dump_paths(1, 1)
    o0, d0, _, pix0 = R.dump_paths(1, 0)
    # some first-bounce survivors start INSIDE the sphere (transmitted rays): centre (-1,4,-1), radius 1.5
    inside = np.linalg.norm(o - np.array([-1, 4, -1], np.float32), axis=1) < 1.5
    assert inside.sum() > 20
    assert not np.any(np.isnan(d)) and np.allclose(np.linalg.norm(d, axis=1), 1, atol=1e-5)
