"""The camera-ray culling tables (pixel rectangles, their union, per-row hull spans: pt_init's build_camera_cull, host code) against the
oracle's own camera rays and intersection tests, on the CPU: a hit from a pixel the tables would skip is a violation.  The GPU suite
repeats this on >= 10^4 cases with every pixel on the device (tests/test_gpu_camera_cull.py); the cases here are the first of the same
generator, at the frame shapes a Python loop can afford -- among them the one that showed, in round 3, that the tables have to
contain the REFERENCE's hits and not the primitive's projection: a 100 : 1 ellipsoid seen from 20 000 object units through a 1.5-degree
lens, whose fp32 sphere test (src/intersections.h:101-143) reports hits 60 columns off the ellipsoid's projection."""
import numpy as np

from test_gpu_camera_cull import _case


def test_culling_tables_hold_every_hit_of_the_oracle(pt, oracle):
    rng = np.random.default_rng(20261004)                     # (the GPU sweep's seed: case 21 is the ellipsoid)
    mats = np.zeros(1, oracle.MATERIAL_DTYPE)
    hits = culled = checked = 0
    for k in range(40):
        cam, geoms = _case(oracle, rng)
        W, H = (int(v) for v in cam["resolution"][0])
        if W * H > 14000:
            continue
        rects, scene, spans = pt.camera_cull_tables(cam.view(pt.CAMERA_DTYPE), geoms.view(pt.GEOM_DTYPE))
        o, d, _, pix = oracle.Renderer(cam, geoms, mats, 1).dump_paths(1, 0)
        assert len(pix) == W * H
        for i in range(len(pix)):
            x, y = int(pix[i]) % W, int(pix[i]) // W
            in_scene = scene[0] <= x <= scene[2] and scene[1] <= y <= scene[3]
            for gi in range(len(geoms)):
                reach = in_scene and rects[gi][0] <= x <= rects[gi][2] and rects[gi][1] <= y <= rects[gi][3] \
                    and spans[y, gi, 0] <= x <= spans[y, gi, 1]
                t = oracle.intersect(geoms[gi:gi + 1], tuple(o[i]) + tuple(d[i]))[0]
                assert reach or not t > 0, "case %d: pixel (%d, %d) hits primitive %d, which the tables skip there" % (k, x, y, gi)
                hits += bool(t > 0)
                culled += not reach
        checked += 1
    assert checked >= 10 and hits > 10000 and culled > 100000      # (not vacuous)
