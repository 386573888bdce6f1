"""The two clusters of spheres a sphere-heavy scene's survivors are binned by (pt_api.hip: build_sphere_clusters -- host code, k_bounce's
CLUSTER variants), on the CPU: the table holds every sphere once, cluster 0 first; each cluster's box holds its spheres; and a half-line
that misses a box in exact arithmetic misses, in the oracle (src/intersections.h:101-143), every sphere behind it -- the certificate's
statement, here with double-precision slabs against the reference's own fp32 test.  The device's fp32 form of the certificate is swept on
the GPU (tests/test_gpu_parity.py::test_sphere_cluster_boxes_never_reject_a_hit, 2^28 rays per scene)."""
import os

import numpy as np

from conftest import SCENES


def _field(oracle, rng, n):
    geoms = [oracle.make_geom(1, 1, (0, 0, 0), (0, 0, 0), (12, 0.01, 12)), oracle.make_geom(1, 1, (0, 10, 0), (0, 0, 0), (12, 0.01, 12)),
             oracle.make_geom(1, 0, (0, 9.8, 0), (0, 0, 0), (3, 0.3, 3))]
    for _ in range(n):
        d = rng.uniform(0.3, 1.4)
        s = (d, d, d) if rng.random() < 0.7 else tuple(d * rng.uniform(0.7, 1.3, 3))
        geoms.append(oracle.make_geom(0, 1, tuple(rng.uniform(-4, 4, 3) + np.array([0, 5, 0])), tuple(rng.uniform(-180, 180, 3)), s))
    order = rng.permutation(len(geoms))
    return np.concatenate([geoms[i] for i in order])


def _misses_box(o, d, lo, hi):
    """exact-arithmetic (double) slab test of the half-line o + t d, t >= 0, against [lo, hi]: True = certainly outside"""
    with np.errstate(divide="ignore", invalid="ignore"):
        t1, t2 = (lo - o) / d, (hi - o) / d
    par = d == 0
    if np.any(par & ((o < lo) | (o > hi))):
        return True
    tn = np.where(par, -np.inf, np.minimum(t1, t2)).max()
    tf = np.where(par, np.inf, np.maximum(t1, t2)).min()
    return bool(tf < max(tn, 0.0))


def test_clusters_partition_the_spheres_and_their_boxes_hold_them(pt, oracle):
    rng = np.random.default_rng(77)
    scenes = [oracle.Scene(os.path.join(SCENES, "spheres64.txt")).geoms] + [_field(oracle, rng, n) for n in (5, 6, 9, 17, 40)]
    certified = 0
    for geoms in scenes:
        info, table = pt.sphere_clusters(geoms.view(pt.GEOM_DTYPE))
        sph = [i for i in range(len(geoms)) if int(geoms["type"][i]) == 0]
        n0 = info["n0"]
        assert n0 % 2 == 0 and 2 <= n0 < len(table) and info["omax"] > 0          # (pt_init pads the table's END to an even count itself)
        c0, c1 = list(table[:n0]), list(table[n0:])
        assert sorted(set(c0) | set(c1)) == sph and not set(c0) & set(c1)                 # a partition ...
        assert len(set(c0)) >= len(c0) - 1 and len(set(c1)) >= len(c1) - 1                 # ... with at most one padding copy per cluster
        assert abs(len(set(c0)) - len(set(c1))) <= 1                                       # split at the median
        for g, members in enumerate((c0, c1)):
            lo, hi = info["boxes"][g][:3].astype(np.float64), info["boxes"][g][3:].astype(np.float64)
            for i in set(members):
                c = geoms["translation"][i].astype(np.float64)
                r = 0.5 * float(np.abs(geoms["scale"][i]).max())
                assert np.all(c - r > lo) and np.all(c + r < hi), (g, i)
        # the certificate's statement against the reference's own test: rays off the spheres' surfaces and across the scene
        for _ in range(1500):
            i = int(rng.choice(sph))
            c = geoms["translation"][i].astype(np.float64)
            if rng.random() < 0.6:
                n = rng.normal(size=3)
                n /= np.linalg.norm(n)
                o = c + n * (0.5 * float(np.abs(geoms["scale"][i]).max()) + 1e-3)
            else:
                o = rng.uniform(-5, 5, 3) + np.array([0, 5, 0])
            d = rng.normal(size=3)
            d /= np.linalg.norm(d)
            if rng.random() < 0.2:
                d[int(rng.integers(0, 3))] = 0.0
                d /= np.linalg.norm(d)
            o32, d32 = o.astype(np.float32), d.astype(np.float32)
            if float(np.abs(o32).sum()) > info["omax"]:
                continue
            for g, members in enumerate((c0, c1)):
                lo, hi = info["boxes"][g][:3].astype(np.float64), info["boxes"][g][3:].astype(np.float64)
                if _misses_box(o32.astype(np.float64), d32.astype(np.float64), lo, hi):
                    certified += 1
                    for j in set(members):
                        t = oracle.intersect(geoms[j:j + 1], tuple(o32) + tuple(d32))[0]
                        assert not t > 0, "cluster %d certified as missed, sphere %d is hit (t = %r)" % (g, j, t)
    assert certified > 2000          # (not vacuous)
