"""Golden vectors for the triangle test.  TEST INFRASTRUCTURE; runs ONLY in the authoring container.

README.md:116 points mesh loaders to `glm::intersectRayTriangle`; the reference vendors it (glm 0.9.6.3,
external/include/glm/gtx/intersect.inl:36-72) but never calls it.  This script drives that function, compiled in place from
/root/reference by oracle/Makefile (oracle/_ref/libptref.so, ref_harness.cpp: ref_intersect_ray_triangle), over seeded
(ray, triangle) pairs and writes inputs + its outputs to tests/golden/triangles.npz.  tests/test_golden.py checks the oracle's
two-sided restatement against them: identical wherever glm reports a hit, and a back-side hit or a miss wherever it does not.

    python oracle/gen_golden_mesh.py
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")


def main():
    if not os.path.isdir("/root/reference"):
        sys.exit("gen_golden_mesh.py needs /root/reference (authoring container only)")
    subprocess.check_call(["make", "-s", "-C", HERE, "reflib"])
    L = C.CDLL(os.path.join(HERE, "_ref", "libptref.so"))
    L.ref_intersect_ray_triangle.argtypes = [C.c_void_p] * 6
    rng = np.random.default_rng(20151116)
    n = 6144
    V = (rng.normal(size=(n, 3, 3)) * rng.choice([0.05, 1, 1, 4], size=(n, 1, 1))).astype(np.float32)
    O = (rng.normal(size=(n, 3)) * 6).astype(np.float32)
    D = np.empty((n, 3), np.float32)
    for i in range(n):
        v = V[i].astype(np.float64)
        kind = i % 8
        if kind < 4:        # aimed at a point of the triangle's plane, inside or just outside it
            a, b = rng.uniform(-0.15, 1.15, 2)
            if kind == 1:
                b = 1.0 - a + rng.normal() * 1e-6          # on the hypotenuse
            if kind == 2:
                a = rng.normal() * 1e-6                    # on an edge through v0
            tgt = v[0] + a * (v[1] - v[0]) + b * (v[2] - v[0])
            d = tgt - O[i]
        elif kind == 4:     # through a vertex, exactly representable
            d = v[rng.integers(0, 3)] - O[i]
        elif kind == 5:     # (almost) parallel to the plane
            e = v[1] - v[0] + rng.uniform(-1, 1) * (v[2] - v[0])
            d = e + np.cross(v[1] - v[0], v[2] - v[0]) * rng.choice([0.0, 1e-7, 1e-5])
        elif kind == 6:     # pointing away
            d = O[i] - v.mean(axis=0)
        else:
            d = rng.normal(size=3)
        nd = np.linalg.norm(d)
        D[i] = (d / nd if nd > 0 and rng.random() < 0.8 else d).astype(np.float32)
    V[-8:, 2] = V[-8:, 1]                                    # degenerate triangles
    hit = np.zeros(n, np.int32)
    bary = np.full((n, 3), -7.0, np.float32)                 # glm leaves baryPosition components it did not reach
    for i in range(n):
        args = [np.ascontiguousarray(a) for a in (O[i], D[i], V[i, 0], V[i, 1], V[i, 2])]
        hit[i] = L.ref_intersect_ray_triangle(*[a.ctypes.data_as(C.c_void_p) for a in args], bary[i].ctypes.data_as(C.c_void_p))
    np.savez_compressed(os.path.join(GOLD, "triangles.npz"), origin=O, direction=D, v=V, hit=hit, bary=bary)
    print("triangles.npz: %d pairs, %d hits by glm" % (n, int(hit.sum())))


if __name__ == "__main__":
    main()
