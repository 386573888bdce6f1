#!/usr/bin/env python3
"""Renders the reference's shipped configurations at full spec (800x800, 5000 spp, depth 8: scenes/cornell.txt:53-56,
scenes/sphere.txt:13-16) on the HIP path and prints how the 50x50 block means of the 8-bit image compare with the staff
renders' (tests/golden/reference_png_stats.npz): the numbers behind the tolerances of
tests/test_gpu_parity.py::test_full_spec_renders_against_the_reference_pngs."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pt = ge.load_package()
z = np.load(os.path.join(ROOT, "tests", "golden", "reference_png_stats.npz"))
for name in ("cornell", "sphere"):
    sc = pt.Scene(os.path.join(ROOT, "scenes", name + ".txt"))
    W, H = (int(v) for v in sc.camera["resolution"][0])
    n = sc.iterations
    pt.pathtraceFree()
    pt.pathtraceInit(sc, max_batch=64, pipeline_depth=2)
    t0 = time.time()
    it = 1
    while it <= n:
        k = min(64, n - it + 1)
        pt.pathtrace_batch(None, 0, it, k)
        it += k
    img = pt.readback(W * H).reshape(H, W, 3) / np.float32(n)
    dt = time.time() - t0
    pt.pathtraceFree()
    png = (np.clip(img, 0, 1) * np.float32(255)).astype(np.uint8)[:, ::-1].astype(np.float64)
    blocks = png.reshape(50, 16, 50, 16, 3).mean(axis=(1, 3))
    ref = z[name].astype(np.float64)
    print("==", name, "%dx%d %d spp depth %d in %.2f s" % (W, H, n, sc.traceDepth, dt))
    print("global mean", png.reshape(-1, 3).mean(0), "ref", z[name + "_mean"], "ratio", png.reshape(-1, 3).mean(0) / z[name + "_mean"])
    d = blocks - ref
    # flat blocks: the reference varies by < 3 levels over the 3x3 neighbourhood
    pad = np.pad(ref, ((1, 1), (1, 1), (0, 0)), mode="edge")
    nb = np.stack([pad[i:i + 50, j:j + 50] for i in range(3) for j in range(3)])
    flat = (nb.max(0) - nb.min(0)).max(-1) < 3.0
    lit = ref.max(-1) > 8
    print("blocks flat", flat.sum(), "flat&lit", (flat & lit).sum())
    for msk, label in ((flat & lit, "flat lit"), (flat & ~lit, "flat dark"), (~flat, "edges")):
        if msk.sum() == 0: continue
        a = np.abs(d[msk]); r = np.abs(d[msk] / np.maximum(ref[msk], 1))
        print("  %-10s n=%4d abs: mean %.2f p95 %.2f max %.2f | rel: mean %.3f p95 %.3f max %.3f | signed mean %.2f" % (label, msk.sum(), a.mean(), np.percentile(a, 95), a.max(), r.mean(), np.percentile(r, 95), r.max(), d[msk].mean()))
    if name == "cornell":
        regions = {"back wall": (slice(20, 30), slice(20, 30)), "left wall": (slice(20, 30), slice(3, 8)), "right wall": (slice(20, 30), slice(42, 47)),
                   "floor": (slice(42, 47), slice(20, 30)), "ceiling": (slice(3, 6), slice(8, 15)), "sphere": (slice(26, 32), slice(17, 23)),
                   "light": (slice(0, 2), slice(22, 28))}
        for k, (ys, xs) in regions.items():
            print("  region %-10s ours %s ref %s ratio %s" % (k, np.round(blocks[ys, xs].mean((0, 1)), 2), np.round(ref[ys, xs].mean((0, 1)), 2), np.round(blocks[ys, xs].mean((0, 1)) / np.maximum(ref[ys, xs].mean((0, 1)), 1e-9), 3)))
    else:
        on = ref[:, :, 0] > 128
        print("  sphere blocks fully lit in ref:", on.sum(), "ours:", (blocks[:, :, 0] > 128).sum(), "centroid ref", np.argwhere(on).mean(0), "ours", np.argwhere(blocks[:, :, 0] > 128).mean(0))
        print("  sum of levels ours %.1f ref %.1f" % (blocks[:, :, 0].sum(), ref[:, :, 0].sum()))
    np.save(os.path.join(ROOT, "gpurun_out", "r02", "blocks_%s.npy" % name), blocks)
