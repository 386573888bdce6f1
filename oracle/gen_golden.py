"""Golden-vector generator.  TEST INFRASTRUCTURE; runs ONLY in the authoring container.

Drives oracle/_ref/libptref.so -- the reference's own src/intersections.h, src/scene.cpp,
src/utilities.cpp and vendored glm compiled in place from /root/reference by oracle/Makefile --
plus rocThrust's minstd/uniform_real_distribution (oracle/rng_thrust_harness.cpp), and writes
small fixtures under tests/golden/.  The fixtures are DATA (inputs + expected outputs); no
reference source travels.  tests/test_golden.py checks the CPU oracle (oracle/pt_oracle.cpp)
against them bit for bit, here and on the GPU box.

    python oracle/gen_golden.py
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"
sys.path.insert(0, HERE)
from oracle import GEOM_DTYPE, MATERIAL_DTYPE, CAMERA_DTYPE  # noqa: E402  (dtypes only)


def load_ref(name="libptref.so"):
    L = C.CDLL(os.path.join(HERE, "_ref", name))
    L.ref_utilhash.restype = C.c_uint32
    L.ref_utilhash.argtypes = [C.c_uint32]
    for n in ("ref_box", "ref_sphere"):
        getattr(L, n).restype = C.c_float
        getattr(L, n).argtypes = [C.c_void_p] * 4 + [C.POINTER(C.c_int)]
    L.ref_dot.restype = C.c_float
    L.ref_length.restype = C.c_float
    L.ref_dot.argtypes = [C.c_void_p] * 2
    L.ref_length.argtypes = [C.c_void_p]
    L.ref_normalize.argtypes = [C.c_void_p] * 2
    L.ref_cross.argtypes = [C.c_void_p] * 3
    L.ref_reflect.argtypes = [C.c_void_p] * 3
    L.ref_refract.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
    L.ref_mulmv.argtypes = [C.c_void_p] * 3
    L.ref_point_on_ray.argtypes = [C.c_void_p, C.c_float, C.c_void_p]
    L.ref_build_transform.argtypes = [C.c_void_p] * 6
    L.ref_scene_load.restype = C.c_void_p
    L.ref_scene_load.argtypes = [C.c_char_p]
    for n in ("ref_scene_num_geoms", "ref_scene_num_materials", "ref_scene_iterations", "ref_scene_depth",
              "ref_scene_image_len"):
        getattr(L, n).argtypes = [C.c_void_p]
        getattr(L, n).restype = C.c_int
    for n in ("ref_scene_copy_geoms", "ref_scene_copy_materials", "ref_scene_copy_camera"):
        getattr(L, n).argtypes = [C.c_void_p, C.c_void_p]
    L.ref_scene_image_name.argtypes = [C.c_void_p]
    L.ref_scene_image_name.restype = C.c_char_p
    return L


def p(a):
    return a.ctypes.data_as(C.c_void_p)


def ref_geom(L, gtype, mat, t, r, s):
    g = np.zeros(1, GEOM_DTYPE)
    g["type"], g["materialid"] = gtype, mat
    t, r, s = (np.array(v, np.float32) for v in (t, r, s))
    g["translation"], g["rotation"], g["scale"] = t, r, s
    xf, inv, it = (np.empty(16, np.float32) for _ in range(3))
    L.ref_build_transform(p(t), p(r), p(s), p(xf), p(inv), p(it))
    g["transform"], g["inverseTransform"], g["invTranspose"] = xf, inv, it
    return g


def ref_scene(L, path):
    h = L.ref_scene_load(path.encode())
    ng, nm = L.ref_scene_num_geoms(h), L.ref_scene_num_materials(h)
    geoms = np.zeros(ng, GEOM_DTYPE)
    mats = np.zeros(nm, MATERIAL_DTYPE)
    cam = np.zeros(1, CAMERA_DTYPE)
    L.ref_scene_copy_geoms(h, p(geoms))
    L.ref_scene_copy_materials(h, p(mats))
    L.ref_scene_copy_camera(h, p(cam))
    meta = dict(iterations=L.ref_scene_iterations(h), depth=L.ref_scene_depth(h),
                image_len=L.ref_scene_image_len(h), image_name=L.ref_scene_image_name(h).decode())
    return geoms, mats, cam, meta


def gen_rays(rng, geom, n):
    """Ray mix for one geom: far origins aimed near the object, origins inside, random, and
    degenerate (axis-parallel directions with exact zeros, origins on slab planes)."""
    t = geom["translation"][0].astype(np.float64)
    s = np.abs(geom["scale"][0]).astype(np.float64)
    rays = np.zeros((n, 6), np.float32)
    for i in range(n):
        kind = i % 8
        if kind in (0, 1, 2):      # outside, aimed at a point in/near the object
            o = rng.uniform(-12, 12, 3)
            tgt = t + rng.uniform(-0.7, 0.7, 3) * s
            d = tgt - o
        elif kind == 3:            # origin inside the object's bounding box
            o = t + rng.uniform(-0.45, 0.45, 3) * np.minimum(s, 50)
            d = rng.normal(size=3)
        elif kind == 4:            # fully random (mostly misses)
            o = rng.uniform(-12, 12, 3)
            d = rng.normal(size=3)
        elif kind == 5:            # camera-like origin, unnormalised direction
            o = np.array([0.0, 5.0, 10.5]) + rng.normal(size=3) * 0.01
            d = (t + rng.uniform(-0.6, 0.6, 3) * s - o) * rng.uniform(0.1, 7.0)
        elif kind == 6:            # axis-parallel direction with exact zeros (inf/NaN slab paths)
            o = t + rng.uniform(-0.7, 0.7, 3) * s
            ax = rng.integers(0, 3)
            o[ax] = t[ax] + rng.choice([-1, 1]) * (s[ax] * 0.5 + rng.uniform(0.5, 6))
            d = np.zeros(3)
            d[ax] = -np.sign(o[ax] - t[ax]) * rng.uniform(0.2, 3) * rng.choice([1, 1, 1, -1])
            if rng.random() < 0.3:
                d[(ax + 1) % 3] = rng.choice([0.0, -0.0])
        else:                      # origin exactly on a slab plane of an axis-aligned unit box
            o = t + rng.uniform(-0.5, 0.5, 3) * s
            ax = rng.integers(0, 3)
            o[ax] = t[ax] + 0.5 * s[ax] * rng.choice([-1, 1])
            d = rng.normal(size=3)
            if rng.random() < 0.5:
                d[ax] = 0.0
        if kind not in (5,):
            nd = np.linalg.norm(d)
            if nd > 0 and rng.random() < 0.8:
                d = d / nd
        rays[i, :3] = o
        rays[i, 3:] = d
    # make the first lanes normalised IN FLOAT (what the pipeline actually feeds)
    return rays


def run_isect(L, geom, rays):
    n = len(rays)
    t = np.empty(n, np.float32)
    P = np.full((n, 3), -7.0, np.float32)
    N = np.full((n, 3), -7.0, np.float32)
    O = np.full(n, 1, np.int32)
    f = L.ref_sphere if int(geom["type"][0]) == 0 else L.ref_box
    g = np.ascontiguousarray(geom)
    for i in range(n):
        o = C.c_int(1)
        r = np.ascontiguousarray(rays[i])
        t[i] = f(p(g), p(r), p(P[i]), p(N[i]), C.byref(o))
        O[i] = o.value
    return t, P, N, O


def main():
    if not os.path.isdir(REF):
        sys.exit("gen_golden.py needs /root/reference (authoring container only)")
    subprocess.check_call(["make", "-s", "-C", HERE, "reflib"])    # (the libraries alone: no product build needed)
    os.makedirs(GOLD, exist_ok=True)
    L = load_ref()
    Ld = load_ref("libptref_dpow.so")
    rng = np.random.default_rng(20151003)
    prov = {"generator": "oracle/gen_golden.py", "reference": "CIS565-Fall-2015/Project3-CUDA-Path-Tracer @ /root/reference",
            "ref_build": "g++ -O2 -ffp-contract=off, sources compiled in place (oracle/Makefile target ref)"}

    # ---- a6 utilhash --------------------------------------------------------
    xs = np.concatenate([np.array([0, 1, 2, 0x80000000, 0x80000001, 0xFFFFFFFF, 0x7FFFFFFF, 921599], np.uint32),
                         rng.integers(0, 2**32, 4088, dtype=np.uint64).astype(np.uint32)])
    hs = np.array([L.ref_utilhash(int(x)) for x in xs], np.uint32)
    np.savez_compressed(os.path.join(GOLD, "utilhash.npz"), x=xs, h=hs)

    # ---- a19 / a8 / a9 glm ops ------------------------------------------------
    n = 2048
    A = rng.normal(size=(n, 3)).astype(np.float32) * rng.choice([1e-3, 1, 1, 1, 30], size=(n, 1)).astype(np.float32)
    B = rng.normal(size=(n, 3)).astype(np.float32)
    Bn = (B / np.linalg.norm(B, axis=1, keepdims=True)).astype(np.float32)
    An = (A / np.linalg.norm(A, axis=1, keepdims=True)).astype(np.float32)
    eta = rng.choice(np.array([1 / 1.5, 1.5, 1 / 1.33, 1.33, 1.0, 2.4], np.float32), size=n)
    Mx = rng.normal(size=(n, 16)).astype(np.float32)
    V4 = np.concatenate([rng.normal(size=(n, 3)).astype(np.float32) * 5,
                         rng.choice(np.array([0.0, 1.0], np.float32), size=(n, 1))], axis=1).astype(np.float32)
    tt = rng.uniform(-2, 30, n).astype(np.float32)
    rays = np.concatenate([A, B], axis=1).astype(np.float32)
    out = {k: np.empty((n, 3), np.float32) for k in ("normalize", "cross", "reflect", "refract", "mulmv", "point_on_ray")}
    dots = np.empty(n, np.float32)
    lens = np.empty(n, np.float32)
    for i in range(n):
        a, b, an, bn = (np.ascontiguousarray(v[i]) for v in (A, B, An, Bn))
        L.ref_normalize(p(a), p(out["normalize"][i]))
        L.ref_cross(p(a), p(b), p(out["cross"][i]))
        dots[i] = L.ref_dot(p(a), p(b))
        lens[i] = L.ref_length(p(a))
        L.ref_reflect(p(an), p(bn), p(out["reflect"][i]))
        L.ref_refract(p(an), p(bn), C.c_float(float(eta[i])), p(out["refract"][i]))
        L.ref_mulmv(p(np.ascontiguousarray(Mx[i])), p(np.ascontiguousarray(V4[i])), p(out["mulmv"][i]))
        L.ref_point_on_ray(p(np.ascontiguousarray(rays[i])), C.c_float(float(tt[i])), p(out["point_on_ray"][i]))
    np.savez_compressed(os.path.join(GOLD, "glm_ops.npz"), A=A, B=B, An=An, Bn=Bn, eta=eta, M=Mx, V4=V4, t=tt,
                        dot=dots, length=lens, **out)

    # ---- TRS / inverse / inverseTranspose (utilities.cpp:65-72, scene.cpp:82-85) -----
    n = 512
    T = rng.uniform(-10, 10, (n, 3)).astype(np.float32)
    R = rng.uniform(-180, 180, (n, 3)).astype(np.float32)
    S = rng.uniform(0.01, 10, (n, 3)).astype(np.float32)
    R[:64] = rng.choice(np.array([0, 90, -90, 180, 45], np.float32), size=(64, 3))
    XF, INV, IT = (np.empty((n, 16), np.float32) for _ in range(3))
    for i in range(n):
        L.ref_build_transform(p(np.ascontiguousarray(T[i])), p(np.ascontiguousarray(R[i])),
                              p(np.ascontiguousarray(S[i])), p(XF[i]), p(INV[i]), p(IT[i]))
    np.savez_compressed(os.path.join(GOLD, "transforms.npz"), T=T, R=R, S=S, transform=XF, inverse=INV, invTranspose=IT)

    # ---- scene loader dumps (scene.cpp) -------------------------------------------
    scenes = {}
    for name in ("cornell", "sphere"):
        g, m, c, meta = ref_scene(L, f"{REF}/scenes/{name}.txt")
        g2, m2, c2, meta2 = ref_scene(L, os.path.join(ROOT, "scenes", f"{name}.txt"))
        assert g.tobytes() == g2.tobytes() and m.tobytes() == m2.tobytes() and c.tobytes() == c2.tobytes() \
            and meta == meta2, f"authored scenes/{name}.txt differs from the reference's under the reference loader"
        scenes[name] = (g, m, c, meta)
        np.savez_compressed(os.path.join(GOLD, f"scene_{name}.npz"), geoms=g.view(np.uint8), materials=m.view(np.uint8),
                            camera=c.view(np.uint8), meta=json.dumps(meta))
    # authored scenes (C4, C5) through the reference loader as well
    for name in ("cornell_glass", "spheres64", "rotated"):
        g, m, c, meta = ref_scene(L, os.path.join(ROOT, "scenes", f"{name}.txt"))
        np.savez_compressed(os.path.join(GOLD, f"scene_{name}.npz"), geoms=g.view(np.uint8), materials=m.view(np.uint8),
                            camera=c.view(np.uint8), meta=json.dumps(meta))

    # ---- a10 / a11 intersections ----------------------------------------------------
    cornell_geoms = scenes["cornell"][0]
    geoms = [cornell_geoms[i:i + 1] for i in range(len(cornell_geoms))]
    geoms.append(ref_geom(L, 0, 0, (1, 2, 3), (30, 45, 60), (1, 2, 3)))          # ellipsoid
    geoms.append(ref_geom(L, 1, 0, (-2, 3, 1), (20, -35, 70), (2, 0.5, 3)))      # rotated box
    geoms.append(ref_geom(L, 1, 0, (0, 0, 0), (0, 0, 0), (1, 1, 1)))             # unit box (exact slab planes)
    geoms.append(ref_geom(L, 0, 0, (0, 0, 0), (0, 0, 0), (3, 3, 3)))             # sphere.txt sphere
    geoms.append(ref_geom(L, 0, 0, (2.5, 6, -2), (0, 0, 0), (0.8, 0.8, 0.8)))    # small sphere
    G = np.concatenate(geoms)
    n = 1536
    all_rays, all_t, all_p, all_n, all_o = [], [], [], [], []
    ndiff = 0
    for gi in range(len(G)):
        g = G[gi:gi + 1]
        rays = gen_rays(rng, g, n)
        t, P, N, O = run_isect(L, g, rays)
        if int(g["type"][0]) == 0:
            t2, P2, N2, O2 = run_isect(Ld, g, rays)
            ndiff += int(np.sum((t.view(np.uint32) != t2.view(np.uint32)) | np.any(P.view(np.uint32) != P2.view(np.uint32), axis=1)))
        all_rays.append(rays); all_t.append(t); all_p.append(P); all_n.append(N); all_o.append(O)
    prov["sphere_vectors_changed_by_host_double_pow"] = ndiff
    prov["sphere_vectors_total"] = int(sum(len(r) for r, g in zip(all_rays, G) if int(g["type"]) == 0))
    np.savez_compressed(os.path.join(GOLD, "intersections.npz"), geoms=G.view(np.uint8), rays=np.stack(all_rays),
                        t=np.stack(all_t), p=np.stack(all_p), n=np.stack(all_n), outside=np.stack(all_o))

    # ---- a7 RNG via rocThrust ---------------------------------------------------------
    exe = os.path.join(HERE, "_ref", "rng_thrust")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-host-only", "-O2", "-ffp-contract=off", "-w",
                           os.path.join(HERE, "rng_thrust_harness.cpp"), "-o", exe])
    seeds = np.concatenate([np.array([0, 1, 2147483646, 2147483647, 2147483648, 0xFFFFFFFF, 1948881963, 854875198,
                                      3235770143], np.uint32),
                            rng.integers(0, 2**32, 503, dtype=np.uint64).astype(np.uint32)])
    nd = 16
    inp = f"{len(seeds)} {nd}\n" + "\n".join(str(int(s)) for s in seeds) + "\n"
    txt = subprocess.run([exe], input=inp.encode(), stdout=subprocess.PIPE, check=True).stdout.decode()
    u = np.zeros((len(seeds), nd), np.uint32)
    for i, line in enumerate(txt.strip().split("\n")):
        tok = line.split()
        assert int(tok[0]) == int(seeds[i])
        u[i] = [int(t, 16) for t in tok[1:]]
    np.savez_compressed(os.path.join(GOLD, "rng_thrust.npz"), seeds=seeds, u01_bits=u)
    prov["thrust"] = "rocThrust shipped with ROCm 7.2.0 (/opt/rocm/include/thrust), host side, hipcc --offload-host-only"

    # ---- end-to-end statistics of the reference's staff-solution renders (img/REFERENCE_*.png):
    #      50x50 box-downsampled 8-bit means (data derived from the PNGs, not the PNGs themselves)
    from PIL import Image
    stats = {}
    for name in ("cornell", "sphere"):
        im = np.asarray(Image.open(f"{REF}/img/REFERENCE_{name}.5000samp.png").convert("RGB")).astype(np.float64)
        assert im.shape == (800, 800, 3)
        stats[name] = im.reshape(50, 16, 50, 16, 3).mean(axis=(1, 3)).astype(np.float32)
        stats[name + "_mean"] = im.reshape(-1, 3).mean(axis=0).astype(np.float32)
    np.savez_compressed(os.path.join(GOLD, "reference_png_stats.npz"), **stats)

    json.dump(prov, open(os.path.join(GOLD, "PROVENANCE.json"), "w"), indent=1)
    print("golden fixtures written to", GOLD)
    print(json.dumps(prov, indent=1))


if __name__ == "__main__":
    main()
