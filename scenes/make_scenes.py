"""Authors scenes/*.txt (run once; outputs are committed).

cornell.txt / sphere.txt carry the same numeric content as the reference's example scenes
(own layout and comments; tests/test_golden.py checks that the reference loader produced
byte-identical structs from both).  cornell_glass.txt (BASELINE config C4) and spheres64.txt
(config C5) are authored by this build as SURVEY.md section 8(d) specifies.  cornell_mesh.txt / mesh_small.txt use the
scene format's third object type, "mesh" (README.md:236), with OBJ files generated here under scenes/models/.
"""
import math
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def material(i, name, rgb, specex, specrgb, refl, refr, ior, emit):
    return (f"// material {i}: {name}\nMATERIAL {i}\nRGB {rgb}\nSPECEX {specex}\nSPECRGB {specrgb}\n"
            f"REFL {refl}\nREFR {refr}\nREFRIOR {ior}\nEMITTANCE {emit}\n\n")


def camera(res, fovy, iters, depth, name, eye="0.0 5 10.5", view="0 0 -1", up="0 1 0"):
    return (f"// pinhole camera (FOVY is the vertical half-angle in degrees)\nCAMERA\nRES {res}\nFOVY {fovy}\n"
            f"ITERATIONS {iters}\nDEPTH {depth}\nFILE {name}\nEYE {eye}\nVIEW {view}\nUP {up}\n\n")


def obj(i, name, kind, mat, t, r, s):
    return f"// object {i}: {name}\nOBJECT {i}\n{kind}\nmaterial {mat}\nTRANS {t}\nROTAT {r}\nSCALE {s}\n\n"


CORNELL_MATS = [
    ("ceiling light (emissive)", "1 1 1", 0, "0 0 0", 0, 0, 0, 5),
    ("diffuse white", ".98 .98 .98", 0, "0 0 0", 0, 0, 0, 0),
    ("diffuse red", ".85 .35 .35", 0, "0 0 0", 0, 0, 0, 0),
    ("diffuse green", ".35 .85 .35", 0, "0 0 0", 0, 0, 0, 0),
    ("specular white", ".98 .98 .98", 0, ".98 .98 .98", 1, 0, 0, 0),
]
CORNELL_OBJS = [
    ("ceiling light", "cube", 0, "0 10 0", "0 0 0", "3 .3 3"),
    ("floor", "cube", 1, "0 0 0", "0 0 0", "10 .01 10"),
    ("ceiling", "cube", 1, "0 10 0", "0 0 90", ".01 10 10"),
    ("back wall", "cube", 1, "0 5 -5", "0 90 0", ".01 10 10"),
    ("left wall", "cube", 2, "-5 5 0", "0 0 0", ".01 10 10"),
    ("right wall", "cube", 3, "5 5 0", "0 0 0", ".01 10 10"),
    ("sphere", "sphere", 4, "-1 4 -1", "0 0 0", "3 3 3"),
]
GLASS = ("glass (Schlick Fresnel)", ".98 .98 .98", 0, ".98 .98 .98", 0, 1, 1.5, 0)


def cornell(mats, name, title, closed=False):
    s = f"// {title}\n\n"
    for i, m in enumerate(mats):
        s += material(i, *m)
    # closed: the eye moves inside the box (z = 4.5; the box spans z in [-5, 5]) and a front wall closes it behind the camera
    s += camera("800 800", 45, 5000, 8, name, eye="0.0 5 4.5") if closed else camera("800 800", 45, 5000, 8, name)
    objs = CORNELL_OBJS + ([("front wall (behind the camera)", "cube", 1, "0 5 5", "0 90 0", ".01 10 10")] if closed else [])
    for i, o in enumerate(objs):
        s += obj(i, *o)
    return s


def spheres64(seed=565):
    x = seed % 2147483647 or 1

    def u01():
        nonlocal x
        x = (x * 48271) % 2147483647
        return (x - 1) / 2147483648.0

    mats = CORNELL_MATS + [GLASS]
    s = f"// 64-sphere stress scene (BASELINE config C5; authored by this build; minstd seed {seed})\n\n"
    for i, m in enumerate(mats):
        s += material(i, *m)
    s += camera("800 800", 45, 5000, 8, "spheres64")
    for i, o in enumerate(CORNELL_OBJS[:6]):
        s += obj(i, *o)
    cyc = [1, 2, 3, 4, 5]  # diffuse white / red / green / mirror mix / glass
    k = 6
    for ix in range(4):
        for iy in range(4):
            for iz in range(4):
                cx = -4 + (ix + 0.5) * 2 + (u01() - 0.5) * 0.6
                cy = 1 + (iy + 0.5) * 2 + (u01() - 0.5) * 0.6
                cz = -4 + (iz + 0.5) * 2 + (u01() - 0.5) * 0.6
                d = 2 * (0.3 + 0.3 * u01())
                s += obj(k, f"sphere {k - 6}", "sphere", cyc[(k - 6) % 5], f"{cx:.4f} {cy:.4f} {cz:.4f}",
                         "0 0 0", f"{d:.4f} {d:.4f} {d:.4f}")
                k += 1
    return s


# ---- scenes OUTSIDE the Cornell shape (round 5: what do the kernels' scene-keyed heuristics cost elsewhere?) ------------------------------
def lattice_scene(title, name, kind, n, seed, rotate):
    """Cornell's room and light with n^3 primitives of `kind` on a jittered lattice inside it: sizes, (for cubes) orientations and the
    material cycle from a minstd stream, like spheres64.txt"""
    x = seed % 2147483647 or 1

    def u01():
        nonlocal x
        x = (x * 48271) % 2147483647
        return (x - 1) / 2147483648.0

    mats = CORNELL_MATS + [GLASS]
    s = f"// {title} (authored by this build; minstd seed {seed})\n\n"
    for i, m in enumerate(mats):
        s += material(i, *m)
    s += camera("800 800", 45, 5000, 8, name)
    for i, o in enumerate(CORNELL_OBJS[:6]):
        s += obj(i, *o)
    cyc = [1, 2, 3, 4, 5]
    k = 6
    step = 8.0 / n
    for ix in range(n):
        for iy in range(n):
            for iz in range(n):
                cx = -4 + (ix + 0.5) * step + (u01() - 0.5) * 0.3 * step
                cy = 1 + (iy + 0.5) * step + (u01() - 0.5) * 0.3 * step
                cz = -4 + (iz + 0.5) * step + (u01() - 0.5) * 0.3 * step
                d = step * (0.3 + 0.3 * u01())
                rot = "%.2f %.2f %.2f" % (360 * u01(), 360 * u01(), 360 * u01()) if rotate else "0 0 0"
                sc = f"{d:.4f} {d * (0.5 + u01()):.4f} {d:.4f}" if rotate else f"{d:.4f} {d:.4f} {d:.4f}"
                s += obj(k, f"{kind} {k - 6}", kind, cyc[(k - 6) % 5], f"{cx:.4f} {cy:.4f} {cz:.4f}", rot, sc)
                k += 1
    return s


def room_tilted():
    """A room none of whose walls is axis-aligned (every wall rotated about two axes, the whole room turned against the camera), a tilted
    light, three spheres and two rotated cubes: nothing of Cornell's box shape for the wall certificates to lean on"""
    mats = CORNELL_MATS + [GLASS]
    s = "// tilted room: no axis-aligned wall, oblique camera (authored by this build)\n\n"
    for i, m in enumerate(mats):
        s += material(i, *m)
    s += camera("800 800", 40, 5000, 8, "room_tilted", eye="1.5 5.5 11", view="-0.12 -0.05 -1", up="0.05 1 0")
    objs = [
        ("light, tilted", "cube", 0, "0.3 9.4 -0.5", "6 20 -4", "3.5 .3 3"),
        ("floor", "cube", 1, "0 0 0", "3 17 -2", "12 .05 12"),
        ("ceiling", "cube", 1, "0 10 0", "-4 10 2", "12 .05 12"),
        ("back wall", "cube", 1, "0 5 -5.5", "85 3 12", "12 .05 11"),
        ("left wall", "cube", 2, "-5.5 5 0", "0 15 88", "11 .05 12"),
        ("right wall", "cube", 3, "5.5 5 0", "7 -12 93", "11 .05 12"),
        ("mirror-mix sphere", "sphere", 4, "-1.5 3.5 -1", "0 0 0", "3 3 3"),
        ("glass ellipsoid", "sphere", 5, "2 2.5 1", "30 45 60", "2 3 2"),
        ("small white sphere", "sphere", 1, "0.5 6.5 -2", "0 0 0", "1.5 1.5 1.5"),
        ("red block", "cube", 2, "-3 1.5 1.5", "10 35 5", "1.5 3 1.5"),
        ("green block", "cube", 3, "3 1 -2.5", "0 60 20", "2 2 2"),
    ]
    for i, o in enumerate(objs):
        s += obj(i, *o)
    return s


# ---- OBJ models (generated: no network, no third-party assets) ---------------------------------------------
def icosphere(subdiv, radius=0.5):
    """Unit-diameter icosphere: 20 * 4^subdiv counter-clockwise (outward) triangles."""
    t = (1 + 5 ** 0.5) / 2
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
         (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7),
         (9, 8, 1)]
    norm = lambda p: tuple(radius * c / math.sqrt(sum(x * x for x in p)) for c in p)
    v = [norm(p) for p in v]
    for _ in range(subdiv):
        mid, nf = {}, []

        def m(a, b):
            k = (min(a, b), max(a, b))
            if k not in mid:
                v.append(norm(tuple((v[a][c] + v[b][c]) / 2 for c in range(3))))
                mid[k] = len(v) - 1
            return mid[k]
        for a, b, c in f:
            ab, bc, ca = m(a, b), m(b, c), m(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return v, f


def torus(nu, nv, R=0.35, r=0.15):
    """Torus around the y axis as quads (the loaders fan them into triangles), outward counter-clockwise."""
    v, f = [], []
    for i in range(nu):
        a = 2 * math.pi * i / nu
        for j in range(nv):
            b = 2 * math.pi * j / nv
            v.append(((R + r * math.cos(b)) * math.cos(a), r * math.sin(b), -(R + r * math.cos(b)) * math.sin(a)))
    for i in range(nu):
        for j in range(nv):
            a, b = i * nv + j, ((i + 1) % nu) * nv + j
            c, d = ((i + 1) % nu) * nv + (j + 1) % nv, i * nv + (j + 1) % nv
            f.append((a, b, c, d))
    return v, f


def write_obj(name, title, v, f, vn=None, usemtl=None):
    """vn: one normal per vertex (faces then read `i//i`); usemtl: {first face index: scene material} -- `usemtl <k>` statements"""
    os.makedirs(os.path.join(HERE, "models"), exist_ok=True)
    with open(os.path.join(HERE, "models", name), "w") as fp:
        fp.write(f"# {title} (generated by scenes/make_scenes.py)\n")
        for p in v:
            fp.write("v %.6f %.6f %.6f\n" % p)
        for p in vn or []:
            fp.write("vn %.6f %.6f %.6f\n" % p)
        for k, face in enumerate(f):
            if usemtl and k in usemtl:
                fp.write("usemtl %s\n" % usemtl[k])
            fp.write("f " + " ".join(("%d//%d" % (i + 1, i + 1)) if vn else str(i + 1) for i in face) + "\n")


def unit_cube():
    """[-0.5, 0.5]^3 as six outward counter-clockwise quads: -x, +x, -y, +y, -z, +z"""
    v = [(x, y, z) for x in (-0.5, 0.5) for y in (-0.5, 0.5) for z in (-0.5, 0.5)]          # index = 4 x + 2 y + z
    f = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    return v, f


def cornell_mesh():
    mats = CORNELL_MATS + [GLASS]
    s = "// Cornell box whose sphere is a 1280-triangle icosphere, plus a glass torus (object type \"mesh\", README.md:236)\n\n"
    for i, m in enumerate(mats):
        s += material(i, *m)
    s += camera("800 800", 45, 5000, 8, "cornell_mesh")
    for i, o in enumerate(CORNELL_OBJS[:6]):
        s += obj(i, *o)
    s += obj(6, "icosphere, where Cornell's sphere is", "mesh models/icosphere3.obj", 4, "-1 4 -1", "0 0 0", "3 3 3")
    s += obj(7, "glass torus", "mesh models/torus.obj", 5, "2.2 2.2 1.5", "50 0 25", "4 4 4")
    return s


def mesh_small():
    s = "// small mesh scene for the CPU oracle: an 80-triangle icosphere and a 128-triangle torus under a light\n\n"
    for i, m in enumerate(CORNELL_MATS + [GLASS]):
        s += material(i, *m)
    s += camera("96 96", 45, 16, 6, "mesh_small")
    s += obj(0, "ceiling light", "cube", 0, "0 10 0", "0 0 0", "6 .3 6")
    s += obj(1, "floor", "cube", 1, "0 0 0", "0 0 0", "10 .01 10")
    s += obj(2, "back wall", "cube", 2, "0 5 -5", "0 90 0", ".01 10 10")
    s += obj(3, "icosphere (mirror mix)", "mesh models/icosphere1.obj", 4, "-1.5 3 0", "10 20 30", "4 3 4")
    s += obj(4, "glass torus", "mesh models/torus_small.obj", 5, "2 3.5 1", "60 10 0", "5 5 5")
    return s


def mesh_attributes():
    """vertex normals and per-face materials (README.md:112-116: `vn`, `usemtl`): a smooth-shaded icosphere, a flat one beside it, and a
    cube mesh whose six faces take six materials (one of them emissive, one a mirror mix)"""
    s = "// mesh attributes: an icosphere with vertex normals (smooth), the same icosphere without (flat), a cube with a material per face\n\n"
    for i, m in enumerate(CORNELL_MATS + [GLASS]):
        s += material(i, *m)
    s += camera("96 96", 45, 16, 6, "mesh_attributes")
    s += obj(0, "ceiling light", "cube", 0, "0 10 0", "0 0 0", "6 .3 6")
    s += obj(1, "floor", "cube", 1, "0 0 0", "0 0 0", "10 .01 10")
    s += obj(2, "back wall", "cube", 2, "0 5 -5", "0 90 0", ".01 10 10")
    s += obj(3, "icosphere with vertex normals (mirror mix)", "mesh models/icosphere1_vn.obj", 4, "-2.5 3 0", "10 20 30", "4 3 4")
    s += obj(4, "the same icosphere, flat", "mesh models/icosphere1.obj", 1, "2.5 6.5 -1", "10 20 30", "3 3 3")
    s += obj(5, "cube mesh, a material per face", "mesh models/cube_materials.obj", 1, "2.2 2.5 1", "20 35 10", "3 3 3")
    return s


def main():
    w = lambda n, s: open(os.path.join(HERE, n), "w").write(s)
    v1, f1 = icosphere(1)
    write_obj("icosphere1_vn.obj", "icosphere, 1 subdivision: 80 triangles, diameter 1, with vertex normals (the radial directions)", v1, f1,
              vn=[tuple(c / 0.5 for c in p) for p in v1])
    write_obj("cube_materials.obj", "unit cube, six quads, a scene material per face (usemtl <k>; the last face keeps the object's)",
              *unit_cube(), usemtl={0: 2, 1: 3, 2: 4, 3: 0, 4: 5, 5: "object"})
    w("mesh_attributes.txt", mesh_attributes())
    write_obj("icosphere3.obj", "icosphere, 3 subdivisions: 1280 triangles, diameter 1", *icosphere(3))
    write_obj("icosphere1.obj", "icosphere, 1 subdivision: 80 triangles, diameter 1", *icosphere(1))
    write_obj("torus.obj", "torus, 48 x 24 quads = 2304 triangles", *torus(48, 24))
    write_obj("torus_small.obj", "torus, 8 x 8 quads = 128 triangles", *torus(8, 8))
    w("cornell_mesh.txt", cornell_mesh())
    w("mesh_small.txt", mesh_small())
    w("cornell.txt", cornell(CORNELL_MATS, "cornell", "Cornell box, same values as the reference's scenes/cornell.txt"))
    w("cornell_closed.txt", cornell(CORNELL_MATS, "cornell_closed",
                                    "Cornell box CLOSED by a front wall, camera inside: no light can escape (the reference's analysis, README.md:284-293; "
                                    "authored by this build)", closed=True))
    g = list(CORNELL_MATS)
    g[4] = GLASS
    w("cornell_glass.txt", cornell(g, "cornell_glass", "Cornell box with a glass sphere (BASELINE config C4; authored by this build)"))
    s = "// single emissive sphere, same values as the reference's scenes/sphere.txt\n\n"
    s += material(0, "emissive white", "1 1 1", 0, "0 0 0", 0, 0, 0, 5)
    s += camera("800 800", 45, 5000, 8, "sphere")
    s += obj(0, "sphere", "sphere", 0, "0 0 0", "0 0 0", "3 3 3")
    w("sphere.txt", s)
    w("spheres64.txt", spheres64())
    w("cubes64.txt", lattice_scene("64 rotated cubes in Cornell's room", "cubes64", "cube", 4, 7001, True))
    w("spheres512.txt", lattice_scene("512 spheres in Cornell's room", "spheres512", "sphere", 8, 7002, False))
    w("room_tilted.txt", room_tilted())


if __name__ == "__main__":
    main()
