"""Authors scenes/*.txt (run once; outputs are committed).

cornell.txt / sphere.txt carry the same numeric content as the reference's example scenes
(own layout and comments; tests/test_golden.py checks that the reference loader produced
byte-identical structs from both).  cornell_glass.txt (BASELINE config C4) and spheres64.txt
(config C5) are authored by this build as SURVEY.md section 8(d) specifies.
"""
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


def cornell(mats, name, title):
    s = f"// {title}\n\n"
    for i, m in enumerate(mats):
        s += material(i, *m)
    s += camera("800 800", 45, 5000, 8, name)
    for i, o in enumerate(CORNELL_OBJS):
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


def main():
    w = lambda n, s: open(os.path.join(HERE, n), "w").write(s)
    w("cornell.txt", cornell(CORNELL_MATS, "cornell", "Cornell box, same values as the reference's scenes/cornell.txt"))
    g = list(CORNELL_MATS)
    g[4] = GLASS
    w("cornell_glass.txt", cornell(g, "cornell_glass", "Cornell box with a glass sphere (BASELINE config C4; authored by this build)"))
    s = "// single emissive sphere, same values as the reference's scenes/sphere.txt\n\n"
    s += material(0, "emissive white", "1 1 1", 0, "0 0 0", 0, 0, 0, 5)
    s += camera("800 800", 45, 5000, 8, "sphere")
    s += obj(0, "sphere", "sphere", 0, "0 0 0", "0 0 0", "3 3 3")
    w("sphere.txt", s)
    w("spheres64.txt", spheres64())


if __name__ == "__main__":
    main()
