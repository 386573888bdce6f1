"""Triangle meshes, CPU side (no GPU): the two OBJ / scene loaders (host library and oracle) agree; the oracle's mesh test
behaves like the analytic primitives it approximates; the hierarchy the product builds (pt_test_mesh_bvh, host code) has the
properties the traversal's equivalence with the brute-force rule rests on -- and a CPU walk of it, following the kernel's
rule in fp32, returns the oracle's brute-force result."""
import os

import numpy as np
import pytest

from conftest import SCENES

f32 = np.float32


def test_loaders_agree_on_mesh_scenes(pt, oracle):
    for name in ("mesh_small.txt", "cornell_mesh.txt"):
        a = pt.Scene(os.path.join(SCENES, name))
        b = oracle.Scene(os.path.join(SCENES, name))
        assert a.geoms.tobytes() == b.geoms.tobytes() and a.materials.tobytes() == b.materials.tobytes()
        assert sorted(a.meshes) == sorted(b.meshes) and len(a.meshes) == 2
        for g in a.meshes:
            assert a.geoms["type"][g] == 2 and a.meshes[g].tobytes() == b.meshes[g].tobytes()
    assert pt.Scene(os.path.join(SCENES, "cornell_mesh.txt")).meshes[6].shape == (1280, 9)
    assert pt.Scene(os.path.join(SCENES, "cornell_mesh.txt")).meshes[7].shape == (2304, 9)     # quads fanned into triangles


def test_obj_statements(pt, oracle, tmp_path):
    (tmp_path / "m.obj").write_text(
        "# comment\no thing\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nvt 0 0\n"
        "f 1/1/1 2/1/1 3/1/1 4/1/1\n"          # a quad with texture / normal references: two triangles
        "v 0 0 1\nf -1 1//1 2//1\n"            # negative = relative to the vertices read so far
        "f 1 2 99\nf 1 2\ns off\n")            # out-of-range reference and a two-vertex face: ignored
    (tmp_path / "s.txt").write_text(
        "MATERIAL 0\nRGB 1 1 1\nSPECEX 0\nSPECRGB 0 0 0\nREFL 0\nREFR 0\nREFRIOR 0\nEMITTANCE 1\n\n"
        "CAMERA\nRES 8 8\nFOVY 45\nITERATIONS 1\nDEPTH 2\nFILE x\nEYE 0 0 5\nVIEW 0 0 -1\nUP 0 1 0\n\n"
        "OBJECT 0\nmesh m.obj\nmaterial 0\nTRANS 0 0 0\nROTAT 0 0 0\nSCALE 1 1 1\n\n"
        "OBJECT 1\nflubber\nmaterial 0\nTRANS 0 0 0\nROTAT 0 0 0\nSCALE 1 1 1\n\n")
    want = np.array([[0, 0, 0, 1, 0, 0, 1, 1, 0], [0, 0, 0, 1, 1, 0, 0, 1, 0], [0, 0, 1, 0, 0, 0, 1, 0, 0]], f32)
    for mod in (pt, oracle):
        sc = mod.Scene(str(tmp_path / "s.txt"))
        assert list(sc.meshes) == [0] and np.array_equal(sc.meshes[0], want)
        assert list(sc.geoms["type"]) == [2, 0]       # an unknown type line leaves the default type (the reference's behaviour)
    # a mesh object whose file is missing, unreadable or holds no triangle fails the LOAD (it used to render as a unit sphere
    # with the mesh's transform and material)
    (tmp_path / "empty.obj").write_text("# no faces\nv 0 0 0\nv 1 0 0\n")
    for bad in ("missing.obj", "empty.obj"):
        (tmp_path / "bad.txt").write_text((tmp_path / "s.txt").read_text().replace("flubber", "mesh " + bad))
        for mod in (pt, oracle):
            with pytest.raises(IOError):
                mod.Scene(str(tmp_path / "bad.txt"))


def test_loaders_agree_on_mesh_attributes(pt, oracle, tmp_path):
    # `vn` + the third field of a face corner = vertex normals (kept only when every corner of every face names one), `usemtl <k>` = a
    # scene material per face: the product's host loader and the oracle's read the same arrays
    name = os.path.join(SCENES, "mesh_attributes.txt")
    a, b = pt.Scene(name), oracle.Scene(name)
    assert a.geoms.tobytes() == b.geoms.tobytes() and sorted(a.meshes) == sorted(b.meshes) == [3, 4, 5]
    assert sorted(a.mesh_normals) == sorted(b.mesh_normals) == [3] and sorted(a.mesh_materials) == sorted(b.mesh_materials) == [5]
    assert a.mesh_normals[3].shape == (80, 9) and a.mesh_normals[3].tobytes() == b.mesh_normals[3].tobytes()
    # (the model's normals are its vertices' radial directions: vertices on a sphere of diameter 1)
    assert np.allclose(a.mesh_normals[3], 2 * a.meshes[3], atol=2e-6)
    want = np.repeat(np.array([2, 3, 4, 0, 5, -1], np.int32), 2)               # six quads = twelve triangles, `usemtl object` = the object's
    assert np.array_equal(a.mesh_materials[5], want) and np.array_equal(b.mesh_materials[5], want)
    # a face corner without a normal makes the whole mesh flat; a name that is no integer returns to the object's material
    (tmp_path / "m.obj").write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nusemtl 1\nf 1//1 2//1 3//1\nusemtl shiny\nf 1//1 3//1 4\n")
    (tmp_path / "s.txt").write_text(
        "MATERIAL 0\nRGB 1 1 1\nSPECEX 0\nSPECRGB 0 0 0\nREFL 0\nREFR 0\nREFRIOR 0\nEMITTANCE 1\n\n"
        "MATERIAL 1\nRGB 1 0 0\nSPECEX 0\nSPECRGB 0 0 0\nREFL 0\nREFR 0\nREFRIOR 0\nEMITTANCE 0\n\n"
        "CAMERA\nRES 8 8\nFOVY 45\nITERATIONS 1\nDEPTH 2\nFILE x\nEYE 0 0 5\nVIEW 0 0 -1\nUP 0 1 0\n\n"
        "OBJECT 0\nmesh m.obj\nmaterial 0\nTRANS 0 0 0\nROTAT 0 0 0\nSCALE 1 1 1\n\n")
    for mod in (pt, oracle):
        sc = mod.Scene(str(tmp_path / "s.txt"))
        assert sc.mesh_normals == {} and np.array_equal(sc.mesh_materials[0], np.array([1, -1], np.int32))


def test_vertex_normals_of_an_icosphere_are_the_spheres_normals(oracle):
    # The model's vertex normals are the radial directions of its vertices, which lie on the sphere of diameter 1: the barycentric
    # blend of the normals at a hit is then parallel to the same blend of the vertices -- the hit point itself.  So the shading
    # normal of the smooth icosphere must be the normal the SPHERE primitive's formula gives at that object-space point,
    # normalize(invTranspose * (p_obj, 0)) (src/intersections.h:137-140), whatever the (anisotropic) transform -- and the flat
    # icosphere's normal must not (it is constant per face).
    sc = oracle.Scene(os.path.join(SCENES, "mesh_attributes.txt"))
    tris, normals = sc.meshes[3], sc.mesh_normals[3]
    rng = np.random.default_rng(5)
    worst_smooth, worst_flat, hits = 0.0, 0.0, 0
    for scale in ((4, 3, 4), (1, 1, 1), (0.5, 2.0, 1.0)):
        geom = oracle.make_geom(2, 0, (-1.5, 3, 0), (10, 20, 30), scale)
        M = geom["transform"][0].astype(np.float64).reshape(4, 4).T          # (column-major in the struct)
        invT = np.linalg.inv(M[:3, :3]).T
        for _ in range(400):
            o = np.array([-1.5, 3, 0]) + rng.normal(size=3) * 6
            target = np.array([-1.5, 3, 0]) + rng.normal(size=3) * 0.8
            d = (target - o) / np.linalg.norm(target - o)
            ray = tuple(o.astype(f32)) + tuple(d.astype(f32))
            t, p, n, outside, tri = oracle.mesh_intersect(geom, tris, ray, normals=normals)
            tf, pf, nf, of, trif = oracle.mesh_intersect(geom, tris, ray)
            assert np.float32(t).view(np.uint32) == np.float32(tf).view(np.uint32) and tri == trif       # the normals change the shading only
            if t <= 0:
                continue
            assert np.array_equal(p.view(np.uint32), pf.view(np.uint32)) and outside == of
            pobj = np.linalg.inv(M) @ np.append(p.astype(np.float64), 1.0)
            want = invT @ pobj[:3]
            want /= np.linalg.norm(want)
            if not outside:
                want = -want
            worst_smooth = max(worst_smooth, float(np.abs(n - want).max()))
            worst_flat = max(worst_flat, float(np.abs(nf - want).max()))
            hits += 1
    assert hits > 300
    assert worst_smooth < 2e-3, worst_smooth          # (the hit point is offset by 1e-4 along the ray: 2e-4 of the radius at most)
    assert worst_flat > 0.05                           # an 80-triangle icosphere's face normals are up to ~15 degrees off


def test_face_materials_in_the_oracle(oracle):
    # `usemtl`: a face takes the scene material it names.  (i) naming the object's own material for every face changes nothing, bit for
    # bit; (ii) the cube mesh of mesh_attributes.txt has an emissive face (material 0): paths end there as on a light
    sc = oracle.Scene(os.path.join(SCENES, "mesh_attributes.txt"))
    sc.set_resolution(48, 48)

    def render(mats):
        ref = oracle.Renderer(sc.camera, sc.geoms, sc.materials, 5, meshes=sc.meshes, mesh_normals=sc.mesh_normals, mesh_materials=mats)
        img = np.zeros(48 * 48 * 3, f32)
        lights = 0
        for it in (1, 2, 3):
            lights += ref.iterate(it, img).lightHits
        return img, lights
    base, lights_base = render({})
    own = {5: np.full(12, int(sc.geoms["materialid"][5]), np.int32)}
    same, lights_same = render(own)
    assert np.array_equal(base.view(np.uint32), same.view(np.uint32)) and lights_base == lights_same
    with_faces, lights_faces = render(sc.mesh_materials)
    assert lights_faces > lights_base + 50 and not np.array_equal(base, with_faces)


def _cube_mesh():
    c = np.array([[x, y, z] for x in (-.5, .5) for y in (-.5, .5) for z in (-.5, .5)], f32)
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]      # outward counter-clockwise
    tris = []
    for a, b, cc, d in quads:
        tris += [np.concatenate([c[a], c[b], c[cc]]), np.concatenate([c[a], c[cc], c[d]])]
    return np.array(tris, f32)


def test_mesh_of_a_cube_matches_the_box_primitive(oracle):
    tris = _cube_mesh()
    rng = np.random.default_rng(5)
    box = oracle.make_geom(1, 0, (1, 2, 3), (20, 30, 40), (2, 1, 3))
    mesh = box.copy()
    mesh["type"] = 2
    hits = inside = 0
    for i in range(600):
        if i % 3 == 2:
            o = np.array([1, 2, 3]) + rng.uniform(-0.3, 0.3, 3)                # inside
        else:
            o = rng.uniform(-8, 8, 3)
        d = np.array([1, 2, 3]) + rng.uniform(-1.2, 1.2, 3) - o if i % 3 != 2 else rng.normal(size=3)
        d = d / np.linalg.norm(d)
        ray = np.concatenate([o, d]).astype(f32)
        tb, pb, nb, ob = oracle.intersect(box, ray)
        tm, pm, nm, om, tri = oracle.mesh_intersect(mesh, tris, ray)
        assert (tb > 0) == (tm > 0), (i, tb, tm)
        if tb > 0:
            assert abs(tb - tm) < 1e-4 * max(1, tb) and np.allclose(pb, pm, atol=2e-4) and ob == om
            assert np.allclose(nb, nm, atol=1e-4), (nb, nm)
            hits += 1
            inside += 1 - om
    assert hits > 250 and inside > 100


def test_icosphere_mesh_approximates_the_sphere(oracle):
    sc = oracle.Scene(os.path.join(SCENES, "cornell_mesh.txt"))
    tris = sc.meshes[6]
    sph = oracle.make_geom(0, 0, (-1, 4, -1), (0, 0, 0), (3, 3, 3))
    mesh = sph.copy()
    mesh["type"] = 2
    rng = np.random.default_rng(6)
    n = 0
    for _ in range(300):
        o = rng.uniform(-9, 9, 3)
        d = np.array([-1, 4, -1]) + rng.normal(size=3) * 0.6 - o
        ray = np.concatenate([o, d / np.linalg.norm(d)]).astype(f32)
        ts, ps, ns, os_ = oracle.intersect(sph, ray)
        tm, pm, nm, om, tri = oracle.mesh_intersect(mesh, tris, ray)
        if ts > 0 and tm > 0 and -(ray[3:] @ ns) > 0.5:                                # (away from the silhouette)
            assert abs(ts - tm) < 0.03 and os_ == om and nm @ ns > 0.99          # 1280 facets of a radius-1.5 ball
            n += 1
    assert n > 150


def test_hierarchy_invariants(pt):
    sc = pt.Scene(os.path.join(SCENES, "cornell_mesh.txt"))
    cases = [(t, o) for t in sc.meshes.values() for o in (0, 5)] + [(_cube_mesh(), o) for o in range(8)] + [(_cube_mesh()[:1], 3)]
    LEAF = pt.MESH_LEAF
    for tris, octant in cases:
        nt = len(tris)
        recs, nodes, first, need = pt.mesh_bvh(tris, octant)
        assert len(recs) == nt and len(nodes) == max(nt - 1, 1)
        margin = f32(1e-5) * np.abs(tris).max()
        v = tris.reshape(nt, 3, 3)
        neg = np.array([(octant >> a) & 1 for a in range(3)], bool)
        for i, r in enumerate(recs):                          # triangle i, file order: its vertices and the mesh's margin
            assert np.array_equal(r["v0"], v[i][0]) and np.array_equal(r["v1"], v[i][1]) and np.array_equal(r["v2"], v[i][2])
            assert r["margin"] == margin
        box_lo = lambda t: v[t].min(0) - margin                # a triangle's own box, as the kernel derives it
        box_hi = lambda t: v[t].max(0) + margin
        def planes_of(lo, hi):
            # the six halves of a box for this octant: entry planes, exit planes; lo rounded down, hi up (never inwards, and by
            # less than one half-precision step)
            lo16 = lo.astype(np.float16)
            lo16 = np.where(lo16.astype(f32) > lo, np.nextafter(lo16, np.float16(-np.inf)), lo16)
            hi16 = hi.astype(np.float16)
            hi16 = np.where(hi16.astype(f32) < hi, np.nextafter(hi16, np.float16(np.inf)), hi16)
            return np.concatenate([np.where(neg, hi16, lo16), np.where(neg, lo16, hi16)])

        if nt == 1:
            # a root whose near child is the triangle and whose far child is a box no ray passes (entered at +inf, left at -inf)
            nd = nodes[0]
            assert need == 0 and int(nd["ref"]) == LEAF
            assert np.array_equal(nd["planes"], planes_of(box_lo(0), box_hi(0)))
            fp = nd["far_planes"].astype(f32)
            assert np.all(np.isinf(fp)) and np.all((fp[:3] > 0) != neg) and np.all((fp[3:] < 0) != neg)
            continue
        # walk from the root (inner node 0): every triangle exactly once, every inner node exactly once; a child's box in its
        # parent is the exact union of what lies below it, rounded outwards to half precision and stored as the planes a ray of
        # the octant enters / leaves through; near child first
        seen_tri, seen_node = [], []

        def box_of(ref):
            ref = int(ref)
            if ref & LEAF:
                t = (ref & ~LEAF) // 3
                assert (ref & ~LEAF) % 3 == 0
                seen_tri.append(t)
                return box_lo(t), box_hi(t), 0, 0
            assert (ref - first) % 2 == 0
            k = (ref - first) // 2
            assert 0 <= k < len(nodes)
            seen_node.append(k)
            nd = nodes[k]
            nlo, nhi, nd_, nn = box_of(nd["ref"])
            flo, fhi, fd_, fn = box_of(nd["far_ref"])
            assert np.array_equal(nd["planes"], planes_of(nlo, nhi)) and np.array_equal(nd["far_planes"], planes_of(flo, fhi))
            # the first child is the nearer one for this octant along some axis: its centre does not lie behind the second's
            cl, cr = nlo + nhi, flo + fhi
            assert any((cl[a] >= cr[a]) if (octant >> a) & 1 else (cl[a] <= cr[a]) for a in range(3))
            # depth-first layout: the near child's inner nodes follow their parent directly, the far child's come after them
            if not int(nd["ref"]) & LEAF:
                assert (int(nd["ref"]) - first) // 2 == k + 1
            return np.minimum(nlo, flo), np.maximum(nhi, fhi), 1 + max(nd_, fd_), max(1 + nn, fn)

        _, _, depth, need_here = box_of(first)
        assert sorted(seen_tri) == list(range(nt)) and sorted(seen_node) == list(range(nt - 1))
        assert depth <= 3 * int(np.ceil(np.log2(max(nt, 1)))) + 2                      # no side of a split below an eighth: logarithmic
        # the far children that can wait at once on a lane's stack: this copy's need is within what pt_init reserves for all eight
        assert need_here <= need <= depth and need <= 24


def test_median_rebuild_bounds_the_stack(pt, oracle, monkeypatch):
    # a hierarchy that would need more stack levels than the threshold (24; lowered here through the tests' switch) is rebuilt by
    # median splits alone: need <= ceil(log2 ntris), the same invariants, the same winners as the loop over every triangle
    sc = pt.Scene(os.path.join(SCENES, "cornell_mesh.txt"))
    tris = sc.meshes[7]
    _, _, _, need_sah = pt.mesh_bvh(tris, 0)
    monkeypatch.setenv("PT_AMD_MESH_STACK_MAX", str(need_sah - 1))
    copies = [pt.mesh_bvh(tris, o) for o in range(8)]
    monkeypatch.delenv("PT_AMD_MESH_STACK_MAX")
    need = copies[0][3]
    assert need <= int(np.ceil(np.log2(len(tris)))) and need < need_sah + 1
    ident = oracle.make_geom(2, 0, (0, 0, 0), (0, 0, 0), (1, 1, 1))
    rng = np.random.default_rng(11)
    hits = 0
    for i in range(200):
        o = (rng.normal(size=3) * 2.0).astype(f32)
        tgt = tris[rng.integers(len(tris))].reshape(3, 3).mean(0) + rng.normal(size=3) * 0.02
        d = ((tgt - o) / np.linalg.norm(tgt - o)).astype(f32)
        rd = oracle.normalize(d)
        wt, wp, wn, wo, wtri = oracle.mesh_intersect(ident, tris, np.concatenate([o, d]))
        octant = int(np.signbit(rd[0])) | int(np.signbit(rd[1])) << 1 | int(np.signbit(rd[2])) << 2
        best, tbest, vis = _walk(*copies[octant][:3], tris, oracle, o, rd, copies[octant][3])
        assert best == wtri
        hits += best >= 0
    assert hits > 150


def test_half_precision_planes_round_outwards_everywhere(pt):
    # halfBitsDirected (pt_mesh.h) on boxes from 1e-9 to 1e6, both signs: a node's planes contain the exact box, by less than
    # one step of half precision; beyond the half range they are the largest finite half or infinity, on the permitted side
    rng = np.random.default_rng(5)
    for scale in (1e-9, 3e-7, 6e-5, 1e-3, 1.0, 300.0, 65504.0, 7e4, 1e6):
        tris = (rng.uniform(-1, 1, (6, 9)) * scale).astype(f32)
        recs, nodes, first, need = pt.mesh_bvh(tris, 0)
        margin = f32(1e-5) * np.abs(tris).max()
        v = tris.reshape(-1, 3, 3)
        for nd in nodes:
            for ref, planes in ((int(nd["ref"]), nd["planes"]), (int(nd["far_ref"]), nd["far_planes"])):
                if not ref & pt.MESH_LEAF:
                    continue
                t = (ref & ~pt.MESH_LEAF) // 3
                lo, hi = v[t].min(0) - margin, v[t].max(0) + margin
                pl = planes.astype(np.float64)
                assert np.all(pl[:3] <= lo) and np.all(pl[3:] >= hi)                      # (octant 0: entry = lo, exit = hi)
                lo16, hi16 = planes[:3], planes[3:]
                fin = np.isfinite(lo16.astype(np.float64))
                assert np.all(np.nextafter(lo16, np.float16(np.inf)).astype(np.float64)[fin] > lo[fin])   # the nearest half below
                fin = np.isfinite(hi16.astype(np.float64))
                assert np.all(np.nextafter(hi16, np.float16(-np.inf)).astype(np.float64)[fin] < hi[fin])
                exact = np.concatenate([lo, hi]).astype(np.float64)
                big = np.abs(exact) > 65504.0                                             # beyond the half range
                assert np.all((np.abs(pl[big]) == 65504.0) | np.isinf(pl[big]))


def _walk(recs, nodes, first, tris, oracle, ro, rd, need):
    """The kernel's traversal (ptd::meshIntersectionTest) in numpy fp32 on object-space rays; triangle test by the oracle."""
    LEAF = 0x80000000
    nt = len(recs)
    up, dn = f32(1.00001), f32(0.99999)
    g = np.where(np.abs(rd) < f32(1e-30), np.copysign(f32(1e-30), rd), rd).astype(f32)
    inv = (f32(1.0) / g).astype(f32)
    c = (-(ro * inv)).astype(f32)
    fma = lambda x: (x.astype(np.float64) * inv.astype(np.float64) + c.astype(np.float64)).astype(f32)   # exact in fp64, rounded once
    best, tbest, visited, deepest = -1, f32(0), 0, 0

    def box_pass(lo, hi):
        a, b = fma(lo), fma(hi)
        tn = np.max(np.minimum(a, b))
        tf = np.min(np.maximum(a, b))
        tmin = f32(tn * dn)
        return bool(f32(tf * up) >= tmin and tf >= 0 and (best < 0 or not tmin > tbest)), tmin

    def planes_pass(pl):
        pl = pl.astype(f32)
        tn, tf = np.max(fma(pl[:3])), np.min(fma(pl[3:]))
        tmin = f32(tn * dn)
        return bool(f32(tf * up) >= tmin and tf >= 0 and (best < 0 or not tmin > tbest))

    ref = first
    stack = []
    margin = recs[0]["margin"]
    while True:
        visited += 1
        pop = True
        if ref & LEAF:
            tri = (ref & ~LEAF) // 3
            vv = np.stack([recs[tri]["v0"], recs[tri]["v1"], recs[tri]["v2"]])
            ok, tmin = box_pass(vv.min(0) - margin, vv.max(0) + margin)
            if ok:
                v = tris[tri]
                hit, tuv, front = oracle.mesh_triangle(ro, rd, v[0:3], v[3:6], v[6:9])
                t = f32(tuv[0])
                if hit and t >= tmin and (best < 0 or t < tbest or (t == tbest and tri < best)):
                    best, tbest = tri, t
        else:
            nd = nodes[(int(ref) - first) // 2]
            pn = planes_pass(nd["planes"])
            pf = planes_pass(nd["far_planes"])
            if pn and pf:
                stack.append(int(nd["far_ref"]))
                deepest = max(deepest, len(stack))
            if pn or pf:
                ref = int(nd["ref"]) if pn else int(nd["far_ref"])
                pop = False
        if pop:
            if not stack:
                break
            ref = stack.pop()
    assert deepest <= need
    return best, tbest, visited


def test_hierarchy_walk_equals_the_brute_force_rule(pt, oracle):
    sc = pt.Scene(os.path.join(SCENES, "cornell_mesh.txt"))
    ident = oracle.make_geom(2, 0, (0, 0, 0), (0, 0, 0), (1, 1, 1))       # identity: world space = object space
    rng = np.random.default_rng(7)
    for g in (6, 7):
        tris = sc.meshes[g]
        copies = [pt.mesh_bvh(tris, o) for o in range(8)]
        hits = visited = 0
        for i in range(400):
            o = (rng.normal(size=3) * (0.2 if i % 4 == 0 else 2.0)).astype(f32)
            tgt = tris[rng.integers(len(tris))].reshape(3, 3).mean(0) + rng.normal(size=3) * 0.02
            d = (tgt - o) if i % 5 else rng.normal(size=3)
            d = (d / np.linalg.norm(d)).astype(f32)
            if i % 7 == 3:                        # axis-parallel components, +0 and -0: the copy is chosen by the SIGN BIT of the reciprocal
                d[i % 3] = f32(0.0) if i % 2 else f32(-0.0)
                d = (d / np.linalg.norm(d)).astype(f32)
            # (the oracle normalises once more in object space; feed it the direction it will actually use)
            rd = oracle.normalize(d)
            wt, wp, wn, wo, wtri = oracle.mesh_intersect(ident, tris, np.concatenate([o, d]))
            octant = int(np.signbit(rd[0])) | int(np.signbit(rd[1])) << 1 | int(np.signbit(rd[2])) << 2
            best, tbest, vis = _walk(*copies[octant][:3], tris, oracle, o, rd, copies[octant][3])
            assert best == wtri, (g, i, best, wtri)
            hits += best >= 0
            visited += vis
        assert hits > 250
        assert visited / 400 < 0.05 * (2 * len(tris) - 1)      # ... and it is a hierarchy: a fraction of the records per ray
        print('mesh', g, 'records fetched per ray', visited / 400)
