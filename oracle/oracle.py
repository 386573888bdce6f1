"""ctypes binding of the CPU oracle (oracle/libptoracle.so).  TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libptoracle.so")

# byte-identical to reference src/sceneStructs.h:18-47
GEOM_DTYPE = np.dtype([
    ("type", "<i4"), ("materialid", "<i4"),
    ("translation", "<f4", 3), ("rotation", "<f4", 3), ("scale", "<f4", 3),
    ("transform", "<f4", 16), ("inverseTransform", "<f4", 16), ("invTranspose", "<f4", 16),
])
MATERIAL_DTYPE = np.dtype([
    ("color", "<f4", 3), ("specExponent", "<f4"), ("specColor", "<f4", 3),
    ("hasReflective", "<f4"), ("hasRefractive", "<f4"), ("indexOfRefraction", "<f4"),
    ("emittance", "<f4"),
])
CAMERA_DTYPE = np.dtype([
    ("resolution", "<i4", 2), ("position", "<f4", 3), ("view", "<f4", 3), ("up", "<f4", 3),
    ("fov", "<f4", 2),
])
assert GEOM_DTYPE.itemsize == 236 and MATERIAL_DTYPE.itemsize == 44 and CAMERA_DTYPE.itemsize == 52


class Counters(C.Structure):
    _fields_ = [("live", C.c_int64 * 64), ("lightHits", C.c_int64), ("misses", C.c_int64),
                ("depthKilled", C.c_int64)]


def build(force=False):
    """Compile the oracle (and oracle/_ref when /root/reference exists)."""
    if force or not os.path.exists(LIB_PATH) or \
            os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(HERE, "pt_oracle.cpp")):
        subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    if os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        fp = C.POINTER(C.c_float)
        L.orc_utilhash.restype = C.c_uint32
        L.orc_utilhash.argtypes = [C.c_uint32]
        L.orc_seed.restype = C.c_uint32
        L.orc_seed.argtypes = [C.c_int] * 3
        L.orc_rng_stream.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_rng_stream_from_seed.argtypes = [C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_sincos.argtypes = [C.c_float, fp, fp]
        for name in ("orc_normalize",):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p]
        L.orc_reflect.argtypes = [C.c_void_p] * 3
        L.orc_refract.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
        L.orc_mulmv.argtypes = [C.c_void_p] * 3
        L.orc_point_on_ray.argtypes = [C.c_void_p, C.c_float, C.c_void_p]
        L.orc_build_transform.argtypes = [C.c_void_p] * 6
        for name in ("orc_box_intersect", "orc_sphere_intersect"):
            f = getattr(L, name)
            f.restype = C.c_float
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.orc_hemisphere_seeded.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_scene_load.restype = C.c_void_p
        L.orc_scene_load.argtypes = [C.c_char_p]
        L.orc_scene_free.argtypes = [C.c_void_p]
        for name in ("orc_scene_num_geoms", "orc_scene_num_materials", "orc_scene_iterations",
                     "orc_scene_depth"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = C.c_int
        for name in ("orc_scene_geoms", "orc_scene_materials", "orc_scene_camera"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = C.c_void_p
        L.orc_scene_image_name.argtypes = [C.c_void_p]
        L.orc_scene_image_name.restype = C.c_char_p
        L.orc_camera_set_resolution.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_scene_num_meshes.argtypes = [C.c_void_p]
        for name in ("orc_scene_mesh_geom", "orc_scene_mesh_ntris"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_int]
        L.orc_scene_mesh_tris.argtypes = [C.c_void_p, C.c_int]
        L.orc_scene_mesh_tris.restype = C.c_void_p
        L.orc_render_set_mesh.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_render_set_mesh_attributes.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        for name in ("orc_scene_mesh_normals", "orc_scene_mesh_materials"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_int]
            getattr(L, name).restype = C.c_void_p
        L.orc_mesh_intersect_attr.restype = C.c_float
        L.orc_mesh_intersect_attr.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_mesh_intersect.restype = C.c_float
        L.orc_mesh_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_mesh_margin.restype = C.c_float
        L.orc_mesh_margin.argtypes = [C.c_void_p, C.c_int]
        L.orc_mesh_triangle.argtypes = [C.c_void_p] * 6 + [C.POINTER(C.c_int)]
        L.orc_render_create.restype = C.c_void_p
        L.orc_render_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.orc_render_free.argtypes = [C.c_void_p]
        L.orc_render_set_extras.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_int]
        L.orc_pow.argtypes = [C.c_float, C.c_float]
        L.orc_pow.restype = C.c_float
        L.orc_render_iterate.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                         C.POINTER(Counters)]
        L.orc_render_dump_paths.restype = C.c_int
        L.orc_render_dump_paths.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_camera_ray.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_to_rgba8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_scan_exclusive_i32.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.orc_compact_nonzero_i32.restype = C.c_int64
        L.orc_compact_nonzero_i32.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def f32(*v):
    return np.ascontiguousarray(np.array(v, dtype=np.float32).reshape(-1))


# ---------------------------------------------------------------- primitives
def utilhash(a):
    return int(lib().orc_utilhash(C.c_uint32(a & 0xFFFFFFFF)))


def seed(it, index, depth):
    return int(lib().orc_seed(it, index, depth))


def rng_stream(it, index, depth, n):
    u = np.empty(n, np.float32)
    s = np.empty(n, np.uint32)
    lib().orc_rng_stream(it, index, depth, n, _p(u), _p(s))
    return u, s


def rng_stream_from_seed(sd, n):
    u = np.empty(n, np.float32)
    s = np.empty(n, np.uint32)
    lib().orc_rng_stream_from_seed(C.c_uint32(sd & 0xFFFFFFFF), n, _p(u), _p(s))
    return u, s


def sincos(x):
    s, c = C.c_float(), C.c_float()
    lib().orc_sincos(C.c_float(x), C.byref(s), C.byref(c))
    return np.float32(s.value), np.float32(c.value)


def normalize(v):
    o = np.empty(3, np.float32)
    lib().orc_normalize(_p(f32(*v)), _p(o))
    return o


def reflect(i, n):
    o = np.empty(3, np.float32)
    lib().orc_reflect(_p(f32(*i)), _p(f32(*n)), _p(o))
    return o


def refract(i, n, eta):
    o = np.empty(3, np.float32)
    lib().orc_refract(_p(f32(*i)), _p(f32(*n)), C.c_float(eta), _p(o))
    return o


def mulmv(m16, v4):
    o = np.empty(3, np.float32)
    lib().orc_mulmv(_p(f32(*m16)), _p(f32(*v4)), _p(o))
    return o


def point_on_ray(ray6, t):
    o = np.empty(3, np.float32)
    lib().orc_point_on_ray(_p(f32(*ray6)), C.c_float(t), _p(o))
    return o


def build_transform(t, r, s):
    xf, inv, it = (np.empty(16, np.float32) for _ in range(3))
    lib().orc_build_transform(_p(f32(*t)), _p(f32(*r)), _p(f32(*s)), _p(xf), _p(inv), _p(it))
    return xf, inv, it


def make_geom(gtype, materialid, t, r, s):
    g = np.zeros(1, GEOM_DTYPE)
    g["type"] = gtype
    g["materialid"] = materialid
    g["translation"], g["rotation"], g["scale"] = f32(*t), f32(*r), f32(*s)
    g["transform"], g["inverseTransform"], g["invTranspose"] = build_transform(t, r, s)
    return g


def intersect(geom, ray6, sphere=None):
    """geom: 1-element GEOM_DTYPE array.  Returns (t, p, n, outside); p, n, outside keep their
    input sentinel (-7 / -7) when the test misses (outputs untouched on miss)."""
    g = np.ascontiguousarray(geom).reshape(-1)[:1]
    if sphere is None:
        sphere = int(g["type"][0]) == 0
    p = np.full(3, -7.0, np.float32)
    n = np.full(3, -7.0, np.float32)
    o = C.c_int(-7 & 1)
    f = lib().orc_sphere_intersect if sphere else lib().orc_box_intersect
    t = f(_p(g), _p(f32(*ray6)), _p(p), _p(n), C.byref(o))
    return np.float32(t), p, n, int(o.value)


def mesh_intersect(geom, tris, ray6, normals=None):
    """One ray against one mesh geom (tris: n x 9 floats, object space), brute force over every triangle.  `normals`: n x 9 vertex
    normals (smooth shading) or None (flat).  Returns (t, p, n, outside, triangle); p, n, outside keep their sentinels on a miss,
    triangle = -1."""
    g = np.ascontiguousarray(geom).reshape(-1)[:1]
    tr = np.ascontiguousarray(tris, np.float32).reshape(-1, 9)
    p = np.full(3, -7.0, np.float32)
    n = np.full(3, -7.0, np.float32)
    o, k = C.c_int(1), C.c_int(-1)
    if normals is None:
        t = lib().orc_mesh_intersect(_p(g), _p(tr), len(tr), _p(f32(*ray6)), _p(p), _p(n), C.byref(o), C.byref(k))
    else:
        nn = np.ascontiguousarray(normals, np.float32).reshape(len(tr), 9)
        t = lib().orc_mesh_intersect_attr(_p(g), _p(tr), len(tr), _p(nn), _p(f32(*ray6)), _p(p), _p(n), C.byref(o), C.byref(k))
    return np.float32(t), p, n, int(o.value), int(k.value)


def mesh_triangle(o, d, v0, v1, v2):
    """The two-sided triangle test alone.  Returns (hit, (t, u, v), front); t, u, v are NaN where not evaluated."""
    tuv = np.full(3, np.nan, np.float32)
    front = C.c_int(0)
    hit = lib().orc_mesh_triangle(_p(f32(*o)), _p(f32(*d)), _p(f32(*v0)), _p(f32(*v1)), _p(f32(*v2)), _p(tuv), C.byref(front))
    return bool(hit), tuv, bool(front.value)


def mesh_margin(tris):
    tr = np.ascontiguousarray(tris, np.float32).reshape(-1, 9)
    return np.float32(lib().orc_mesh_margin(_p(tr), len(tr)))


def hemisphere_seeded(n, it, index, depth):
    o = np.empty(3, np.float32)
    lib().orc_hemisphere_seeded(_p(f32(*n)), it, index, depth, _p(o))
    return o


# ---------------------------------------------------------------- scene
class Scene:
    def __init__(self, path):
        h = lib().orc_scene_load(path.encode())
        if not h:
            raise IOError("Error reading from file - aborting! (%s)" % path)
        L = lib()
        ng, nm = L.orc_scene_num_geoms(h), L.orc_scene_num_materials(h)
        self.geoms = np.frombuffer(C.string_at(L.orc_scene_geoms(h), 236 * ng), GEOM_DTYPE).copy() \
            if ng else np.zeros(0, GEOM_DTYPE)
        self.materials = np.frombuffer(C.string_at(L.orc_scene_materials(h), 44 * nm), MATERIAL_DTYPE).copy() \
            if nm else np.zeros(0, MATERIAL_DTYPE)
        self.camera = np.frombuffer(C.string_at(L.orc_scene_camera(h), 52), CAMERA_DTYPE).copy()
        self.iterations = L.orc_scene_iterations(h)
        self.depth = L.orc_scene_depth(h)
        self.image_name = L.orc_scene_image_name(h).decode()
        self.meshes = {}            # geom index -> (ntris, 9) float32, object space
        self.mesh_normals = {}      # geom index -> (ntris, 9) float32 vertex normals (`vn`); absent: flat shading
        self.mesh_materials = {}    # geom index -> (ntris,) int32 scene material per face (`usemtl <k>`, -1 = the object's); absent: none
        for i in range(L.orc_scene_num_meshes(h)):
            nt = L.orc_scene_mesh_ntris(h, i)
            g = L.orc_scene_mesh_geom(h, i)
            self.meshes[g] = np.frombuffer(C.string_at(L.orc_scene_mesh_tris(h, i), 36 * nt), np.float32).reshape(nt, 9).copy()
            nptr, mptr = L.orc_scene_mesh_normals(h, i), L.orc_scene_mesh_materials(h, i)
            if nptr:
                self.mesh_normals[g] = np.frombuffer(C.string_at(nptr, 36 * nt), np.float32).reshape(nt, 9).copy()
            if mptr:
                self.mesh_materials[g] = np.frombuffer(C.string_at(mptr, 4 * nt), np.int32).copy()
        L.orc_scene_free(h)

    def set_resolution(self, w, h):
        lib().orc_camera_set_resolution(_p(self.camera), w, h)


class Renderer:
    def __init__(self, camera, geoms, materials, depth, meshes=None, mesh_normals=None, mesh_materials=None):
        self.camera = np.ascontiguousarray(camera).copy()
        self.geoms = np.ascontiguousarray(geoms)
        self.materials = np.ascontiguousarray(materials)
        self.depth = depth
        self.W, self.H = (int(v) for v in self.camera["resolution"][0])
        self.h = lib().orc_render_create(_p(self.camera), _p(self.geoms), len(self.geoms),
                                         _p(self.materials), len(self.materials), depth)
        for g, tris in (meshes or {}).items():
            tr = np.ascontiguousarray(tris, np.float32).reshape(-1, 9)
            lib().orc_render_set_mesh(self.h, int(g), _p(tr), len(tr))
            nn = (mesh_normals or {}).get(g)
            mm = (mesh_materials or {}).get(g)
            if nn is not None or mm is not None:
                nn = None if nn is None else np.ascontiguousarray(nn, np.float32).reshape(len(tr), 9)
                mm = None if mm is None else np.ascontiguousarray(mm, np.int32).reshape(len(tr))
                lib().orc_render_set_mesh_attributes(self.h, int(g), None if nn is None else _p(nn), None if mm is None else _p(mm))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_render_free(self.h)
            self.h = None

    def set_extras(self, lens_radius=0.0, focal_distance=0.0, direct_lighting=False):
        lib().orc_render_set_extras(self.h, C.c_float(lens_radius), C.c_float(focal_distance), 1 if direct_lighting else 0)

    def set_variant(self, scatter_offset=0.001, mirror_mode=0, emit_color_mode=0):
        """STUDY variants of the build-defined scatter (never used by a parity test): offset of the new origin along the normal;
        REFL > 0 materials as 50/50 energy conserving (0, the build's choice), 50/50 with 1 / p weights (1), a pure mirror (2) or split
        by the intensities of the two colours with 1 / p weights (3, src/interactions.h:56-59); emit_color_mode 1 = an emitter hit
        contributes throughput x emittance without the emitter's own colour (`color *= m.color` AFTER the emitter test)."""
        L = lib()
        L.orc_render_set_variant.argtypes = [C.c_void_p, C.c_float, C.c_int]
        L.orc_render_set_variant(self.h, C.c_float(scatter_offset), mirror_mode)
        L.orc_render_set_emit_variant.argtypes = [C.c_void_p, C.c_int]
        L.orc_render_set_emit_variant(self.h, emit_color_mode)

    def iterate(self, it, image, rank=0, count=1):
        c = Counters()
        assert image.dtype == np.float32 and image.size == self.W * self.H * 3 and image.flags.c_contiguous
        lib().orc_render_iterate(self.h, it, _p(image), rank, count, C.byref(c))
        return c

    def dump_paths(self, it, bounces, rank=0, count=1):
        n = self.W * self.H
        o, d, c = (np.empty((n, 3), np.float32) for _ in range(3))
        pix = np.empty(n, np.int32)
        k = lib().orc_render_dump_paths(self.h, it, bounces, rank, count, _p(o), _p(d), _p(c), _p(pix))
        return o[:k], d[:k], c[:k], pix[:k]

    def camera_ray(self, it, index):
        r = np.empty(6, np.float32)
        lib().orc_camera_ray(self.h, it, index, _p(r))
        return r


def to_rgba8(image, it):
    image = np.ascontiguousarray(image, np.float32)
    n = image.size // 3
    out = np.empty((n, 4), np.uint8)
    lib().orc_to_rgba8(_p(image), n, it, _p(out))
    return out


def scan_exclusive(a):
    a = np.ascontiguousarray(a, np.int32)
    out = np.empty_like(a)
    lib().orc_scan_exclusive_i32(_p(a), _p(out), a.size)
    return out


def compact_nonzero(a):
    a = np.ascontiguousarray(a, np.int32)
    out = np.empty_like(a)
    k = lib().orc_compact_nonzero_i32(_p(a), _p(out), a.size)
    return out[:k]
