"""project3-cuda-path-tracer_amd -- MI355X-native hot path of CIS565 Project3-CUDA-Path-Tracer.

Python host-side mirror of the reference's renderer interface (src/pathtrace.h:6-8):

    pathtraceInit(scene)                 reference src/pathtrace.cu:75-85
    pathtrace(pbo, frame, iteration)     reference src/pathtrace.cu:123-174
    pathtraceFree()                      reference src/pathtrace.cu:87-92

over the C ABI declared in include/pt_amd.h (csrc/libpt_amd.so, hand-written HIP for gfx950) and
the C++ scene loader in host/ (libpt_host.so).  There is NO CPU fallback: importing works
anywhere (so the library's symbols can be checked), but every compute entry point fails loudly
when the extension or a GPU is missing.  This package never touches oracle/.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# PT_AMD_LIB: another build of the same library (A/B measurements of kernel variants); never a different backend
LIB_PATH = os.environ.get("PT_AMD_LIB") or os.path.join(HERE, "csrc", "libpt_amd.so")
HOST_LIB_PATH = os.path.join(HERE, "host", "libpt_host.so")

# byte-identical to reference src/sceneStructs.h:18-47 (include/pt_amd.h PtGeom/PtMaterial/PtCamera)
GEOM_DTYPE = np.dtype([
    ("type", "<i4"), ("materialid", "<i4"),
    ("translation", "<f4", 3), ("rotation", "<f4", 3), ("scale", "<f4", 3),
    ("transform", "<f4", 16), ("inverseTransform", "<f4", 16), ("invTranspose", "<f4", 16),
])
MATERIAL_DTYPE = np.dtype([
    ("color", "<f4", 3), ("specularExponent", "<f4"), ("specularColor", "<f4", 3),
    ("hasReflective", "<f4"), ("hasRefractive", "<f4"), ("indexOfRefraction", "<f4"),
    ("emittance", "<f4"),
])
CAMERA_DTYPE = np.dtype([
    ("resolution", "<i4", 2), ("position", "<f4", 3), ("view", "<f4", 3), ("up", "<f4", 3),
    ("fov", "<f4", 2),
])
assert GEOM_DTYPE.itemsize == 236 and MATERIAL_DTYPE.itemsize == 44 and CAMERA_DTYPE.itemsize == 52

PT_MAX_DEPTH = 62
PT_MAX_BATCH = 256
PT_FLAG_KERNEL_TIMING = 1
PT_FLAG_ACCUM_SHARD_ROWS = 2
PT_FLAG_DIRECT_LIGHTING = 4
PT_FLAG_TRACE_AHEAD = 8
PT_FLAG_MIXTURE_WEIGHTED = 16

# second link target of the same source: + the test-only entry points of include/pt_amd_test.h (tests/ and profiles/ only)
TEST_LIB_PATH = os.path.join(HERE, "csrc", "libpt_amd_test.so")

# every symbol include/pt_amd.h declares: the product's whole exported surface
ABI_SYMBOLS = [
    "pt_init", "pt_iterate", "pt_iterate_batch", "pt_sync", "pt_readback", "pt_readback_rgba8", "pt_counters",
    "pt_counters_reset", "pt_free", "pt_last_error", "pt_device_count",
    "pt_scan_exclusive_i32", "pt_compact_nonzero_i32", "pt_pin_host", "pt_unpin_host", "pt_set_meshes", "pt_set_meshes_sized", "pt_abi_version",
    "pt_ctx_create", "pt_ctx_make_current", "pt_ctx_current", "pt_ctx_destroy",
    "pt_group_create", "pt_group_destroy", "pt_group_size", "pt_group_collective", "pt_group_set_meshes", "pt_group_init", "pt_group_iterate_batch",
    "pt_group_iterate", "pt_group_reduce", "pt_group_sync", "pt_group_readback", "pt_group_counters",
]
PT_AMD_ABI_VERSION = 6
# every symbol include/pt_amd_test.h declares: libpt_amd_test.so only -- the product library must NOT export them
TEST_ABI_SYMBOLS = [
    "pt_debug_trace_paths",
    "pt_test_utilhash", "pt_test_rng", "pt_test_intersect", "pt_test_hemisphere", "pt_test_sincos", "pt_test_reflect_refract",
    "pt_test_slab_quotients", "pt_test_slab_quotients_sweep", "pt_test_box_fast_sweep", "pt_test_sphere_cull_sweep", "pt_test_unscaled_sqrt_sweep",
    "pt_test_force_fault", "pt_test_pow", "pt_test_wall_box_sweep", "pt_test_mesh_intersect", "pt_test_mesh_bvh",
    "pt_test_mesh_cull_sweep", "pt_test_camera_cull_sweep", "pt_test_camera_cull_tables",
    "pt_test_wall_plane_sweep", "pt_test_wall_planes", "pt_test_sphere_halfline_sweep", "pt_test_sphere_cluster_sweep", "pt_test_sphere_clusters", "pt_test_camera_cull_margin",
    "pt_test_group_fail_next_reduce", "pt_test_sphere_group_sweep",
]


class PtOptions(C.Structure):
    _fields_ = [("shard_rank", C.c_int32), ("shard_count", C.c_int32), ("device", C.c_int32),
                ("flags", C.c_int32), ("pipeline_depth", C.c_int32), ("max_batch", C.c_int32),
                ("stream", C.c_void_p), ("accum_dev", C.c_void_p), ("lens_radius", C.c_float), ("focal_distance", C.c_float)]


class PtMesh(C.Structure):
    _fields_ = [("geom", C.c_int32), ("ntris", C.c_int32), ("tris", C.c_void_p), ("normals", C.c_void_p), ("materials", C.c_void_p)]


class PtCounters(C.Structure):
    _fields_ = [("live", C.c_int64 * (PT_MAX_DEPTH + 2)), ("light_hits", C.c_int64), ("misses", C.c_int64),
                ("iterations", C.c_int64), ("bounce_launches", C.c_int64), ("bounce_kernel_ms", C.c_double),
                ("raygen_kernel_ms", C.c_double), ("raygen_launches", C.c_int64), ("ended_early", C.c_int64 * (PT_MAX_DEPTH + 2))]


class PtError(RuntimeError):
    pass


_lib = None
_test = None
_last_init = None           # the arguments of the last pathtraceInit (debug_trace_paths re-creates the scene in the test library)
_host = None
_atexit_registered = False


def _bind(L, with_tests):
    vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
    L.pt_init.argtypes = [vp, vp, i32, vp, i32, i32, C.POINTER(PtOptions)]
    L.pt_iterate.argtypes = [i32, i32, vp]
    L.pt_iterate_batch.argtypes = [i32, i32, i32, vp]
    L.pt_sync.argtypes = []
    L.pt_readback.argtypes = [vp]
    L.pt_readback_rgba8.argtypes = [i32, vp]
    L.pt_counters.argtypes = [C.POINTER(PtCounters)]
    L.pt_counters_reset.argtypes = []
    L.pt_free.argtypes = []
    L.pt_free.restype = None
    L.pt_last_error.restype = C.c_char_p
    L.pt_device_count.argtypes = []
    L.pt_scan_exclusive_i32.argtypes = [vp, vp, i64, vp]
    L.pt_compact_nonzero_i32.argtypes = [vp, vp, i64, vp, vp]
    L.pt_pin_host.argtypes = [vp, C.c_size_t]
    L.pt_unpin_host.argtypes = []
    L.pt_set_meshes.argtypes = [C.POINTER(PtMesh), i32]
    L.pt_set_meshes_sized.argtypes = [C.POINTER(PtMesh), i32, C.c_size_t]
    L.pt_abi_version.argtypes = []
    L.pt_ctx_create.argtypes = []
    L.pt_ctx_create.restype = vp
    L.pt_ctx_make_current.argtypes = [vp]
    L.pt_ctx_current.argtypes = []
    L.pt_ctx_current.restype = vp
    L.pt_ctx_destroy.argtypes = [vp]
    L.pt_group_create.argtypes = [C.POINTER(vp), i32, C.POINTER(C.c_int32)]
    L.pt_group_destroy.argtypes = [vp]
    L.pt_group_destroy.restype = None
    L.pt_group_size.argtypes = [vp]
    L.pt_group_collective.argtypes = [vp]
    L.pt_group_collective.restype = C.c_char_p
    L.pt_group_set_meshes.argtypes = [vp, C.POINTER(PtMesh), i32]
    L.pt_group_init.argtypes = [vp, vp, vp, i32, vp, i32, i32, C.POINTER(PtOptions)]
    L.pt_group_iterate_batch.argtypes = [vp, i32, i32, i32]
    L.pt_group_iterate.argtypes = [vp, i32, i32]
    L.pt_group_reduce.argtypes = [vp]
    L.pt_group_sync.argtypes = [vp]
    L.pt_group_readback.argtypes = [vp, vp]
    L.pt_group_counters.argtypes = [vp, C.POINTER(PtCounters)]
    if with_tests:
        L.pt_debug_trace_paths.argtypes = [i32, i32, vp, vp, vp, vp, C.POINTER(C.c_int32)]
        u64p = C.POINTER(C.c_uint64)
        L.pt_test_utilhash.argtypes = [vp, vp, i32]
        L.pt_test_rng.argtypes = [vp, i32, i32, vp]
        L.pt_test_intersect.argtypes = [vp, i32, vp, vp, i32, vp, vp, vp, vp]
        L.pt_test_hemisphere.argtypes = [vp, vp, i32, vp]
        L.pt_test_sincos.argtypes = [vp, i32, vp, vp]
        L.pt_test_reflect_refract.argtypes = [vp, vp, vp, i32, vp, vp]
        L.pt_test_slab_quotients.argtypes = [vp, vp, i32, vp, vp, vp, vp]
        L.pt_test_slab_quotients_sweep.argtypes = [C.c_uint64, i64, u64p]
        L.pt_test_box_fast_sweep.argtypes = [vp, i32, C.c_uint64, i64, u64p, u64p]
        L.pt_test_sphere_cull_sweep.argtypes = [vp, i32, C.c_uint64, i64, u64p, u64p]
        L.pt_test_sphere_halfline_sweep.argtypes = [vp, i32, C.c_uint64, i64, u64p, u64p, u64p]
        L.pt_test_sphere_cluster_sweep.argtypes = [vp, i32, C.c_uint64, i64, u64p, u64p, vp]
        L.pt_test_sphere_clusters.argtypes = [vp, i32, vp, vp, i32, vp]
        L.pt_test_unscaled_sqrt_sweep.argtypes = [u64p]
        L.pt_test_force_fault.argtypes = [i32]
        L.pt_test_pow.argtypes = [vp, vp, i32, vp]
        L.pt_test_wall_box_sweep.argtypes = [vp, i32, C.c_uint64, i64, u64p, u64p]
        L.pt_test_mesh_intersect.argtypes = [vp, vp, i32, i32, vp, i32, vp, vp, vp, vp, vp]
        L.pt_test_mesh_bvh.argtypes = [vp, i32, i32, vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.pt_test_mesh_cull_sweep.argtypes = [vp, vp, i32, C.c_uint64, i64] + [u64p] * 3
        L.pt_test_camera_cull_sweep.argtypes = [vp, vp, i32, i32] + [u64p] * 3
        L.pt_test_camera_cull_tables.argtypes = [vp, vp, i32, vp, vp, vp]
        L.pt_test_camera_cull_margin.argtypes = [vp, vp, i32, i32, C.POINTER(C.c_double), u64p]
        L.pt_test_wall_plane_sweep.argtypes = [vp, i32, C.c_uint64, i64, C.POINTER(C.c_int32)] + [u64p] * 3
        L.pt_test_wall_planes.argtypes = [vp, i32, vp, vp] + [C.POINTER(C.c_int32)] * 3
        L.pt_test_group_fail_next_reduce.argtypes = [vp, i32]
        L.pt_test_sphere_group_sweep.argtypes = [vp, i32, C.c_uint64, i64, u64p, u64p, C.POINTER(C.c_int32)]
    return L


def _load(path, with_tests):
    if not os.path.exists(path):
        raise PtError("HIP extension missing: %s (run __graft_entry__.build()); there is no CPU fallback" % path)
    # torch (device memory / streams / RCCL plumbing) bundles its own libamdhip64.so; it has to be
    # loaded FIRST so that this library binds to the same HIP runtime instance (same soname) and
    # torch stream handles / device pointers are valid here.  Two runtimes in one process break both.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    return _bind(C.CDLL(path), with_tests)


def lib():
    """The HIP extension (the product library).  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        _lib = _load(LIB_PATH, False)
    return _lib


def test_lib():
    """libpt_amd_test.so: the same source linked with the test-only entry points of include/pt_amd_test.h.  It holds its own
    renderer instance; the product path never loads it."""
    global _test
    if _test is None:
        _test = _load(TEST_LIB_PATH, True)
    return _test


class renderer_from_test_library:
    """Context manager for the one test that has to reach INTO a renderer (pt_test_force_fault): inside it the renderer API of
    this module drives the instance that lives in libpt_amd_test.so."""

    def __enter__(self):
        global _lib
        self._saved = lib()
        _lib = test_lib()
        return self

    def __exit__(self, *exc):
        global _lib
        _lib.pt_free()
        _lib = self._saved
        return False


def _check(rc, L=None):
    if rc != 0:
        raise PtError("pt_amd error %d: %s" % (rc, (L or lib()).pt_last_error().decode()))


def _tcheck(rc):
    _check(rc, test_lib())


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def device_count():
    return lib().pt_device_count()


# --------------------------------------------------------------------------- scene (host/ loader)
def host_lib():
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise PtError("host library missing: %s (run __graft_entry__.build())" % HOST_LIB_PATH)
        H = C.CDLL(HOST_LIB_PATH)
        vp = C.c_void_p
        H.pth_scene_load.restype = vp
        H.pth_scene_load.argtypes = [C.c_char_p]
        H.pth_scene_free.argtypes = [vp]
        H.pth_scene_free.restype = None
        for n in ("pth_scene_num_geoms", "pth_scene_num_materials", "pth_scene_iterations", "pth_scene_depth"):
            getattr(H, n).argtypes = [vp]
            getattr(H, n).restype = C.c_int
        for n in ("pth_scene_geoms", "pth_scene_materials", "pth_scene_camera"):
            getattr(H, n).argtypes = [vp]
            getattr(H, n).restype = vp
        H.pth_scene_image_name.argtypes = [vp]
        H.pth_scene_image_name.restype = C.c_char_p
        H.pth_scene_set_resolution.argtypes = [vp, C.c_int, C.c_int]
        H.pth_scene_set_resolution.restype = None
        H.pth_scene_num_meshes.argtypes = [vp]
        for n in ("pth_scene_mesh_geom", "pth_scene_mesh_ntris"):
            getattr(H, n).argtypes = [vp, C.c_int]
        H.pth_scene_mesh_tris.argtypes = [vp, C.c_int]
        H.pth_scene_mesh_tris.restype = vp
        for n in ("pth_scene_mesh_normals", "pth_scene_mesh_materials"):
            getattr(H, n).argtypes = [vp, C.c_int]
            getattr(H, n).restype = vp
        for n in ("pth_save_png", "pth_save_hdr"):
            getattr(H, n).argtypes = [C.c_char_p, vp, C.c_int, C.c_int, C.c_float]
            getattr(H, n).restype = C.c_int
        _host = H
    return _host


class Scene:
    """Mirror of the reference's `Scene` (src/scene.h:13-26): geoms, materials, state."""

    def __init__(self, filename):
        H = host_lib()
        h = H.pth_scene_load(os.fsencode(filename))
        if not h:
            raise IOError("Error reading from file - aborting!")  # reference src/scene.cpp:12-15
        self._h = h
        self._refresh()

    def _refresh(self):
        H, h = host_lib(), self._h
        ng, nm = H.pth_scene_num_geoms(h), H.pth_scene_num_materials(h)
        self.geoms = np.frombuffer(C.string_at(H.pth_scene_geoms(h), 236 * ng), GEOM_DTYPE).copy() if ng \
            else np.zeros(0, GEOM_DTYPE)
        self.materials = np.frombuffer(C.string_at(H.pth_scene_materials(h), 44 * nm), MATERIAL_DTYPE).copy() if nm \
            else np.zeros(0, MATERIAL_DTYPE)
        self.camera = np.frombuffer(C.string_at(H.pth_scene_camera(h), 52), CAMERA_DTYPE).copy()
        self.iterations = H.pth_scene_iterations(h)
        self.traceDepth = H.pth_scene_depth(h)
        self.imageName = H.pth_scene_image_name(h).decode()
        # `mesh <file.obj>` objects (README.md:236): geom index -> (ntris, 9) float32 triangles in object space
        self.meshes = {}
        self.mesh_normals = {}      # geom index -> (ntris, 9) vertex normals (`vn`); absent: flat shading
        self.mesh_materials = {}    # geom index -> (ntris,) int32 scene material per face (`usemtl <k>`, -1 = the object's); absent: none
        for i in range(H.pth_scene_num_meshes(h)):
            nt = H.pth_scene_mesh_ntris(h, i)
            g = H.pth_scene_mesh_geom(h, i)
            self.meshes[g] = np.frombuffer(C.string_at(H.pth_scene_mesh_tris(h, i), 36 * nt), np.float32).reshape(nt, 9).copy()
            nptr, mptr = H.pth_scene_mesh_normals(h, i), H.pth_scene_mesh_materials(h, i)
            if nptr:
                self.mesh_normals[g] = np.frombuffer(C.string_at(nptr, 36 * nt), np.float32).reshape(nt, 9).copy()
            if mptr:
                self.mesh_materials[g] = np.frombuffer(C.string_at(mptr, 4 * nt), np.int32).copy()
        w, hh = (int(v) for v in self.camera["resolution"][0])
        self.image = np.zeros((hh, w, 3), np.float32)   # RenderState::image (src/sceneStructs.h:53)

    def set_resolution(self, w, h):
        """RES override; fov.x is recomputed as src/scene.cpp:133-136 does."""
        host_lib().pth_scene_set_resolution(self._h, w, h)
        self._refresh()

    def __del__(self):
        if getattr(self, "_h", None) and _host is not None:   # _host is gone during interpreter shutdown
            _host.pth_scene_free(self._h)
            self._h = None


# --------------------------------------------------------------------------- renderer API
_scene = None


def pathtraceInit(scene, shard_rank=0, shard_count=1, stream=0, accum_dev=0, device=-1, flags=0, traceDepth=None,
                  pipeline_depth=0, max_batch=0, lens_radius=0.0, focal_distance=0.0, direct_lighting=False, trace_ahead=False,
                  mixture_weighted=False):
    """reference src/pathtrace.cu:75-85.  `scene` is borrowed until pathtraceFree().
    lens_radius / focal_distance / direct_lighting: the README extras (depth of field, direct lighting), off by default.
    trace_ahead: PT_FLAG_TRACE_AHEAD -- pathtrace(pbo, frame, iter) called once per iteration draws on batches of max_batch
    iterations traced ahead (same image, bit for bit)."""
    global _scene
    if direct_lighting:
        flags |= PT_FLAG_DIRECT_LIGHTING
    if trace_ahead:
        flags |= PT_FLAG_TRACE_AHEAD
    if mixture_weighted:        # the REFL > 0 mixture with its 1 / p weights (src/interactions.h:54-58 to the letter; default: without)
        flags |= PT_FLAG_MIXTURE_WEIGHTED
    opt = PtOptions(shard_rank, shard_count, device, flags, pipeline_depth, max_batch, stream or None, accum_dev or None,
                    lens_radius, focal_distance)
    geoms = np.ascontiguousarray(scene.geoms)
    mats = np.ascontiguousarray(scene.materials)
    cam = np.ascontiguousarray(scene.camera)
    depth = scene.traceDepth if traceDepth is None else traceDepth
    set_meshes(getattr(scene, "meshes", None) or {}, getattr(scene, "mesh_normals", None), getattr(scene, "mesh_materials", None))
    global _atexit_registered
    if not _atexit_registered:
        # an interpreter that exits with a live renderer (an exception between pathtrace and pathtraceFree): drain the streams
        # and release everything while the interpreter, torch and the HIP runtime are still whole (the library registers its own
        # exit handler as well, for hosts that are not Python)
        import atexit
        atexit.register(pathtraceFree)
        _atexit_registered = True
    _check(lib().pt_init(_p(cam), _p(geoms), len(geoms), _p(mats), len(mats), depth, C.byref(opt)))
    _scene = scene
    global _last_init
    _last_init = (cam, geoms, mats, depth, (shard_rank, shard_count, device, flags, lens_radius, focal_distance),
                  (getattr(scene, "meshes", None) or {}, getattr(scene, "mesh_normals", None), getattr(scene, "mesh_materials", None)))


def _mesh_array(meshes, normals=None, materials=None):
    """(PtMesh array, count, the arrays it points into) for pt_set_meshes / pt_group_set_meshes"""
    keep = [(int(g), np.ascontiguousarray(t, np.float32).reshape(-1, 9)) for g, t in sorted(meshes.items())]
    extra = []                                       # (keeps the attribute arrays alive across the call)
    arr = (PtMesh * max(len(keep), 1))()
    for i, (g, t) in enumerate(keep):
        nn = (normals or {}).get(g)
        mm = (materials or {}).get(g)
        nn = None if nn is None else np.ascontiguousarray(nn, np.float32).reshape(len(t), 9)
        mm = None if mm is None else np.ascontiguousarray(mm, np.int32).reshape(len(t))
        extra += [nn, mm]
        arr[i] = PtMesh(g, len(t), t.ctypes.data, None if nn is None else nn.ctypes.data, None if mm is None else mm.ctypes.data)
    return arr, len(keep), (keep, extra)


def set_meshes(meshes, normals=None, materials=None, L=None):
    """pt_set_meshes: {geom index: (ntris, 9) triangles in object space} for the next pathtraceInit (an empty dict clears); optionally
    {geom index: (ntris, 9) vertex normals} and {geom index: (ntris,) int32 face materials}.  Through the sized entry point: a
    library built against another PtMesh refuses instead of striding through garbage."""
    arr, n, alive = _mesh_array(meshes, normals, materials)
    L = L or lib()
    _check(L.pt_set_meshes_sized(arr, n, C.sizeof(PtMesh)), L)


def pathtrace(pbo, frame, iteration, readback=True):
    """reference src/pathtrace.cu:123-174: one iteration; `pbo` is a DEVICE pointer (int) or None.
    With readback=True the running sum lands in scene.image like the reference's D2H copy (:170-171)."""
    _check(lib().pt_iterate(frame, iteration, pbo or None))
    if readback and _scene is not None:
        _check(lib().pt_readback(_p(_scene.image)))


def pathtrace_batch(pbo, frame, first_iteration, count, readback=False):
    """`count` iterations in one wavefront (pt_iterate_batch); same result as `count` pathtrace() calls."""
    _check(lib().pt_iterate_batch(frame, first_iteration, count, pbo or None))
    if readback and _scene is not None:
        _check(lib().pt_readback(_p(_scene.image)))


def pathtraceFree():
    """reference src/pathtrace.cu:87-92; legal before the first pathtraceInit (src/main.cpp:91-94)."""
    global _scene
    if _lib is not None:          # (nothing to free in a library that was never loaded)
        _lib.pt_free()
    _scene = None


def sync():
    _check(lib().pt_sync())


def force_fault(which):
    """Diagnostics (test library only, see renderer_from_test_library): set the renderer's device fault word by hand (2), or clear it (0)."""
    _tcheck(test_lib().pt_test_force_fault(which))


def readback(npixels):
    """Un-normalised running sum of the WHOLE frame (W*H*3 floats; with PT_FLAG_ACCUM_SHARD_ROWS the other shards'
    rows are zero).  `npixels` must be the frame's pixel count: pt_readback always writes that many."""
    if _scene is not None:
        res = _scene.camera["resolution"][0]
        if int(res[0]) * int(res[1]) != npixels:
            raise PtError("readback: the frame has %d pixels, not %d" % (int(res[0]) * int(res[1]), npixels))
    out = np.empty(npixels * 3, np.float32)
    _check(lib().pt_readback(_p(out)))
    return out


def readback_rgba8(iteration, npixels):
    out = np.empty((npixels, 4), np.uint8)
    _check(lib().pt_readback_rgba8(iteration, _p(out)))
    return out


def counters():
    c = PtCounters()
    _check(lib().pt_counters(C.byref(c)))
    return c


def counters_reset():
    _check(lib().pt_counters_reset())


def debug_trace_paths(iteration, bounces, npixels):
    """State of the paths alive after `bounces` bounces of `iteration` (sorted by pixel).  A diagnostic of the TEST library (the product
    exports none since round 5): the scene of the last pathtraceInit is initialised in libpt_amd_test.so's own renderer -- the same
    kernels, its own instance -- traced there and freed again; the product's renderer is not touched."""
    if _last_init is None:
        raise PtError("debug_trace_paths before pathtraceInit")
    T = test_lib()
    cam, geoms, mats, depth, (rank, count, device, flags, lens_radius, focal_distance), meshes = _last_init
    if _lib is T:                                    # (renderer_from_test_library: the renderer IS the test library's)
        own = False
    else:
        own = True
        set_meshes(*meshes, L=T)
        opt = PtOptions(rank, count, device, flags & ~(PT_FLAG_TRACE_AHEAD | PT_FLAG_KERNEL_TIMING), 1, 1, None, None, lens_radius, focal_distance)
        _tcheck(T.pt_init(_p(cam), _p(geoms), len(geoms), _p(mats), len(mats), depth, C.byref(opt)))
    try:
        o, d, c = (np.empty((npixels, 3), np.float32) for _ in range(3))
        pix = np.empty(npixels, np.int32)
        n = C.c_int32(0)
        _tcheck(T.pt_debug_trace_paths(iteration, bounces, _p(o), _p(d), _p(c), _p(pix), C.byref(n)))
    finally:
        if own:
            T.pt_free()
    k = n.value
    return o[:k], d[:k], c[:k], pix[:k]


class Context:
    """pt_ctx_*: a renderer instance of its own.  `with ctx:` makes it the calling thread's current context -- the renderer API of
    this module then drives it -- and puts the previous one back."""

    def __init__(self):
        self.handle = lib().pt_ctx_create()
        if not self.handle:
            raise PtError("pt_ctx_create failed: %s" % lib().pt_last_error().decode())

    def __enter__(self):
        self._prev = lib().pt_ctx_current()
        _check(lib().pt_ctx_make_current(self.handle))
        return self

    def __exit__(self, *exc):
        _check(lib().pt_ctx_make_current(self._prev))
        return False

    def destroy(self):
        if self.handle:
            _check(lib().pt_ctx_destroy(self.handle))
            self.handle = None


class Group:
    """pt_group_*: n contexts rendering the row shards y % n of one frame, member i on devices[i] (default: i % device count)."""

    def __init__(self, n, devices=None):
        h = C.c_void_p()
        dv = None if devices is None else (C.c_int32 * n)(*devices)
        _check(lib().pt_group_create(C.byref(h), n, dv))
        self.handle, self.n, self._scene = h, n, None

    @property
    def collective(self):
        return lib().pt_group_collective(self.handle).decode()

    def init(self, scene, traceDepth=None, flags=0, pipeline_depth=0, max_batch=0, lens_radius=0.0, focal_distance=0.0):
        geoms, mats, cam = (np.ascontiguousarray(x) for x in (scene.geoms, scene.materials, scene.camera))
        arr, nm, alive = _mesh_array(getattr(scene, "meshes", None) or {}, getattr(scene, "mesh_normals", None), getattr(scene, "mesh_materials", None))
        _check(lib().pt_group_set_meshes(self.handle, arr, nm))
        opt = PtOptions(0, 1, -1, flags, pipeline_depth, max_batch, None, None, lens_radius, focal_distance)
        _check(lib().pt_group_init(self.handle, _p(cam), _p(geoms), len(geoms), _p(mats), len(mats),
                                   scene.traceDepth if traceDepth is None else traceDepth, C.byref(opt)))
        self._scene = scene

    def iterate_batch(self, first_iteration, count, frame=0):
        _check(lib().pt_group_iterate_batch(self.handle, frame, first_iteration, count))

    def iterate(self, iteration, frame=0):
        """config C3 as written: one iteration on every member, then the frame's (asynchronous) assembly"""
        _check(lib().pt_group_iterate(self.handle, frame, iteration))

    def reduce(self):
        """assemble the frame from what has been committed so far (asynchronous)"""
        _check(lib().pt_group_reduce(self.handle))

    def sync(self):
        _check(lib().pt_group_sync(self.handle))

    def readback(self):
        res = self._scene.camera["resolution"][0]
        out = np.empty(int(res[0]) * int(res[1]) * 3, np.float32)
        _check(lib().pt_group_readback(self.handle, _p(out)))
        return out

    def counters(self):
        c = PtCounters()
        _check(lib().pt_group_counters(self.handle, C.byref(c)))
        return c

    def destroy(self):
        if self.handle:
            lib().pt_group_destroy(self.handle)
            self.handle = None


def save_png(basename, image_sum, samples):
    """saveImage + image::savePNG (reference src/main.cpp:49-70, src/image.cpp:22-39): divide by the
    sample count, mirror X, clamp, x255, truncate; writes <basename>.png."""
    img = np.ascontiguousarray(image_sum, np.float32)
    h, w = img.shape[0], img.shape[1]
    rc = host_lib().pth_save_png(os.fsencode(basename), _p(img), w, h, C.c_float(samples))
    if rc != 0:
        raise IOError("could not write %s.png" % basename)


def save_hdr(basename, image_sum, samples):
    """saveImage + image::saveHDR (reference src/main.cpp:49-70, src/image.cpp:41-45): writes <basename>.hdr."""
    img = np.ascontiguousarray(image_sum, np.float32)
    h, w = img.shape[0], img.shape[1]
    if host_lib().pth_save_hdr(os.fsencode(basename), _p(img), w, h, C.c_float(samples)) != 0:
        raise IOError("could not write %s.hdr" % basename)


# --------------------------------------------------------------------------- primitive / scan entry points (host arrays)
def test_utilhash(x):
    x = np.ascontiguousarray(x, np.uint32)
    out = np.empty_like(x)
    _tcheck(test_lib().pt_test_utilhash(_p(x), _p(out), x.size))
    return out


def test_rng(seeds, ndraws):
    seeds = np.ascontiguousarray(seeds, np.uint32)
    out = np.empty((seeds.size, ndraws), np.float32)
    _tcheck(test_lib().pt_test_rng(_p(seeds), seeds.size, ndraws, _p(out)))
    return out


def test_intersect(geoms, geom_index, rays, sentinel=-7.0):
    geoms = np.ascontiguousarray(geoms)
    gi = np.ascontiguousarray(geom_index, np.int32)
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
    n = len(rays)
    t = np.empty(n, np.float32)
    p = np.full((n, 3), sentinel, np.float32)
    nn = np.full((n, 3), sentinel, np.float32)
    o = np.ones(n, np.int32)
    _tcheck(test_lib().pt_test_intersect(_p(geoms), len(geoms), _p(gi), _p(rays), n, _p(t), _p(p), _p(nn), _p(o)))
    return t, p, nn, o


def test_mesh_intersect(geom, tris, rays, flat=False, sentinel=-7.0):
    """Rays against one mesh geom on the GPU (hierarchy, or flat=True: plain triangle list).  Returns t, p, n, outside, culled."""
    g = np.ascontiguousarray(geom).reshape(-1)[:1]
    tr = np.ascontiguousarray(tris, np.float32).reshape(-1, 9)
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
    n = len(rays)
    t = np.empty(n, np.float32)
    p = np.full((n, 3), sentinel, np.float32)
    nn = np.full((n, 3), sentinel, np.float32)
    o = np.ones(n, np.int32)
    culled = np.zeros(n, np.int32)
    _tcheck(test_lib().pt_test_mesh_intersect(_p(g), _p(tr), len(tr), 1 if flat else 0, _p(rays), n, _p(t), _p(p), _p(nn), _p(o), _p(culled)))
    return t, p, nn, o, culled


def test_mesh_cull_sweep(geom, tris, seed, rays):
    """Device sweep of the bounding-ball test of one mesh geom.  Returns (culled, violations, hits)."""
    g = np.ascontiguousarray(geom).reshape(-1)[:1]
    tr = np.ascontiguousarray(tris, np.float32).reshape(-1, 9)
    c, v, h = C.c_uint64(), C.c_uint64(), C.c_uint64()
    _tcheck(test_lib().pt_test_mesh_cull_sweep(_p(g), _p(tr), len(tr), seed, rays, C.byref(c), C.byref(v), C.byref(h)))
    return c.value, v.value, h.value


MESH_LEAF = 0x80000000
# the two kinds of 64-byte record of a mesh (pt_device.h: MeshRec)
MESH_TRI_DTYPE = np.dtype([("v0", "<f4", 3), ("v1", "<f4", 3), ("v2", "<f4", 3), ("margin", "<f4"), ("pad", "<u4", 2)])
MESH_NODE_DTYPE = np.dtype([("planes", "<f2", 6), ("ref", "<u4"), ("far_planes", "<f2", 6), ("far_ref", "<u4")])
MESH_TRI_UNITS, MESH_NODE_UNITS = 3, 2


def mesh_bvh(tris, octant=0):
    """The hierarchy pt_init builds for a mesh (host only), in the layout for rays of direction octant `octant` (bit a set:
    component a negative) -> (triangle records [ntris], inner nodes [max(ntris - 1, 1)], first unit of the inner nodes, stack
    levels a lane needs).  A node holds the boxes and refs of its near (planes, ref) and far (far_planes, far_ref) child, a box
    as six half-precision planes: the three a ray of the octant enters through (lo where its direction is positive, hi where
    negative), then the three it leaves through; lo rounded down, hi up.  Refs count units of 16 bytes: with MESH_LEAF set it is
    triangle (ref & ~MESH_LEAF) / 3, any other is inner node (ref - first unit) / 2."""
    tr = np.ascontiguousarray(tris, np.float32).reshape(-1, 9)
    nt = len(tr)
    out = np.zeros((5 * nt + 4, 4), np.uint32)
    n, need = C.c_int(len(out)), C.c_int(0)
    _tcheck(test_lib().pt_test_mesh_bvh(_p(tr), nt, octant, _p(out), C.byref(n), C.byref(need)))
    first = 3 * nt + (3 * nt) % 2
    assert n.value == first + 2 * max(nt - 1, 1)
    return (out[:3 * nt].reshape(-1).view(MESH_TRI_DTYPE).reshape(-1), out[first:n.value].reshape(-1).view(MESH_NODE_DTYPE).reshape(-1),
            first, need.value)


def test_hemisphere(normals, iter_index_depth):
    normals = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
    iid = np.ascontiguousarray(iter_index_depth, np.int32).reshape(-1, 3)
    out = np.empty_like(normals)
    _tcheck(test_lib().pt_test_hemisphere(_p(normals), _p(iid), len(normals), _p(out)))
    return out


def test_sincos(x):
    x = np.ascontiguousarray(x, np.float32)
    s, c = np.empty_like(x), np.empty_like(x)
    _tcheck(test_lib().pt_test_sincos(_p(x), x.size, _p(s), _p(c)))
    return s, c


def test_pow(x, e):
    x = np.ascontiguousarray(x, np.float32)
    e = np.ascontiguousarray(e, np.float32)
    out = np.empty_like(x)
    _tcheck(test_lib().pt_test_pow(_p(x), _p(e), x.size, _p(out)))
    return out


def test_reflect_refract(I, N, eta):
    I = np.ascontiguousarray(I, np.float32).reshape(-1, 3)
    N = np.ascontiguousarray(N, np.float32).reshape(-1, 3)
    eta = np.ascontiguousarray(eta, np.float32)
    r1, r2 = np.empty_like(I), np.empty_like(I)
    _tcheck(test_lib().pt_test_reflect_refract(_p(I), _p(N), _p(eta), len(I), _p(r1), _p(r2)))
    return r1, r2


def test_slab_quotients(o, d):
    o = np.ascontiguousarray(o, np.float32)
    d = np.ascontiguousarray(d, np.float32)
    out = [np.empty_like(o) for _ in range(4)]
    _tcheck(test_lib().pt_test_slab_quotients(_p(o), _p(d), o.size, *[_p(a) for a in out]))
    return out


def test_slab_quotients_sweep(seed, pairs):
    m = C.c_uint64(0)
    _tcheck(test_lib().pt_test_slab_quotients_sweep(seed, pairs, C.byref(m)))
    return int(m.value)


def test_unscaled_sqrt_sweep():
    m = (C.c_uint64 * 4)()
    _tcheck(test_lib().pt_test_unscaled_sqrt_sweep(m))
    return [int(v) for v in m]


def test_sphere_cull_sweep(geoms, seed, rays):
    geoms = np.ascontiguousarray(geoms)
    culled, bad = C.c_uint64(0), C.c_uint64(0)
    _tcheck(test_lib().pt_test_sphere_cull_sweep(_p(geoms), len(geoms), seed, rays, C.byref(culled), C.byref(bad)))
    return int(culled.value), int(bad.value)


def test_sphere_halfline_sweep(geoms, seed, rays):
    """-> (certified misses, of them with the centre behind the origin, violations)"""
    geoms = np.ascontiguousarray(geoms)
    culled, behind, bad = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    _tcheck(test_lib().pt_test_sphere_halfline_sweep(_p(geoms), len(geoms), seed, rays, C.byref(culled), C.byref(behind), C.byref(bad)))
    return int(culled.value), int(behind.value), int(bad.value)


def test_sphere_cluster_sweep(geoms, seed, rays):
    """geoms: a whole scene (spheres and cubes) -> (certificates issued per cluster [2], violations, {omax, n0, boxes[2][6]})"""
    geoms = np.ascontiguousarray(geoms)
    cert = (C.c_uint64 * 2)()
    bad = C.c_uint64(0)
    info = np.zeros(18, np.float32)
    _tcheck(test_lib().pt_test_sphere_cluster_sweep(_p(geoms), len(geoms), seed, rays, cert, C.byref(bad), info.ctypes.data))
    boxes = info[2:].reshape(2, 8)[:, :6].copy()
    return [int(cert[0]), int(cert[1])], int(bad.value), {"omax": float(info[0]), "n0": int(info[1]), "boxes": boxes}


def test_sphere_group_sweep(geoms, seed, rays):
    """(group certificates issued, violations, groups): pt_test_sphere_group_sweep"""
    g = np.ascontiguousarray(geoms)
    cert, viol, ng = C.c_uint64(), C.c_uint64(), C.c_int32()
    _tcheck(test_lib().pt_test_sphere_group_sweep(_p(g), len(g), seed, rays, C.byref(cert), C.byref(viol), C.byref(ng)))
    return int(cert.value), int(viol.value), int(ng.value)


def sphere_clusters(geoms):
    """The two clusters of spheres pt_init builds for a sphere-heavy scene without meshes (host only).
    -> {omax, n0, boxes (2, 6): lo, hi}, table (primitive index per entry of the sweep's table: cluster 0 first, n0 entries)"""
    geoms = np.ascontiguousarray(geoms)
    info = np.zeros(18, np.float32)
    table = np.zeros(len(geoms) + 2, np.int32)
    n = C.c_int32(0)
    _tcheck(test_lib().pt_test_sphere_clusters(_p(geoms), len(geoms), info.ctypes.data, table.ctypes.data, len(table), C.byref(n)))
    return {"omax": float(info[0]), "n0": int(info[1]), "boxes": info[2:].reshape(2, 8)[:, :6].copy()}, table[:n.value].copy()


def test_wall_box_sweep(geoms, seed, rays):
    geoms = np.ascontiguousarray(geoms)
    culled, bad = C.c_uint64(0), C.c_uint64(0)
    _tcheck(test_lib().pt_test_wall_box_sweep(_p(geoms), len(geoms), seed, rays, C.byref(culled), C.byref(bad)))
    return int(culled.value), int(bad.value)


def camera_cull_tables(camera, geoms):
    """The camera-ray culling tables pt_init derives (host only).  Returns (rects (n, 4), scene rect (4,), spans (H, n, 2))."""
    cam = np.ascontiguousarray(camera)
    geoms = np.ascontiguousarray(geoms)
    H = int(cam["resolution"][0][1])
    rects = np.zeros((len(geoms), 4), np.int32)
    scene = np.zeros(4, np.int32)
    spans = np.zeros((H, len(geoms), 2), np.int32)
    _tcheck(test_lib().pt_test_camera_cull_tables(_p(cam), _p(geoms), len(geoms), _p(rects), _p(scene), _p(spans)))
    return rects, scene, spans


def test_camera_cull_sweep(camera, geoms, samples=1):
    """Device sweep of the camera-ray culling for one (camera, primitive set).  Returns (hits, culled pairs, violations)."""
    cam = np.ascontiguousarray(camera)
    geoms = np.ascontiguousarray(geoms)
    h, c, v = C.c_uint64(), C.c_uint64(), C.c_uint64()
    _tcheck(test_lib().pt_test_camera_cull_sweep(_p(cam), _p(geoms), len(geoms), samples, C.byref(h), C.byref(c), C.byref(v)))
    return int(h.value), int(c.value), int(v.value)


def test_camera_cull_margin(camera, geoms, samples=1):
    """Device measurement of the culling tables' inflation margin for one (camera, primitive set).  Returns (largest fraction of the
    inflation a hit of the reference's test needed, hits that needed any)."""
    cam = np.ascontiguousarray(camera)
    geoms = np.ascontiguousarray(geoms)
    w, n = C.c_double(), C.c_uint64()
    _tcheck(test_lib().pt_test_camera_cull_margin(_p(cam), _p(geoms), len(geoms), samples, C.byref(w), C.byref(n)))
    return float(w.value), int(n.value)


def test_box_fast_sweep(geoms, seed, rays):
    """Device sweep of the box test's fast slab phase against the reference's loop.  Returns (rays, decided by the fast path, hits
    among those, bit mismatches, mismatches of the fast reciprocal, mismatches of the fast quotient)."""
    geoms = np.ascontiguousarray(geoms)
    c = (C.c_uint64 * 4)()
    d = (C.c_uint64 * 2)()
    _tcheck(test_lib().pt_test_box_fast_sweep(_p(geoms), len(geoms), seed, rays, c, d))
    return tuple(int(v) for v in c) + tuple(int(v) for v in d)


def test_wall_plane_sweep(geoms, seed, rays):
    """Device sweep of the one-plane-per-wall certificates.  Returns (walls with a plane, certified, violations, single-wall rays)."""
    geoms = np.ascontiguousarray(geoms)
    n = C.c_int32(0)
    c, v, s1 = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    _tcheck(test_lib().pt_test_wall_plane_sweep(_p(geoms), len(geoms), seed, rays, C.byref(n), C.byref(c), C.byref(v), C.byref(s1)))
    return int(n.value), int(c.value), int(v.value), int(s1.value)


def test_wall_planes(geoms):
    """Host only (no GPU): the planes pt_init keeps for rotated walls.  Returns (planes [nplane][5] = normal, threshold, far; wall_geom, nslot, nwalls)."""
    geoms = np.ascontiguousarray(geoms)
    planes = np.zeros((6, 5), np.float32)
    wg = np.zeros(6, np.int32)
    ns, npl, nw = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    _tcheck(test_lib().pt_test_wall_planes(_p(geoms), len(geoms), _p(planes), _p(wg), C.byref(ns), C.byref(npl), C.byref(nw)))
    return planes[:npl.value].copy(), wg[:nw.value].copy(), int(ns.value), int(nw.value)


def scan_exclusive_dev(in_ptr, out_ptr, n, stream=0):
    _check(lib().pt_scan_exclusive_i32(in_ptr, out_ptr, n, stream or None))


def compact_nonzero_dev(in_ptr, out_ptr, n, count_ptr, stream=0):
    _check(lib().pt_compact_nonzero_i32(in_ptr, out_ptr, n, count_ptr, stream or None))
