"""CPU-only tests of the product's host side: C-ABI surface, scene loader, PNG writer, headless
driver protocol.  No compute call is made (no GPU here); the oracle is used only as a checker."""
import ctypes as C
import json
import os
import re
import struct
import subprocess
import zlib

import numpy as np
import pytest

from conftest import GOLD, ROOT, SCENES


def _exported(path):
    """EVERY dynamic symbol the library defines, whatever its kind (functions, data, weak template instantiations, kernel handles)"""
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
    return sorted(l.split()[2] for l in out.splitlines() if len(l.split()) == 3)


def test_abi_library_exports_every_declared_symbol(pt):
    hdr = open(os.path.join(ROOT, "include", "pt_amd.h")).read()
    declared = sorted(set(re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", hdr)))
    assert declared == sorted(pt.ABI_SYMBOLS)
    L = C.CDLL(pt.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    # ... and NOTHING else: no test hook, no probe, no fault injection -- and (round 6: csrc/pt_amd.map) no kernel handle, no STL
    # instantiation, no class of the library's own among the defined dynamic symbols of ANY kind
    assert _exported(pt.LIB_PATH) == declared


def test_test_library_adds_exactly_the_test_header(pt):
    hdr = open(os.path.join(ROOT, "include", "pt_amd_test.h")).read()
    # (pt_test_*, and the renderer diagnostic pt_debug_trace_paths, which the product exported until round 5)
    declared = sorted(set(re.findall(r"\b(pt_(?:test|debug)_[a-z0-9_]+)\s*\(", hdr)))
    assert declared == sorted(pt.TEST_ABI_SYMBOLS)
    assert _exported(pt.TEST_LIB_PATH) == sorted(pt.ABI_SYMBOLS + pt.TEST_ABI_SYMBOLS)


def test_struct_sizes_match_reference_layout(pt):
    # Ray 24, Geom 236, Material 44, Camera 52 and the field offsets
    g, m, c = pt.GEOM_DTYPE, pt.MATERIAL_DTYPE, pt.CAMERA_DTYPE
    assert [g.fields[k][1] for k in ("type", "materialid", "translation", "rotation", "scale", "transform",
                                     "inverseTransform", "invTranspose")] == [0, 4, 8, 20, 32, 44, 108, 172]
    assert [m.fields[k][1] for k in ("color", "specularExponent", "specularColor", "hasReflective", "hasRefractive",
                                     "indexOfRefraction", "emittance")] == [0, 12, 16, 28, 32, 36, 40]
    assert [c.fields[k][1] for k in ("resolution", "position", "view", "up", "fov")] == [0, 8, 20, 32, 44]


def test_no_gpu_means_loud_failure_not_fallback(pt):
    if pt.device_count() > 0:
        pytest.skip("GPU present")
    sc = pt.Scene(os.path.join(SCENES, "sphere.txt"))
    with pytest.raises(pt.PtError, match="no HIP device"):
        pt.pathtraceInit(sc)
    with pytest.raises(pt.PtError, match="before pt_init"):
        pt.pathtrace(None, 0, 1)
    with pytest.raises(pt.PtError):
        pt.test_utilhash(np.arange(4, dtype=np.uint32))
    pt.pathtraceFree()  # legal before any Init (src/main.cpp:91-94)


@pytest.mark.parametrize("name", ["cornell", "sphere", "cornell_glass", "spheres64", "rotated"])
def test_scene_loader_matches_reference_loader(pt, name):
    z = np.load(os.path.join(GOLD, f"scene_{name}.npz"))
    sc = pt.Scene(os.path.join(SCENES, f"{name}.txt"))
    meta = json.loads(str(z["meta"]))
    assert sc.geoms.tobytes() == z["geoms"].tobytes()
    assert sc.materials.tobytes() == z["materials"].tobytes()
    assert sc.camera.tobytes() == z["camera"].tobytes()
    assert (sc.iterations, sc.traceDepth, sc.imageName) == (meta["iterations"], meta["depth"], meta["image_name"])
    assert sc.image.shape[0] * sc.image.shape[1] == meta["image_len"]


def test_scene_loader_agrees_with_oracle_on_overrides(pt, oracle):
    for res in ((1280, 720), (1920, 1080), (400, 400), (4096, 4096), (96, 64)):
        a = pt.Scene(os.path.join(SCENES, "cornell.txt"))
        b = oracle.Scene(os.path.join(SCENES, "cornell.txt"))
        a.set_resolution(*res)
        b.set_resolution(*res)
        assert a.camera.tobytes() == b.camera.tobytes()
    assert abs(float(a.camera["fov"][0][0]) - 60.6422424) > 1  # 96x64 is not 16:9 ...
    a.set_resolution(1280, 720)
    assert abs(float(a.camera["fov"][0][0]) - 60.6422424) < 5e-6  # ... 1280x720 is


def test_scene_format_quirks(pt, tmp_path):
    # CRLF line ends, no trailing newline, out-of-order ids skipped, unknown keywords ignored
    txt = ("MATERIAL 0\r\nRGB 1 0.5 0.25\r\nSPECEX 3\r\nSPECRGB 0 0 0\r\nREFL 0\r\nREFR 0\r\nREFRIOR 0\r\nEMITTANCE 2\r\n\r\n"
           "MATERIAL 5\r\nRGB 9 9 9\r\nSPECEX 0\r\nSPECRGB 0 0 0\r\nREFL 0\r\nREFR 0\r\nREFRIOR 0\r\nEMITTANCE 0\r\n\r\n"
           "CAMERA\r\nRES 32 16\r\nFOVY 30\r\nITERATIONS 7\r\nDEPTH 3\r\nFILE quirk\r\nEYE 1 2 3\r\nUP 0 1 0\r\nVIEW 0 0 -1\r\n\r\n"
           "OBJECT 0\r\ncube\r\nmaterial 0\r\nSCALE 2 2 2\r\nBOGUS 1 2 3\r\nTRANS 1 0 0\r\n\r\n"
           "OBJECT 0\r\nsphere\r\nmaterial 0\r\nTRANS 0 0 0")
    p = tmp_path / "quirk.txt"
    p.write_bytes(txt.encode())
    sc = pt.Scene(str(p))
    assert len(sc.materials) == 1 and len(sc.geoms) == 1      # id 5 and the duplicate id 0 are skipped
    assert sc.materials["color"][0].tolist() == [1.0, 0.5, 0.25] and sc.materials["specularExponent"][0] == 3
    assert sc.geoms["type"][0] == 1 and sc.geoms["scale"][0].tolist() == [2, 2, 2]
    assert sc.geoms["transform"][0].reshape(4, 4)[3].tolist() == [1, 0, 0, 1]   # column 3 = translation
    assert (sc.iterations, sc.traceDepth, sc.imageName) == (7, 3, "quirk")
    assert sc.camera["position"][0].tolist() == [1, 2, 3]
    with pytest.raises(IOError):
        pt.Scene(str(tmp_path / "missing.txt"))


def _decode_png(path):
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w = 8, b"", None
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert zlib.crc32(tag + body) & 0xFFFFFFFF == struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0]
        if tag == b"IHDR":
            w, h, depth, ctype = struct.unpack(">IIBB", body[:10])
            assert (depth, ctype) == (8, 2)
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    raw = zlib.decompress(idat)
    rows = np.frombuffer(raw, np.uint8).reshape(h, 3 * w + 1)
    assert np.all(rows[:, 0] == 0)
    return rows[:, 1:].reshape(h, w, 3)


def test_save_png_semantics(pt, tmp_path):
    # saveImage: divide by samples, mirror X (src/main.cpp:58); savePNG: clamp, x255, truncate (image.cpp:27-30)
    w, h, samples = 5, 3, 4
    rng = np.random.default_rng(1)
    img = (rng.random((h, w, 3)) * 6 - 0.5).astype(np.float32)
    base = str(tmp_path / "out")
    pt.save_png(base, img, samples)
    got = _decode_png(base + ".png")
    want = (np.clip(img / np.float32(samples), 0, 1) * np.float32(255.0)).astype(np.uint8)[:, ::-1]
    assert np.array_equal(got, want)


def test_save_hdr_round_trip(pt, tmp_path):
    # Radiance RGBE: shared exponent, 8-bit mantissas -> relative error below 1/128 of the brightest channel
    w, h, samples = 7, 4, 2
    rng = np.random.default_rng(2)
    img = (rng.random((h, w, 3)) * np.array([0.01, 5.0, 300.0])).astype(np.float32)
    img[0, 0] = 0
    base = str(tmp_path / "out")
    pt.save_hdr(base, img, samples)
    data = open(base + ".hdr", "rb").read()
    head, _, body = data.partition(b"\n\n")
    assert head.startswith(b"#?RADIANCE") and b"FORMAT=32-bit_rle_rgbe" in head
    dims, _, px = body.partition(b"\n")
    assert dims == b"-Y %d +X %d" % (h, w) and len(px) == 4 * w * h
    rgbe = np.frombuffer(px, np.uint8).reshape(h, w, 4).astype(np.float64)
    dec = rgbe[:, :, :3] / 256.0 * np.exp2(rgbe[:, :, 3:4] - 128.0) * (rgbe[:, :, 3:4] > 0)
    want = (img / np.float32(samples))[:, ::-1].astype(np.float64)      # X mirror like the PNG path
    tol = want.max(axis=2, keepdims=True) / 128.0 + 1e-30
    assert np.all(np.abs(dec - want) <= tol)


def test_headless_driver_without_gpu_reports_and_exits_nonzero(pt, tmp_path):
    if pt.device_count() > 0:
        pytest.skip("GPU present")
    exe = os.path.join(ROOT, "project3-cuda-path-tracer_amd", "host", "pt_render")
    r = subprocess.run([exe, os.path.join(SCENES, "sphere.txt"), "--res", "8", "8", "--iterations", "1"],
                       capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 1 and "pathtraceInit" in r.stderr and "no HIP device" in r.stderr


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "project3-cuda-path-tracer_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".cpp", ".hip", "Makefile")):
                src = open(os.path.join(dp, f), errors="ignore").read()
                code = "\n".join(l for l in src.split("\n") if "never touches oracle/" not in l)
                assert "pt_oracle" not in code and "libptoracle" not in code and "import oracle" not in code, f


def test_foreign_obj_material_names_fall_back_to_the_objects_material(pt, tmp_path, capfd):
    # `usemtl <k>` is a MATERIAL index of the scene file; an OBJ from elsewhere with numeric material names of its own
    # must not silently rebind faces (or fail much later, in pt_init): indices the scene does not have are dropped with a warning
    (tmp_path / "m.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\nusemtl 1\nf 1 2 3\nusemtl 7\nf 1 2 4\nusemtl 250\nf 1 3 4\nusemtl rusty_metal\nf 2 3 4\n")
    scene = ("MATERIAL 0\nRGB 1 1 1\nSPECEX 0\nSPECRGB 0 0 0\nREFL 0\nREFR 0\nREFRIOR 0\nEMITTANCE 5\n\n"
             "MATERIAL 1\nRGB .5 .5 .5\nSPECEX 0\nSPECRGB 0 0 0\nREFL 0\nREFR 0\nREFRIOR 0\nEMITTANCE 0\n\n"
             "CAMERA\nRES 8 8\nFOVY 45\nITERATIONS 1\nDEPTH 2\nFILE t\nEYE 0 0 5\nVIEW 0 0 -1\nUP 0 1 0\n\n"
             "OBJECT 0\nmesh m.obj\nmaterial 0\nTRANS 0 0 0\nROTAT 0 0 0\nSCALE 1 1 1\n")
    (tmp_path / "s.txt").write_text(scene)
    sc = pt.Scene(str(tmp_path / "s.txt"))
    assert sc.mesh_materials[0].tolist() == [1, -1, -1, -1]
    assert "2 faces name a material" in capfd.readouterr().out


def test_bench_group_mode_without_a_gpu_fails_loudly(pt):
    # `bench.py --group M` (the C ABI's device groups, what the `group` blocks of the driver's line run) has no CPU fallback either:
    # without a HIP device it says so and exits non-zero -- no line, no oracle standing in
    if pt.device_count() > 0:
        pytest.skip("GPU present")
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--group", "2", "--steps", "1", "--warmup", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and "needs a GPU" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_no_kernel_of_the_library_uses_scratch(tmp_path):
    # no instantiation of k_bounce / k_mesh_walk / k_commit may park registers in scratch memory (a spilled build of the
    # PLAIN bounce was 7 % slower).  Read from the code object itself: the gfx950 ELF inside the library's offload bundle, its AMDGPU metadata note.
    import re
    import struct
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    lib = os.path.join(ROOT, "project3-cuda-path-tracer_amd", "csrc", "libpt_amd.so")
    if not (os.path.exists(readelf) and os.path.exists(lib)):
        pytest.skip("llvm-readelf or the built library is missing")
    blob = open(lib, "rb").read()
    at = blob.find(b"__CLANG_OFFLOAD_BUNDLE__")
    assert at >= 0
    (entries,) = struct.unpack_from("<Q", blob, at + 24)
    pos, elf = at + 32, None
    for _ in range(entries):
        off, size, tl = struct.unpack_from("<QQQ", blob, pos)
        triple = blob[pos + 24:pos + 24 + tl].decode()
        pos += 24 + tl
        if "gfx950" in triple:
            elf = blob[at + off:at + off + size]
    assert elf, "no gfx950 code object in the library"
    (tmp_path / "co.elf").write_bytes(elf)
    notes = subprocess.run([readelf, "--notes", str(tmp_path / "co.elf")], capture_output=True, text=True, timeout=120).stdout
    names = re.findall(r"\.name:\s+(\S+)", notes)
    scratch = [int(x) for x in re.findall(r"\.private_segment_fixed_size:\s+(\d+)", notes)]
    assert len(names) == len(scratch) >= 30
    assert sum("k_bounce" in n for n in names) >= 20 and any("k_mesh_walk" in n for n in names) and any("k_commit" in n for n in names)
    assert [(n, s) for n, s in zip(names, scratch) if s != 0] == []


def test_bench_prints_its_help():
    # (a bare `%` in an option's help text made argparse raise while FORMATTING the help: `bench.py --help` ended in a traceback)
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "--gpus" in r.stdout and "--steps" in r.stdout and "Traceback" not in r.stderr
