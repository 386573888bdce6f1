"""The drop-in claim of INTEGRATION.md, tested at compile, link and run level.

oracle/_ref/ref_shim_driver = the product's shim (host/pathtrace_shim.cpp, unchanged) compiled against the REFERENCE's
own src/pathtrace.h:1-8, src/scene.h:13-26, src/sceneStructs.h and glm, together with the reference's own loader
(src/scene.cpp, src/utilities.cpp), linked with libpt_amd.so, under a main() that follows runCuda() of
src/main.cpp:72-113 call for call (oracle/ref_shim_driver.cpp).  It is built where /root/reference exists (the
authoring container); the binary travels to the GPU box."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, SCENES

DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_shim_driver")
HAVE_REFERENCE = os.path.isdir("/root/reference/src")


@pytest.mark.skipif(not HAVE_REFERENCE, reason="the reference sources exist only in the authoring container")
def test_shim_compiles_and_links_against_the_reference_headers(built):
    # `built` ran `make -C oracle ref`, which compiles the shim against /root/reference/src/{pathtrace.h,scene.h,...}
    assert os.path.exists(DRIVER)
    src = os.path.join(ROOT, "oracle", "_ref", "shimsrc", "pathtrace_shim.cpp")
    assert os.path.realpath(src) == os.path.join(ROOT, "project3-cuda-path-tracer_amd", "host", "pathtrace_shim.cpp")
    assert os.path.getmtime(DRIVER) >= os.path.getmtime(os.path.realpath(src))
    # the three reference symbols come from the shim, the renderer from libpt_amd.so
    syms = subprocess.run(["nm", "-C", DRIVER], capture_output=True, text=True).stdout
    for s in ("T pathtraceInit(Scene*)", "T pathtraceFree()", "T pathtrace(uchar4*, int, int)", "U pt_init", "U pt_iterate", "U pt_readback"):
        assert s in syms, s
    assert "T Scene::Scene(" in syms and "T Scene::loadGeom(" in syms          # the reference's own loader


@pytest.mark.skipif(not HAVE_REFERENCE, reason="the reference sources exist only in the authoring container")
def test_runcuda_call_order_reaches_the_renderer_and_fails_loudly_without_a_gpu(built, tmp_path, pt):
    if pt.device_count() > 0:
        pytest.skip("a GPU is present: tests/test_gpu_parity.py runs the driver for real")
    r = subprocess.run([DRIVER, os.path.join(SCENES, "cornell.txt"), "2", str(tmp_path / "o.bin")], capture_output=True, text=True)
    # the reference's loader parsed the scene, runCuda's Free -> Init reached pt_init through the shim, and the missing
    # device is reported the way checkCUDAError does: message on stderr, exit(EXIT_FAILURE) -- never a CPU fallback
    assert "Loaded camera!" in r.stdout and "Connecting Geom 6 to Material 4" in r.stdout
    assert r.returncode == 1
    assert "pathtraceInit: pt_init: no HIP device" in r.stderr
    assert not os.path.exists(tmp_path / "o.bin")


def _read_dump(path, oracle):
    raw = open(path, "rb").read()
    cam = np.frombuffer(raw[:52], oracle.CAMERA_DTYPE).copy()
    W, H = (int(v) for v in cam["resolution"][0])
    img = np.frombuffer(raw[52:], np.float32)
    assert img.size == W * H * 3
    return cam, W, H, img


@pytest.mark.gpu
@pytest.mark.parametrize("move,members", [(None, 1), (("2", "0.5", "0.2", "-1", "0.05", "-0.1"), 1), (None, 2), (("2", "0.5", "0.2", "-1", "0.05", "-0.1"), 3)])
def test_reference_host_code_drives_the_hip_path(pt, oracle, tmp_path, move, members):
    """The reference's Scene loader + headers + runCuda order over the shim, on the GPU: the image it leaves in
    scene->state.image equals the oracle's bit for bit -- also after a camera move, which restarts the accumulation
    (iteration = 0 -> pathtraceFree(); pathtraceInit(scene), src/main.cpp:73-94) with the camera the reference's own glm
    arithmetic produced."""
    if pt.device_count() < 1:
        pytest.fail("no HIP device: GPU tests must run on the MI355X box")
    if not os.path.exists(DRIVER):
        pytest.skip("oracle/_ref/ref_shim_driver is built only where /root/reference exists")
    out = str(tmp_path / "o.bin")
    # members > 1: PT_AMD_DEVICES -- the unchanged reference host over a GROUP of renderers that share the frame's rows (pt_group_*:
    # one per device of a node; on this one-GPU box they share the device): the same image, bit for bit
    env = dict(os.environ)
    env.pop("PT_AMD_DEVICES", None)
    if members > 1:
        env["PT_AMD_DEVICES"] = str(members)
    r = subprocess.run([DRIVER, os.path.join(SCENES, "cornell.txt"), "3", out] + (list(move) if move else []),
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert ("%d pathtrace calls" % (5 if move else 3)) in r.stdout and "last iteration 3" in r.stdout
    cam, W, H, got = _read_dump(out, oracle)
    assert (W, H) == (800, 800)                                   # scenes/cornell.txt as shipped (RES 800 800)
    sc = oracle.Scene(os.path.join(SCENES, "cornell.txt"))
    if move:
        assert not np.array_equal(cam["position"], sc.camera["position"]) and not np.array_equal(cam["view"], sc.camera["view"])
    else:
        assert cam.tobytes() == sc.camera.tobytes()               # the build's loader and the reference's agree on the camera
    ref = oracle.Renderer(cam, sc.geoms, sc.materials, 8)
    want = np.zeros(W * H * 3, np.float32)
    for it in (1, 2, 3):
        ref.iterate(it, want)
    assert want.max() > 0 and np.array_equal(got.view(np.uint32), want.view(np.uint32))
