"""A process that leaves with batches in flight and no pathtraceFree().

Round 2's 4-rank rehearsal once ended with "GPU core dump created"; the log was not kept and the cause was never identified.
What round 3 closed are the ways a process can go away while the renderer still has work on the GPU: the library's own exit
handler (registered at the first pt_init, so it runs BEFORE the HIP runtime's), the Python package's atexit hook, bench.py's
try / finally and -- round 4 -- its SIGTERM handler (torchrun ends surviving ranks with SIGTERM, which Python does not turn into
an exception by itself).  This file keeps each of them closed: a FRESH child process enqueues two wavefront batches and leaves
without freeing -- by returning from its main module, by sys.exit(3) in the middle, through the raw C ABI without the Python
package's hook, and killed by SIGTERM inside bench.py -- and must end with the expected code and nothing on stderr but the
runtime's known `amdgpu.ids` line; a following process then renders the same frame and finds the fault word clear.

The reference's exit order is pathtraceFree(); cudaDeviceReset(); exit (src/main.cpp:107-112).  Run once; never looped."""
import os
import signal
import subprocess
import sys
import time

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

PRELUDE = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np
import __graft_entry__ as ge
pt = ge.load_package()
sc = pt.Scene(os.path.join(%r, "scenes", "cornell.txt"))
sc.set_resolution(1280, 720)
""" % (ROOT, ROOT)

# (two batches of 32 iterations at 1280x720 are ~2 ms of GPU work on two internal streams: the process is gone long before)
CHILD_RETURNS = PRELUDE + r"""
pt.pathtraceInit(sc, traceDepth=8, pipeline_depth=2, max_batch=32)
pt.pathtrace_batch(None, 0, 1, 32)
pt.pathtrace_batch(None, 0, 33, 32)
# the main module ends here: no sync, no pathtraceFree
"""

CHILD_EXITS_3 = PRELUDE + r"""
pt.pathtraceInit(sc, traceDepth=8, pipeline_depth=2, max_batch=32)
pt.pathtrace_batch(None, 0, 1, 32)
pt.pathtrace_batch(None, 0, 33, 32)
sys.exit(3)
"""

# the raw C ABI, as a host that is not Python would use it: the package's atexit hook is never registered (pathtraceInit is not
# called), so what drains the streams at exit is the LIBRARY's own handler
CHILD_RAW_ABI = PRELUDE + r"""
import ctypes as C
L = pt.lib()
geoms, mats, cam = np.ascontiguousarray(sc.geoms), np.ascontiguousarray(sc.materials), np.ascontiguousarray(sc.camera)
opt = pt.PtOptions(0, 1, -1, 0, 2, 32, None, None, 0.0, 0.0)
assert L.pt_init(cam.ctypes.data, geoms.ctypes.data, len(geoms), mats.ctypes.data, len(mats), 8, C.byref(opt)) == 0, L.pt_last_error()
assert L.pt_iterate_batch(0, 1, 32, None) == 0
assert L.pt_iterate_batch(0, 33, 32, None) == 0
assert not pt._atexit_registered
"""

# a GROUP the host forgets (round 6): three members with an issuing thread each, the frame's assembly through a one-rank RCCL
# communicator, batches traced ahead -- the library's exit handler stops the threads, frees the members and the communicator
CHILD_FORGETS_GROUP = PRELUDE + r"""
os.environ["PT_AMD_COLLECTIVE"] = "rccl"
g = pt.Group(3)
g.init(sc, traceDepth=8, flags=pt.PT_FLAG_TRACE_AHEAD, pipeline_depth=2, max_batch=32)
for it in range(1, 4):
    g.iterate(it)
img = g.readback()
assert img.max() > 0
# the main module ends here: no sync, no pt_group_destroy
"""

CHILD_FOLLOWING = PRELUDE + r"""
pt.pathtraceInit(sc, traceDepth=8, pipeline_depth=2, max_batch=32)
pt.pathtrace_batch(None, 0, 1, 32)
pt.pathtrace_batch(None, 0, 33, 32)
pt.sync()                                   # (PT_ERR_DEVICE here if the fault word were set)
img = pt.readback(1280 * 720)
c = pt.counters()
assert c.iterations == 64 and c.live[1] == 64 * 1280 * 720 and img.max() > 0
np.save(sys.argv[1], img)
pt.pathtraceFree()
"""


def _clean(stderr):
    """stderr without the HIP runtime's known complaint about a file this image does not ship"""
    return [l for l in stderr.splitlines() if l.strip() and "amdgpu.ids" not in l]


def _run(code, *argv, timeout=600):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, "-c", code] + list(argv), capture_output=True, text=True, timeout=timeout, env=env)


def _no_dump(r):
    text = (r.stdout + r.stderr).lower()
    for word in ("core dump", "memory access fault", "hsa_status_error", "segmentation fault", "aborted"):
        assert word not in text, text[-2000:]


def test_processes_that_leave_with_batches_in_flight(pt, tmp_path):
    if pt.device_count() < 1:
        pytest.fail("no HIP device: GPU tests must run on the MI355X box")
    # 1. the main module simply ends (interpreter shutdown with a live renderer: the package's atexit hook, then the library's)
    r = _run(CHILD_RETURNS)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _clean(r.stderr) == [], r.stderr[-2000:]
    _no_dump(r)
    # 2. sys.exit(3) in the middle of a run
    r = _run(CHILD_EXITS_3)
    assert r.returncode == 3, r.stderr[-2000:]
    assert _clean(r.stderr) == [], r.stderr[-2000:]
    _no_dump(r)
    # 3. the raw C ABI without the package's hook: the library's exit handler alone
    r = _run(CHILD_RAW_ABI)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _clean(r.stderr) == [], r.stderr[-2000:]
    _no_dump(r)
    # 4. a group nobody destroyed: member threads, RCCL communicator, batches traced ahead
    r = _run(CHILD_FORGETS_GROUP)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _clean(r.stderr) == [], r.stderr[-2000:]
    _no_dump(r)
    # ... and a following process renders the same 64 iterations, with a clean fault word, to the frame this process renders
    out = str(tmp_path / "following.npy")
    r = _run(CHILD_FOLLOWING, out)
    assert r.returncode == 0, r.stderr[-2000:]
    _no_dump(r)
    sc = pt.Scene(os.path.join(ROOT, "scenes", "cornell.txt"))
    sc.set_resolution(1280, 720)
    pt.pathtraceFree()
    pt.pathtraceInit(sc, traceDepth=8, pipeline_depth=1, max_batch=64)
    pt.pathtrace_batch(None, 0, 1, 64)
    want = pt.readback(1280 * 720)
    pt.pathtraceFree()
    assert np.array_equal(np.load(out).view(np.uint32), want.view(np.uint32))


def test_bench_rank_killed_by_sigterm_frees_the_renderer_first(pt, tmp_path):
    # what torchrun does to the surviving ranks of a failed job: SIGTERM in the middle of the timed blocks.  bench.py turns it into
    # SystemExit, so cleanup() -- synchronise, pathtraceFree -- runs before the process goes away; exit code 128 + 15.
    if pt.device_count() < 1:
        pytest.fail("no HIP device: GPU tests must run on the MI355X box")
    mark = str(tmp_path / "timing.mark")
    env = dict(os.environ, BENCH_MARK_FILE=mark)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "400", "--warmup", "1", "--repeats", "200", "--cpu-spp", "0"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    try:
        t0 = time.time()
        while not os.path.exists(mark) and p.poll() is None and time.time() - t0 < 300:
            time.sleep(0.05)
        assert os.path.exists(mark) and p.poll() is None, "bench.py never reached its timed blocks"
        time.sleep(0.5)                              # (well inside them: 200 blocks of 400 steps are minutes of work)
        p.send_signal(signal.SIGTERM)
        out, err = p.communicate(timeout=120)
    finally:
        if p.poll() is None:
            p.kill()
    assert p.returncode == 128 + signal.SIGTERM, (p.returncode, err[-2000:])
    assert not [l for l in out.splitlines() if l.startswith("{")]        # no line from a run that did not finish
    text = (out + err).lower()
    for word in ("core dump", "memory access fault", "hsa_status_error", "segmentation fault"):
        assert word not in text, err[-2000:]
    # the GPU is fine afterwards: this process renders and reads the fault word
    sc = pt.Scene(os.path.join(ROOT, "scenes", "cornell.txt"))
    sc.set_resolution(320, 180)
    pt.pathtraceFree()
    pt.pathtraceInit(sc, traceDepth=8, max_batch=8)
    pt.pathtrace_batch(None, 0, 1, 8)
    pt.sync()
    assert pt.readback(320 * 180).max() > 0
    pt.pathtraceFree()
