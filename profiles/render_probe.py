"""Wall time of the headless driver pt_render (the reference's main() / runCuda() / saveImage() through the shim) on the GPU
box: the per-iteration protocol call by call (PT_AMD_TRACE_AHEAD=0), with trace-ahead (the shim's default) and through
pt_iterate_batch (--batch 32: no per-iteration copy).  The PNGs of one configuration must be the same file byte for byte.
    python profiles/render_probe.py"""
import glob, hashlib, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = os.path.join(ROOT, "project3-cuda-path-tracer_amd", "host", "pt_render")
SC = os.path.join(ROOT, "scenes", "cornell.txt")
def run(extra, env=None):
    with tempfile.TemporaryDirectory() as d:
        t0 = time.perf_counter()
        r = subprocess.run([R, SC, "--out", os.path.join(d, "img")] + extra, capture_output=True, text=True, env=dict(os.environ, **(env or {})))
        dt = time.perf_counter() - t0
        if r.returncode != 0:
            sys.exit("pt_render failed: " + r.stderr[-2000:])
        png = glob.glob(os.path.join(d, "*.png"))
        assert len(png) == 1, os.listdir(d)
        return dt, hashlib.sha256(open(png[0], "rb").read()).hexdigest()[:16]
run(["--res", "64", "64", "--iterations", "2"])                      # (pages the binary and the libraries in)
for what, args in (("cornell.txt as shipped (800x800, 5000 spp, depth 8)", []), ("1280x720, 400 iterations", ["--res", "1280", "720", "--iterations", "400"])):
    print(what + ", wall incl. process start, init and PNG:")
    a = run(args, {"PT_AMD_TRACE_AHEAD": "0"}); print("  call by call (PT_AMD_TRACE_AHEAD=0)  %.3f s  png %s" % a)
    b = run(args); print("  trace-ahead (the shim's default)     %.3f s  png %s" % b)
    c = run(args + ["--batch", "32"]); print("  --batch 32 (no per-iteration copy)   %.3f s  png %s" % c)
    assert a[1] == b[1] == c[1]
