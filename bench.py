#!/usr/bin/env python3
"""bench.py -- headline benchmark of the path-tracing hot path on MI355X.

Metric (BASELINE.json): Mpaths/s, paths = pixels x bounces x spp (NOMINAL segments), Cornell box at
1280x720, 8 bounces.

One STEP = one pass of the hot path over one batch of input = `--batch` (64) consecutive iterations (spp) of the
whole frame -- BASELINE config C2 in full: 1280x720, 64 spp, 8 bounces -- camera rays, 8 fused intersect+shade+compact
bounces, ordered accumulation, issued as ONE wavefront batch (pt_iterate_batch: the 64 iterations' paths share the 8
launches; results are identical to one call per iteration), with 3 batches in flight on internal streams.  So
`--steps 20` times 1280 iterations.  Scene, accumulator and path state are resident in HBM before the timed region.
(Rounds 1-3 stepped in batches of 32; a launch carries a fixed ~15 us of ramp-up and tail, which 64 iterations
amortise over twice the paths: `--batch 32` reproduces the old step.)

The timed block -- EXACTLY `--steps` steps between barrier + device synchronisation on both sides -- is repeated
`--repeats` (15) times inside one run (a single block is ~40 ms; fifteen are ~0.6 s of GPU time): `ms_per_step` and
`value` are the MEDIAN block's (ms_per_step x steps = that block's wall), `ms_per_step_min` / `_max` and `value_min` /
`_max` give the spread over the blocks, `ms_per_step_blocks` lists them all.

    python bench.py [--gpus N] [--steps K] [--warmup W]
        N > 1 without a torchrun environment: bench.py starts its N ranks itself (a child `python -m
        torch.distributed.run`, before this process has touched the GPU) and relays rank 0's JSON line
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (what the driver does)

N > 1: image rows are sharded round-robin over the ranks (row y -> rank y % N), every rank accumulates only its
own rows (packed), and ONE collective -- RCCL over xGMI -- assembles the frame at rank 0: the reduce(sum, float32, 3 W H,
root 0) of zero-padded full frames that north_star and SURVEY 8e name (the default since round 6), or `--collective gather`:
the gather of the row blocks (disjoint rows: both are bit-identical to 1 GPU; the gather moves 1/N of the bytes).  The
N > 1 line carries both readings (`multi_gpu.value_collective_gather`).  `--collective-every batch` (default) issues it after every
committed wavefront batch, `--collective-every 1` after every single iteration (BASELINE config C3 as written:
one pt_iterate + one collective per iteration).  The default run also times a bounded sample of the per-iteration
mode and reports it in config.per_iteration_collective.
`--scaling weak` (default): a step on N GPUs is `--batch` x N iterations of the whole frame -- every rank traces its 1/N of
the rows for N times the iterations, as many paths per step as the single GPU (a renderer's weak scaling: N times the
samples per pixel in the same time); value = all ranks' paths / the slowest rank's time.  `--scaling strong`: a step is
`--batch` iterations whatever N (config C3 as written: a fixed number of samples, divided; ranks then fuse steps into
fatter wavefront batches).  At N = 1 the two are the same run.  ONE N > 1 run reports all three readings of "N GPUs",
each over whole steps: `value_weak`, `value_strong` (collective per wavefront batch) and `value_c3_as_written` (strong,
one pt_iterate + one reduce(sum) of zero-padded full frames per ITERATION: BASELINE config C3 to the letter), plus
`collective_bytes_per_call` and the collective's share of a step; `value` = the one `--scaling` names (default weak:
per-GPU work fixed as N grows, which is what "scaling": "weak" in the line says).
`value_c3_as_written` is a FAST path (round 4) and is also measured at N = 1: the iterations come out of batches traced
ahead (PT_FLAG_TRACE_AHEAD: one small commit launch per call), the renderer accumulates into the zero-padded full frame
itself, and the reduce of a double-buffered snapshot runs on a stream of its own, overlapped with the next iteration's
commit (distributed.PerIterationReducer); at N = 1 the reduce goes through a one-rank RCCL group, so the protocol's
per-iteration cost is a hardware number even without a second GPU.  It is timed over the same `--steps` steps as `value`.

Prints ONE JSON line on rank 0, with `roofline` (dominant kernel = the fused bounce kernel, HIP-event timed on the
streams it runs on, against the 8 TB/s HBM peak) and `cpu_baseline` (the single-thread CPU oracle on a bounded
sample of the same workload; N = 1 only).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

# dmabuf IPC: read when the HSA runtime starts, so it must be in the environment before the first GPU call
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PATH_BYTES = 44                # SoA PathSegment: origin 12 + dir 12 + throughput 12 + pixelIndex 4 + remainingBounces 4
ACCUM_BYTES = 12               # radiance of one emitter hit parked in the batch's buffer (vec3 fp32 write)
SIMDS, CLOCK_HZ = 1024, 2.4e9  # 256 CUs x 4 SIMDs; max shader clock


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="steps per timed block (one step = --batch iterations of the whole frame, times N with --scaling weak); default 32 "
                         "on one GPU (1024 spp = 16 x the 64 spp of BASELINE config C2), 20 on N GPUs")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps first (default 4)")
    ap.add_argument("--repeats", type=int, default=15,
                    help="how many times the timed block of --steps steps is repeated inside the run (median reported, min / max beside it)")
    ap.add_argument("--scene", default=os.path.join(ROOT, "scenes", "cornell.txt"))
    ap.add_argument("--res", type=int, nargs=2, default=[1280, 720])
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--cpu-spp", type=int, default=40, help="spp of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--pipeline", type=int, default=3,
                    help="batches in flight (PtOptions.pipeline_depth; 0 = library default 3).  Round 4: 3 -- with steps of 64 iterations a third "
                         "batch in flight is worth +1.5 .. 4 %% (profiles/exp_r4i.sh, exp_r4k.sh); the per-iteration reading of config C3 keeps 2")
    ap.add_argument("--batch", type=int, default=64,
                    help="iterations per step = iterations traced as one wavefront batch (64 = the spp of BASELINE config C2)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1.  weak (default): a step is --batch x N iterations of the whole frame, so a rank's share of a step -- "
                         "its 1/N of the rows of N times the iterations -- is as many paths as the single GPU's step (more GPUs = "
                         "more samples per pixel in the same time); strong: a step is --batch iterations whatever N (BASELINE "
                         "config C3 as written: a fixed number of samples, divided)")
    ap.add_argument("--collective-every", default="batch", choices=["batch", "1"],
                    help="N > 1: assemble the frame at rank 0 after every wavefront batch, or after every iteration")
    ap.add_argument("--collective", default="reduce", choices=["gather", "reduce"],
                    help="N > 1: reduce(sum, float32, 3 W H, root 0) of zero-padded full frames -- the collective north_star and SURVEY 8e name (default "
                         "since round 6) -- or the gather of the packed row blocks (1/N of the bytes: rows are disjoint); the N > 1 line carries BOTH readings")
    ap.add_argument("--per-iteration-sample", type=int, default=None,
                    help="whole steps timed in the per-iteration mode of BASELINE config C3 (strong scaling, one pt_iterate + one "
                         "reduce per iteration, N = 1 included); default = --steps; 0 = skip")
    ap.add_argument("--dump-c3-frame", default=None, help="rank 0 writes the frame the per-iteration mode's last reduce delivered to this .npy file")
    ap.add_argument("--extra-passes", type=int, default=1, help="N > 1: 0 = only the pass `--scaling` names (no value_weak / value_strong pair)")
    ap.add_argument("--pmc-traffic-json", default=os.path.join(ROOT, "profiles", "pmc_traffic.json"),
                    help="PMC counters per bounce-kernel launch from the rocprofv3 --pmc passes (profiles/README.md)")
    ap.add_argument("--valu-rate-json", default=os.path.join(ROOT, "profiles", "valu_issue_rate.json"),
                    help="measured cycles per wave64 vector instruction per SIMD (profiles/valu_issue_rate.hip)")
    ap.add_argument("--dump-frame", default=None, help="rank 0 writes the final frame (float32 W*H*3 running sum) to this .npy file")
    ap.add_argument("--configs", type=int, default=None,
                    help="1: after the headline line's own measurement, measure the other BASELINE configurations that fit one GPU -- C4 (glass, 1920x1080, "
                         "depth 16), C5 on one GPU (64 spheres, 4096x4096, 16 spp per step) and the mesh scene -- a few timed blocks each, every one "
                         "in a child process of its own, and report them under `configs` with a roofline each.  Default: 1 for the plain "
                         "`python bench.py [--steps K --warmup W]` on one GPU, 0 as soon as a scene / frame / depth / batch is given")
    ap.add_argument("--roofline-bound", default="hbm", choices=["hbm", "valu_fp32"],
                    help="what `roofline` prices the bounce kernel against: the 8 TB/s of HBM (the contract's bound; C2, C4, meshes) or the vector "
                         "units' fp32 issue rate (SURVEY 8d: the 64-sphere configuration C5 is VALU-bound; needs that configuration's counters, --pmc-key)")
    ap.add_argument("--pmc-key", default=None, help="take the PMC counters from profiles/pmc_configs.json[<key>] instead of --pmc-traffic-json")
    ap.add_argument("--group-blocks", type=int, default=None,
                    help="1: also measure the C ABI's device groups in child processes (`--group`): at N = 1 eight members on the one device and config C3 as "
                         "written through the library's own collective, at N > 1 a group over the N devices beside the ranks' reading.  Default: as --configs "
                         "at N = 1, as --extra-passes at N > 1")
    ap.add_argument("--group", type=int, default=0,
                    help="M > 0: measure the C ABI's own multi-device host path instead of the ranks -- ONE process, a pt_group of M members (include/pt_amd.h: "
                         "pt_group_*; SURVEY 8e: single process, ncclCommInitAll, one stream per device), member i on device i %% --group-devices, a host thread "
                         "per member, the frame assembled by the library's own collective after every wavefront batch -- and print that block's JSON line "
                         "(bench.py runs itself this way as a child process for the `group` blocks of the driver's line)")
    ap.add_argument("--group-devices", type=int, default=0, help="--group: distinct HIP devices the members are dealt over (0: min(M, devices visible))")
    ap.add_argument("--group-c3", type=int, default=0,
                    help="--group: 1 = config C3 as written instead of the batch mode: one pt_group_iterate (one iteration out of batches traced ahead on every "
                         "member + the frame's reduce) per ITERATION")
    return ap.parse_args(argv)


def self_launch(args):
    """`python bench.py --gpus N` outside torchrun: start the N ranks as children of a `torch.distributed.run` child
    process -- before this process has initialised the GPU -- relay their output and exit with their code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd)
    sys.exit(r.returncode)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


class BoxTelemetry:
    """Shader / memory clock, package power against its cap and temperatures of THIS process's GPU, sampled from the amdgpu hwmon
    files (sysfs: plain reads from a host thread, no GPU call, no child process) every 20 ms while the timed blocks run -- so that a
    line from a slow box says WHY it is slow (a power cap, a hot card, a clock that never leaves its floor) without a second run.
    Boxes of the pool differ by up to 1.8 x on the same build (profiles/r03_sensitivity_experiments.txt)."""
    FILES = {"sclk_mhz": ("freq1_input", 1e-6), "mclk_mhz": ("freq2_input", 1e-6), "power_w": ("power1_input", 1e-6),
             "temp_junction_c": ("temp2_input", 1e-3), "temp_mem_c": ("temp3_input", 1e-3)}

    def __init__(self, torch, device_index):
        import glob
        import threading
        self.hwmon, self.samples, self.err = None, {k: [] for k in self.FILES}, None
        self._stop = threading.Event()
        self._on = threading.Event()
        try:
            p = torch.cuda.get_device_properties(device_index)
            want = "%04x:%02x:%02x." % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
            for d in glob.glob("/sys/class/drm/card*/device"):
                if os.path.basename(os.path.realpath(d)).startswith(want):
                    h = glob.glob(d + "/hwmon/hwmon*")
                    if h:
                        self.hwmon, self.pci = h[0], os.path.basename(os.path.realpath(d))
            if self.hwmon is None:
                self.err = "no amdgpu hwmon directory for PCI device %s*" % want
        except Exception as e:
            self.err = repr(e)
        self._thread = threading.Thread(target=self._run, daemon=True)
        if self.hwmon:
            self._thread.start()

    def _read(self, name):
        try:
            return float(open(os.path.join(self.hwmon, name)).read())
        except (OSError, ValueError):
            return None

    def _run(self):
        while not self._stop.is_set():
            if self._on.is_set():
                for k, (f, scale) in self.FILES.items():
                    v = self._read(f)
                    if v is not None:
                        self.samples[k].append(v * scale)
            self._stop.wait(0.02)

    def start(self):
        self._on.set()

    def pause(self):
        self._on.clear()

    def report(self):
        self._stop.set()
        if not self.hwmon:
            return {"error": self.err}
        out = {"source": "amdgpu hwmon (sysfs) of PCI device %s, sampled every 20 ms during the timed blocks" % self.pci,
               "samples": len(self.samples["sclk_mhz"])}
        for k, v in self.samples.items():
            if v:
                s_ = sorted(v)
                out[k] = {"min": round(s_[0], 1), "median": round(s_[len(s_) // 2], 1), "max": round(s_[-1], 1)}
        cap = self._read("power1_cap")
        out["power_cap_w"] = round(cap * 1e-6, 1) if cap else None
        return out


def box_calibration(pt, torch):
    """What THIS box does on a fixed memory-bound job, beside the line's own numbers: the library's exclusive scan of 2^26 int32
    (8 algorithmic bytes per element; 0.19 ms = 2.8 TB/s on the boxes of rounds 2-3).  Boxes of the pool differ -- the same two
    builds measured 167 G and 228 G paths/s on two boxes within minutes (profiles/r03_sensitivity_experiments.txt) -- and a
    reader of one line cannot tell a slow box from a slow build without it."""
    n = 1 << 26
    try:
        x = torch.ones(n, dtype=torch.int32, device="cuda")
        y = torch.empty_like(x)
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            pt.scan_exclusive_dev(x.data_ptr(), y.data_ptr(), n, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            pt.scan_exclusive_dev(x.data_ptr(), y.data_ptr(), n, st)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        ok = int(y[-1].item()) == n - 1
        del x, y
        return {"job": "pt_scan_exclusive_i32 of 2^26 elements, 20 calls", "ms_per_call": round(ms, 4),
                "algorithmic_GBps": round(8.0 * n / (ms * 1e-3) / 1e9, 1), "result_checked": ok}
    except Exception as e:            # (a calibration must never cost the line)
        return {"job": "pt_scan_exclusive_i32 of 2^26 elements", "error": repr(e)}


def cpu_baseline(args, scene, pt):
    """Single-thread CPU oracle (oracle/, kind 'port') on a bounded sample of the same workload, pinned to ONE core; plus BASELINE
    config C1 (scenes/sphere.txt 400x400, 1 spp, depth 4) in full."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    pinned, saved = None, None
    try:                                                   # one core for the whole baseline (restored afterwards)
        saved = os.sched_getaffinity(0)
        pinned = sorted(saved)[len(saved) // 2]
        os.sched_setaffinity(0, {pinned})
    except (AttributeError, OSError):
        pinned = None
    try:
        W, H = args.res
        ref = orc.Renderer(scene.camera.view(orc.CAMERA_DTYPE), scene.geoms.view(orc.GEOM_DTYPE),
                           scene.materials.view(orc.MATERIAL_DTYPE), args.depth, meshes=getattr(scene, "meshes", None))
        img = np.zeros(W * H * 3, np.float32)
        K = 64                                                 # a probe first: every 64th row of one sample
        t0 = time.perf_counter()
        ref.iterate(1, img, 0, K)                              # (also warms the caches / pages the library in)
        probe = time.perf_counter() - t0
        if probe * K * args.cpu_spp <= 45.0:
            t0 = time.perf_counter()
            for it in range(2, 2 + args.cpu_spp):
                ref.iterate(it, img)
            dt = time.perf_counter() - t0
            pixels, what = W * H * args.cpu_spp, "%d spp" % args.cpu_spp
        else:
            # a slow scene (brute-force meshes): bound the sample to ~15 s of row slices (rows y with y % 64 == r) of sample 2
            nslices = max(1, min(K - 1, int(15.0 / max(probe, 1e-3))))
            t0 = time.perf_counter()
            for r in range(1, 1 + nslices):
                ref.iterate(2, img, r, K)
            dt = time.perf_counter() - t0
            pixels = sum((H - r + K - 1) // K for r in range(1, 1 + nslices)) * W
            what = "%d of every %d rows of 1 spp" % (nslices, K)
        # BASELINE config C1: scenes/sphere.txt at 400x400, 1 spp, depth 4, the whole thing -- it takes milliseconds, so it is run
        # 15 times and the median is reported
        c1 = None
        c1_path = os.path.join(ROOT, "scenes", "sphere.txt")
        if os.path.exists(c1_path):
            s1 = pt.Scene(c1_path)
            s1.set_resolution(400, 400)
            r1 = orc.Renderer(s1.camera.view(orc.CAMERA_DTYPE), s1.geoms.view(orc.GEOM_DTYPE), s1.materials.view(orc.MATERIAL_DTYPE), 4)
            i1 = np.zeros(400 * 400 * 3, np.float32)
            ts = []
            for k in range(15):
                i1[:] = 0
                t0 = time.perf_counter()
                r1.iterate(1, i1)
                ts.append(time.perf_counter() - t0)
            ts.sort()
            c1 = {"config": "scenes/sphere.txt 400x400, 1 spp, 4 bounces (BASELINE config C1, run in full)", "value": round(400 * 400 * 4 / ts[7] / 1e6, 2),
                  "unit": "Mpaths/s", "ms": round(ts[7] * 1e3, 3), "ms_min": round(ts[0] * 1e3, 3), "ms_max": round(ts[-1] * 1e3, 3), "runs": 15}
    finally:
        if saved is not None and pinned is not None:
            os.sched_setaffinity(0, saved)
    return {"value": round(pixels * args.depth / dt / 1e6, 3), "unit": "Mpaths/s", "cores": 1,
            "kind": "port", "cpu_model": cpu_model(), "host_cores": os.cpu_count(), "pinned_to_core": pinned,
            "sample": "%s of %s %dx%d depth %d (%.1f s, single thread, g++ -O2 -ffp-contract=off)"
                      % (what, os.path.basename(args.scene), W, H, args.depth, dt),
            "c1": c1}


# the other BASELINE configurations that fit one GPU (BASELINE.json configs[3], configs[4] on one of its eight GPUs) and the README's mesh
# extra: (key, bench arguments, what the block says about itself)
OTHER_CONFIGS = [
    ("c4", ["--scene", "scenes/cornell_glass.txt", "--res", "1920", "1080", "--depth", "16", "--steps", "4", "--warmup", "1"],
     "BASELINE config C4: Cornell + glass sphere, 1920x1080, 16 bounces, its 256 spp as 4 steps of 64"),
    ("c5_one_gpu", ["--scene", "scenes/spheres64.txt", "--res", "4096", "4096", "--depth", "8", "--batch", "16", "--steps", "2", "--warmup", "1",
                    "--roofline-bound", "valu_fp32"],
     "BASELINE config C5 on ONE of its eight GPUs: 64-sphere scene, 4096x4096, 8 bounces, its 16 spp as one step"),
    ("mesh", ["--scene", "scenes/cornell_mesh.txt", "--steps", "4", "--warmup", "1"],
     "README extra (SURVEY 8f-4): Cornell box with two triangle meshes, 1280x720, 8 bounces, 64 spp per step"),
]


def other_configs(args):
    """One child `bench.py` per configuration (its own process: its own renderer, pools and device memory, released when it ends), a few
    timed blocks each; what comes back is that run's own line, cut down to the number, its spread, the workload and the roofline."""
    out = {}
    for key, extra, what in OTHER_CONFIGS:
        extra = [os.path.join(ROOT, a) if a.startswith("scenes/") else a for a in extra]     # (whatever directory the driver runs from)
        cmd = [sys.executable, os.path.abspath(__file__)] + extra + ["--repeats", "3", "--cpu-spp", "0", "--per-iteration-sample", "0",
                                                                      "--configs", "0", "--pmc-key", key, "--pipeline", str(args.pipeline)]
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not line:
                out[key] = {"what": what, "error": "rc %d: %s" % (r.returncode, r.stderr.strip().splitlines()[-1] if r.stderr.strip() else "no output")}
                continue
            d = json.loads(line[-1])
            out[key] = {"what": what, "value": d["value"], "unit": d["unit"], "value_min": d["value_min"], "value_max": d["value_max"],
                        "steps": d["steps"], "warmup": d["warmup"], "repeats": d["repeats"], "ms_per_step": d["ms_per_step"],
                        "workload": d["config"]["workload"], "live_segments_per_iteration": d["config"]["live_segments_per_iteration"],
                        "roofline": d["roofline"], "command": "python bench.py " + " ".join(cmd[2:]),
                        "wall_s_of_the_child_process": round(time.perf_counter() - t0, 1)}
        except Exception as e:                               # (an extra block must never cost the line)
            out[key] = {"what": what, "error": repr(e)}
    return out


def valu_issue_rate(path, waves_per_simd):
    """cycles per wave64 vector instruction per SIMD from the committed microbenchmark: the kernel-like instruction mix at
    the residency nearest to the kernel's, and the plain v_fma_f32 rate (wall-clock based columns)."""
    try:
        rows = json.load(open(path))["rows"]
    except Exception:
        return None
    def pick(prefix):
        c = [r for r in rows if r["op"].startswith(prefix)]
        if not c:
            return None
        r = min(c, key=lambda r: abs(r["waves_per_simd"] - waves_per_simd))
        return r["cycles_per_instruction_per_simd_from_wall"], r["waves_per_simd"]
    mix, fma = pick("mix"), pick("v_fma_f32")
    if not mix or not fma:
        return None
    return {"mix": mix[0], "fma": fma[0], "waves_per_simd": mix[1]}


def run_group(args):
    """`bench.py --group M`: the hot path through the C ABI's device groups, one process.  A step is `--batch` x D iterations of the whole frame
    (D = distinct devices: the weak reading of bench.py; on ONE device a step is the single GPU's step, its rows divided over the members);
    the members trace it in wavefront batches of min(PT_MAX_BATCH, --batch x M) iterations (a member's launch then carries as many paths as
    the single GPU's, capped by PT_MAX_BATCH -- the ranks of `--gpus N` fuse steps the same way), pt_group_reduce after every batch.
    Timed like the headline: `--warmup` steps, then `--repeats` blocks of EXACTLY `--steps` steps between pt_group_sync on both sides."""
    import numpy as np
    import __graft_entry__ as ge
    pt = ge.load_package()
    ndev_visible = pt.device_count()
    if ndev_visible < 1:
        print("bench.py --group needs a GPU (the hot path has no CPU fallback)", file=sys.stderr)
        return 1
    M = args.group
    nd = args.group_devices if args.group_devices > 0 else min(M, ndev_visible)
    if nd > ndev_visible:
        print("bench.py --group: %d devices asked for, %d visible" % (nd, ndev_visible), file=sys.stderr)
        return 2
    W, H = args.res
    D, B = args.depth, args.batch
    steps = args.steps if args.steps is not None else 32
    warmup = args.warmup if args.warmup is not None else 4
    scene = pt.Scene(args.scene)
    scene.set_resolution(W, H)
    P = W * H
    I = B * nd                                            # iterations of the whole frame per step
    g = pt.Group(M, devices=[i % nd for i in range(M)])
    out = {"members": M, "devices": nd, "members_per_device": (M + nd - 1) // nd, "collective": g.collective,
           "issue_threads": 0 if (M == 1 or os.environ.get("PT_AMD_GROUP_THREADS") == "0") else M}
    try:
        if args.group_c3:
            g.init(scene, traceDepth=D, flags=pt.PT_FLAG_TRACE_AHEAD, pipeline_depth=min(args.pipeline, 2) if args.pipeline > 0 else 2, max_batch=B)
            it = 1
            for _ in range(max(warmup, 1) * B):
                g.iterate(it)
                it += 1
            g.sync()
            walls, enq = [], []
            for _ in range(args.repeats):
                t0 = time.perf_counter()
                for _ in range(steps * I):
                    g.iterate(it)
                    it += 1
                t1 = time.perf_counter()
                g.sync()
                t2 = time.perf_counter()
                walls.append(t2 - t0)
                enq.append(t1 - t0)
            walls.sort()
            dt = walls[len(walls) // 2]
            out.update({"value": round(P * D * I * steps / dt / 1e6, 2), "unit": "Mpaths/s", "steps": steps, "iterations_per_step": I,
                        "iterations_per_wavefront_batch": B, "ms_per_step": round(dt / steps * 1e3, 4),
                        "ms_per_iteration": round(dt / (steps * I) * 1e3, 5), "ms_per_iteration_min": round(walls[0] / (steps * I) * 1e3, 5),
                        "ms_per_iteration_max": round(walls[-1] / (steps * I) * 1e3, 5), "repeats": len(walls),
                        "host_enqueue_us_per_iteration": round(sorted(enq)[len(enq) // 2] / (steps * I) * 1e6, 2),
                        "mode": "strong scaling, one pt_group_iterate per iteration: every member commits one iteration of a wavefront batch traced ahead "
                                "(PT_FLAG_TRACE_AHEAD) into the accumulator AND into one of two snapshot frames (k_commit), the library's collective "
                                "reads the snapshot on its own stream"})
            frame = g.readback()
        else:
            maxb = min(pt.PT_MAX_BATCH, B * M)
            g.init(scene, traceDepth=D, pipeline_depth=min(args.pipeline, 2) if (M > nd and args.pipeline > 0) else args.pipeline, max_batch=maxb)

            def run(first, nsteps):
                it, end, nb = first, first + nsteps * I, 0
                while it < end:
                    n = min(maxb, end - it)
                    g.iterate_batch(it, n)
                    g.reduce()
                    it += n
                    nb += 1
                return it, nb

            it, _ = run(1, warmup)
            g.sync()
            walls, enq, nb = [], [], 1
            for _ in range(args.repeats):
                t0 = time.perf_counter()
                it, nb = run(it, steps)
                t1 = time.perf_counter()
                g.sync()
                t2 = time.perf_counter()
                walls.append(t2 - t0)
                enq.append(t1 - t0)
            blocks = list(walls)
            walls.sort()
            dt = walls[len(walls) // 2]
            nominal = P * D * I * steps
            out.update({"value": round(nominal / dt / 1e6, 2), "unit": "Mpaths/s", "value_min": round(nominal / walls[-1] / 1e6, 2),
                        "value_max": round(nominal / walls[0] / 1e6, 2), "steps": steps, "warmup": warmup, "repeats": len(walls),
                        "iterations_per_step": I, "iterations_per_wavefront_batch": maxb, "ms_per_step": round(dt / steps * 1e3, 4),
                        "ms_per_step_blocks": [round(w / steps * 1e3, 4) for w in blocks],
                        "host_enqueue_us_per_wavefront_batch": round(sorted(enq)[len(enq) // 2] / max(nb, 1) * 1e6, 1),
                        "gpu_ms_per_wavefront_batch": round(dt / max(nb, 1) * 1e3, 4),
                        "mode": "%s, pt_group_iterate_batch + pt_group_reduce per wavefront batch" % ("weak scaling over %d devices" % nd if nd > 1 else
                                                                                                    "one device, its rows divided over the members")})
            frame = g.readback()
        c = g.counters()
        out["iterations_committed"] = int(c.iterations)
        out["workload"] = "%s %dx%d, %d bounces" % (os.path.relpath(args.scene, ROOT), W, H, D)
        if args.dump_frame:
            np.save(args.dump_frame, frame)
    finally:
        g.destroy()
    print(json.dumps(out), flush=True)
    return 0


def group_block(args, members, devices, c3=False, extra_env=None, timeout=300):
    """one `bench.py --group` child process (its own HIP runtime, queues and device memory); returns its line, or {"error": ...}"""
    cmd = [sys.executable, os.path.abspath(__file__), "--group", str(members), "--group-devices", str(devices), "--group-c3", "1" if c3 else "0",
           "--scene", args.scene, "--res", str(args.res[0]), str(args.res[1]), "--depth", str(args.depth), "--batch", str(args.batch),
           "--steps", str(args.steps), "--warmup", str(args.warmup), "--repeats", str(5 if not c3 else 3), "--pipeline", str(args.pipeline)]
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    env.update(extra_env or {})
    t0 = time.perf_counter()
    p = None
    try:
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
        so, se = p.communicate(timeout=timeout)
        line = [l for l in so.splitlines() if l.startswith("{")]
        if p.returncode != 0 or not line:
            return {"error": "rc %d: %s" % (p.returncode, se.strip().splitlines()[-1] if se.strip() else "no output")}
        d = json.loads(line[-1])
        d["command"] = "python bench.py " + " ".join(cmd[2:]) + ("" if not extra_env else "   (env: %s)" % " ".join("%s=%s" % kv for kv in sorted(extra_env.items())))
        d["wall_s_of_the_child_process"] = round(time.perf_counter() - t0, 1)
        return d
    except subprocess.TimeoutExpired:
        if p is not None:
            p.kill()                                         # (the exact child this call started)
            p.communicate()
        return {"error": "no line within %d s: killed" % timeout}
    except Exception as e:                                   # (an extra block must never cost the line)
        return {"error": repr(e)}


def _on_sigterm(signum, frame):
    # torchrun ends the surviving ranks of a failed job with SIGTERM, which Python does not turn into an exception by itself: raise
    # one, so that main()'s `finally: cleanup()` drains the streams and frees the renderer before the process goes away
    raise SystemExit(128 + signum)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.configs is None:
        # the driver's plain command (flags --gpus / --steps / --warmup only) reports every one-GPU configuration; any other run is about ONE
        d = parse([])
        args.configs = 1 if (args.gpus == 1 and world == 1 and all(getattr(args, k) == getattr(d, k) for k in
                                                                     ("scene", "res", "depth", "batch", "pipeline", "scaling", "cpu_spp", "repeats"))) else 0
    if args.group > 0:
        sys.exit(run_group(args))
    if world == 1 and args.gpus > 1:
        self_launch(args)                                   # never returns
    import signal
    signal.signal(signal.SIGTERM, _on_sigterm)
    # Every exit path of a rank drains and frees the renderer BEFORE the process goes away (an exception between pt_iterate
    # and pathtraceFree used to leave launches in flight at context teardown), and a failed rank exits non-zero.
    ctx = {}
    rc = 1
    try:
        run(args, ctx)
        rc = 0
    except SystemExit as e:
        rc = e.code if isinstance(e.code, int) else (0 if e.code is None else 1)
        if rc and not isinstance(e.code, int):
            print(e.code, file=sys.stderr, flush=True)
    except BaseException:
        import traceback
        traceback.print_exc()
        rc = 1
    finally:
        cleanup(ctx, rc == 0)
    sys.exit(rc)


def cleanup(ctx, ok):
    """Orderly exit of a rank: wait for its own GPU work, free the renderer, and -- when every rank got here in good order --
    leave together: nobody tears its HIP context down while another rank still works (sync, barrier, tear-down).  A rank that
    failed skips the barrier (the others may never reach it) and exits non-zero; the launcher then ends the others."""
    torch = ctx.get("torch")
    pt = ctx.get("pt")
    try:
        if torch is not None and torch.cuda.is_available():
            torch.cuda.synchronize()
    except Exception:
        pass
    try:
        if pt is not None:
            pt.pathtraceFree()                              # (synchronises the renderer's own streams first)
    except Exception:
        pass
    dist = ctx.get("dist")
    if dist is not None and dist.is_initialized():
        try:
            if ok and dist.get_world_size() > 1:
                # leave together, but never wait for ever: the group was formed with a timeout (distributed.init_process_group), which
                # RCCL's watchdog enforces on this barrier too; gloo takes it explicitly
                import datetime
                if ctx.get("backend") == "gloo":
                    dist.monitored_barrier(timeout=datetime.timedelta(seconds=120))
                else:
                    dist.barrier()
            for k in ("accum", "frame", "bufs", "reducer", "accum_full"):
                ctx.pop(k, None)
            dist.destroy_process_group()
        except Exception:
            pass


def run(args, ctx):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world

    import numpy as np
    import torch
    import torch.distributed as dist
    ctx["torch"] = torch

    B = args.batch
    if args.warmup is None:
        args.warmup = 4
    if args.steps is None:
        args.steps = 32 if world == 1 else 20
    if args.repeats < 1:
        sys.exit("bench.py: --repeats must be at least 1")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the hot path has no CPU fallback)")
    # BENCH_BACKEND=gloo lets the N > 1 path be rehearsed with several ranks on ONE GPU (RCCL needs one
    # GPU per rank); the real runs use nccl = RCCL over xGMI.
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    ngpu = torch.cuda.device_count()
    if world > 1 and backend == "nccl" and ngpu < world:
        # (local_rank % device_count would silently stack ranks on one device, which RCCL cannot serve)
        print("bench.py: %d ranks need %d GPUs under RCCL (backend nccl), this node shows %d; BENCH_BACKEND=gloo rehearses "
              "several ranks on one GPU" % (world, world, ngpu), file=sys.stderr, flush=True)
        sys.exit(2)
    device_index = local_rank % ngpu
    torch.cuda.set_device(device_index)
    import __graft_entry__ as ge
    pt = ge.load_package()
    ctx["pt"] = pt
    ptdist = ge.load_submodule("distributed")
    ctx["backend"] = backend
    if world > 1:
        ptdist.init_process_group(backend)
        ctx["dist"] = dist

    W, H = args.res
    scene = pt.Scene(args.scene)
    scene.set_resolution(W, H)
    P = W * H
    D = args.depth
    if B < 1 or B > pt.PT_MAX_BATCH:
        sys.exit("bench.py: --batch must be 1..%d" % pt.PT_MAX_BATCH)
    # accumulator as a torch tensor so RCCL can move it.  N = 1: the full frame.  N > 1: this rank's rows
    # only (packed, padded to the largest shard); rank 0 assembles `frame` from the gathered blocks.
    if world > 1:
        accum = torch.zeros(ptdist.padded_block_floats(W, H, world), dtype=torch.float32, device="cuda")
        frame = torch.zeros(P * 3, dtype=torch.float32, device="cuda") if rank == 0 else None
        bufs = ptdist.make_gather_buffers(accum, world, rank)
        shard_flag = pt.PT_FLAG_ACCUM_SHARD_ROWS
    else:
        accum = torch.zeros(P * 3, dtype=torch.float32, device="cuda")
        frame, bufs, shard_flag = None, None, 0
    ctx.update(accum=accum, frame=frame, bufs=bufs)
    stream = torch.cuda.current_stream()

    def init(flags, pipeline, max_batch, full_frame=None):
        """`full_frame`: accumulate into this zero-padded FULL frame (own rows in place, zeros elsewhere) instead of the packed row
        block -- the per-iteration mode of config C3, whose reduce then needs no scatter"""
        pt.pathtraceFree()
        pt.pathtraceInit(scene, shard_rank=rank, shard_count=world, stream=stream.cuda_stream,
                         accum_dev=(accum if full_frame is None else full_frame).data_ptr(), device=device_index,
                         flags=flags | (shard_flag if full_frame is None else 0),
                         traceDepth=D, pipeline_depth=pipeline, max_batch=max_batch)

    def collect(collective):
        # the single collective of the data path: the row blocks travel to rank 0 over xGMI
        ptdist.gather_frame(accum, bufs, frame, W, H, dst=0, collective=collective)

    def run_steps(first_iter, steps, I, maxb, every, collective, reducer=None):
        """`steps` steps of I iterations from `first_iter`.  every == "batch": wavefront batches of `maxb` iterations (one or
        several steps each, or a part of one), the frame assembled at rank 0 after every batch; every == "1": I single-iteration
        calls per step (each commits one iteration of a batch of `maxb` traced ahead), each followed by the collective -- the
        gather / reduce on the caller's stream, or, with a `reducer`, config C3's overlapped reduce of full frames."""
        it = first_iter
        end = first_iter + steps * I
        while it < end:
            if every == "batch":
                n = min(maxb, end - it)
                pt.pathtrace_batch(None, 0, it, n)
                if world > 1:
                    collect(collective)
                it += n
            else:
                pt.pathtrace(None, 0, it, readback=False)
                if reducer is not None:
                    reducer.collect()
                elif world > 1:
                    collect(collective)
                it += 1
        return it

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(dt):
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def sum_over_ranks(values):
        if world > 1:
            t = torch.tensor([float(v) for v in values], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return [float(v) for v in t.tolist()]
        return [float(v) for v in values]

    def timed(first_iter, steps, I, maxb, every, collective, reducer=None):
        """EXACTLY `steps` steps between barrier + torch.cuda.synchronize() on both sides; the slowest rank's time."""
        barrier()
        t0 = time.perf_counter()
        nxt = run_steps(first_iter, steps, I, maxb, every, collective, reducer)
        if reducer is not None:
            reducer.finish()      # (gloo's asynchronous reduces are not the device's work: the closing barrier + synchronize would not wait for them)
        barrier()
        return max_over_ranks(time.perf_counter() - t0), nxt

    def plan(scaling, every):
        """iterations of the whole frame per step, and the iterations a rank issues as one wavefront batch"""
        # (per-iteration modes: the single-iteration calls draw on batches of B iterations traced ahead, PT_FLAG_TRACE_AHEAD)
        if scaling == "weak":
            # a step is I = B x N iterations, traced as one wavefront batch (in pieces of PT_MAX_BATCH should it be larger)
            I = B * world
            return I, (min(I, pt.PT_MAX_BATCH) if every == "batch" else B)
        # N ranks share the frame's rows, so a rank's launches cover 1/N of the paths: with the collective once per batch a rank
        # traces `fuse` consecutive steps as ONE wavefront batch (at most PT_MAX_BATCH iterations), which keeps its launches fat
        # -- but not so few batches that the two in flight never overlap: at least 8 per timed block
        fuse = max(1, min(world, pt.PT_MAX_BATCH // B, max(1, args.steps // 8))) if every == "batch" else 1
        return B, (B * fuse if every == "batch" else B)

    def measure(scaling, every, collective, steps, repeats, warmup, flags=0, pipeline=None, c3=False):
        """one configuration: init, warm up, `repeats` timed blocks of `steps` steps; returns the blocks' walls (s) etc.
        `c3`: BASELINE config C3 as written -- accumulation into the zero-padded full frame, one overlapped reduce per iteration"""
        I, maxb = plan(scaling, every)
        reducer = None
        if every == "1":
            flags |= pt.PT_FLAG_TRACE_AHEAD
        if c3:
            if ctx.get("accum_full") is None:
                ctx["accum_full"] = torch.zeros(P * 3, dtype=torch.float32, device="cuda")
            ctx["accum_full"].zero_()
            init(flags, args.pipeline if pipeline is None else pipeline, maxb, full_frame=ctx["accum_full"])
            reducer = ptdist.PerIterationReducer(ctx["accum_full"], dst=0, always_collective=True)
            ctx["reducer"] = reducer
        else:
            accum.zero_()
            init(flags, args.pipeline if pipeline is None else pipeline, maxb)
        nxt = run_steps(1, warmup, I, maxb, every, collective, reducer)
        barrier()
        pt.counters_reset()
        walls = []
        for _ in range(repeats):
            dt, nxt = timed(nxt, steps, I, maxb, every, collective, reducer)
            walls.append(dt)
        return {"I": I, "maxb": maxb, "walls": walls, "counters": pt.counters(), "next_iter": nxt, "reducer": reducer}

    def median(v):
        s_ = sorted(v)
        return s_[len(s_) // 2] if len(s_) % 2 else 0.5 * (s_[len(s_) // 2 - 1] + s_[len(s_) // 2])

    every = args.collective_every
    # ---- pass A: the headline number (`--scaling`, `--collective-every`, `--collective`), repeated blocks --------------
    telemetry = BoxTelemetry(torch, device_index) if rank == 0 else None
    if telemetry:
        telemetry.start()
    if os.environ.get("BENCH_MARK_FILE") and rank == 0:    # (tests: "the timed blocks are about to start")
        open(os.environ["BENCH_MARK_FILE"], "w").close()
    A = measure(args.scaling, every, args.collective, args.steps, args.repeats, args.warmup)
    if telemetry:
        telemetry.pause()
    I, maxb = A["I"], A["maxb"]
    dt = median(A["walls"])
    cntA = A["counters"]
    if args.dump_frame and rank == 0:
        np.save(args.dump_frame, (frame if world > 1 else accum).cpu().numpy())

    def reading(scaling, ev, collective, steps, warmup, c3=False, pipeline=None):
        m = measure(scaling, ev, collective, steps, 1, warmup, c3=c3, pipeline=pipeline)
        w = m["walls"][0]
        r = {"value": round(P * D * m["I"] * steps / w / 1e6, 2), "unit": "Mpaths/s", "steps": steps, "iterations_per_step": m["I"],
             "iterations_per_wavefront_batch": m["maxb"], "ms_per_step": round(w / steps * 1e3, 4),
             "mode": "%s scaling, one %s per %s" % (scaling, collective, "wavefront batch" if ev == "batch" else "iteration")}
        if ev == "1":
            r["ms_per_iteration"] = round(w / (steps * m["I"]) * 1e3, 5)
            r["calls"] = "one pt_iterate per iteration, committing one iteration of a wavefront batch traced ahead (PT_FLAG_TRACE_AHEAD)"
        if c3:
            r["mode"] += " of zero-padded full frames, snapshot double-buffered, the reduce on its own stream"
        return r

    # ---- N > 1: the collective on its own (its bytes and its share of a step), then the other readings of "N GPUs" -------
    multi = None
    if world > 1:
        calls = 10
        barrier()
        t0 = time.perf_counter()
        for _ in range(calls):
            collect(args.collective)
        barrier()
        coll_ms = max_over_ranks(time.perf_counter() - t0) / calls * 1e3
        block_bytes = accum.numel() * 4
        coll_calls_per_step = (I + maxb - 1) // maxb if every == "batch" else I
        multi = {"backend": backend if backend != "nccl" else "nccl (RCCL)", "ranks_per_device": max(1, (world + max(ngpu, 1) - 1) // max(ngpu, 1)),
                 "collective": args.collective, "collective_every": every,
                 "collective_bytes_per_call": {"sent_by_each_rank": block_bytes if args.collective == "gather" else P * 12,
                                               "received_by_rank_0": (world - 1) * (block_bytes if args.collective == "gather" else P * 12)},
                 "collective_ms_per_call": round(coll_ms, 4), "collective_calls_per_step": coll_calls_per_step,
                 "collective_share_of_step": round(coll_ms * coll_calls_per_step / (dt / args.steps * 1e3), 4)}

        if args.extra_passes:
            other = "strong" if args.scaling == "weak" else "weak"
            multi["value_" + other] = reading(other, every, args.collective, args.steps, min(args.warmup, 2))
            # ... and the headline's own pass once more with the OTHER collective (reduce: the contract's; gather: 1/N of the bytes)
            other_coll = "gather" if args.collective == "reduce" else "reduce"
            multi["value_collective_" + other_coll] = reading(args.scaling, every, other_coll, args.steps, min(args.warmup, 2))

    # ---- BASELINE config C3 to the letter, N >= 1: a fixed number of samples divided over the ranks, one pt_iterate and one
    #      reduce(sum) of zero-padded full frames per ITERATION (the reference's per-iteration full-frame transfer,
    #      src/pathtrace.cu:170-171), as the overlapped fast path of distributed.PerIterationReducer.  N = 1: through a one-rank
    #      RCCL group, so that the protocol's per-iteration cost is a hardware number without a second GPU.
    c3 = None
    c3_steps = args.steps if args.per_iteration_sample is None else args.per_iteration_sample
    if c3_steps > 0 and (world > 1 or args.extra_passes):
        try:
            if world == 1 and not dist.is_initialized():
                ptdist.init_process_group(backend, single_rank=True)
                ctx["dist"] = dist
            c3 = reading("strong", "1", "reduce", c3_steps, 1, c3=True, pipeline=min(args.pipeline, 2) if args.pipeline > 0 else 2)
            red = ctx.get("reducer")
            c3["collective_bytes_per_call"] = red.bytes_per_call() if red is not None else None
            c3["collective_backend"] = (backend if backend != "nccl" else "nccl (RCCL)") + (", one-rank group" if world == 1 else "")
            if args.dump_c3_frame and rank == 0 and red is not None:
                np.save(args.dump_c3_frame, red.frame().cpu().numpy())
        except Exception as e:                               # (the extra reading must never cost the line)
            if world > 1:
                raise
            c3 = {"error": repr(e)}
        ctx.pop("reducer", None)

    # ---- pass B: same steps with HIP events around every launch (roofline of the bounce kernel); one
    #      batch in flight, so that a launch's duration is the kernel's own and not its share of a GPU it
    #      co-occupies with the neighbouring batches' launches
    Bp = measure(args.scaling, every, args.collective, args.steps, 1, min(args.warmup, 2), flags=pt.PT_FLAG_KERNEL_TIMING, pipeline=1)
    dtB = Bp["walls"][0]
    cnt = Bp["counters"]
    pt.pathtraceFree()

    # ---- the C ABI's own multi-device host path (pt_group_*: ONE process, a host thread and a renderer per member, the library's collective;
    #      SURVEY 8e), each reading in a child process of its own (`bench.py --group`), while this process's renderers are freed and its ranks wait:
    #      N = 1: eight members on the one device against the headline, and config C3 as written through the library's own per-iteration reduce
    #             (a one-rank RCCL communicator: the call path a one-GPU box can run);
    #      N > 1: rank 0 starts a group over the N devices -- what the reference's single-process host (src/main.cpp:72-113) would drive.
    groups = None
    want_groups = args.group_blocks if args.group_blocks is not None else (args.configs if world == 1 else args.extra_passes)
    def wait_for_rank0(key):
        """the other ranks wait ON THE HOST for rank 0's child process (a key of the rendezvous store): a collective barrier would spin on
        their GPUs for as long as the group measurement -- which uses those very GPUs -- takes"""
        if world == 1:
            return
        try:
            import datetime
            from torch.distributed.distributed_c10d import _get_default_store
            store = _get_default_store()
            if rank == 0:
                store.set(key, "1")
            else:
                store.wait([key], datetime.timedelta(seconds=900))
        except Exception:
            pass                                            # (the barrier below still brings the ranks together)

    if want_groups:
        barrier()
        if rank == 0:
            groups = {}
            if world == 1:
                groups["group_8_members_one_device"] = group_block(args, 8, 1)
                groups["c3_as_written"] = group_block(args, 1, 1, c3=True, extra_env={"PT_AMD_COLLECTIVE": "rccl"})
            else:
                nd = min(world, ngpu)
                groups["group"] = group_block(args, world, nd)
                groups["group_c3_as_written"] = group_block(args, world, nd, c3=True)
        wait_for_rank0("pt_amd_group_blocks_done")
        barrier()

    iters_block = args.steps * I                             # iterations of one timed block
    liveA = [int(cntA.live[d]) for d in range(D + 2)]
    live = [int(cnt.live[d]) for d in range(D + 2)]
    # paths that actually travel through the path pools: a survivor that certainly misses everything ends at its scatter and
    # is counted in live[d + 1] (it does enter that bounce and miss) without ever being written or read
    moved = [int(cnt.live[d]) - int(cnt.ended_early[d]) for d in range(D + 2)]
    hits = int(cnt.light_hits)
    # algorithmic HBM bytes of the bounce launches: read every queued path (bounce 1 builds its camera
    # rays in registers and reads nothing), write every queued survivor (nothing is written after the last
    # bounce), park the radiance of every emitter hit
    bounce_bytes = sum(PATH_BYTES * moved[d] for d in range(2, D + 1)) \
        + sum(PATH_BYTES * moved[d + 1] for d in range(1, D)) + ACCUM_BYTES * hits
    launches = max(int(cnt.bounce_launches), 1)
    iters_per_launch = maxb      # (N > 1: the last launch of a run may carry fewer steps; the PMC figures are N = 1 only)
    avg_ms = cnt.bounce_kernel_ms / launches
    achieved = bounce_bytes / launches / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    # whole-frame path counts: summed over the ranks (every rank holds the counts of its own rows)
    live_sum, moved_sum, liveA_sum = (sum_over_ranks([sum(live[1:D + 1]), sum(moved[1:D + 1]), sum(liveA[1:D + 1])]))
    # PMC counters of the bounce kernel from the committed rocprofv3 passes.  They are stored PER ITERATION of a launch's
    # batch together with the configuration they were collected on; used only for that very configuration (frame,
    # depth, iterations per launch), scaled by the iterations a launch of THIS run carries -- else null.
    traffic, valu_insts, lds_conf, pmc_src, valu_util = None, None, None, None, None
    try:
        pmc = json.load(open(args.pmc_traffic_json))
        if args.pmc_key:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_configs.json")))[args.pmc_key]
        here = ["%s %dx%d" % (os.path.relpath(args.scene, ROOT), W, H), "%d bounces" % D]
        if world == 1 and pmc.get("workload") == here and pmc.get("iterations_per_launch") == iters_per_launch:
            traffic = pmc["hbm_bytes_per_launch_iteration"] * iters_per_launch
            valu_insts = pmc["valu_wave_insts_per_launch_iteration"] * iters_per_launch
            lds_conf = pmc.get("lds_bank_conflict_cycles_per_launch")
            valu_util = pmc.get("valu_utilisation_counter_derived")
            pmc_src = pmc.get("source")
    except Exception:
        pass
    rate = valu_issue_rate(args.valu_rate_json, 6)

    if rank == 0:
        nominal = P * D * iters_block
        walls = sorted(A["walls"])
        rf = {"bound": "hbm", "kernel": ("k_mesh_walk + k_bounce (the walks of the bounce's rays through the meshes' hierarchies, then the fused bounce: one pair "
                                         "of launches per bounce and batch, timed together)") if getattr(scene, "meshes", None) else
                                        "k_bounce (fused [camera rays+]intersect+shade+compact, one launch per bounce and batch)",
              "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
              "frac": round(achieved / HBM_PEAK_GBS, 5),
              "traffic": traffic,
              "algorithmic_bytes_per_launch": round(bounce_bytes / launches, 1),
              "traffic_over_algorithmic": round(traffic / (bounce_bytes / launches), 3) if traffic else None,
              "avg_launch_ms": round(avg_ms, 5), "launches": launches, "iterations_per_launch": iters_per_launch,
              # the PIPELINED effective rate (labelled as such; `frac` above is the per-launch one): the same algorithmic bytes per step
              # over the headline pass's ms_per_step, i.e. with the batches in flight overlapping each other's ramp-up and tail
              "frac_pipelined": round(bounce_bytes / args.steps / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 5),
              "bounce_kernel_share_of_step": round(cnt.bounce_kernel_ms / (dtB * 1e3), 4),
              "ms_per_step_with_events": round(dtB / args.steps * 1e3, 4),
              "pmc_source": pmc_src}
        if world > 1:
            rf["scope"] = "rank 0's shard (its launches, its rows' paths)"
        if valu_insts and rate:
            # the kernel is VALU-issue-bound, not HBM-bound: wave64 vector instructions per launch (PMC)
            # x MEASURED cycles per instruction per SIMD (profiles/valu_issue_rate.json) / (1024 SIMDs x 2.4 GHz)
            t_mix = valu_insts * rate["mix"] / (SIMDS * CLOCK_HZ) * 1e3
            t_fma = valu_insts * rate["fma"] / (SIMDS * CLOCK_HZ) * 1e3
            rf["valu"] = {"wave_instructions_per_launch": round(valu_insts),
                          # from the counters alone: SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) of the profiled run
                          "utilisation_counter_derived": round(valu_util, 4) if valu_util else None,
                          "cycles_per_instruction_per_simd_measured": {"kernel_like_mix": rate["mix"], "v_fma_f32": rate["fma"],
                                                                       "waves_per_simd": rate["waves_per_simd"]},
                          "issue_bound_ms_per_launch": {"kernel_like_mix": round(t_mix, 5), "v_fma_f32": round(t_fma, 5)},
                          "frac_of_issue_bound": {"kernel_like_mix": round(t_mix / avg_ms, 4), "v_fma_f32": round(t_fma / avg_ms, 4)},
                          "lds_bank_conflict_cycles_per_launch": lds_conf}
        if args.roofline_bound == "valu_fp32":
            # a configuration whose kernel is bound by the vector units (C5: 70 primitives, ~7 kflop per segment) is priced
            # against the vector-fp32 peak (MI355X_MICROARCH.md: 157.3 TFLOP/s = 1024 SIMDs x 64 FLOP per clock x 2.4 GHz, an FMA
            # counting two).  `achieved` counts every wave64 vector instruction of the launch (PMC: SQ_INSTS_VALU of THIS configuration)
            # as one such issue slot -- 64 lanes x 2 FLOP -- so `frac` is the share of the SIMDs' issue slots the kernel fills at the
            # maximum clock: an upper bound of the arithmetic actually done (compares, selects and moves fill slots too).
            hbm = {k: rf[k] for k in ("achieved", "peak", "unit", "frac", "frac_pipelined", "algorithmic_bytes_per_launch", "traffic_over_algorithmic")}
            v_ach = valu_insts * 128.0 / (avg_ms * 1e-3) / 1e12 if (valu_insts and avg_ms > 0) else None
            rf.update({"bound": "valu_fp32", "achieved": round(v_ach, 2) if v_ach else None, "peak": 157.3, "unit": "TFLOP/s",
                       "frac": round(v_ach / 157.3, 5) if v_ach else None,
                       "counted": "SQ_INSTS_VALU per launch x 64 lanes x 2 FLOP (one FMA issue slot each) / the launch's HIP-event duration",
                       "hbm": hbm})
            rf.pop("frac_pipelined", None)
        out = {
            "metric": "Mpaths/sec (paths = pixels x bounces x spp) at %dx%d, %d bounces" % (W, H, D),
            "value": round(nominal / dt / 1e6, 2), "unit": "Mpaths/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            # the timed block (exactly `steps` steps between barrier + synchronise) repeated inside this run: median above
            "repeats": len(walls),
            "ms_per_step_min": round(walls[0] / args.steps * 1e3, 4), "ms_per_step_max": round(walls[-1] / args.steps * 1e3, 4),
            "value_min": round(nominal / walls[-1] / 1e6, 2), "value_max": round(nominal / walls[0] / 1e6, 2),
            "ms_per_step_blocks": [round(w / args.steps * 1e3, 4) for w in A["walls"]],
            "config": {"workload": "%s %dx%d, %d bounces, %d spp per step x %d steps%s" % (
                           os.path.relpath(args.scene, ROOT), W, H, D, I, args.steps,
                           "" if world == 1 else ", rows sharded y%%%d + %s %s of the row blocks per %s" % (
                               world, "RCCL" if backend == "nccl" else backend, args.collective, "batch" if every == "batch" else "iteration")),
                       "iterations_per_step": I,
                       "paths_per_step_nominal": P * D * I,
                       "ms_per_iteration": round(dt / iters_block * 1e3, 5),
                       "iterations_per_wavefront_batch": maxb,
                       "batches_in_flight": args.pipeline if args.pipeline > 0 else 3,
                       "collective_every": None if world == 1 else every,
                       "live_segments_per_iteration": round(live_sum / max(iters_block, 1), 1),
                       # unterminated paths entering bounce d = 1 .. depth, per iteration (rank 0's rows when N > 1): the reference's
                       # compaction analysis, README.md:284-293
                       "live_per_bounce_per_iteration": [round(live[d] / max(iters_block, 1), 1) for d in range(1, D + 1)],
                       "queued_segments_per_iteration": round(moved_sum / max(iters_block, 1), 1),
                       "live_Msegments_per_s": round(liveA_sum / len(walls) / dt / 1e6, 2)},
            "roofline": rf,
        }
        if multi:
            out["value_" + args.scaling] = {"value": out["value"], "unit": "Mpaths/s", "steps": args.steps, "iterations_per_step": I,
                                            "iterations_per_wavefront_batch": maxb, "ms_per_step": out["ms_per_step"],
                                            "mode": "%s scaling, one %s per %s" % (args.scaling, args.collective,
                                                                                 "wavefront batch" if every == "batch" else "iteration")}
            for k in ("value_weak", "value_strong"):
                if k in multi:
                    out[k] = multi.pop(k)
            out["multi_gpu"] = multi
        if c3 is not None:
            out["value_c3_as_written"] = c3
        if groups:
            one = groups.get("group_8_members_one_device")
            if one and "value" in one:
                one["of_the_one_context_rate"] = round(one["value"] / out["value"], 4)
            gc3 = groups.pop("c3_as_written", None)
            if gc3 is not None and world == 1:
                # N = 1: the protocol through the LIBRARY's own collective is the reading; the torch.distributed one stands beside it
                if c3 is not None:
                    out["value_c3_as_written_torch_distributed"] = c3
                out["value_c3_as_written"] = gc3
            out.update(groups)
            out["multi_device_note"] = ("pt_group's RCCL reduce has only ever run in a ONE-rank communicator: no multi-GPU node was available to any build round; "
                                        "N > 1 figures of either path are unmeasured unless this line's n_gpus says otherwise")
        out["box_calibration"] = box_calibration(pt, torch)
        tm = telemetry.report() if telemetry else None
        out["box_calibration"]["telemetry_during_timed_blocks"] = tm
        # the vector-issue bound once more at the shader clock the card actually ran at (the figures above price it at the 2.4 GHz maximum;
        # under this load the card settles some 10 % below it, at ~1.2 of its 1.4 kW cap)
        try:
            sclk = tm["sclk_mhz"]["median"] * 1e6
            v = rf.get("valu")
            if v and sclk > 0:
                v["frac_of_issue_bound_at_measured_sclk"] = {k: round(x * CLOCK_HZ / sclk, 4) for k, x in v["frac_of_issue_bound"].items()}
                v["measured_sclk_mhz"] = tm["sclk_mhz"]["median"]
        except (TypeError, KeyError):
            pass
        if world == 1 and args.cpu_spp > 0:
            out["cpu_baseline"] = cpu_baseline(args, scene, pt)
        if world == 1 and args.configs:
            out["configs"] = other_configs(args)
        # a block measured by a child process that failed stands in the line with an `error` key -- and is NAMED here, so that a line that
        # lacks a configuration cannot pass for a complete one
        failed = [k for k, v in out.get("configs", {}).items() if "error" in v]
        failed += [k for k in ("group_8_members_one_device", "group", "group_c3_as_written", "value_c3_as_written") if isinstance(out.get(k), dict) and "error" in out[k]]
        if "configs" in out or groups:
            out["configs_failed"] = failed
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
