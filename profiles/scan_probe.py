"""Throughput of the stream-compaction library (pt_scan_exclusive_i32 / pt_compact_nonzero_i32, the reference's empty
stream_compaction/ stub, README.md:83-86) on device buffers.     python profiles/scan_probe.py [log2 n] [reps]
Algorithmic bytes: scan 8 n (read + write), compaction 4 n + 4 kept (+ 8 B count).  Under rocprofv3 (profiles/run_scan.sh) the
same run gives the kernels' durations, HBM traffic and LDS bank conflicts."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pt = ge.load_package()
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 26
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = 1 << lg
light = os.environ.get("SCAN_PROBE_LIGHT") == "1"      # under a rocprofv3 counter pass: few framework kernels, no self-check
if light:
    i = torch.arange(n, device="cuda", dtype=torch.int32)
    vals = torch.where(i % 10 < 3, i % 997 + 1, torch.zeros_like(i))         # 30 % survivors
    del i
else:
    g = torch.Generator(device="cuda").manual_seed(565)
    flags = (torch.rand(n, device="cuda", generator=g) < 0.3).to(torch.int32)    # 30 % survivors, like a bounce
    vals = torch.randint(0, 1000, (n,), device="cuda", dtype=torch.int32, generator=g) * flags
out = torch.empty_like(vals)
cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
for name, fn, nbytes in (("scan", lambda: pt.scan_exclusive_dev(vals.data_ptr(), out.data_ptr(), n), 8 * n),
                         ("compact", lambda: pt.compact_nonzero_dev(vals.data_ptr(), out.data_ptr(), n, cnt.data_ptr()), None)):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    if nbytes is None:
        nbytes = 4 * n + 4 * int(cnt.item()) + 8
    print("%s: n = 2^%d, %.3f ms per call, %.0f GB/s algorithmic (%.1f %% of 8 TB/s)" % (name, lg, dt * 1e3, nbytes / dt / 1e9, nbytes / dt / 8e10))
if light:
    sys.exit(0)
ref = torch.cumsum(vals.to(torch.int64), 0) - vals
pt.scan_exclusive_dev(vals.data_ptr(), out.data_ptr(), n)
torch.cuda.synchronize()
assert torch.equal(out.to(torch.int64) & 0xffffffff, ref & 0xffffffff), "scan mismatch"
print("scan equals torch.cumsum (mod 2^32); compaction kept %d of %d" % (int(cnt.item()), n))
