"""GPU occupancy of a pipelined run from a rocprofv3 kernel trace: over a window of camera-ray launches (default: the 3rd to
the last-but-2nd) -- wall span, time with at least one kernel running (union), time with two or more,
idle gaps, and the sums per kernel.     python profiles/timeline.py <dir containing */*kernel_trace.csv>"""
import csv, glob, os, re, sys
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
first = [i for i, r in enumerate(rows) if 'k_bounce<true' in r['Kernel_Name']]
# optional: python profiles/timeline.py <dir> <a> <b> = from the a-th to the b-th camera-ray launch (bench.py runs a pipelined
# pass, then a second pass with one batch in flight: `--steps 20 --warmup 5` -> launches 0..24 are the pipelined pass)
a, b = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2, len(first) - 3)
lo, hi = int(rows[first[a]]['Start_Timestamp']), int(rows[first[b]]['Start_Timestamp'])
ev, per = [], {}
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    s, e = max(s, lo), min(e, hi)
    if e <= s:
        continue
    ev += [(s, 1), (e, -1)]
    m = re.search(r'(k_\w+(<[^>]*>)?)', r['Kernel_Name'])
    k = m.group(1) if m else r['Kernel_Name'][:30]
    per[k] = per.get(k, 0) + (e - s)
ev.sort()
depth, last, busy1, busy2, gaps = 0, lo, 0, 0, []
for t, d in ev:
    if depth >= 1:
        busy1 += t - last
    if depth >= 2:
        busy2 += t - last
    if depth == 0 and t > last:
        gaps.append(t - last)
    depth += d
    last = t
span = hi - lo
nb = b - a
print('window %.3f ms, %d batches: %.4f ms per batch' % (span / 1e6, nb, span / 1e6 / nb))
print('>= 1 kernel running %.1f %%, >= 2 running %.1f %%, idle %.1f %% in %d gaps (largest %.1f us)' % (
    100 * busy1 / span, 100 * busy2 / span, 100 * (span - busy1) / span, len(gaps), max(gaps or [0]) / 1e3))
for k, v in sorted(per.items(), key=lambda kv: -kv[1]):
    print('  %-40s %.4f ms per batch' % (k, v / 1e6 / nb))
