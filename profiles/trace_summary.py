"""One batch (its 8 bounce launches + commit) from the middle of a rocprofv3 kernel trace: per-launch durations, gaps,
grids and register counts.   python profiles/trace_summary.py <dir containing */*kernel_trace.csv>"""
import csv, glob, re, sys
import os
f = sorted(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))


def short(name):
    m = re.search(r'(k_\w+(<[^>]*>)?)', name)
    return m.group(1) if m else name[:40]


# (a batch starts with the camera-ray bounce -- in a scene with meshes, with the walks that run ahead of it)
first = 'k_mesh_walk<true' if any('k_mesh_walk<true' in r['Kernel_Name'] for r in rows) else 'k_bounce<true'
idx = [i for i, r in enumerate(rows) if first in r['Kernel_Name']]
i0 = idx[len(idx) // 2]
n = (idx[len(idx) // 2 + 1] - i0) if len(idx) > len(idx) // 2 + 1 else 10
prev_end = None
for r in rows[i0:i0 + n]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%-26s dur %8d ns  gap %6s  grid %7s wg %4s vgpr %3s sgpr %3s lds %6s scratch %s' % (
        short(r['Kernel_Name']), en - st, (st - prev_end) if prev_end else '-', r.get('Grid_Size_X', r.get('Grid_Size')),
        r.get('Workgroup_Size_X', r.get('Workgroup_Size')), r.get('VGPR_Count'), r.get('SGPR_Count'), r.get('LDS_Block_Size'),
        r.get('Scratch_Size', r.get('Private_Segment_Size'))))
    prev_end = en
print('batch span ns (first launch start -> next batch\'s first launch start):', int(rows[i0 + n]['Start_Timestamp']) - int(rows[i0]['Start_Timestamp']) if i0 + n < len(rows) else 'n/a')
