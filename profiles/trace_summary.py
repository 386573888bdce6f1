import csv, glob, sys
f = glob.glob(sys.argv[1]+'/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx = [i for i,r in enumerate(rows) if 'generate' in r['Kernel_Name'] or 'k_bounce<true' in r['Kernel_Name']]
i0 = idx[len(idx)//2]
prev_end=None
n = (idx[len(idx)//2+1]-i0+1) if len(idx) > len(idx)//2+1 else 10
for r in rows[i0:i0+n]:
    st,en=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print('%-34s dur %7d ns  gap %s  grid %s vgpr %s sgpr %s lds %s' % (r['Kernel_Name'].split('::')[-1].split('(')[0][:34], en-st, (st-prev_end) if prev_end else '-', r['Grid_Size_X'], r['VGPR_Count'], r['SGPR_Count'], r['LDS_Block_Size']))
    prev_end=en
print('iteration span ns:', int(rows[i0+n-1]['Start_Timestamp'])-int(rows[i0]['Start_Timestamp']))
