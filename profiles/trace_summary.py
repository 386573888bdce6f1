import csv, glob, sys
f = glob.glob(sys.argv[1]+'/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx = [i for i,r in enumerate(rows) if 'generate' in r['Kernel_Name']]
i0 = idx[len(idx)//2]
prev_end=None
for r in rows[i0:i0+10]:
    st,en=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print('%-28s dur %7d ns  gap %s  grid %s vgpr %s sgpr %s lds %s' % (r['Kernel_Name'].split('::')[-1][:28], en-st, (st-prev_end) if prev_end else '-', r['Grid_Size_X'], r['VGPR_Count'], r['SGPR_Count'], r['LDS_Block_Size']))
    prev_end=en
print('iteration span ns:', int(rows[i0+9]['Start_Timestamp'])-int(rows[i0]['Start_Timestamp']))
