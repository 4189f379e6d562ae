import csv,glob,sys
for f in sorted(glob.glob('/root/repo/gpurun_out/kt_d*/vm/*kernel_trace.csv')):
    print(f)
    rows=list(csv.DictReader(open(f)))
    rows.sort(key=lambda r:int(r['Start_Timestamp']))
    sel=[r for r in rows if 'ransac' in r['Kernel_Name']]
    idx=max(i for i,r in enumerate(sel) if 'prep' in r['Kernel_Name'])
    t0=int(sel[idx]['Start_Timestamp'])
    for r in sel[idx:]:
        s=int(r['Start_Timestamp']);e=int(r['End_Timestamp'])
        if (e-s)>30e3: print(f"  {(s-t0)/1e3:9.1f} us +{(e-s)/1e3:9.1f} us  grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}  {r['Kernel_Name'][:70]}")
