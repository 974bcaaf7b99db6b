import csv, glob, sys, re
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last step: find the last cast kernel with WRITE=true and grid 1M
idx = [i for i, r in enumerate(rows) if 'rt2d_cast_kernel<true' in r['Kernel_Name']]
i0 = idx[-1]
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if e - s < 20000: continue
    name = re.sub(r'\(.*\)$', '', re.sub(r'^void ', '', r['Kernel_Name']))[:60]
    print(f"{(s - t0) / 1e3:9.1f} us +{(e - s) / 1e3:8.1f} us  grid {int(r['Grid_Size_X']):>9}  q{r.get('Queue_Id','?')}  {name}")
