"""Every kernel of the LAST product step in a rocprofv3 kernel trace (gpurun_out/trace_TAG): start relative to the step's
cast, duration, grid, name.  python tools/timeline_all.py gpurun_out/trace_TAG"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'rt2d_cast_kernel<true' in r['Kernel_Name']]
i0 = idx[-1]; t0 = int(rows[i0]['Start_Timestamp'])
prev_end = None
for r in rows[max(0, i0 - 8):]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = re.sub(r'\(.*\)$', '', re.sub(r'^void ', '', r['Kernel_Name']))[:70]
    gap = '' if prev_end is None else f'(gap {(s - prev_end) / 1e3:6.1f})'
    print(f"{(s - t0) / 1e3:9.1f} us +{(e - s) / 1e3:8.1f} us {gap:14s} grid {int(r['Grid_Size_X']):>9}  {name}")
    prev_end = e
