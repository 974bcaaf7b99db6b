#!/bin/bash
# tools/pmc_write_small.sh -- WRITE_SIZE of the 999-ray storing cast as 250 waves of 4 rays (default) and as 16 waves of 64
# rays (RT2D_CAST_CUS=16: whole 512-byte rows per store): are the bytes beyond 16 B per point partial-row write-backs?
REPO=$(pwd); OUT=$REPO/gpurun_out/write_small; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
( timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/lanes4" -o w -- python3 $REPO/tools/cast_write_traffic.py 999 > "$OUT/lanes4.log" 2>&1 ) || exit 1
( export RT2D_CAST_CUS=16; timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/lanes64" -o w -- python3 $REPO/tools/cast_write_traffic.py 999 > "$OUT/lanes64.log" 2>&1 ) || exit 1
( export RT2D_CAST_CUS=64; timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/lanes16" -o w -- python3 $REPO/tools/cast_write_traffic.py 999 > "$OUT/lanes16.log" 2>&1 ) || exit 1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
for tag in ('lanes4', 'lanes16', 'lanes64'):
    acc = collections.defaultdict(list)
    for f in glob.glob(f'{sys.argv[1]}/{tag}/**/w_counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'rt2d_cast_kernel' in r['Kernel_Name'] and r['Counter_Name'] == 'WRITE_SIZE':
                acc[(r['Kernel_Name'].split('(')[0][-40:], int(r['Grid_Size']))].append(float(r['Counter_Value']))
    for k, v in acc.items():
        print(tag, k, [f'{x * 1024 / 1e6:.3f} MB' for x in v])
    print([l for l in open(f'{sys.argv[1]}/{tag}.log').read().splitlines() if l.startswith('rays')][-1])
PY
