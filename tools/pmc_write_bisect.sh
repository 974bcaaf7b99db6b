#!/bin/bash
# tools/pmc_write_bisect.sh -- WRITE_SIZE of the 1M-ray storing cast under the stepping kernel's switches (one --pmc pass each)
REPO=$(pwd); OUT=$REPO/gpurun_out/write_bisect; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" true; ( export "$@"; timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/$tag" -o w -- python3 $REPO/tools/cast_write_traffic.py > "$OUT/$tag.log" 2>&1 ) || return 1; }
run default RT2D_X=0 && run tight0 RT2D_CAST_TIGHT=0 && run tab1 RT2D_CAST_TAB=1 && run tab0 RT2D_CAST_TAB=0
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
for tag in ('default', 'tight0', 'tab1', 'tab0'):
    acc = collections.defaultdict(list)
    for f in glob.glob(f'{sys.argv[1]}/{tag}/**/w_counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'rt2d_cast_kernel' in r['Kernel_Name'] and r['Counter_Name'] == 'WRITE_SIZE':
                acc[(r['Kernel_Name'].split('(')[0][-40:], int(r['Grid_Size']))].append(float(r['Counter_Value']))
    for k, v in acc.items():
        print(tag, k, [f'{x * 1024 / 1e9:.3f} GB' for x in v])
    print(open(f'{sys.argv[1]}/{tag}.log').read().strip().splitlines()[-1])
PY
