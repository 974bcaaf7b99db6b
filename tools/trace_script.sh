#!/bin/bash
# tools/trace_script.sh TAG SCRIPT [args] -- rocprofv3 kernel trace of a Python script of this repository; prints, per kernel
# and grid, the number of launches, the median and the total of their durations.  Run through gpurun from the repo root.
set -e
TAG=${1:?tag}; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/trace_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o t -- python3 $REPO/$* > "$OUT/trace.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, re
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    g = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])
    name = re.sub(r'\(.*\)$', '', re.sub(r'^void ', '', r['Kernel_Name']))[:70]
    d[(name, g)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for (k, g), v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:24]:
    v.sort()
    print(f'{k:70s} grid {g:>10d} n={len(v):4d} med={v[len(v)//2]:10.1f} us total={sum(v)/1e3:9.2f} ms')
PY
