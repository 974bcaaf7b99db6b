#!/bin/bash
# tools/pmc_custom.sh TAG "COUNTER COUNTER ..." SCRIPT [args] -- one rocprofv3 --pmc pass (at most 8 SQ counters) of a Python
# script of this repository, summed per rt2d kernel and grid.  Run through gpurun from the repo root.
set -e
TAG=${1:?tag}; CTRS=${2:?counters}; shift; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_custom_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 170 rocprofv3 --pmc $CTRS --output-format csv -d "$OUT" -o p -- python3 $REPO/$* > "$OUT/run.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, sys, collections, re, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
def short(n): return re.sub(r'\(.*\)$', '', re.sub(r'^void ', '', n))[:60]
for f in glob.glob(f'{sys.argv[1]}/**/p_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'rt2d_' in r['Kernel_Name']:
            acc[(short(r['Kernel_Name']), int(r['Grid_Size']))][r['Counter_Name']].append(float(r['Counter_Value']))
for key, d in sorted(acc.items(), key=lambda kv: -max(sum(v) for v in kv[1].values()))[:8]:
    n = len(next(iter(d.values())))
    print(f'{key[0]} grid={key[1]}: {n} launches')
    print('    ' + '  '.join(f'{k}={sum(v)/len(v):.4g}' for k, v in sorted(d.items())))
PY
