#!/bin/bash
# tools/pmc_any.sh TAG SCRIPT [args] -- SQ / TCC counter passes (each its own run, no tracing domains) of a Python script,
# summarised per rt2d kernel (and grid size).  Run through gpurun from the repo root.
set -e
TAG=${1:?tag}; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_any_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 170 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/t" -o p -- python3 $REPO/$* > "$OUT/t.log" 2>&1
timeout -s KILL 170 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d "$OUT/p1" -o p -- python3 $REPO/$* > "$OUT/p1.log" 2>&1
timeout -s KILL 170 rocprofv3 --pmc TCC_ATOMIC_sum TCC_REQ_sum --output-format csv -d "$OUT/p2" -o p -- python3 $REPO/$* > "$OUT/p2.log" 2>&1
python3 - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv, sys, collections, re, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
def short(n): return re.sub(r'\(.*\)$', '', re.sub(r'^void ', '', n))
for f in glob.glob(f'{sys.argv[1]}/t/**/p_kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'rt2d_' in r['Kernel_Name']:
            g = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])
            dur[(short(r['Kernel_Name']), g)].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for sub in ('p1', 'p2'):
    for f in glob.glob(f'{sys.argv[1]}/{sub}/**/p_counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'rt2d_' in r['Kernel_Name']:
                acc[(short(r['Kernel_Name']), int(r['Grid_Size']))][r['Counter_Name']].append(float(r['Counter_Value']))
for key, d in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    if sum(d) < 2e5: continue
    m = {k: sum(v) / len(v) for k, v in acc.get(key, {}).items()}
    print(f'{key[0]} grid={key[1]}: {len(d)} launches, mean {sum(d)/len(d)/1e3:.1f} us, total {sum(d)/1e6:.2f} ms')
    if m: print('    ' + '  '.join(f'{k}={v:.4g}' for k, v in sorted(m.items())))
PY
