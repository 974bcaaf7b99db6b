#!/bin/bash
# tools/pmc_small.sh TAG -- where a lone wave's cycles go in the 999-ray storing cast: two counter passes of their own
# (SQ wave / wait / issue cycles, then the instruction cache), summary on stdout.  Run through gpurun from the repo root.
set -e
TAG=${1:?tag}
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_small_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_IFETCH SQ_WAIT_INST_LDS --output-format csv -d "$OUT/sq" -o s -- python3 $REPO/bench.py --no-cpu --steps 3 --warmup 1 --large-rays 0 > "$OUT/sq.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES --output-format csv -d "$OUT/ic" -o i -- python3 $REPO/bench.py --no-cpu --steps 3 --warmup 1 --large-rays 0 > "$OUT/ic.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, sys, collections
for sub, f in (('sq', 's_counter_collection.csv'), ('ic', 'i_counter_collection.csv')):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f'{sys.argv[1]}/{sub}/{f}')):
        if 'rt2d_cast_kernel<true' in r['Kernel_Name']:
            acc[r['Grid_Size']][r['Counter_Name']].append(float(r['Counter_Value']))
    for g, c in acc.items():
        print(sub, 'grid', g, {k: round(sum(v) / len(v)) for k, v in c.items()})
PY
