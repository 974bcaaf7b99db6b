#!/bin/bash
# tools/pmc_mid.sh -- counters of the tile binner on a 65 536-ray fan (tools/prof_multifreq.py): two --pmc passes of their own
set -e
R=$(pwd); OUT=$R/gpurun_out/pmc_mid; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR --output-format csv -d $OUT/sq -o s -- python3 $R/tools/prof_multifreq.py > $OUT/sq.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc TCC_ATOMIC_sum FETCH_SIZE --output-format csv -d $OUT/tcc -o c -- python3 $R/tools/prof_multifreq.py > $OUT/tcc.log 2>&1
python3 - $OUT <<'PY'
import csv, sys, collections
for sub, f in (('sq', 's_counter_collection.csv'), ('tcc', 'c_counter_collection.csv')):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f'{sys.argv[1]}/{sub}/{f}')):
        if 'heatmap_fast' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(sub, {k: (round(min(v)), round(sum(v) / len(v)), round(max(v))) for k, v in acc.items()})
PY
