#!/bin/bash
# tools/pmc_count_default.sh TAG [args of tools/count_default.py] -- where the cycles of the count pass at default step
# sizes go: three counter passes (each its own run), summarised per launch.  Run through gpurun from the repo root.
set -e
TAG=${1:?tag}; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_count_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="$REPO/tools/count_default.py $*"
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/t" -o p -- python3 $CMD > "$OUT/t.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d "$OUT/p1" -o p -- python3 $CMD > "$OUT/p1.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU --output-format csv -d "$OUT/p2" -o p -- python3 $CMD > "$OUT/p2.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_IFETCH SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F64 SQ_THREAD_CYCLES_VALU --output-format csv -d "$OUT/p3" -o p -- python3 $CMD > "$OUT/p3.log" 2>&1 || echo "pass 3 failed (a counter name?)"
python3 - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv, sys, collections, re, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
def short(n): return re.sub(r'\(.*\)$', '', re.sub(r'^void ', '', n))
for f in glob.glob(f'{sys.argv[1]}/t/**/p_kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'rt2d_cast' in r['Kernel_Name']:
            g = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])
            dur[(short(r['Kernel_Name']), g)].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for sub in ('p1', 'p2', 'p3'):
    for f in glob.glob(f'{sys.argv[1]}/{sub}/**/p_counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'rt2d_cast' in r['Kernel_Name']:
                acc[(short(r['Kernel_Name']), int(r['Grid_Size']))][r['Counter_Name']].append(float(r['Counter_Value']))
for key, d in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    m = {k: sum(v) / len(v) for k, v in acc.get(key, {}).items()}
    print(f'{key[0]} grid={key[1]}: {len(d)} launches, mean {sum(d)/len(d)/1e3:.1f} us')
    for k, v in sorted(m.items()): print(f'    {k} = {v:.5g}')
PY
