#!/bin/bash
# tools/pmc_large.sh TAG [extra bench.py args] -- where the wave-cycles of the 1M-ray kernels go: SQ counter passes of their
# own (no tracing domains), summary per (kernel, grid) on stdout and in gpurun_out/pmc_large_TAG/summary.txt.
# Run through gpurun from the repo root.
set -e
TAG=${1:?tag}
shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_large_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="$REPO/bench.py --no-cpu --steps 2 --warmup 1 --large-steps 1 $*"
timeout -s KILL 170 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d "$OUT/p1" -o p -- python3 $BENCH > "$OUT/p1.log" 2>&1
echo "pass 1 done"
timeout -s KILL 170 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d "$OUT/p2" -o p -- python3 $BENCH > "$OUT/p2.log" 2>&1
echo "pass 2 done"
timeout -s KILL 170 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d "$OUT/p3" -o p -- python3 $BENCH > "$OUT/p3.log" 2>&1
echo "pass 3 done"
python3 - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv, sys, collections, re, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ('p1', 'p2', 'p3'):
    for f in glob.glob(f'{sys.argv[1]}/{sub}/**/p_counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'rt2d_' not in r['Kernel_Name']:
                continue
            name = re.sub(r'\(.*\)$', '', re.sub(r'^void ', '', r['Kernel_Name']))
            acc[(name, int(r['Grid_Size']))][r['Counter_Name']].append(float(r['Counter_Value']))
for (name, grid), c in sorted(acc.items(), key=lambda kv: -kv[0][1]):
    if grid < 100000:
        continue
    m = {k: sum(v) / len(v) for k, v in c.items()}
    w = m.get('SQ_WAVES', 0) or 1
    print(f'{name} grid={grid} launches={len(next(iter(c.values())))}')
    print('   ' + '  '.join(f'{k}={v:.4g}' for k, v in sorted(m.items())))
    if 'SQ_WAVE_CYCLES' in m:
        wc = m['SQ_WAVE_CYCLES']
        print('   of wave-cycles: ' + '  '.join(f'{k[3:]}={m[k] / wc:.1%}' for k in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_SCA', 'SQ_ACTIVE_INST_LDS', 'SQ_WAIT_INST_LDS') if k in m))
PY
