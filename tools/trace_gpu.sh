#!/bin/bash
# tools/trace_gpu.sh TAG [extra bench.py args] -- kernel trace only (the first pass of tools/profile_gpu.sh), then a
# per-kernel table of the large-fan launches on stdout.  Run on the GPU box through gpurun, from the repo root.
set -e
TAG=${1:?tag}
shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/trace_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o t -- python3 $REPO/bench.py --no-cpu --steps 10 --warmup 3 --large-steps 3 "$@" > "$OUT/trace.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r['Kernel_Name'][:90]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
for k, v in sorted(d.items(), key=lambda kv: -max(kv[1])):
    v.sort()
    print(f'{k:90s} n={len(v):4d} max={v[-1]:8.3f} med={v[len(v)//2]:8.3f} ms')
PY
