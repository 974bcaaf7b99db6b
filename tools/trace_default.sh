#!/bin/bash
# tools/trace_default.sh [rays] -- kernel timeline of the 65 536-ray fan at the reference's default step sizes through
# cast(store=False) + heatmap() (fused path): the launches longer than 2 ms with their start, duration, grid and LDS.
# Run through gpurun from the repo root; copies the rocprofv3 stats to gpurun_out/prof_def/.
set -e
RAYS=${1:-65536}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_def
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o def -- python3 $REPO/tools/default_res.py --rays $RAYS --reps 1 --paths fused > "$OUT/run.log" 2>&1
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/def_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
t0 = min(int(r['Start_Timestamp']) for r in rows)
big = [r for r in rows if int(r['End_Timestamp']) - int(r['Start_Timestamp']) > 2e6]
for r in big[-16:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s - t0) / 1e6:9.1f} ms +{(e - s) / 1e6:7.1f} ms  grid {r['Grid_Size_X']:>8} wg {r['Workgroup_Size_X']:>4} lds {r.get('LDS_Block_Size', '?'):>7}  {r['Kernel_Name'][:80]}")
PY
