#!/bin/bash
# tools/profile_gpu.sh TAG [extra bench.py args]  -- run on the GPU box (through gpurun), from the repo root.
#
# Collects, for the bench.py workload, under gpurun_out/prof_TAG/:
#   trace/   rocprofv3 --kernel-trace --stats   (per-dispatch durations: 999-ray fan and the 1M-ray large fan)
#   fetch/   rocprofv3 --pmc FETCH_SIZE         (HBM read traffic per dispatch; its own pass, no tracing domains)
#   write/   rocprofv3 --pmc WRITE_SIZE         (HBM write traffic per dispatch; its own pass)
#   valu/    rocprofv3 --pmc 8 SQ counters      (vector / scalar instructions issued, FP64 op mix, waves; its own pass)
#   atomic/  rocprofv3 --pmc TCC_ATOMIC_sum ... (L2 atomic requests per dispatch; its own pass)
# tools/summarize_profile.py turns them into the files committed under profiles/.
# Every step has its own time limit and the chain stops at the first failure.
set -e
TAG=${1:?tag}
shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
# the build these passes measure (hash of the kernel sources; the git commit it was built at, if the library was built
# in the development container): stamped into every summary by tools/summarize_profile.py
python3 -c "import sys, json; sys.path.insert(0, '$REPO'); import rt2d_loader; rt = rt2d_loader.load(); print(json.dumps(rt._lib.build_info()))" > "$OUT/build.json"
cd /tmp && export TMPDIR=/tmp
BENCH="$REPO/bench.py --no-cpu --profile-pass $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 $BENCH --steps 30 --warmup 3 --large-steps 3 > "$OUT/trace.log" 2>&1
echo "trace done"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o f -- python3 $BENCH --steps 3 --warmup 1 --large-steps 2 > "$OUT/fetch.log" 2>&1
echo "fetch done"
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o w -- python3 $BENCH --steps 3 --warmup 1 --large-steps 2 > "$OUT/write.log" 2>&1
echo "write done"
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS --output-format csv -d "$OUT/valu" -o v -- python3 $BENCH --steps 3 --warmup 1 --large-steps 2 > "$OUT/valu.log" 2>&1
echo "valu done"
timeout -s KILL 240 rocprofv3 --pmc TCC_ATOMIC_sum TCC_REQ_sum --output-format csv -d "$OUT/atomic" -o a -- python3 $BENCH --steps 3 --warmup 1 --large-steps 2 > "$OUT/atomic.log" 2>&1
echo "atomic done"
ls -R "$OUT" | head -40
