set -e
cd /root/repo
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_distributed.py tests/test_gpu_chunked.py tests/test_gpu_unsorted.py -x -q --timeout 300 --timeout-method thread
timeout -k 10 600 python3 bench.py --gpus 2 --steps 20 --warmup 3 --large-steps 1 --no-cpu > gpurun_out/r2_bench_h2.log 2> gpurun_out/r2_bench_h2.err
timeout -k 10 600 python3 bench.py --force-collectives --steps 30 --warmup 3 --large-rays 0 --no-cpu > gpurun_out/r2_bench_h1.log 2> gpurun_out/r2_bench_h1.err
timeout -k 10 600 python3 bench.py --steps 30 --warmup 3 --large-rays 0 --no-cpu > gpurun_out/r2_bench_h0.log 2> gpurun_out/r2_bench_h0.err
