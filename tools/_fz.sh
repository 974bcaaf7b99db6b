set -e
cd /root/repo
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2_tests_m.log 2>&1
timeout -k 10 200 python3 tools/small_fan_timeline.py > gpurun_out/small0.log 2>&1
