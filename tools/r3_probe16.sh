mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_live_erosion.py tests/test_gpu_parity.py -m gpu -q -x -k "live or pool or erosion or config4" > gpurun_out/t5.log 2>&1; echo "rc live $?" >> gpurun_out/t5.log
tail -n 4 gpurun_out/t5.log
timeout -k 10 600 python tools/bench_config4.py --json gpurun_out/r03_config4b.json > gpurun_out/config4.txt 2>&1; grep -A12 '"per_job_ms"' gpurun_out/config4.txt | tail -n 30; grep "wet\|cycle_ms\|driver" gpurun_out/config4.txt
