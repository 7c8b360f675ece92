mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_live_erosion.py tests/test_gpu_parity.py -m gpu -q -x -k "live or config4 or cpp_host" > gpurun_out/t5.log 2>&1; echo "rc live $?" >> gpurun_out/t5.log
tail -n 4 gpurun_out/t5.log
python tools/bench_config4.py 2>/dev/null | grep -E "driver|cycle_ms"
python tools/bench_config4.py --serial-branch 2>/dev/null | grep -E "driver|cycle_ms"
