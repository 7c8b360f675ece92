mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_live_erosion.py -m gpu -q -x > gpurun_out/t5.log 2>&1; echo "rc live $?" >> gpurun_out/t5.log
tail -n 8 gpurun_out/t5.log
timeout -k 10 600 python tools/bench_config4.py --at 1,100,1000 --json gpurun_out/r03_config4.json > gpurun_out/config4.txt 2>&1; tail -n 120 gpurun_out/config4.txt
