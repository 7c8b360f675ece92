set -x
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_bench.py tests/test_gpu_fast.py -x -q > gpurun_out/r05_bench_tests.log 2>&1; tail -15 gpurun_out/r05_bench_tests.log
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k config5 > gpurun_out/r05_config5.log 2>&1; tail -5 gpurun_out/r05_config5.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_driver_cmd.json 2> gpurun_out/r05_bench_driver_cmd.err; tail -3 gpurun_out/r05_bench_driver_cmd.err
timeout -k 10 600 python bench.py --as-rank 3 8 > gpurun_out/r05_bench_rank3of8.json 2> gpurun_out/r05_bench_rank3of8.err; tail -3 gpurun_out/r05_bench_rank3of8.err
