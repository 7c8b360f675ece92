set -x
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/t1.log 2>&1; echo "rc default suite $?" >> gpurun_out/t1.log
NZ_FLOW_STREAM=2 timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "flow or pipeline or metric or sweep or sharded or stripe or batch or rw or smoke or demo" > gpurun_out/t2.log 2>&1; echo "rc stream-all $?" >> gpurun_out/t2.log
tail -3 gpurun_out/t1.log gpurun_out/t2.log
python tools/bench_stage.py all --res 4096 > gpurun_out/stage_default.txt 2>&1
NZ_FLOW_STREAM=0 python tools/bench_stage.py flow >> gpurun_out/stage_default.txt 2>&1
for W in 2048 3072 4096 6144 9216; do echo "WAVES $W" >> gpurun_out/stage_default.txt; NZ_FLOW_STREAM_WAVES=$W python tools/bench_stage.py flow >> gpurun_out/stage_default.txt 2>&1; done
cat gpurun_out/stage_default.txt
python bench.py --no-extras > gpurun_out/bench1.json 2> gpurun_out/bench1.err; tail -c 1500 gpurun_out/bench1.json
