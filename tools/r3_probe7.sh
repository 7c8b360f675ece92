set -x
mkdir -p gpurun_out
NZ_CONV_STREAM=2 NZ_FLOW_STREAM=2 timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "filter or gauss or blur or conv or pipeline or metric or sweep or sharded or stripe or batch or rw or smoke or demo" > gpurun_out/t3.log 2>&1; echo "rc stream-all $?" >> gpurun_out/t3.log
tail -n 3 gpurun_out/t3.log
rm -f gpurun_out/stage.txt
for W in 2048 3072 4096 6144; do echo "WAVES $W" >> gpurun_out/stage.txt; NZ_CONV_STREAM_WAVES=$W python tools/bench_stage.py gauss --reps 300 >> gpurun_out/stage.txt 2>&1; done
echo "TILE" >> gpurun_out/stage.txt; NZ_CONV_STREAM=0 python tools/bench_stage.py gauss --reps 300 >> gpurun_out/stage.txt 2>&1
for C in 3 4 6; do echo "TCAP $C" >> gpurun_out/stage.txt; NZ_CONV_STREAM_WAVES=3072 NZ_CONV_TCAP=$C python tools/bench_stage.py gauss --reps 300 >> gpurun_out/stage.txt 2>&1; done
grep -v amdgpu.ids gpurun_out/stage.txt
