set -x
mkdir -p gpurun_out
NZ_FLOW_STREAM=2 timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "flow or pipeline or metric or sweep or sharded or stripe or batch or rw or smoke or demo" > gpurun_out/t2.log 2>&1; echo "rc stream-all $?" >> gpurun_out/t2.log
tail -n 3 gpurun_out/t2.log
python bench.py --no-extras --no-cpu-baseline > gpurun_out/bench2.json 2> gpurun_out/bench2.err; python -c "
import json;d=json.loads(open('gpurun_out/bench2.json').read().strip().split('\n')[-1]);print(d['value'],d['ms_per_step'],{k:v['ms'] for k,v in d['stages'].items()})"
bash tools/probe_flow_stream.sh > gpurun_out/flow_probe.txt 2>&1; cat gpurun_out/flow_probe.txt
