set -x
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_bench.py -q -x > gpurun_out/t4.log 2>&1; echo "rc bench tests $?" >> gpurun_out/t4.log
tail -n 15 gpurun_out/t4.log
bash tools/collect_profiles.sh r03_v1 $(cat gpurun_out/.commit 2>/dev/null || echo unknown) swap > gpurun_out/collect_r03_v1.log 2>&1; tail -n 3 gpurun_out/collect_r03_v1.log
python -c "
import json;d=json.loads(open('gpurun_out/r03_v1_bench.json').read().strip().split('\n')[-1]);print(d['value'],d['ms_per_step'],{k:(v['ms'],v.get('valu_issue_frac'),v.get('useful_valu_frac')) for k,v in d['stages'].items()});print(d['roofline']);print(d.get('step_valu'))"
