set -e
mkdir -p gpurun_out/r6
python -m pytest tests/test_reference_screenshots.py tests/test_gpu_bench.py -x -q -m gpu > gpurun_out/r6/t1.log 2>&1 || { tail -30 gpurun_out/r6/t1.log; exit 1; }
tail -3 gpurun_out/r6/t1.log
python bench.py > gpurun_out/r6/bench0.json 2> gpurun_out/r6/bench0.err; echo "bench rc $?"
python -c "
import json; d=json.load(open('gpurun_out/r6/bench0.json'))
print(d['value'], d['ms_per_step'], d['verified'], {k:v['ms'] for k,v in d['stages'].items()})
print({m:(d['float_modes'][m]['ms_per_step'], d['float_modes'][m].get('stages_within_1e-5')) for m in ('strict','fast','relaxed')})
"
