set -x
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/t1.log 2>&1; echo "rc default suite $?" >> gpurun_out/t1.log
tail -n 12 gpurun_out/t1.log
python bench.py > gpurun_out/bench4.json 2> gpurun_out/bench4.err; python -c "
import json;d=json.loads(open('gpurun_out/bench4.json').read().strip().split('\n')[-1]);print(d['value'],d['ms_per_step'],d.get('stage_by_stage'),{k:v['ms'] for k,v in d['stages'].items()});print(d.get('verified'),d.get('verified_detail'));print({k:d.get(k) for k in ('in_place_entries','two_tiles_in_flight','tile_as_two_stripes')})"
tail -n 5 gpurun_out/bench4.err
