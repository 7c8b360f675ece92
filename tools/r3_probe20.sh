mkdir -p gpurun_out

rm -f gpurun_out/fuse.txt
run() { echo "== $1 | $2" >> gpurun_out/fuse.txt; env $1 python bench.py $2 --no-extras --no-cpu-baseline --steps 200 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().split('\n')[-1]);print(d['value'],d['ms_per_step'],(d.get('stage_by_stage') or {}).get('ms_per_step'),{k:v['ms'] for k,v in d['stages'].items()})" >> gpurun_out/fuse.txt; }
run "X=1" "--schedule pipeline"
run "X=1" "--schedule stages"
run "NZ_FLOW_STREAM_WAVES=4096" "--schedule pipeline"
run "NZ_CONV_CHAIN=0" "--schedule pipeline"
run "X=1" "--schedule pipeline"
run "X=1" "--schedule stages"
cat gpurun_out/fuse.txt
