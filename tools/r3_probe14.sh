mkdir -p gpurun_out
rm -f gpurun_out/fuse.txt
run() { echo "== $1" >> gpurun_out/fuse.txt; env $1 python bench.py --no-extras --no-cpu-baseline --steps 100 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().split('\n')[-1]);print(d['value'],d['ms_per_step'],(d.get('stage_by_stage') or {}).get('ms_per_step'),{k:v['ms'] for k,v in d['stages'].items()})" >> gpurun_out/fuse.txt; }
run "X=1"
run "NZ_FLOW_STREAM=0"
run "NZ_FLOW_STREAM_WAVES=6144"
run "NZ_FLOW_STREAM_WAVES=4608"
run "NZ_CONV_TCAP=4"
run "NZ_PIPELINE_STRIPES=0"
cat gpurun_out/fuse.txt
