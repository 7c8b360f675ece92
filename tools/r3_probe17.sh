mkdir -p gpurun_out
rm -f gpurun_out/stripe.txt
run() { echo "== $1 | $2" >> gpurun_out/stripe.txt; env $1 python bench.py $2 --no-cpu-baseline --steps 30 --warmup 10 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().split('\n')[-1]);print(d['value'],d['ms_per_step'],{k:v['ms'] for k,v in d.get('stages',{}).items()}, d.get('grid_16384'))" >> gpurun_out/stripe.txt; }
run "X=1" "--as-rank 3 8 --no-extras"
run "NZ_CONV_STREAM=2" "--as-rank 3 8 --no-extras"
run "NZ_CONV_STREAM=0 NZ_FLOW_STREAM=0" "--as-rank 3 8 --no-extras"
run "X=1" "--as-rank 0 1 --stripe-rows 16384 --no-extras --steps 10 --warmup 3"
run "NZ_CONV_STREAM=0" "--as-rank 0 1 --stripe-rows 16384 --no-extras --steps 10 --warmup 3"
run "NZ_CONV_STREAM=0 NZ_FLOW_STREAM=0" "--as-rank 0 1 --stripe-rows 16384 --no-extras --steps 10 --warmup 3"
cat gpurun_out/stripe.txt
