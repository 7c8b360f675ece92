set -x
mkdir -p gpurun_out
rm -f gpurun_out/stage.txt
for R in 2048 8192; do
for M in 0 1; do echo "RES $R CONV_STREAM $M" >> gpurun_out/stage.txt; NZ_CONV_STREAM=$M python tools/bench_stage.py gauss --res $R --reps 100 >> gpurun_out/stage.txt 2>&1; done
for M in 0 1; do echo "RES $R FLOW_STREAM $M" >> gpurun_out/stage.txt; NZ_FLOW_STREAM=$M python tools/bench_stage.py flow --res $R --reps 100 >> gpurun_out/stage.txt 2>&1; done
done
echo "RES 8192 CONV_STREAM 1 waves 4096" >> gpurun_out/stage.txt; NZ_CONV_STREAM_WAVES=4096 python tools/bench_stage.py gauss --res 8192 --reps 100 >> gpurun_out/stage.txt 2>&1
echo "RES 8192 CONV_STREAM 1 waves 3072 tcap 6" >> gpurun_out/stage.txt; NZ_CONV_TCAP=6 NZ_CONV_STREAM_WAVES=3072 python tools/bench_stage.py gauss --res 8192 --reps 100 >> gpurun_out/stage.txt 2>&1
grep -v amdgpu.ids gpurun_out/stage.txt
NZ_CONV_STREAM=0 python bench.py --no-cpu-baseline > gpurun_out/bench3.json 2> gpurun_out/bench3.err; python -c "
import json;d=json.loads(open('gpurun_out/bench3.json').read().strip().split('\n')[-1]);print(d['value'],d['ms_per_step'],{k:v['ms'] for k,v in d['stages'].items()});print({k:d.get(k) for k in ('in_place_entries','two_tiles_in_flight','tile_as_two_stripes','grid_16384')})"
