set -x
TAG=$1
mkdir -p gpurun_out
C=${2:-unknown}
bash tools/collect_profiles.sh $TAG $C swap > gpurun_out/collect_$TAG.log 2>&1; tail -n 6 gpurun_out/collect_$TAG.log
bash tools/collect_stalls.sh $TAG > gpurun_out/${TAG}_stalls.txt 2>&1; tail -n 3 gpurun_out/${TAG}_stalls.txt
python bench.py --flush copy --no-cpu-baseline > gpurun_out/${TAG}_bench_flush_copy.json 2>/dev/null
python bench.py --schedule pipeline --no-cpu-baseline --no-extras > gpurun_out/${TAG}_bench_one_call.json 2>/dev/null
python tools/bench_next.py > gpurun_out/${TAG}_next_rows.txt 2>&1; tail -n 3 gpurun_out/${TAG}_next_rows.txt
python tools/bench_tiles.py > gpurun_out/${TAG}_tiles.txt 2>&1; tail -n 3 gpurun_out/${TAG}_tiles.txt
bash tools/collect_config4.sh $TAG > gpurun_out/${TAG}_c4_collect.txt 2>&1; tail -n 3 gpurun_out/${TAG}_c4_collect.txt
python tools/bench_config4.py --at 1,100,1000 --json gpurun_out/${TAG}_config4.json > /dev/null 2>&1
ls gpurun_out | grep $TAG
