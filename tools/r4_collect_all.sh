#!/bin/bash
# Round 4's evidence set, on the GPU box:  tools/r4_collect_all.sh <tag> <commit> [part ...]   (parts: metric sharded config4 next)
# -> gpurun_out/<tag>_*; the files worth judging are copied into profiles/ by hand.
TAG=$1
C=${2:-unknown}
shift 2
PARTS=${*:-metric sharded config4 next}
mkdir -p gpurun_out
one() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['ms_per_step'], d['value'], {k:v['ms'] for k,v in d['stages'].items()}, d.get('verified'))"; }
for part in $PARTS; do
case $part in
metric)
  bash tools/collect_profiles.sh $TAG $C swap > gpurun_out/collect_$TAG.log 2>&1; tail -n 5 gpurun_out/collect_$TAG.log
  bash tools/collect_stalls.sh $TAG > gpurun_out/${TAG}_stalls.txt 2>&1; tail -n 2 gpurun_out/${TAG}_stalls.txt
  # the bench line: three runs on this box, all kept; the middle one is the one to commit
  for i in 1 2 3; do python bench.py > gpurun_out/${TAG}_bench_run$i.json 2> gpurun_out/${TAG}_bench_run$i.err; one < gpurun_out/${TAG}_bench_run$i.json; done
  python bench.py --flush copy --no-cpu-baseline --no-extras > gpurun_out/${TAG}_bench_flush_copy.json 2>/dev/null
  ;;
sharded)
  # rank 3 of 8 of the 16384^2 grid rehearsed on this GPU: ghost rows recomputed, exchanged (three schedules), exchanged once
  python bench.py --as-rank 3 8 --halo recompute --no-extras --marked-steps 20 > gpurun_out/${TAG}_rank3of8_recompute.json 2>/dev/null; one < gpurun_out/${TAG}_rank3of8_recompute.json
  for o in 0 1 2; do python bench.py --as-rank 3 8 --halo exchange --overlap $o --no-extras --marked-steps 20 > gpurun_out/${TAG}_rank3of8_exchange_o$o.json 2>/dev/null; one < gpurun_out/${TAG}_rank3of8_exchange_o$o.json; done
  python bench.py --as-rank 3 8 --halo exchange_once --no-extras --marked-steps 20 > gpurun_out/${TAG}_rank3of8_exchange_once.json 2>/dev/null; one < gpurun_out/${TAG}_rank3of8_exchange_once.json
  for o in 0 1 2; do bash tools/trace_sharded.sh ${TAG}_o$o exchange --overlap $o --marked-steps 0; done
  bash tools/trace_sharded.sh ${TAG}_recompute recompute --marked-steps 0
  ;;
config4)
  python tools/bench_config4.py --at 1,100,1000 --json gpurun_out/${TAG}_config4.json > /dev/null 2>&1
  bash tools/stats_config4.sh > gpurun_out/${TAG}_config4_kernel_stats.txt 2>&1; tail -n 4 gpurun_out/${TAG}_config4_kernel_stats.txt
  ;;
next)
  python tools/bench_next.py > gpurun_out/${TAG}_next_rows.txt 2>&1; tail -n 3 gpurun_out/${TAG}_next_rows.txt
  ;;
esac
done
ls gpurun_out | grep $TAG
