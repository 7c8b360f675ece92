#!/bin/bash
# Round-6 evidence, part A (one GPU call): kernel stats + counters of the metric step in the three float modes, the bench lines.
#   tools/r6_collect_a.sh <tag> <commit>
TAG=${1:-r06_v2}
COMMIT=${2:-unknown}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
mkdir -p gpurun_out
for MODE in fast relaxed; do
  NZ_SKIP_BENCH=1 bash tools/collect_profiles.sh ${TAG}_${MODE} $COMMIT swap $MODE > gpurun_out/${TAG}_${MODE}_collect.log 2>&1
  echo "$MODE counters done"
done
bash tools/collect_profiles.sh $TAG $COMMIT swap strict > gpurun_out/${TAG}_collect.log 2>&1
echo "strict counters + bench done"
python3 bench.py --steps 20 --warmup 5 --no-extras > gpurun_out/${TAG}_bench_driver_command.json 2> gpurun_out/${TAG}_bench_driver_command.err
python3 bench.py --flush copy --no-cpu-baseline --no-extras > gpurun_out/${TAG}_bench_flush_copy.json 2>/dev/null
python3 tools/bench_modes.py --rounds 3 > gpurun_out/${TAG}_modes.txt 2>&1
tail -4 gpurun_out/${TAG}_modes.txt
python3 -c "
import json
d=json.load(open('gpurun_out/${TAG}_bench.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'verified', d['verified'], 'roofline', d['roofline'].get('frac'), d['roofline'].get('bound'))
print({k:(v['ms_per_step'], v['roofline'].get('frac')) for k,v in d['float_modes'].items() if k!='note'})
print('grid', d['grid_16384']['recompute'])
"
