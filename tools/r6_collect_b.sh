#!/bin/bash
# Round-6 evidence, part B (one GPU call): stalls, conv phase profile, next rows with counters, tiles, config 4, the rank rehearsal.
#   tools/r6_collect_b.sh <tag> <commit>
TAG=${1:-r06_v2}
COMMIT=${2:-unknown}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
mkdir -p gpurun_out
bash tools/collect_stalls.sh ${TAG} strict > gpurun_out/${TAG}_stalls.log 2>&1; echo "stalls strict done"
bash tools/collect_stalls.sh ${TAG}_fast fast > gpurun_out/${TAG}_fast_stalls.log 2>&1; echo "stalls fast done"
bash tools/collect_next.sh ${TAG} $COMMIT strict > /dev/null 2>&1; echo "next rows strict done"
python3 tools/bench_next.py --float-mode fast > gpurun_out/${TAG}_fast_next_rows.txt 2>&1; echo "next rows fast done"
python3 tools/bench_tiles.py --res 256 512 768 1024 2048 --streams 1 2 4 > gpurun_out/${TAG}_tiles.txt 2>&1
for r in 256 512 768 1024 2048; do python3 tools/bench_stage.py all --pair --res $r --reps 200 2>/dev/null | tail -5 >> gpurun_out/${TAG}_tiles.txt; done; echo "tiles done"
python3 bench.py --as-rank 3 8 --halo recompute > gpurun_out/${TAG}_rank3of8_recompute.json 2>/dev/null
python3 bench.py --as-rank 3 8 --halo exchange > gpurun_out/${TAG}_rank3of8_exchange_o0.json 2>/dev/null
echo "rank rehearsal done"
AT=100 bash tools/collect_config4.sh ${TAG} > gpurun_out/${TAG}_config4_collect.log 2>&1; echo "config 4 counters done"
python3 tools/bench_config4.py --at 1,100 --json gpurun_out/${TAG}_config4.json > gpurun_out/${TAG}_config4.log 2>&1; echo "config 4 bench done"
# the stamped build last: it replaces the library in this scratch copy
for m in 0 1; do EXTRA="" bash tools/probe_conv_phases.sh 4096 17 $m > gpurun_out/${TAG}_conv_phases_mode$m.txt 2>&1; done
echo "conv phases done"
tail -5 gpurun_out/${TAG}_config4_collect.log
