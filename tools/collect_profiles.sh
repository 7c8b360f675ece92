#!/bin/bash
# Collects the evidence set profiles/README.md describes, on the GPU box:
#   tools/collect_profiles.sh <tag>      ->  gpurun_out/<tag>_{bench.json,kernel_stats.csv,pmc.json}
# Counter passes are separate runs with --kernel-trace only (no other trace domains).
set -e
TAG=${1:-r01_vX}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd "$ROOT" && python3 bench.py > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 60 --warmup 20 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$TAG/stats" -o run -- python3 "$ROOT/bench.py" $ARGS > "$OUT/${TAG}_stats.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/prof_$TAG/fetch" -o run -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/${TAG}_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/prof_$TAG/write" -o run -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/${TAG}_write.log" 2>&1
cd "$ROOT"
cp "$(find "$OUT/prof_$TAG/stats" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_kernel_stats.csv"
python3 tools/pmc_summary.py "$(find "$OUT/prof_$TAG/fetch" -name '*counter_collection.csv' | head -1)" \
    "$(find "$OUT/prof_$TAG/write" -name '*counter_collection.csv' | head -1)" "$OUT/${TAG}_pmc.json" \
    "bench.py --steps 5 --warmup 2, 4096^2 metric tile, $TAG kernels"
rm -rf "$OUT/prof_$TAG"
