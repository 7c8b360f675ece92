#!/bin/bash
# Kernel timeline of the sharded step (rank R of P rehearsed on one GPU): tools/trace_sharded.sh <tag> <halo> [extra bench args]
#   -> gpurun_out/<tag>_timeline.txt, gpurun_out/<tag>_kernel_stats.csv
set -e
TAG=${1:-r04_sharded}
HALO=${2:-exchange}
shift 2 || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P="$OUT/prof_$TAG"
rm -rf "$P"
rocprofv3 --kernel-trace --stats --output-format csv -d "$P" -o run -- python3 "$ROOT/bench.py" --as-rank 3 8 --halo "$HALO" --steps 20 --warmup 10 --no-extras "$@" > "$OUT/${TAG}_trace.log" 2>&1
cd "$ROOT"
cp "$(find "$P" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_kernel_stats.csv"
python3 tools/trace_timeline.py "$(find "$P" -name '*kernel_trace.csv' | head -1)" > "$OUT/${TAG}_timeline.txt"
rm -rf "$P"
