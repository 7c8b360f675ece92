#!/bin/bash
# Kernel timeline of one live-erosion cycle as the host driver enqueues it: tools/trace_config4.sh <tag> [driver args]
#   -> gpurun_out/<tag>_timeline.txt
set -e
TAG=${1:-r04_config4}
shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P="$OUT/prof_$TAG"
rm -rf "$P"
rocprofv3 --kernel-trace --output-format csv -d "$P" -o run -- python3 "$ROOT/tools/driver_config4.py" "$@" > "$OUT/${TAG}_trace.log" 2>&1
cd "$ROOT"
python3 tools/trace_timeline.py "$(find "$P" -name '*kernel_trace.csv' | head -1)" --first fill_queue --step -4 --width 60 > "$OUT/${TAG}_timeline.txt"
rm -rf "$P"
