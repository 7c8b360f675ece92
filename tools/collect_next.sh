#!/bin/bash
# Counters of the stages outside the metric pipeline, on the GPU box:
#   tools/collect_next.sh <tag> [commit] [float mode]  ->  gpurun_out/<tag>_next_{rows.txt,counters.json,table.txt}
# One rocprofv3 pass per counter group over tools/bench_next.py (kernel trace only), folded by tools/fold_counters.py.
set -e
TAG=${1:-rXX}
COMMIT=${2:-unknown}
MODE=${3:-strict}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
python3 "$ROOT/tools/bench_next.py" --float-mode $MODE > "$OUT/${TAG}_next_rows.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
P="$OUT/prof_next_$TAG"
rm -rf "$P"
ARGS="--reps 2 --float-mode $MODE"
rocprofv3 --kernel-trace --stats --output-format csv -d "$P/stats" -o run -- python3 "$ROOT/tools/bench_next.py" --reps 6 --float-mode $MODE > "$OUT/${TAG}_next_stats.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$P/fetch" -o run -- python3 "$ROOT/tools/bench_next.py" $ARGS > "$OUT/${TAG}_next_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$P/write" -o run -- python3 "$ROOT/tools/bench_next.py" $ARGS > "$OUT/${TAG}_next_write.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d "$P/sq" -o run -- python3 "$ROOT/tools/bench_next.py" $ARGS > "$OUT/${TAG}_next_sq.log" 2>&1
cd "$ROOT"
python3 tools/fold_counters.py --fetch "$(find "$P/fetch" -name '*counter_collection.csv' | head -1)" \
    --write "$(find "$P/write" -name '*counter_collection.csv' | head -1)" \
    --sq "$(find "$P/sq" -name '*counter_collection.csv' | head -1)" --stats "$(find "$P/stats" -name '*kernel_stats.csv' | head -1)" \
    --out "$OUT/${TAG}_next_counters.json" --res 4096 --flush next-rows --float-mode $MODE --commit "$COMMIT" \
    --note "$TAG: tools/bench_next.py (every stage outside the metric pipeline); means over every launch of a kernel, whatever its arguments" > /dev/null
python3 tools/next_counters_table.py "$OUT/${TAG}_next_counters.json" > "$OUT/${TAG}_next_table.txt"
rm -rf "$P"
cat "$OUT/${TAG}_next_table.txt"
