#!/bin/bash
# Collects the evidence set profiles/README.md describes, on the GPU box:
#   tools/collect_profiles.sh <tag> [commit] [flush] [float mode]   ->  gpurun_out/<tag>_{bench.json,kernel_stats.csv,counters.json}
# (float mode strict / fast / relaxed: one evidence set per mode; NZ_SKIP_BENCH=1 leaves the final bench.py run out)
# Every counter group is its own rocprofv3 run with --kernel-trace only (no other trace domains).
set -e
TAG=${1:-r02_vX}
COMMIT=${2:-unknown}
FLUSH=${3:-swap}
MODE=${4:-strict}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-extras --flush $FLUSH --float-mode $MODE"
P="$OUT/prof_$TAG"
rm -rf "$P"
rocprofv3 --kernel-trace --stats --output-format csv -d "$P/stats" -o run -- python3 "$ROOT/bench.py" --steps 60 --warmup 20 $ARGS > "$OUT/${TAG}_stats.log" 2>&1
echo "stats pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$P/fetch" -o run -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 $ARGS > "$OUT/${TAG}_fetch.log" 2>&1
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$P/write" -o run -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 $ARGS > "$OUT/${TAG}_write.log" 2>&1
echo "write pass done"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d "$P/sq" -o run -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 $ARGS > "$OUT/${TAG}_sq.log" 2>&1
echo "sq pass done"
cd "$ROOT"
cp "$(find "$P/stats" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_kernel_stats.csv"
python3 tools/fold_counters.py --fetch "$(find "$P/fetch" -name '*counter_collection.csv' | head -1)" \
    --write "$(find "$P/write" -name '*counter_collection.csv' | head -1)" \
    --sq "$(find "$P/sq" -name '*counter_collection.csv' | head -1)" --stats "$OUT/${TAG}_kernel_stats.csv" \
    --out "$OUT/${TAG}_counters.json" --res 4096 --flush $FLUSH --float-mode $MODE --commit "$COMMIT" --note "$TAG kernels"
rm -rf "$P"
# the bench line of these kernels, now that their counter summary exists (bench.py reads profiles/*_counters.json)
cp "$OUT/${TAG}_counters.json" "$ROOT/profiles/${TAG}_counters.json"
if [ -z "$NZ_SKIP_BENCH" ]; then
  python3 bench.py --flush $FLUSH --float-mode $MODE > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
fi
