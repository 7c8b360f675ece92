#!/bin/bash
# Where does a slow step of the metric pipeline lose its time?  Kernel trace of `bench.py` (timed steps only matter), then
# per kernel: the slowest launches with the gap before them.   tools/trace_outliers.sh <tag> [env assignments...]
TAG=${1:-outl}; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
P="$OUT/prof_$TAG"; rm -rf "$P"
rocprofv3 --kernel-trace --output-format csv -d "$P" -o run -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-extras --grid 0 --steps 400 > "$OUT/${TAG}_trace.log" 2>&1
python3 - "$(find "$P" -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys, collections, re
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"::(\w+(?:<[^>]*>)?)", r["Kernel_Name"])
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:40], r.get("Queue_Id", "?")))
rows.sort()
by = collections.defaultdict(list)
prev_end = rows[0][0]
for i, (s, e, n, q) in enumerate(rows):
    by[n].append(((e - s) / 1e3, (s - prev_end) / 1e3, i))
    prev_end = max(prev_end, e)
for n, v in by.items():
    if len(v) < 50:
        continue
    d = sorted(x[0] for x in v)
    print("%-60s n %5d  median %8.1f us  p99 %8.1f  max %8.1f" % (n, len(v), d[len(d) // 2], d[int(len(d) * 0.99)], d[-1]))
    for dur, gap, i in sorted(v, reverse=True)[:3]:
        print("      launch #%d: %8.1f us, gap before it %8.1f us, queue %s" % (i, dur, gap, rows[i][3]))
gaps = sorted(((rows[i][0] - max(r[1] for r in rows[max(0, i - 8):i])) / 1e3, i) for i in range(1, len(rows)))[-5:]
print("largest gaps (us) before launch # (the first dozen launches are the start-up):", [(round(g, 1), i, rows[i][2][-30:]) for g, i in gaps])
PY
rm -rf "$P"
