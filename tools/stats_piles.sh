#!/bin/bash
# pile_kernel launch statistics of the config-4 driver loop: tools/stats_piles.sh <tag> [driver args]
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P="$OUT/prof_$TAG"; rm -rf "$P"
rocprofv3 --kernel-trace --stats --output-format csv -d "$P" -o run -- python3 "$ROOT/tools/driver_config4.py" "$@" > "$OUT/${TAG}_trace.log" 2>&1
python3 - "$(find "$P" -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("pile_", "descent_kernel", "disperse_list", "flow_from_track")):
        print("%-40s calls %5s  avg %9.1f us  min %9.1f  max %9.1f" % (n.split("(")[-2].split("::")[-1][:40] if "(" in n else n[:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
rm -rf "$P"
