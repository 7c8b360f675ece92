#!/bin/bash
# per-kernel times of BASELINE config 4's cycle around cycle 100 (kernel trace only), on the GPU box: tools/stats_config4.sh
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/c4s
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c4s -o run -- python3 $ROOT/tools/bench_config4.py --at 100 --skip-fresh > /tmp/c4s.log 2>&1
python3 - <<'PY'
import csv, glob, re
f = glob.glob("/tmp/c4s/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].replace("(anonymous namespace)::", "")
    m = re.search(r"(\w+(?:<[^>(]*>)?)\(", n)
    print("%-40s calls %6s  avg %9.1f us  total %9.2f ms" % ((m.group(1) if m else n)[:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
