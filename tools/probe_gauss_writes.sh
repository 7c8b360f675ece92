#!/bin/bash
# Where does the fifth plane of writes of the chained Gauss launch come from?  WRITE_SIZE / FETCH_SIZE per launch
# (rocprofv3 --pmc, kernel trace only) of Gauss5 x16/x17 as one chained grid and as separate launches, on a grid with
# partial edge tiles (4096) and on one whose tiles divide it exactly (4032 = 36 x 112 at T = 4).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -f "$OUT/gauss_writes.txt"
run() {  # label, env..., -- args
  label=$1; shift
  for C in WRITE_SIZE FETCH_SIZE; do
    P=/tmp/gw_$$; rm -rf $P
    env "$@" rocprofv3 --kernel-trace --pmc $C --output-format csv -d $P -o run -- python3 "$ROOT/tools/bench_stage.py" gauss --reps 3 $ARGS > /dev/null 2>&1 || true
    python3 - "$P" "$label" "$C" >> "$OUT/gauss_writes.txt" <<'PY'
import collections, csv, glob, sys
d = collections.defaultdict(list)
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "conv" in r["Kernel_Name"]:
            d[r["Kernel_Name"].split("(")[0].replace("(anonymous namespace)::", "")[:60]].append(float(r["Counter_Value"]))
for k, v in d.items():
    print("%-34s %-10s %-62s launches %3d  mean %10.1f KiB" % (sys.argv[2], sys.argv[3], k, len(v), sum(v) / len(v)))
PY
    rm -rf $P
  done
}
ARGS="--res 4096 --gauss 17" run "4096 x17 chain" X=1
ARGS="--res 4096 --gauss 17" run "4096 x17 separate" NZ_CONV_CHAIN=0
ARGS="--res 4032 --gauss 17" run "4032 x17 chain" X=1
ARGS="--res 4032 --gauss 17" run "4032 x17 separate" NZ_CONV_CHAIN=0
cat "$OUT/gauss_writes.txt"
