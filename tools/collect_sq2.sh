#!/bin/bash
# Wave-level stall counters of one stage (default gauss): where do the waves of the conv kernel wait?
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
STAGE=${1:-gauss}
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU"; do
  rm -rf /tmp/pf2
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pf2 -o run -- python3 "$ROOT/tools/bench_stage.py" $STAGE --reps 2 > /dev/null 2>&1 || echo "pass failed: $set"
  python3 - <<'PY'
import collections, csv, glob
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pf2/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"][28:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in d.items():
    if "conv_reg" in k or "flow_fused" in k or "simplex" in k:
        print(k, {n: "%.3e" % (sum(v) / len(v)) for n, v in c.items()})
PY
done
