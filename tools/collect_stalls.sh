#!/bin/bash
# Where do the waves of the metric pipeline's kernels spend their cycles?  SQ counters in separate rocprofv3 passes
# (kernel trace only), folded into gpurun_out/<tag>_stalls.json (means per launch).
#   tools/collect_stalls.sh <tag> [float mode]
set -e
TAG=${1:-stalls}
MODE=${2:-strict}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P="$OUT/prof_stalls"
rm -rf "$P"
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-extras --float-mode $MODE"
i=0
for GROUP in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH SQ_IFETCH" \
             "GRBM_GUI_ACTIVE SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $GROUP --output-format csv -d "$P/g$i" -o run -- python3 "$ROOT/bench.py" $ARGS > "$OUT/${TAG}_g$i.log" 2>&1 || echo "group $i failed"
  echo "group $i done"
done
python3 - "$P" "$OUT/${TAG}_stalls.json" <<'PY'
import collections, csv, glob, json, re, sys
d = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(sys.argv[1] + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        m = re.search(r"(\w+(?:<[^>(]*>)?)\(", name)
        d[m.group(1) if m else name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in d.items()}
json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)
for k, e in out.items():
    wc = e.get("SQ_WAVE_CYCLES", 0) or 1
    print(k)
    for c in sorted(e):
        print("   %-28s %14.4g  %6.3f of wave cycles" % (c, e[c], e[c] / wc))
PY
rm -rf "$P"
