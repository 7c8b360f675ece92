#!/bin/bash
# SQ counters of the metric pipeline's kernels (one rocprofv3 --pmc pass, kernel trace only):
#   tools/collect_sq.sh <tag>  ->  gpurun_out/<tag>_sq.json
set -e
TAG=${1:-r01_vX}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf "$OUT/prof_sq"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv \
    -d "$OUT/prof_sq" -o run -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/${TAG}_sq.log" 2>&1
cd "$ROOT"
python3 - "$(find "$OUT/prof_sq" -name '*counter_collection.csv' | head -1)" "$OUT/${TAG}_sq.json" <<'PY'
import collections, csv, json, re, sys
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(\w+(?:<[^>(]*>)?)\(", r["Kernel_Name"].replace("(anonymous namespace)::", ""))
    d[m.group(1) if m else r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, c in d.items():
    e = {n: sum(v) / len(v) for n, v in c.items()}
    e["launches_sampled"] = len(next(iter(c.values())))
    # a wave64 VALU instruction occupies its SIMD-32 for 2 cycles.  SQ_INSTS_VALU counts the whole chip;
    # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (8 x the kernel's cycles), so the SIMD-cycles available
    # are GRBM_GUI_ACTIVE x 128 SIMDs per XCD
    if e.get("GRBM_GUI_ACTIVE") and e.get("SQ_INSTS_VALU"):
        e["valu_issue_utilisation"] = round(e["SQ_INSTS_VALU"] * 2.0 / (e["GRBM_GUI_ACTIVE"] * 128), 4)
    out[k] = e
json.dump({"note": "means per launch; valu_issue_utilisation = SQ_INSTS_VALU * 2 cycles / (GRBM_GUI_ACTIVE [summed over 8 XCDs] * 128 SIMDs per XCD)",
           "kernels": out}, open(sys.argv[2], "w"), indent=1)
for k, e in out.items():
    print("%-36s valu insts %.3e  gui cycles %.3e  valu issue util %s" % (k, e.get("SQ_INSTS_VALU", 0), e.get("GRBM_GUI_ACTIVE", 0),
                                                                       e.get("valu_issue_utilisation")))
PY
rm -rf "$OUT/prof_sq"
