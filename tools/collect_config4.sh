#!/bin/bash
# Per-kernel times and counters of BASELINE config 4's cycle in its long-run regime (around cycle 100 of one run), on the
# GPU box:   tools/collect_config4.sh <tag>   ->  gpurun_out/<tag>_config4_kernel_stats.csv, <tag>_config4_kernel_counters.json
# Every counter group is its own rocprofv3 run with --kernel-trace only.
set -e
TAG=${1:-r03_vX}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P="$OUT/prof_c4_$TAG"
rm -rf "$P"
CMD="$ROOT/tools/bench_config4.py --at ${AT:-100} --skip-fresh"
rocprofv3 --kernel-trace --stats --output-format csv -d "$P/stats" -o run -- python3 $CMD > "$OUT/${TAG}_c4_stats.log" 2>&1
echo "stats pass done"
i=0
for GROUP in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $GROUP --output-format csv -d "$P/g$i" -o run -- python3 $CMD > "$OUT/${TAG}_c4_g$i.log" 2>&1 || echo "group $i failed"
  echo "group $i done"
done
cp "$(find "$P/stats" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_config4_kernel_stats.csv"
python3 - "$P" "$OUT/${TAG}_config4_kernel_counters.json" "$OUT/${TAG}_config4_kernel_stats.csv" <<'PY'
import collections, csv, glob, json, re, sys
def short(name):
    name = name.replace("(anonymous namespace)::", "")
    m = re.search(r"(\w+(?:<[^>(]*>)?)\(", name)
    return m.group(1) if m else name
d = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(sys.argv[1] + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        d[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
stats = {}
for r in csv.DictReader(open(sys.argv[3])):
    stats[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "total_ms": float(r["TotalDurationNs"]) / 1e6,
                               "share_pct": float(r["Percentage"])}
out = {}
for k, cs in d.items():
    e = {c: sum(v) / len(v) for c, v in cs.items()}
    if "FETCH_SIZE" in e or "WRITE_SIZE" in e:  # KiB; FETCH_SIZE reports half the bytes of wide coalesced reads on gfx950
        e["hbm_bytes_per_launch"] = (2.0 * e.get("FETCH_SIZE", 0.0) + e.get("WRITE_SIZE", 0.0)) * 1024.0
    e.update(stats.get(k, {}))
    if "avg_us" in e and "SQ_INSTS_VALU" in e and e["avg_us"] > 0:
        e["valu_issue_frac"] = e["SQ_INSTS_VALU"] * 2.0 / (1024 * 2.4e9) / (e["avg_us"] * 1e-6)
        e["hbm_traffic_frac"] = e.get("hbm_bytes_per_launch", 0.0) / (e["avg_us"] * 1e-6) / 8e12
    out[k] = e
json.dump({"config": "bench_config4.py --at 100 --skip-fresh (8192^2, 10 000 particles per cycle, WATER_STEPS 10): means per launch "
                     "over one long run", "kernels": out}, open(sys.argv[2], "w"), indent=1, sort_keys=True)
for k, e in sorted(out.items(), key=lambda kv: -kv[1].get("total_ms", 0)):
    print("%-44s calls %6d avg %9.1f us total %8.2f ms  valu %.3g  issue %.3f  hbm %.3f" % (
        k[:44], e.get("calls", 0), e.get("avg_us", 0), e.get("total_ms", 0), e.get("SQ_INSTS_VALU", 0), e.get("valu_issue_frac", 0), e.get("hbm_traffic_frac", 0)))
PY
rm -rf "$P"
