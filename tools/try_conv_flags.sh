#!/bin/bash
# XCD-aware tile order for the register conv / erosion kernels: time + HBM fetch per launch
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/noize_job_amd/csrc"
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -fno-slp-vectorize"
mkdir -p build
for extra in "-DNZ_XCD_REMAP=0" "-DNZ_XCD_REMAP=1"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $BASE $extra -c nz_filter.hip -o build/nz_filter.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libnoize_hip.so
  echo "== flags: [$extra]"
  python3 "$ROOT/tools/bench_stage.py" gauss --reps 40 2>/dev/null
  python3 "$ROOT/tools/bench_stage.py" erosion --reps 40 2>/dev/null
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pf && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o run -- python3 "$ROOT/tools/bench_stage.py" gauss --reps 2 > /dev/null 2>&1; python3 - <<'PY'
import csv, glob, collections
d = collections.defaultdict(list)
for f in glob.glob("/tmp/pf/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            d[r["Kernel_Name"][:40]].append(float(r["Counter_Value"]))
for k, v in d.items():
    print("   fetch %-40s %7.1f MB per launch (x2 corrected)" % (k, 2 * 1024 * sum(v) / len(v) / 1e6))
PY
  )
done
