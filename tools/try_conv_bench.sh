#!/bin/bash
# compile-time variants of nz_filter.hip under bench.py's timed steps (the chained launch inside the whole step), on the GPU
# box's scratch copy of the tree:  tools/try_conv_bench.sh "-DNZ_DPP_OLD=1" ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/noize_job_amd/csrc"
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -fno-slp-vectorize"
mkdir -p build
restore() { rm -f build/nz_filter.o; make >/dev/null 2>&1 || true; }
trap restore EXIT
for extra in "" "$@" ""; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $BASE $extra -c nz_filter.hip -o build/nz_filter.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -ldl -o ../libnoize_hip.so
  echo "== flags: [$extra]"
  for i in 1 2 3; do
    python3 "$ROOT/bench.py" --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['ms_per_step'], {k:v['ms'] for k,v in d['stages'].items()})"
  done
done
