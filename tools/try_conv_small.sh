#!/bin/bash
# compile-time variants of nz_filter.hip timed on small tiles (one tile at a time, one stream):
#   tools/try_conv_small.sh "-DNZ_CONV_NT_WIDE=256" ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/noize_job_amd/csrc"
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -fno-slp-vectorize"
mkdir -p build
restore() { rm -f build/nz_filter.o; make >/dev/null 2>&1 || true; }
trap restore EXIT
for extra in "" "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $BASE $extra -c nz_filter.hip -o build/nz_filter.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -ldl -o ../libnoize_hip.so
  echo "== flags: [$extra]"
  for r in 256 512 1024 2048; do python3 "$ROOT/tools/bench_stage.py" gauss --res $r --reps 100 2>/dev/null | tail -1; done
done
