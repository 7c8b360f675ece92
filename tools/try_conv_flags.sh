#!/bin/bash
# compile-time variants of nz_filter.hip timed on the GPU box (its scratch copy of the tree):
#   tools/try_conv_flags.sh "-DNZ_CONV5_WAVES=4" "-DNZ_XCD_REMAP=0" ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/noize_job_amd/csrc"
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -fno-slp-vectorize"
mkdir -p build
for extra in "" "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $BASE $extra -c nz_filter.hip -o build/nz_filter.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -ldl -o ../libnoize_hip.so
  echo "== flags: [$extra]"
  python3 "$ROOT/tools/bench_stage.py" gauss --reps 300 2>/dev/null | tail -1
done
