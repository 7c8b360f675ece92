#!/bin/bash
# compile-time variants of nz_fractal.hip timed on the GPU box (its scratch copy of the tree):
#   tools/try_fractal_flags.sh "-DNZ_OCT_UNROLL=2" ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/noize_job_amd/csrc"
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -fno-slp-vectorize"
mkdir -p build
for extra in "" "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $BASE $extra -c nz_fractal.hip -o build/nz_fractal.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -ldl -o ../libnoize_hip.so
  echo "== flags: [$extra]"
  python3 "$ROOT/tools/bench_stage.py" noise --reps 400 2>/dev/null | tail -1
done
