#!/bin/bash
# compile-time variants of nz_flow.hip timed on the GPU box (its scratch copy of the tree):
#   tools/try_flow_flags.sh "-DNZ_FT_NT=768 -DNZ_FT_OCC=6" ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/noize_job_amd/csrc"
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -fno-slp-vectorize"
mkdir -p build
for extra in "" "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $BASE $extra -c nz_flow.hip -o build/nz_flow.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -ldl -o ../libnoize_hip.so
  echo "== flags: [$extra]"
  python3 "$ROOT/tools/bench_stage.py" flow --reps 40 2>/dev/null | tail -1
done
