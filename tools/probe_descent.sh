#!/bin/bash
# builds the stamped variant of nz_live.hip into the in-tree library (on the GPU box's scratch copy) and prints where a
# descent step's shader clocks go:  tools/probe_descent.sh [extra -D flags]
set -e
# the instrumented object replaces the stock one in csrc/build: whatever happens, it is removed again and the stock
# library rebuilt, so that a later `make` (what the tests and bench.py run) never finds an up-to-date probe object
restore() { rm -f "$ROOT/noize_job_amd/csrc/build/nz_live.o"; make -C "$ROOT/noize_job_amd/csrc" >/dev/null 2>&1 || true; }
ROOT=$(cd "$(dirname "$0")/.." && pwd)
trap restore EXIT
cd "$ROOT/noize_job_amd/csrc"
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -fno-slp-vectorize"
mkdir -p build
/opt/rocm/bin/hipcc --offload-arch=gfx950 $BASE -DNZ_DESCENT_PROBE "$@" -mllvm -amdgpu-atomic-optimizer-strategy=None -c nz_live.hip -o build/nz_live.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -ldl -o ../libnoize_hip.so
python3 "$ROOT/tools/probe_descent.py"
