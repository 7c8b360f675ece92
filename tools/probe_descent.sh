#!/bin/bash
# builds the stamped variant of nz_live.hip into the in-tree library (on the GPU box's scratch copy) and prints where a
# descent step's shader clocks go:  tools/probe_descent.sh [extra -D flags]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/noize_job_amd/csrc"
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -fno-slp-vectorize"
mkdir -p build
/opt/rocm/bin/hipcc --offload-arch=gfx950 $BASE -DNZ_DESCENT_PROBE "$@" -mllvm -amdgpu-atomic-optimizer-strategy=None -c nz_live.hip -o build/nz_live.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libnoize_hip.so
python3 "$ROOT/tools/probe_descent.py"
