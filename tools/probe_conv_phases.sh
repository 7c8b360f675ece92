#!/bin/bash
# builds the stamped variant of nz_filter.hip into the in-tree library (on the GPU box's scratch copy) and prints the timeline
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/noize_job_amd/csrc"
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -fno-slp-vectorize"
mkdir -p build
/opt/rocm/bin/hipcc --offload-arch=gfx950 $BASE -DNZ_CONV_PROBE $EXTRA -c nz_filter.hip -o build/nz_filter.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -ldl -o ../libnoize_hip.so
python3 "$ROOT/tools/probe_conv_phases.py" "$@"
