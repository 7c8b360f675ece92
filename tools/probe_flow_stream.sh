#!/bin/bash
# builds the stamped variant of nz_flow.hip into the in-tree library (on the GPU box's scratch copy) and prints the timeline
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/noize_job_amd/csrc"
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -fno-slp-vectorize"
mkdir -p build
/opt/rocm/bin/hipcc --offload-arch=gfx950 $BASE -DNZ_FLOW_PROBE $EXTRA -c nz_flow.hip -o build/nz_flow.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libnoize_hip.so
python3 "$ROOT/tools/probe_flow_stream.py" "$@"
