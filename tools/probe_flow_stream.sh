#!/bin/bash
# builds the stamped variant of nz_flow_stream.hip into the in-tree library (on the GPU box's scratch copy) and prints the timeline
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/noize_job_amd/csrc"
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -fno-slp-vectorize"
mkdir -p build
/opt/rocm/bin/hipcc --offload-arch=gfx950 $BASE -DNZ_FLOW_PROBE $EXTRA -mllvm -amdgpu-sched-strategy=max-ilp -c nz_flow_stream.hip -o build/nz_flow_stream.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libnoize_hip.so
python3 "$ROOT/tools/probe_flow_stream.py" "$@"
