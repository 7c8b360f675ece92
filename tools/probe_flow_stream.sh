#!/bin/bash
# builds the stamped variant of nz_flow_stream.hip into the in-tree library (on the GPU box's scratch copy) and prints the timeline
set -e
# the instrumented object replaces the stock one in csrc/build: whatever happens, it is removed again and the stock
# library rebuilt, so that a later `make` (what the tests and bench.py run) never finds an up-to-date probe object
restore() { rm -f "$ROOT/noize_job_amd/csrc/build/nz_flow_stream.o"; make -C "$ROOT/noize_job_amd/csrc" >/dev/null 2>&1 || true; }
ROOT=$(cd "$(dirname "$0")/.." && pwd)
trap restore EXIT
cd "$ROOT/noize_job_amd/csrc"
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -fno-slp-vectorize"
mkdir -p build
/opt/rocm/bin/hipcc --offload-arch=gfx950 $BASE -DNZ_FLOW_PROBE $EXTRA -mllvm -amdgpu-sched-strategy=max-ilp -c nz_flow_stream.hip -o build/nz_flow_stream.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -ldl -o ../libnoize_hip.so
python3 "$ROOT/tools/probe_flow_stream.py" "$@"
