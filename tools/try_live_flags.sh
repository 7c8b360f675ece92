#!/bin/bash
# compile-time variants of nz_live.hip timed on the GPU box (config 4's per-job times at cycle 100): tools/try_live_flags.sh "-DX" ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/noize_job_amd/csrc"
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -fno-slp-vectorize -mllvm -amdgpu-atomic-optimizer-strategy=None -mllvm -amdgpu-sched-strategy=max-ilp"
for extra in "" "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $BASE $extra -c nz_live.hip -o build/nz_live.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -ldl -o ../libnoize_hip.so
  echo "== flags: [$extra]"
  python3 "$ROOT/tools/bench_config4.py" --at 100 --skip-fresh 2>/dev/null | grep -E "descent|erode|cycle_ms|events_per"
done
