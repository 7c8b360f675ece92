#!/bin/bash
# conv_reg_kernel workgroup size for 5..9 taps: 512 threads (128-row tiles) vs 1024 threads (256-row tiles)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/noize_job_amd/csrc"
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -fno-slp-vectorize"
mkdir -p build
for extra in "-DNZ_CONV_NT_WIDE=512" "-DNZ_CONV_NT_WIDE=1024"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $BASE $extra -c nz_filter.hip -o build/nz_filter.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libnoize_hip.so
  for cap in 5 6 8; do
    echo "== flags: [$extra] NZ_CONV_TCAP=$cap"
    NZ_CONV_TCAP=$cap python3 "$ROOT/tools/bench_stage.py" gauss --reps 40 2>/dev/null
  done
done
