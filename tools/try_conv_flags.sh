#!/bin/bash
# rebuilds nz_filter.hip with each flag set on the GPU box and times the metric's Gauss5 x17 stage per fusion cap
set -e
cd "$(dirname "$0")/../noize_job_amd/csrc"
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -fno-slp-vectorize"
mkdir -p build
for extra in "-DNZ_CONV_NT=256" "-DNZ_CONV_NT=512"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $BASE $extra -c nz_filter.hip -o build/nz_filter.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libnoize_hip.so
  for cap in 3 4 5 6; do
    echo "== flags: [$extra] NZ_CONV_TCAP=$cap"
    NZ_CONV_TCAP=$cap python3 ../../tools/bench_stage.py gauss --reps 30 2>/dev/null
  done
done
