#!/bin/bash
# conv_reg_kernel register budget for the 5-tap kernel: 4 waves per SIMD (two 512-thread workgroups per CU) vs 6 (three)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/noize_job_amd/csrc"
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -fno-slp-vectorize"
mkdir -p build
for extra in "-DNZ_CONV5_WAVES=4" "-DNZ_CONV5_WAVES=6"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $BASE $extra -c nz_filter.hip -o build/nz_filter.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libnoize_hip.so
  echo "== flags: [$extra]"
  python3 "$ROOT/tools/bench_stage.py" gauss --reps 40 2>/dev/null | tail -1
  python3 "$ROOT/tools/bench_next.py" 2>/dev/null | grep "filter Gauss[3579]_S1 x6"
done
