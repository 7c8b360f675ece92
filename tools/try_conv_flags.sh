#!/bin/bash
set -e
cd "$(dirname "$0")/../noize_job_amd/csrc"
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -fno-slp-vectorize"
for extra in "-DNZ_CONV_NT=256" "-DNZ_CONV_NT=512" "-DNZ_CONV_NT=1024"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $BASE $extra -c nz_filter.hip -o build/nz_filter.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libnoize_hip.so
  echo "== flags: [$extra]"
  for tc in 3 4; do NZ_CONV_TCAP=$tc python3 ../../tools/bench_stage.py gauss --reps 10 2>/dev/null; done
done
cd ../.. && python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "kernel_filter or gauss" 2>&1 | tail -1
