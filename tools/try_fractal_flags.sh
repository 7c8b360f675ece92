#!/bin/bash
# rebuilds nz_fractal.o with extra flags on the GPU box and times the noise stage
set -e
cd "$(dirname "$0")/../noize_job_amd/csrc"
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -fno-slp-vectorize"
for extra in "-DNZ_EXACT_FMA=0 -DNZ_FR_VEC=2" "-DNZ_EXACT_FMA=1 -DNZ_FR_VEC=2" "-DNZ_EXACT_FMA=1 -DNZ_FR_VEC=1" "-DNZ_EXACT_FMA=1 -DNZ_FR_VEC=4"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $BASE $extra -c nz_fractal.hip -o build/nz_fractal.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libnoize_hip.so
  echo "== flags: [$extra]"
  python3 ../../tools/bench_stage.py noise --reps 10 2>/dev/null
done
cd ../.. && python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "fractal" 2>&1 | tail -2
