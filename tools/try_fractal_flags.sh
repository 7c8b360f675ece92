#!/bin/bash
# rebuilds nz_fractal.hip with each flag set on the GPU box and times the noise stage
set -e
cd "$(dirname "$0")/../noize_job_amd/csrc"
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -fno-slp-vectorize"
mkdir -p build
for extra in "-DNZ_FT_VEC=2" "-DNZ_FT_VEC=4" "-DNZ_FT_VEC=1"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $BASE $extra -c nz_fractal.hip -o build/nz_fractal.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libnoize_hip.so
  echo "== flags: [$extra]"
  for r in 4096 8192; do python3 ../../tools/bench_stage.py noise --res $r --reps 20 2>/dev/null; done
done
