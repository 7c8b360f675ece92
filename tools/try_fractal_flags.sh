#!/bin/bash
set -e
cd "$(dirname "$0")/../noize_job_amd/csrc"
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -fno-slp-vectorize"
for extra in "-DNZ_PSR_FMOD=0" "-DNZ_PSR_FMOD=1"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $BASE $extra -c nz_fractal.hip -o build/nz_fractal.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libnoize_hip.so
  echo "== flags: [$extra]"
  for b in 2 4 0 6; do python3 ../../tools/bench_stage.py noise --basis $b --reps 10 2>/dev/null; done
done
cd ../.. && python3 -m pytest tests -m gpu -q -k "fractal or fixtures or degenerate" 2>&1 | tail -1
