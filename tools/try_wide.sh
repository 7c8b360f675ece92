#!/bin/bash
# wide-blur tile height A/B: tools/try_wide.sh  (on the GPU box)
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_sweep.py -q -m gpu -x -k "blur or sweep" > gpurun_out/wide_tests.log 2>&1 || { tail -20 gpurun_out/wide_tests.log; exit 1; }
tail -2 gpurun_out/wide_tests.log
for bf in 17 11 99; do
  echo "== NZ_WIDE_BIG_FROM=$bf"
  NZ_WIDE_BIG_FROM=$bf python tools/bench_next.py 2>&1 | grep -i "blur"
done | tee gpurun_out/wide_ab.txt
