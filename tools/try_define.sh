#!/bin/bash
# One source rebuilt with extra defines on the GPU box, stage times after each: an A/B of compile-time knobs in one call.
#   tools/try_define.sh <source.hip> "<defines A>" "<defines B>" ...      ("" = the defaults)
cd "$(dirname "$0")/.."
src=$1; shift
base="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-function"
case $src in
  nz_flow_stream.hip) base="$base -mllvm -amdgpu-sched-strategy=max-ilp";;
  nz_live.hip) base="$base -mllvm -amdgpu-atomic-optimizer-strategy=None -mllvm -amdgpu-sched-strategy=max-ilp";;
esac
for defs in "$@"; do
  echo "== $src [$defs]"
  (cd noize_job_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 $base $defs -c $src -o build/${src%.hip}.o && make -s) || exit 1
  python3 tools/bench_modes.py --rounds ${ROUNDS:-2} ${BENCH_ARGS:-} || exit 1
done
