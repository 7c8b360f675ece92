#!/bin/bash
# compile-time variants of nz_flow_stream.hip under bench.py's timed steps, on the GPU box's scratch copy of the tree:
#   tools/try_flow_bench.sh "-DNZ_FS_WPE=2:2048" ...      (flags : NZ_FLOW_STREAM_WAVES)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/noize_job_amd/csrc"
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp"
mkdir -p build
restore() { rm -f build/nz_flow_stream.o; make >/dev/null 2>&1 || true; }
trap restore EXIT
for spec in ":0" "$@" ":0"; do
  extra="${spec%%:*}"; waves="${spec##*:}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $BASE $extra -c nz_flow_stream.hip -o build/nz_flow_stream.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -ldl -o ../libnoize_hip.so
  echo "== flags: [$extra] waves: $waves"
  for i in 1 2 3; do
    NZ_FLOW_STREAM_WAVES=$waves python3 "$ROOT/bench.py" --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['ms_per_step'], {k:v['ms'] for k,v in d['stages'].items()})"
  done
done
