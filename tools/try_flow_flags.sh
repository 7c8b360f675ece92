#!/bin/bash
set -e
cd "$(dirname "$0")/../noize_job_amd/csrc"
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -fno-slp-vectorize"
for extra in "-DNZ_FT_NT=512 -DNZ_FT_OCC=4" "-DNZ_FT_NT=768 -DNZ_FT_OCC=3" "-DNZ_FT_NT=768 -DNZ_FT_OCC=6" "-DNZ_FT_NT=768 -DNZ_FT_OCC=4"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $BASE $extra -c nz_flow.hip -o build/nz_flow.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A12 "flow_fused_kernelILb1ELb1ELi" | grep -E "VGPRs:|ScratchSize|Occupancy" | sed -E 's/.*remark: +//; s/ \[-R.*//' | head -3 | paste - - -
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libnoize_hip.so
  echo "== flags: [$extra]"
  python3 ../../tools/bench_stage.py flow --reps 10 2>/dev/null
done
cd ../.. && python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "flow" 2>&1 | tail -1
