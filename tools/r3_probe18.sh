mkdir -p gpurun_out
rm -f gpurun_out/wpe.txt
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp"
for V in "-DNZ_FS_WPE=3" "-DNZ_FS_WPE=2"; do
  (cd noize_job_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 $BASE $V -Rpass-analysis=kernel-resource-usage -c nz_flow_stream.hip -o build/nz_flow_stream.o 2>&1 | grep -A8 "flow_stream_kernelILi5" | grep -E "VGPRs:|Spill" ; /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libnoize_hip.so) >> gpurun_out/wpe.txt 2>&1
  for W in 2048 3072; do
  echo "VARIANT $V WAVES $W" >> gpurun_out/wpe.txt
  NZ_FLOW_STREAM_WAVES=$W python bench.py --no-extras --no-cpu-baseline --steps 100 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().split('\n')[-1]);print(d['value'],d['ms_per_step'],{k:v['ms'] for k,v in d['stages'].items()})" >> gpurun_out/wpe.txt
  done
done
cat gpurun_out/wpe.txt
