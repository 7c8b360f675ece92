mkdir -p gpurun_out
rm -f gpurun_out/delay.txt
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp"
for V in "-DNZ_FS_DELAY=0 -DNZ_FS_WPE=2" "-DNZ_FS_DELAY=10 -DNZ_FS_WPE=2" "-DNZ_FS_DELAY=5 -DNZ_FS_WPE=2" "-DNZ_FS_DELAY=15 -DNZ_FS_WPE=2" "-DNZ_FS_DELAY=4 -DNZ_FS_WPE=2"; do
  (cd noize_job_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 $BASE $V -c nz_flow_stream.hip -o build/nz_flow_stream.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libnoize_hip.so) >> gpurun_out/delay.txt 2>&1
  echo "VARIANT $V" >> gpurun_out/delay.txt
  NZ_FLOW_STREAM=2 NZ_FLOW_STREAM_WAVES=2048 timeout -k 10 300 python -m pytest tests -m gpu -q -x -k "flow or metric_pipeline or rw_pair" 2>&1 | tail -n 1 >> gpurun_out/delay.txt
  for W in 2048 2560; do
  NZ_FLOW_STREAM_WAVES=$W python bench.py --no-extras --no-cpu-baseline --steps 100 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().split('\n')[-1]);print('waves $W',d['value'],d['ms_per_step'],{k:v['ms'] for k,v in d['stages'].items()})" >> gpurun_out/delay.txt
  done
done
cat gpurun_out/delay.txt
