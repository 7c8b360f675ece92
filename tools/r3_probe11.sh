set -x
mkdir -p gpurun_out
rm -f gpurun_out/prio.txt
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp"
for V in "-DNZ_FS_PRIO=0" "-DNZ_FS_PRIO=2 -DNZ_FS_PRIO_SHIFT=6" "-DNZ_FS_PRIO=2 -DNZ_FS_PRIO_SHIFT=8" "-DNZ_FS_PRIO=2 -DNZ_FS_PRIO_SHIFT=10"; do
  (cd noize_job_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 $BASE $V -c nz_flow_stream.hip -o build/nz_flow_stream.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libnoize_hip.so)
  echo "VARIANT $V" >> gpurun_out/prio.txt
  python tools/bench_stage.py flow --reps 300 >> gpurun_out/prio.txt 2>&1
  python bench.py --no-extras --no-cpu-baseline --steps 100 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().split('\n')[-1]);print(d['value'],d['ms_per_step'],{k:v['ms'] for k,v in d['stages'].items()})" >> gpurun_out/prio.txt
done
EXTRA="-DNZ_FS_PRIO=2 -DNZ_FS_PRIO_SHIFT=8" bash tools/probe_flow_stream.sh >> gpurun_out/prio.txt 2>&1
grep -v amdgpu.ids gpurun_out/prio.txt
