set -x
mkdir -p gpurun_out
rm -f gpurun_out/stage.txt
for W in 3072 3600 4096 4608 6144; do echo "WAVES $W" >> gpurun_out/stage.txt; NZ_FLOW_STREAM_WAVES=$W python tools/bench_stage.py flow --reps 300 >> gpurun_out/stage.txt 2>&1; done
cd noize_job_amd/csrc
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -fno-slp-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 $BASE -DNZ_FS_PRIO=0 -c nz_flow.hip -o build/nz_flow.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libnoize_hip.so
cd ../..
echo "NO PRIO" >> gpurun_out/stage.txt
for W in 3072 4096; do echo "WAVES $W" >> gpurun_out/stage.txt; NZ_FLOW_STREAM_WAVES=$W python tools/bench_stage.py flow --reps 300 >> gpurun_out/stage.txt 2>&1; done
cd noize_job_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 $BASE -mllvm -amdgpu-sched-strategy=max-ilp -c nz_flow.hip -o build/nz_flow.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libnoize_hip.so
cd ../..
echo "MAX-ILP" >> gpurun_out/stage.txt
for W in 3072; do echo "WAVES $W" >> gpurun_out/stage.txt; NZ_FLOW_STREAM_WAVES=$W python tools/bench_stage.py flow --reps 300 >> gpurun_out/stage.txt 2>&1; done
grep -v amdgpu.ids gpurun_out/stage.txt
