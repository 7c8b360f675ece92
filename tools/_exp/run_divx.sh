set -e
mkdir -p gpurun_out/r6
cp noize_job_amd/libnoize_hip.so /tmp/base.so
for v in base divx1 divx2 base; do
  if [ $v = base ]; then cp /tmp/base.so noize_job_amd/libnoize_hip.so; else cp tools/_exp/libnoize_hip_$v.so noize_job_amd/libnoize_hip.so; fi
  echo "== $v"
  python tools/bench_stage.py flow --res 4096 --reps 200 --pair
  python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "metric_pipeline_4096_rw_pair" -p no:cacheprovider 2>&1 | tail -1
done
cp /tmp/base.so noize_job_amd/libnoize_hip.so
