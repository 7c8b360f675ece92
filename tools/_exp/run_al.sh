set -e
ROOT=$GRAFT_REPO_ROOT
cp $ROOT/noize_job_amd/libnoize_hip.so /tmp/base.so
for v in base al base al; do
  if [ $v = base ]; then cp /tmp/base.so $ROOT/noize_job_amd/libnoize_hip.so; else cp $ROOT/tools/_exp/libnoize_hip_$v.so $ROOT/noize_job_amd/libnoize_hip.so; fi
  echo "== $v"
  python tools/bench_stage.py gauss --res 4096 --reps 300 --pair
done
python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "metric_pipeline_4096" -p no:cacheprovider 2>&1 | tail -1
python tools/bench_stage.py gauss --res 2048 --reps 300 --pair
python tools/bench_stage.py gauss --res 8192 --reps 100 --pair
cp /tmp/base.so $ROOT/noize_job_amd/libnoize_hip.so
python tools/bench_stage.py gauss --res 2048 --reps 300 --pair
python tools/bench_stage.py gauss --res 8192 --reps 100 --pair
