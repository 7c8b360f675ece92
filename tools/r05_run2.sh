set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_fast.py -x -q -s > gpurun_out/r05_fast_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05_fast_tests.log
timeout -k 10 200 python tools/bench_modes.py --rounds 2 > gpurun_out/r05_modes.log 2>&1
for k in "NZ_CONV_TCAP=6" "NZ_CONV_TCAP=4" "NZ_CONV_STREAM=2" "NZ_CONV_CHAIN=0" "NZ_CONV_TCAP=6 NZ_CONV_CHAIN=0"; do
  echo "== $k" >> gpurun_out/r05_modes_knobs.log
  env $k timeout -k 10 120 python tools/bench_modes.py --rounds 1 >> gpurun_out/r05_modes_knobs.log 2>&1
done
for m in 0 1; do EXTRA="" timeout -k 10 300 bash tools/probe_conv_phases.sh 4096 17 $m > gpurun_out/r05_conv_phases_mode$m.txt 2>&1; done
NZ_CONV_CHAIN=0 timeout -k 10 300 python tools/probe_conv_phases.py 4096 17 1 > gpurun_out/r05_conv_phases_separate_mode1.txt 2>&1
tail -3 gpurun_out/r05_fast_tests.log
