set -x
cd $GRAFT_REPO_ROOT
L=gpurun_out/r05_static_chain.log; : > $L
for k in "NZ_CHAIN_TICKET=0" "NZ_CHAIN_TICKET=1" "NZ_CHAIN_TICKET=0 NZ_CONV_TCAP=6" "NZ_CHAIN_TICKET=0 NZ_CONV_CHAIN=2 NZ_CONV_TCAP=6"; do
  echo "== $k" >> $L
  env $k timeout -k 10 120 python tools/bench_modes.py --rounds 2 >> $L 2>&1
done
grep -v amdgpu.ids $L
timeout -k 10 600 python -m pytest tests/test_gpu_sweep.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -k "not config5" > gpurun_out/r05_static_tests.log 2>&1; tail -3 gpurun_out/r05_static_tests.log
NZ_CONV_CHAIN=2 timeout -k 10 600 python -m pytest tests/test_gpu_sweep.py tests/test_gpu_parity.py -x -q > gpurun_out/r05_static_tests_chain2.log 2>&1; tail -3 gpurun_out/r05_static_tests_chain2.log
EXTRA="" timeout -k 10 300 bash tools/probe_conv_phases.sh 4096 17 1 > gpurun_out/r05_conv_phases_mode1_static.txt 2>&1
tail -8 gpurun_out/r05_conv_phases_mode1_static.txt
