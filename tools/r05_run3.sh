set -x
cd $GRAFT_REPO_ROOT
L=gpurun_out/r05_stream_knobs.log; : > $L
for k in "NZ_CONV_STREAM=2" "NZ_CONV_STREAM=2 NZ_CONV_TCAP=6" "NZ_CONV_STREAM=2 NZ_CONV_TCAP=6 NZ_CONV_STREAM_WAVES=3072" "NZ_CONV_STREAM=2 NZ_CONV_TCAP=9" "NZ_CONV_STREAM=2 NZ_CONV_TCAP=9 NZ_CONV_STREAM_WAVES=3072" "NZ_CONV_STREAM=2 NZ_CONV_TCAP=9 NZ_CONV_STREAM_WAVES=2048" "NZ_CONV_STREAM=2 NZ_CONV_TCAP=7" "NZ_CONV_TCAP=6"; do
  echo "== $k" >> $L
  env $k timeout -k 10 120 python tools/bench_modes.py --rounds 1 >> $L 2>&1
done
EXTRA="" timeout -k 10 300 bash tools/probe_conv_phases.sh 4096 17 1 > gpurun_out/r05_conv_phases_mode1.txt 2>&1
NZ_CONV_TCAP=6 timeout -k 10 300 python tools/probe_conv_phases.py 4096 17 1 > gpurun_out/r05_conv_phases_mode1_t6.txt 2>&1
grep -v amdgpu.ids $L
tail -6 gpurun_out/r05_conv_phases_mode1.txt
