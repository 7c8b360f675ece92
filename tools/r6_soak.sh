#!/bin/bash
# Soaks on the round's final kernel sources (one GPU call): the random sweeps of tests/test_gpu_sweep.py with fresh seeds and
# random live-erosion set-ups, each against the oracle bit for bit.   tools/r6_soak.sh <minutes each>
M=${1:-7}
mkdir -p gpurun_out
SOAK_SEED=${SOAK_SEED:-60000} python3 tests/soak.py $M > gpurun_out/r06_soak.log 2>&1 &
P1=$!
python3 tests/soak_live.py --minutes $M --seed ${LIVE_SEED:-606} > gpurun_out/r06_soak_live.log 2>&1 &
P2=$!
while kill -0 $P1 2>/dev/null || kill -0 $P2 2>/dev/null; do sleep 60; echo "soaking: $(tail -1 gpurun_out/r06_soak.log | cut -c1-100)"; done
wait $P1; R1=$?; wait $P2; R2=$?
tail -2 gpurun_out/r06_soak.log; tail -2 gpurun_out/r06_soak_live.log
exit $((R1 + R2))
