#!/bin/bash
# The whole GPU suite under every launch-form knob of DESIGN.md section 6 (none of them may change a result).
#   tools/run_knob_matrix.sh [first [last]]   -- knobs first..last of the list (0-based), default all
cd "$(dirname "$0")/.."
# Eleven knobs are left: NZ_RCCL_LIB names a library; the other ten FORCE a launch form at sizes where it is not
# the default, so that the small-grid parity tests cover every form that is the default somewhere.
KNOBS=(NZ_CONV_CHAIN=0 NZ_CONV_CHAIN=2 NZ_CONV_STREAM=0 NZ_CONV_STREAM=2 NZ_CONV_SMALL=0 NZ_CONV_SMALL=2 "NZ_CONV_SMALL=2 NZ_CONV_CHAIN=2"
       NZ_FLOW_STREAM=0 NZ_FLOW_STREAM=2 NZ_FLOW_TINY=0 NZ_FLOW_TINY=2 NZ_FLOW_TINY=3 NZ_FLOW_NMAX=1 NZ_FLOW_NMAX=3
       NZ_EROSION_EMAX=1 NZ_EROSION_EMAX=4 NZ_POOL_RUNS=0 NZ_POOL_SPARSE=0 NZ_POOL_SPARSE=2 NZ_PILE_TICKET=0)
first=${1:-0}; last=${2:-$((${#KNOBS[@]} - 1))}
for ((i = first; i <= last && i < ${#KNOBS[@]}; i++)); do
  kv=${KNOBS[$i]}
  echo "== $kv"
  log=gpurun_out/knob_$i.log
  mkdir -p gpurun_out
  env $kv timeout -k 10 300 python3 -m pytest tests -m gpu -q -rf > $log 2>&1
  grep -E "^FAILED|^ERROR" $log
  tail -1 $log
done
