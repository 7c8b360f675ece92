#!/bin/bash
# times the fixed-table filters (K = 3..9) for each fusion cap on the GPU box: NZ_CONV_TCAP=<T> tools/try_conv_flags.sh
set -e
cd "$(dirname "$0")/.."
for cap in 1 2 3 4; do
  echo "== NZ_CONV_TCAP=$cap"
  NZ_CONV_TCAP=$cap python3 tools/bench_next.py 2>/dev/null | grep "filter Gauss[79]_S1 x6"
done
