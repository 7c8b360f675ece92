#!/bin/bash
# Gauss5 x17 on a READ / WRITE pair: fusion depth per launch (the pair frees the launch count from the even rule)
cd "$(dirname "$0")/.."
for t in 3 4 5 6 7 8; do
  echo "== NZ_CONV_TCAP=$t"
  NZ_CONV_TCAP=$t python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stages']['gauss'])"
done
