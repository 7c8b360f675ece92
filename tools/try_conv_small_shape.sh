#!/bin/bash
# nz_filter.hip with other shapes of a small grid's 64-row tile (threads x rows per thread), stage times and tile rates after each
#   tools/try_conv_small_shape.sh "<defines A>" "<defines B>" ...      ("" = the defaults)
cd "$(dirname "$0")/.."
base="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-function"
for defs in "$@"; do
  echo "== [$defs]"
  (cd noize_job_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 $base $defs -c nz_filter.hip -o build/nz_filter.o && make -s) || exit 1
  for r in ${SIZES:-256 512 1024 2048}; do python3 tools/bench_stage.py gauss --res $r --reps 200 2>/dev/null | tail -1; done
  python3 tools/bench_tiles.py --res ${TILE_SIZES:-512 1024} --streams 1 2 --tiles 512 --batch 2>/dev/null | grep -v "^$" | tail -6
done
