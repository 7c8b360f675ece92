#!/bin/bash
# The CPU oracle (test infrastructure) under AddressSanitizer + UndefinedBehaviorSanitizer: builds oracle/*.c with
# -fsanitize=address,undefined into /tmp and runs every CPU test that calls it.  GPU sanitizers are not available on this pool;
# the oracle is the code the parity claims rest on, so its memory accesses and arithmetic are checked here.
#   tools/oracle_sanitize.sh  ->  profiles/r06_oracle_sanitizers.txt
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/profiles/r06_oracle_sanitizers.txt
SO=/tmp/libnoize_oracle_asan.so
FLAGS="-O1 -g -march=x86-64-v3 -std=c11 -fPIC -ffp-contract=off -fno-fast-math -fexcess-precision=standard -fopenmp -fsanitize=address,undefined -fno-sanitize-recover=undefined -Wall -Wextra"
gcc $FLAGS "$ROOT/oracle/noize_oracle.c" "$ROOT/oracle/noize_oracle_live.c" -o $SO -shared -fopenmp -lm
{
  echo "# gcc $(gcc -dumpversion) $FLAGS"
  echo "# LD_PRELOAD=libasan, ASAN_OPTIONS=detect_leaks=0 (the interpreter's own allocations), NZO_LIB=$SO"
  cd "$ROOT"
  LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    NZO_LIB=$SO NZO_MAX_THREADS=4 python3 -m pytest tests/test_oracle_kat.py tests/test_golden.py tests/test_oracle_tables.py \
    tests/test_live_erosion.py tests/test_reference_screenshots.py tests/test_reference_constants.py -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -15
} | tee "$OUT"
